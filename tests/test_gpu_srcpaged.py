"""Step pairs while big source lists inject, on the device (round 6; FDTD_OPT_SRC_PAGED — VERDICT round 5, item 5):

  (a) the parity cases `tfsf_box` and `tfsf_angled_box` (a TFSF box around a sphere inside CPML; normal and oblique incidence: one to
      four incident-grid entries per correction leg) at three times their size THROUGH pairs <= 2e-5 from the fp64 oracle;
  (b) a 320^3 Mie-like problem (BASELINE config 4's set-up: TFSF box, dielectric or Lorentz sphere, CPML, a closed flux box with a
      running DFT, random initial fields) and a strip waveguide with a current sheet through the layers (config 3's source as the
      engine sees it, laid out with the sheet normal to x and to y): pairs == single steps == the round-5 schedule, bit for bit.
The emulator holds the kernel logic to single steps on small grids (tests/test_emu_srcpaged.py)."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

from cases import CASES

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.mark.parametrize("name,w,zc", [("tfsf_box", 8, 8), ("tfsf_box", 16, 32), ("tfsf_angled_box", 6, 5)])
def test_tfsf_cases_through_pairs_vs_oracle(name, w, zc, hip_lib):
    from oracle.fdtd_numpy import OracleFdtd
    fn = CASES[name]
    sim = fn(tuple(3 * n for n in fn.__defaults__[0]))
    disc = discretize(sim, n_steps=100)
    o = OracleFdtd(disc.spec)
    ref = o.run()
    with HipEngine(disc.spec, lib=hip_lib, axis_shift=0) as e:
        e.set_option(L.OPT_TWOSTEP, w + 64 * zc)
        e.set_option(L.OPT_SHELL_PAIRS, 1)
        e.set_option(L.OPT_SHELL2, 1)
        st = e.run(100)
        got = e.results()
        f = [e.get_field(c) for c in range(6)]
    assert int(st.src_paged_pairs) >= 30, (int(st.fused2_pairs), int(st.src_paged_pairs))
    scale = max(np.linalg.norm(v) / np.sqrt(v.size) for v in ref.values())
    for k in ref:
        den = max(np.linalg.norm(ref[k]), 0.5 * scale * np.sqrt(ref[k].size))
        assert np.linalg.norm(np.asarray(got[k]) - ref[k]) / den < TOL, k
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.H))
    for c in range(3):
        assert np.linalg.norm(f[c] - o.E[c]) / en < TOL, c
        assert np.linalg.norm(f[3 + c] - o.H[c]) / hn < TOL, c


def _problem(kind, n):
    dl = 0.05
    pulse = td.GaussianPulse(freq0=2e14, fwidth=4e13)
    pml = td.BoundarySpec.all_sides(td.PML(num_layers=12))
    if kind in ("mie", "mie_lorentz"):
        size = ((n - 24) * dl - 1e-6 * dl,) * 3
        r = 0.18 * n * dl
        med = td.Medium(permittivity=2.56) if kind == "mie" else td.Lorentz(eps_inf=2.0, coeffs=[(2.0, 4e14, 2e13)])
        box = 2 * r + 20 * dl
        return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                             structures=[td.Structure(geometry=td.Sphere(radius=r), medium=med)],
                             sources=[td.TFSF(center=(0, 0, 0), size=(box,) * 3, source_time=pulse, injection_axis=2, direction="+")],
                             monitors=[td.FluxMonitor(center=(0, 0, 0), size=(box + 20 * dl,) * 3, freqs=[1.8e14, 2e14, 2.2e14], name="sca")],
                             boundary_spec=pml, shutoff=0)
    # a strip through the whole domain along `axis`, a current sheet across it that runs through the layers
    axis = {"sheet_x": 0, "sheet_y": 1}[kind]
    N = [n - 24, (n - 24) // 2, (n - 24) // 2]
    if axis == 1:
        N = [N[1], N[0], N[2]]
    size = tuple(m * dl - 1e-6 * dl for m in N)
    bar = [0.5, 0.5, 0.3]
    bar[axis] = td.inf
    ssz = [td.inf, td.inf, td.inf]
    ssz[axis] = 0
    c = [0.0, 0.0, 0.0]
    c[axis] = -0.3 * size[axis]
    return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                         structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=tuple(bar)), medium=td.Medium(permittivity=12.0))],
                         sources=[td.UniformCurrentSource(center=tuple(c), size=tuple(ssz), source_time=pulse, polarization="Ez"),
                                  td.UniformCurrentSource(center=tuple(c), size=tuple(ssz), source_time=pulse, polarization="Hx" if axis == 1 else "Hy")],
                         monitors=[td.FluxMonitor(center=tuple(-v for v in c), size=tuple(ssz), freqs=[2e14], name="through")],
                         boundary_spec=pml, shutoff=0)


def _run(spec, lib, runs, twostep=-1, paged=-1):
    with HipEngine(spec, lib=lib, variant=L.VARIANT_FUSED, axis_shift=0) as e:
        e.set_option(L.OPT_PLACEMENT_TRIES, 0)
        e.set_option(L.OPT_TWOSTEP, twostep)
        e.set_option(L.OPT_SHELL_PAIRS, 1)
        if paged >= 0:
            e.set_option(L.OPT_SRC_PAGED, paged)
        rng = np.random.default_rng(11)
        for c in range(6):
            f = e.get_field(c)
            e.set_field(c, ((1e-3 if c < 3 else 1e-3 / 376.73) * rng.uniform(-1, 1, size=f.shape)).astype(np.float32))
        pairs = sp = dp = 0
        for r in runs:
            st = e.run(r)
            pairs += int(st.fused2_pairs)
            sp += int(st.src_paged_pairs)
            dp += int(st.disp_pairs)
        return [e.get_field(c) for c in range(6)], e.results(), pairs, sp, dp


@pytest.mark.parametrize("kind,n", [("mie", 320), ("mie_lorentz", 320), ("sheet_x", 344), ("sheet_y", 344)])
def test_pairs_while_lists_inject_bit_identical_to_single_steps(kind, n, hip_lib):
    runs = (7, 30, 9)
    disc = discretize(_problem(kind, n), n_steps=sum(runs) + 16)
    disc.spec.decay_every = 0
    ref = _run(disc.spec, hip_lib, runs, twostep=0)
    got = _run(disc.spec, hip_lib, runs)
    old = _run(disc.spec, hip_lib, runs, paged=0)
    assert ref[2] == 0 and got[3] >= sum(r // 2 for r in runs) - 1, got[2:]
    assert old[3] == 0
    if kind == "mie_lorentz":
        assert got[4] == got[2], got[2:]
    for res in (got, old):
        for c in range(6):
            assert np.array_equal(res[0][c], ref[0][c]), c
        for k in ref[1]:
            assert np.array_equal(np.asarray(res[1][k]), np.asarray(ref[1][k])), k
