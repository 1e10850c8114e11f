"""Shell pairs on the device (fdtd_capi.hip): step pairs on grids walled by CPML — the two-step sweep over the bulk, the shell
(CPML slabs + collar) as two single steps beside it on the second stream — held to single steps bit for bit and to the fp64
oracle directly:

  (a) the parity cases `pml_box` and `stable_pml_box` (three times their size) THROUGH pairs <= 2e-5 from the oracle;
  (b) a 520 x 72 x 72 grid — three x tiles, odd x layer counts 5 / 3, CPML on all six faces, a lossy bar through the seam and
      into the low-x layers, a PEC box, dipoles next to the seams and to both x slabs, monitors that reach into the shell
      (their pairs give way) — pairs == single steps bit for bit (fields and records), and <= 2e-5 from the oracle;
  (c) the bench V2 spec (materials + 12 CPML layers on six faces, random initial fields) at 320^3 and at BASELINE's 512^3:
      pairs == single steps bit for bit, three times over (two streams: a race would show as differing bits).
Round 5: the same tests cover both forms of a shell pair — the shell as shell2_step_kernel launches (two steps per sweep with the
CPML recursions carried, fdtd_shell2.hpp; the default wherever it applies) and as two single steps through the third field set.
"""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

from cases import CASES, DL, PULSE, rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


S2_PAIRS = [0]      # shell2 pairs of the last _run


def _run(spec, lib, twostep=-1, init=None, steps=None, runs=None, shell=1, shell2=1):
    """shell = 1: shell pairs whatever the cost model says of the grid (these tests are about bits, not about speed); -1: its call.
    shell2 = 1: the shell as shell2_step_kernel launches (two steps per sweep, psi carried) wherever that form applies — the
    round-4 form (two single steps through the third set) where it does not (sources inside the shell while they inject,
    periodic faces, dispersive cells); 0: the round-4 form only"""
    with HipEngine(spec, lib=lib, axis_shift=0) as e:
        e.set_option(L.OPT_TWOSTEP, twostep)
        e.set_option(L.OPT_SHELL_PAIRS, shell)
        e.set_option(L.OPT_SHELL2, shell2)
        S2_PAIRS[0] = 0
        if init is not None:
            for c in range(6):
                e.set_field(c, init[c])
        pairs = 0
        for r in (runs or [steps]):
            st = e.run(r)
            pairs += int(st.shell_pairs)
            S2_PAIRS[0] += int(st.shell2_pairs)
        return [e.get_field(c) for c in range(6)], e.results(), pairs


@pytest.mark.parametrize("name,w,zc", [("pml_box", 5, 4), ("pml_box", 16, 32), ("stable_pml_box", 8, 8),
                                       # dispersive media (z holes of the bulk) running into the layers; the Au array of config 5 in miniature:
                                       # periodic x / y, CPML z, 5 pole pairs, a plane wave (TFSF corrections + incident grid: a z hole too)
                                       ("drude_in_pml", 6, 5), ("au_array", 5, 6)])
def test_parity_cases_through_shell_pairs_vs_oracle(name, w, zc, hip_lib):
    from oracle.fdtd_numpy import OracleFdtd
    fn = CASES[name]
    d0 = fn.__defaults__[0]
    sim = fn(tuple(int(n * 3) for n in d0)) if name != "au_array" else fn(tuple(int(n * 3) for n in d0), fn.__defaults__[1] / 3)
    disc = discretize(sim, n_steps=100)
    o = OracleFdtd(disc.spec)
    ref = o.run()
    f, got, pairs = _run(disc.spec, hip_lib, twostep=w + 64 * zc, shell2=(w + zc) % 2)
    assert pairs > (20 if name != "drude_in_pml" else 10), (name, pairs)      # (drude_in_pml: its DFT plane spans the layers, records end pairs)
    if name in ("pml_box", "stable_pml_box") and (w + zc) % 2:               # (CPML only: the shell2 form, once the dipole's margin allows)
        print(f"\n[{name}] {pairs} shell pairs, {S2_PAIRS[0]} of them in the shell2 form")
    scale = max(np.linalg.norm(v) / np.sqrt(v.size) for v in ref.values())
    for k in ref:
        den = max(np.linalg.norm(ref[k]), 0.5 * scale * np.sqrt(ref[k].size))
        assert np.linalg.norm(np.asarray(got[k]) - ref[k]) / den < TOL, k
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.H))
    for c in range(3):
        assert np.linalg.norm(f[c] - o.E[c]) / en < TOL, c
        assert np.linalg.norm(f[3 + c] - o.H[c]) / hn < TOL, c


def three_x_tile_cpml_sim(N=(512, 64, 64)):
    sx, sy, sz = (n * DL for n in N)
    structures = [
        td.Structure(geometry=td.Sphere(center=(0.35 * sx, 0.05, 0.1), radius=0.9), medium=td.Medium(permittivity=2.2)),
        # a lossy bar that runs through the x tile boundary at cell 256 and into the low-x CPML
        td.Structure(geometry=td.Box(center=(-0.2 * sx, -0.4, -0.3), size=(0.7 * sx, 0.8, 0.6)),
                     medium=td.Medium(permittivity=3.0, conductivity=0.02)),
        td.Structure(geometry=td.Box(center=(0.1 * sx, 0.6, 0.5), size=(0.5, 0.4, 0.3)), medium=td.PEC)]
    bspec = td.BoundarySpec(x=td.Boundary(minus=td.PML(num_layers=5), plus=td.PML(num_layers=3)),
                            y=td.Boundary.pml(num_layers=4),
                            z=td.Boundary(minus=td.PML(num_layers=3), plus=td.PML(num_layers=5)))
    sources = [td.PointDipole(center=(-0.5 * sx + 251.3 * DL, 0.13, 0.07), source_time=PULSE, polarization="Ez"),
               td.PointDipole(center=(-0.5 * sx + 3.2 * DL, -0.21, 0.33), source_time=PULSE, polarization="Hy"),
               td.PointDipole(center=(0.5 * sx - 6.6 * DL, 0.4, -0.5), source_time=PULSE, polarization="Ex"),
               td.PointDipole(center=(0.0, -1.2, 1.1), source_time=PULSE, polarization="Hz")]
    monitors = [td.FieldTimeMonitor(center=(0, 0, 0), size=(td.inf, 0.4, 0), name="t", colocate=False, interval=9),
                td.FieldMonitor(center=(0, 0, 0.2), size=(td.inf, td.inf, 0), freqs=[2.5e14, 3e14], name="f"),
                td.FieldTimeMonitor(center=(0.2, 0.1, -0.1), size=(0, 0, 0), name="probe", colocate=False, interval=1),
                td.FieldMonitor(center=(0, 0.1, 0), size=(8.0, 0, 1.5), freqs=[2.5e14, 3e14], name="f_in", colocate=False)]
    return td.Simulation(size=(sx, sy, sz), grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12,
                         structures=structures, sources=sources, monitors=monitors, boundary_spec=bspec, shutoff=0)


def test_three_x_tiles_shell_pairs_bit_identical_and_oracle(hip_lib):
    from oracle.fdtd_numpy import OracleFdtd
    disc = discretize(three_x_tile_cpml_sim(), n_steps=60)
    spec = disc.spec
    assert spec.shape == (520, 72, 72), spec.shape
    ref_f, ref_m, p0 = _run(spec, hip_lib, twostep=0, runs=[25, 35])
    assert p0 == 0
    for ts, s2 in ((-1, 1), (5 + 64 * 7, 0), (16 + 64 * 32, 1)):
        f, m, p1 = _run(spec, hip_lib, twostep=ts, runs=[25, 35], shell2=s2)
        assert p1 > 8, (ts, p1)
        for c in range(6):
            assert np.array_equal(f[c], ref_f[c]), (ts, c, float(np.abs(f[c] - ref_f[c]).max()))
        for k in ref_m:
            assert np.array_equal(m[k], ref_m[k]), (ts, k)
    o = OracleFdtd(spec)
    om = o.run()
    for k in om:
        assert rel_err(ref_m[k], om[k]) < TOL, (k, rel_err(ref_m[k], om[k]))
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.H))
    for c in range(3):
        assert np.linalg.norm(ref_f[c] - o.E[c]) / en < TOL, c
        assert np.linalg.norm(ref_f[3 + c] - o.H[c]) / hn < TOL, c


def _bench_init(n):
    out = []
    for c in range(6):
        arr = np.empty((n, n, n), dtype=np.float32)
        for k in range(n):
            arr[k] = np.random.default_rng(c * 100003 + k).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
        out.append(arr)
    return out


@pytest.mark.parametrize("n,steps", [(320, 14), (512, 12)])
def test_bench_v2_spec_shell_pairs_equal_single_steps(n, steps, hip_lib):
    """bench.py's `workloads.v2`: what the CPML throughput number is measured on."""
    from bench import build_spec
    spec = build_spec(n, steps + 4, "v2")
    init = _bench_init(n)
    ref, _, p0 = _run(spec, hip_lib, twostep=0, init=init, steps=steps)
    assert p0 == 0 and all(np.isfinite(x).all() for x in ref) and max(np.abs(x).max() for x in ref) > 0
    for rep in range(4):
        # three times in the shell2 form (two streams: a race would show as differing bits), once in the round-4 form
        got, _, p1 = _run(spec, hip_lib, init=init, steps=steps, shell2=1 if rep < 3 else 0)
        assert p1 == steps // 2 and S2_PAIRS[0] == (p1 if rep < 3 else 0), (p1, S2_PAIRS[0])
        for c in range(6):
            assert np.array_equal(got[c], ref[c]), (rep, c, float(np.abs(got[c] - ref[c]).max()))


@pytest.mark.parametrize("kind", ["current_sheet", "tfsf"])
def test_pairs_resume_when_sources_are_spent(kind, hip_lib):
    """A current sheet of thousands of nodes / a TFSF box keep single steps for the length of their pulse only (the waveforms end
    where the reference says the sources end; a TFSF list rests after four more transits of its incident grid): the rest of the
    run goes out in shell pairs, bit-identical to single steps, with a flux DFT inside the bulk recording all along."""
    N = (300, 120, 112)
    size = tuple(n * DL for n in N)
    pulse = td.GaussianPulse(freq0=3e14, fwidth=1.2e14)
    srcs = {"current_sheet": [td.UniformCurrentSource(center=(0, 0, -1.5), size=(9.0, 3.5, 0), source_time=pulse, polarization="Ex")],
            "tfsf": [td.TFSF(center=(0, 0, 0), size=(9.0, 3.0, 2.6), source_time=pulse, injection_axis=2, direction="+")]}[kind]
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1.3e-13, sources=srcs,
                        structures=[td.Structure(geometry=td.Sphere(center=(0.5, 0.1, 0.0), radius=0.8), medium=td.Medium(permittivity=3.0))],
                        monitors=[td.FluxMonitor(center=(0.5, 0.1, 0), size=(2.4, 2.2, 2.0), freqs=[2.6e14, 3e14], name="flux"),
                                  td.FieldTimeMonitor(center=(1.5, 0.3, 0.2), size=(0, 0, 0), name="probe", interval=3, colocate=False)],
                        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=0)
    disc = discretize(sim)
    disc.spec.decay_every = 0
    spec = disc.spec
    n_src = max([len(s.wave_e) for s in spec.sources] + [len(t.wave) for t in spec.tfsf])
    assert n_src < spec.n_steps - 200, (n_src, spec.n_steps)
    ref_f, ref_m, p0 = _run(spec, hip_lib, twostep=0, steps=spec.n_steps)
    got_f, got_m, p1 = _run(spec, hip_lib, steps=spec.n_steps)
    assert p0 == 0 and p1 >= 0.3 * (spec.n_steps - n_src), (p1, spec.n_steps, n_src)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.abs(ref_m[k]).max() > 0 and np.array_equal(got_m[k], ref_m[k]), k


def test_periodic_unit_cell_in_step_pairs(hip_lib):
    """A metasurface's unit cell — periodic x and y, CPML in z, a dielectric pillar and a lossy film, dipoles of both kinds in the
    first / last rows and columns — goes out in shell pairs: periodic x wraps inside the two-step sweep (the wrap is one more seam
    for the seam kernel, beside the tile seam at column 256), the two rows next to a periodic y face and the CPML planes are the
    shell.  Pairs == single steps bit for bit, and <= 2e-5 from the fp64 oracle directly."""
    from oracle.fdtd_numpy import OracleFdtd
    N = (320, 48, 120)
    size = tuple((n - 1e-6) * DL for n in N)
    hx, hy, hz = (0.5 * v for v in size)
    per = td.Boundary.periodic()
    structures = [td.Structure(geometry=td.Cylinder(center=(0.4, 0.1, 0.0), radius=0.6, length=1.2, axis=2), medium=td.Medium(permittivity=6.0)),
                  td.Structure(geometry=td.Box(center=(0, 0, -1.0), size=(td.inf, td.inf, 0.3)), medium=td.Medium(permittivity=2.1, conductivity=0.05))]
    sources = [td.PointDipole(center=(-hx + 0.3 * DL, -hy + 0.6 * DL, 0.4), source_time=PULSE, polarization="Ey"),
               td.PointDipole(center=(hx - 0.4 * DL, 0.1, -0.3), source_time=PULSE, polarization="Ez"),
               td.PointDipole(center=(hx - 0.6 * DL, hy - 0.3 * DL, 0.1), source_time=PULSE, polarization="Hz"),
               td.PointDipole(center=(-hx + 255.2 * DL, -hy + 1.4 * DL, 0.9), source_time=PULSE, polarization="Ex"),
               td.PointDipole(center=(0.1, 0.0, 0.2), source_time=PULSE, polarization="Ez")]
    monitors = [td.FieldTimeMonitor(center=(0.3, 0.1, 0.5), size=(0, 0, 0), name="probe", interval=1, colocate=False),
                td.FluxMonitor(center=(0, 0, 1.6), size=(td.inf, td.inf, 0), freqs=[2.8e14, 3.2e14], name="T")]
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, structures=structures, sources=sources,
                        monitors=monitors, boundary_spec=td.BoundarySpec(x=per, y=per, z=td.Boundary.pml(num_layers=8)), shutoff=0)
    disc = discretize(sim, n_steps=70)
    disc.spec.decay_every = 0
    assert disc.spec.shape == (320, 48, 136), disc.spec.shape
    ref_f, ref_m, p0 = _run(disc.spec, hip_lib, twostep=0, steps=70)
    got_f, got_m, p1 = _run(disc.spec, hip_lib, steps=70)
    assert p0 == 0 and p1 >= 12, p1          # (the flux plane spans the periodic axes: its records — every 2nd or 3rd step here — end pairs)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.abs(ref_m[k]).max() > 0 and np.array_equal(got_m[k], ref_m[k]), k
    o = OracleFdtd(disc.spec)
    om = o.run()
    for k in om:
        assert rel_err(got_m[k], om[k]) < TOL, (k, rel_err(got_m[k], om[k]))
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.H))
    for c in range(3):
        assert np.linalg.norm(got_f[c] - o.E[c]) / en < TOL, c
        assert np.linalg.norm(got_f[3 + c] - o.H[c]) / hn < TOL, c


def test_shell_pairs_are_taken_where_they_pay(hip_lib):
    """The cost models' calls (fdtd_capi.hip shell2_why_not / shell_why_not).  Round 5: the shell as shell2_step_kernel launches (two
    steps per sweep, psi carried) pays on every size measured (192^3 x 1.25 ... 512^3 x 1.32, profiles/r5/r5h): the bench V2 grid
    goes out in shell2 pairs by default at 512^3 and at 320^3.  With that form switched off the round-4 form decides as before:
    a shell cell then costs two single steps and more — 512^3 (shell: 16 % of the cells) in pairs, 320^3 (25 %) in single steps,
    and the run says why."""
    from bench import build_spec
    for n, want_r4 in ((512, True), (320, False)):
        spec = build_spec(n, 16, "v2")
        with HipEngine(spec, lib=hip_lib, axis_shift=0) as e:
            st = e.run(12)
            assert int(st.shell_pairs) == 6 and int(st.shell2_pairs) == 6 and int(st.fused2_off_reason) == 0, (n, int(st.shell_pairs), int(st.shell2_pairs))
            e.set_option(L.OPT_SHELL2, 0)
            st = e.run(12)
            assert int(st.shell2_pairs) == 0 and (int(st.shell_pairs) == 6) == want_r4, (n, int(st.shell_pairs), int(st.shell2_pairs))
            assert int(st.fused2_off_reason) == (0 if want_r4 else 12), int(st.fused2_off_reason)


@pytest.mark.gpu
def test_reciprocity_through_step_pairs_on_the_device(hip_lib):
    """Lorentz reciprocity (tests/test_reciprocity.py: the fp64 oracle holds it to 1e-11) through the production path at a size no
    oracle run reaches in a test: 320 x 96 x 136 cells + CPML on every face, a lossy block and a dielectric sphere across the
    x-tile seam, shell pairs.  E_y at B driven from A along x == E_x at A driven from B along y, 500 steps, to fp32 rounding —
    a pin of the two-step sweep, the shell's single steps, the seam repair and the CPML that does not go through the oracle."""
    from test_reciprocity import reciprocity_sims, series
    cfg = dict(N=(320, 96, 136), A=(100, 40, 50), ca=0, B=(290, 70, 90), cb=1,
               bspec=td.BoundarySpec.all_sides(td.PML(num_layers=8)),
               structures=[td.Structure(geometry=td.Box(center=(0.5, 0, 0), size=(6.0, 1.5, 2.0)), medium=td.Medium(permittivity=4.0, conductivity=0.01)),
                           td.Structure(geometry=td.Sphere(center=(4.5, 0.5, 1.0), radius=1.2), medium=td.Medium(permittivity=2.2))])
    (d1, at1), (d2, at2) = reciprocity_sims(n_steps=500, **cfg)
    out, pairs = [], []
    for d, at in ((d1, at1), (d2, at2)):
        with HipEngine(d.spec, lib=hip_lib, variant=L.VARIANT_FUSED, axis_shift=0) as e:
            e.set_option(L.OPT_TWOSTEP, 16 + 64 * 16)
            e.set_option(L.OPT_SHELL_PAIRS, 1)
            st = e.run()
            pairs.append(int(st.shell_pairs))
            out.append(series(e.results(), at))
    assert min(pairs) >= 200, pairs
    assert np.abs(out[0]).max() > 0
    err = float(np.abs(out[0] - out[1]).max() / np.abs(out[0]).max())
    print(f"\n[reciprocity through {pairs} shell pairs] max |E_B<-A - E_A<-B| / max = {err:.2e}")
    assert err < 5e-5
