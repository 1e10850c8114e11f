"""More than 1022 media (ref scene.py:52 allows 65530): a `CustomMedium` whose permittivity and conductivity
vary independently needs thousands of (eps, sigma) table entries at the raster's fine quantisation steps.  The material table goes
WIDE — 16-bit indices, two words per cell, coefficients read from global memory — and the run takes the two-pass kernels
(`e_update_kernel`, slab-form CPML: `fdtd_kernels.hpp` MatP / medium_of); nothing is coarsened.  Held to the fp64 oracle, which
indexes the same table: emulator on the CPU, the device in the `-m gpu` suite (5 000 media)."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.data import DataArray
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)
DL = 0.05


def _spatial(values, x, y, z):
    a = DataArray(np.asarray(values), {"x": np.asarray(x, float), "y": np.asarray(y, float), "z": np.asarray(z, float)})
    a.tag = "SpatialDataArray"
    return a


def wide_sim(N, n_data, seed=0, eps_span=1.3, sig_span=3.0):
    """A block of CustomMedium (eps 2 ... 2.6 in 1 % steps x sigma over a factor of 3 in 2 % steps, drawn independently per data point)
    that runs into the CPML on the x faces, a uniform lossy bar beside it, two dipoles, a probe and a DFT plane."""
    rng = np.random.default_rng(seed)
    size = tuple(n * DL for n in N)
    ax = [np.linspace(-0.5 * s_, 0.5 * s_, n_data) for s_ in size]
    eps = 2.0 * eps_span ** rng.random((n_data,) * 3)
    sig = 0.01 * sig_span ** rng.random((n_data,) * 3)
    med = td.CustomMedium(permittivity=_spatial(eps, *ax), conductivity=_spatial(sig, *ax), interp_method="nearest")
    return td.Simulation(
        size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, subpixel=False, shutoff=0,
        structures=[td.Structure(geometry=td.Box(center=(0, 0.1 * size[1], 0), size=(td.inf, 0.5 * size[1], 0.6 * size[2])), medium=med),
                    td.Structure(geometry=td.Box(center=(0, -0.35 * size[1], 0), size=(0.4 * size[0], 0.1 * size[1], td.inf)),
                                 medium=td.Medium(permittivity=3.0, conductivity=0.05))],
        sources=[td.PointDipole(center=(0.02, 0.01, 0.03), source_time=PULSE, polarization="Ez"),
                 td.PointDipole(center=(-0.1 * size[0], 0.15 * size[1], -0.05), source_time=PULSE, polarization="Ex")],
        monitors=[td.FieldTimeMonitor(center=(0.1, 0.1, 0.05), size=(0, 0, 0), name="probe", interval=2, colocate=False),
                  td.FieldMonitor(center=(0, 0, 0.1), size=(td.inf, td.inf, 0), freqs=[2.5e14, 3e14], name="f", colocate=False)],
        boundary_spec=td.BoundarySpec(x=td.Boundary.pml(num_layers=5), y=td.Boundary(minus=td.PECBoundary(), plus=td.PML(num_layers=4)),
                                      z=td.Boundary.pml(num_layers=3)))


def check_against_oracle(disc, lib, tol=2e-5):
    from oracle.fdtd_numpy import OracleFdtd
    o = OracleFdtd(disc.spec)
    ref = o.run()
    with HipEngine(disc.spec, lib=lib) as e:
        assert e.variant == L.VARIANT_ZMARCH             # the wide table leaves the fused sweeps for the two-pass kernels
        st = e.run()
        assert not st.diverged
        got = e.results()
        f = [e.get_field(c) for c in range(6)]
    scale = max(np.linalg.norm(v) / np.sqrt(v.size) for v in ref.values())
    for k in ref:
        den = max(np.linalg.norm(ref[k]), 0.5 * scale * np.sqrt(ref[k].size))
        assert np.linalg.norm(np.asarray(got[k]) - ref[k]) / den < tol, k
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.H))
    assert en > 0
    for c in range(3):
        assert np.linalg.norm(f[c] - o.E[c]) / en < tol, c
        assert np.linalg.norm(f[3 + c] - o.H[c]) / hn < tol, c


def test_wide_material_table_on_the_emulator(emu_lib):
    # (eps over a factor of 2, sigma over a decade: thousands of pairs at the fine levels, and still ~ 2000 after ONE coarsening step —
    #  data that one step brings into the 1022-entry table keep the fused sweeps since round 6, next test)
    disc = discretize(wide_sim((28, 24, 20), 24, eps_span=2.0, sig_span=10.0), n_steps=60)
    assert 1023 < len(disc.spec.media) < 65531, len(disc.spec.media)
    # ONE set of levels for the medium, whichever block met the table's end: the fine ones (1 % in eps) or those one coarsening step
    # above them (2 %)
    eps_of = np.sort(np.unique([m.eps_inf for m in disc.spec.media if m.name.startswith("custom_")]))
    steps = np.diff(np.log(eps_of))
    assert np.ptp(steps) < 1e-6 and (abs(steps[0] - 0.00995) < 1e-4 or abs(steps[0] - 0.0199) < 1e-4), steps[:4]
    check_against_oracle(disc, emu_lib)


def test_one_coarsening_step_keeps_the_fused_sweeps():
    """data whose pairs need ~ 1500 slots at the fine levels but fit the 1022-entry table after one coarsening step
    (0.4 % in permittivity, below the staircase error) keep the fused sweeps and their step pairs instead of the wide table"""
    disc = discretize(wide_sim((28, 24, 20), 24), n_steps=4)
    assert len(disc.spec.media) <= 1023, len(disc.spec.media)
    eps_of = np.sort(np.unique([m.eps_inf for m in disc.spec.media if m.name.startswith("custom_")]))
    assert 0.0198 < np.min(np.diff(np.log(eps_of))) < 0.0200


@pytest.mark.gpu
def test_five_thousand_media_on_the_device(hip_lib):
    disc = discretize(wide_sim((96, 80, 64), 48, eps_span=3.0), n_steps=150)
    n_media = len(disc.spec.media)
    print(f"\n[wide table] {n_media} media on a {disc.spec.shape} grid")
    assert n_media >= 5000, n_media
    check_against_oracle(disc, hip_lib)
