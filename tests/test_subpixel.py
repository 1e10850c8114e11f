"""Sub-pixel smoothing of dielectric interfaces (``Simulation.subpixel``, ref simulation.py:196,
subpixel_spec.py:117-148).  The reference's averaging code is server-side; what is implemented is the
published method its option names refer to (discretize._subpixel_average), so the pins are physical:
the Mie series and the Airy formula, against which staircasing is the baseline to beat."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd.analytic import mie_cross_sections
from tidy3d_amd.data import medium_eps_table
from tidy3d_amd.discretize import discretize, subpixel_mode
from test_tfsf import F0, LAM, solve


def _mie_sim(subpixel, div=16):
    dl = LAM / div
    pulse = td.GaussianPulse(freq0=F0, fwidth=F0 / 5)
    freqs = [0.8 * F0, 0.9 * F0, F0, 1.1 * F0, 1.2 * F0]
    return td.Simulation(
        size=(1.4, 1.4, 1.4), grid_spec=td.GridSpec.uniform(dl=dl), run_time=60 / F0, subpixel=subpixel,
        structures=[td.Structure(geometry=td.Sphere(radius=0.25), medium=td.Medium(permittivity=4.0))],
        sources=[td.TFSF(center=(0, 0, 0), size=(0.8, 0.8, 0.8), source_time=pulse, injection_axis=2, direction="+")],
        monitors=[td.FluxMonitor(center=(0, 0, 0), size=(1.0, 1.0, 1.0), freqs=freqs, name="sca")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=1e-5), freqs


def test_polarized_averaging_beats_staircasing_on_the_mie_sphere():
    """eps = 4 sphere at lambda/16: staircasing misses the scattering cross-section by up to 14 %,
    the polarised average stays within 5 % (measured: <= 4.1 %)."""
    err = {}
    for name, sub in (("stair", False), ("polarized", True)):
        sim, freqs = _mie_sim(sub)
        sd, _ = solve(sim)
        _, ana = mie_cross_sections(0.25, 4.0, freqs)
        err[name] = np.abs(sd["sca"].flux.values / ana - 1)
    assert err["polarized"].max() < 0.05
    assert err["stair"].max() > 0.10
    assert err["polarized"].max() < 0.5 * err["stair"].max()


def test_modes_and_scope():
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
    vol = td.parse({"type": "SubpixelSpec", "dielectric": {"type": "VolumetricAveraging"}})
    stair = td.parse({"type": "SubpixelSpec", "dielectric": {"type": "Staircasing"}})

    def mk(sub, medium):
        return td.Simulation(size=(1, 1, 1), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=1e-14, subpixel=sub,
                             structures=[td.Structure(geometry=td.Sphere(center=(0.013, 0.02, -0.01), radius=0.3), medium=medium)],
                             sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ez")],
                             monitors=[], boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    modes = [subpixel_mode(mk(s, td.Medium())) for s in (True, False, td.SubpixelSpec(), vol, stair)]
    assert modes == ["polarized", "staircase", "polarized", "volumetric", "staircase"]
    diel = td.Medium(permittivity=4.0)
    a = discretize(mk(False, diel), n_steps=2).spec
    b = discretize(mk(stair, diel), n_steps=2).spec
    assert np.array_equal(a.mat_idx, b.mat_idx) and len(a.media) == 3
    p = discretize(mk(True, diel), n_steps=2).spec
    v = discretize(mk(vol, diel), n_steps=2).spec
    for c in range(3):
        ea, ep, ev = (np.real(medium_eps_table(s, 2e14))[s.mat_idx[c]] for s in (a, p, v))
        changed = ep != ea
        assert 0 < changed.mean() < 0.2                         # interface nodes only
        assert np.all((ep[changed] > 1.0) & (ep[changed] < 4.0))  # strictly between the two media
        # harmonic-weighted (polarised) never exceeds the arithmetic (volumetric) average
        assert np.all(ep <= ev * (1 + 0.0025))                  # (0.2 % quantisation steps of the 10-bit material table)
        # the volume average conserves  int (eps - 1) dV  to the quantisation error
        exact = 3.0 * 4 / 3 * np.pi * 0.3 ** 3
        assert (ev - 1).sum() * 0.05 ** 3 == pytest.approx(exact, rel=0.003)
    # lossy, dispersive and PEC interfaces keep the staircase rule (ref subpixel_spec.py:131-137)
    for med in (td.Medium(permittivity=4.0, conductivity=0.1), td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)]),
                td.PECMedium()):
        s0 = discretize(mk(False, med), n_steps=2).spec
        s1 = discretize(mk(True, med), n_steps=2).spec
        assert np.array_equal(s0.mat_idx, s1.mat_idx)
