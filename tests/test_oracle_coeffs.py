"""The oracle derives every update coefficient itself (oracle/fdtd_numpy.py ``_own_*``); the product derives
them in tidy3d_amd/coeffs.py and tidy3d_amd/discretize.py.  Two independent codings of the formulas written
in both docstrings must agree to 1e-13 — so GPU-vs-oracle parity is not common-mode in the coefficients."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from cases import CASES
from oracle import fdtd_numpy as O
from tidy3d_amd import coeffs as P
from tidy3d_amd.constants import EPSILON_0, ETA_0, MU_0
from tidy3d_amd.discretize import discretize

TOL = 1e-13


def _close(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    scale = max(np.max(np.abs(b)), 1e-300) if b.size else 1.0
    assert np.max(np.abs(a - b)) <= TOL * scale if a.size else True


def test_constants():
    assert abs(O._MU0 - MU_0) <= TOL * MU_0 and abs(O._EPS0 - EPSILON_0) <= TOL * EPSILON_0
    assert abs(O._ETA0_SQ - ETA_0 ** 2) <= 1e-12 * ETA_0 ** 2


@pytest.mark.parametrize("name", sorted(CASES))
def test_own_coefficients_equal_the_products(name):
    spec = discretize(CASES[name](), n_steps=40).spec
    # steps
    ip, idl = O._own_inv_steps(spec)
    pip, pidl = P.inv_steps(spec)
    for a in range(3):
        _close(ip[a], pip[a])
        _close(idl[a], pidl[a])
    assert abs(spec.dt / O._MU0 - P.h_coeff(spec.dt)) <= TOL * P.h_coeff(spec.dt)
    # media
    om, pm = O._OwnMaterials(spec.media, spec.dt), P.material_table(spec.media, spec.dt)
    _close(om.ca, pm.ca); _close(om.cb, pm.cb); _close(om.cc, pm.cc)
    for m in range(pm.n_media):
        _close(om.kap[m], pm.kap[m]); _close(om.bet[m], pm.bet[m])
    # CPML
    for a in range(3):
        o, p = O._OwnPml(spec, a), P.pml_axis(spec, a)
        assert (o.n_lo, o.n_hi) == (p.n_lo, p.n_hi)
        for f in ("kinv_e", "b_e", "c_e", "kinv_h", "b_h", "c_h"):
            _close(getattr(o, f), getattr(p, f))
    # absorber
    dm = P.damping_tables(spec)
    if dm is not None:
        for a in range(3):
            _close(np.exp(-2.0 * np.asarray(spec.absorber[a][0])), dm[a].fb)
            _close(np.exp(-2.0 * np.asarray(spec.absorber[a][1])), dm[a].fc)
    # DFT phase tables
    for m in spec.monitors:
        if m.kind == "dft":
            assert m.stride is not None
            pe, ph = O._own_dft_phases(m, spec.dt)
            _close(pe, m.phase_e); _close(ph, m.phase_h)


def test_apodised_dft_phases():
    pulse = td.GaussianPulse(freq0=3e14, fwidth=1e14)
    sim = td.Simulation(size=(1, 1, 1), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=2e-13,
                        sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ez")],
                        monitors=[td.FieldMonitor(center=(0, 0, 0), size=(0.5, 0.5, 0), freqs=[2.5e14, 3.5e14], name="f",
                                                  apodization=td.ApodizationSpec(start=3e-14, end=1.5e-13, width=1e-14))],
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    spec = discretize(sim).spec
    m = spec.monitors[0]
    assert m.apod == (3e-14, 1.5e-13, 1e-14)
    pe, ph = O._own_dft_phases(m, spec.dt)
    _close(pe, m.phase_e); _close(ph, m.phase_h)
    assert np.abs(pe[0]).max() < 0.1 * np.abs(pe[len(pe) // 2]).max()      # the window really tapers
