"""PlaneWave / TFSF injection (ref source.py:1090, :1204-1257): leakage, amplitude (1 W/um^2),
and Mie scattering against the analytic series (BASELINE config 4 in miniature; the full-size run
is a GPU test)."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd.analytic import mie_cross_sections
from tidy3d_amd.constants import C_0, ETA_0
from tidy3d_amd.data import assemble
from tidy3d_amd.discretize import discretize

from oracle.fdtd_numpy import OracleFdtd

LAM = 1.0
F0 = C_0 / LAM


def solve(sim):
    disc = discretize(sim)
    raw = OracleFdtd(disc.spec).run()
    return assemble(disc, raw, log=""), disc


@pytest.mark.parametrize("axis,direction,pol", [(2, "+", 0.0), (0, "-", np.pi / 2), (1, "+", 0.3)])
def test_empty_tfsf_box_leaks_nothing_and_carries_unit_intensity(axis, direction, pol):
    dl = LAM / 16
    pulse = td.GaussianPulse(freq0=F0, fwidth=F0 / 6)
    sim = td.Simulation(
        size=(1.6, 1.6, 1.6), grid_spec=td.GridSpec.uniform(dl=dl), run_time=40 / F0,
        sources=[td.TFSF(center=(0, 0, 0), size=(0.8, 0.8, 0.8), source_time=pulse, injection_axis=axis,
                         direction=direction, pol_angle=pol)],
        monitors=[td.FluxMonitor(center=(0, 0, 0), size=(1.2, 1.2, 1.2), freqs=[F0], name="out"),
                  td.FieldMonitor(center=(0, 0, 0), size=(0, 0, 0), freqs=[0.9 * F0, F0], name="c"),
                  td.FieldTimeMonitor(center=(0.6, 0.55, 0.5), size=(0, 0, 0), name="sf", colocate=False),
                  td.FieldTimeMonitor(center=(0.1, 0.05, 0.0), size=(0, 0, 0), name="tf", colocate=False)],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=1e-6)
    sd, disc = solve(sim)
    assert len(disc.spec.tfsf) == (2 if pol == 0.3 else 1)
    sf = max(np.abs(v.values).max() for v in sd["sf"].field_components.values())
    tf = max(np.abs(sd["tf"][k].values).max() for k in ("Ex", "Ey", "Ez"))
    assert sf < 1e-10 * tf                      # the scattered-field region stays empty
    c = sd["c"]
    e_amp = np.sqrt(sum(np.abs(c[k].values.ravel()) ** 2 for k in ("Ex", "Ey", "Ez")))
    np.testing.assert_allclose(e_amp, np.sqrt(2 * ETA_0), rtol=2e-3)     # 1 W/um^2 (ref source.py:1210)
    # the polarisation follows ref source.py:966-990: pol_angle rotates from the first to the
    # second tangential axis (x,y,z order)
    tang = [a for a in range(3) if a != axis]
    e0 = abs(c["E" + "xyz"[tang[0]]].values.ravel()[1])
    e1 = abs(c["E" + "xyz"[tang[1]]].values.ravel()[1])
    assert np.arctan2(e1, e0) == pytest.approx(pol, abs=2e-3)
    assert abs(sd["out"].flux.values[0]) < 1e-9


def test_planewave_carries_one_watt_per_square_micron():
    pulse = td.GaussianPulse(freq0=F0, fwidth=F0 / 6)
    sim = td.Simulation(
        size=(0, 0, 3.0), grid_spec=td.GridSpec.uniform(dl=LAM / 20), run_time=30 / F0,
        sources=[td.PlaneWave(center=(0, 0, -1.0), size=(td.inf, td.inf, 0), source_time=pulse, direction="+")],
        monitors=[td.FluxMonitor(center=(0, 0, 0.5), size=(td.inf, td.inf, 0), freqs=[0.9 * F0, F0, 1.1 * F0], name="fwd"),
                  td.FluxMonitor(center=(0, 0, -1.3), size=(td.inf, td.inf, 0), freqs=[F0], name="bwd")],
        boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml()))
    sd, _ = solve(sim)
    # the incident E is normalised to sqrt(2 eta0) exactly; the measured flux additionally carries
    # the linear-interpolation (colocation) factor ~cos(k dl/2) per interpolated field and the
    # numerical impedance of the Yee grid: -1.0 .. -1.5 % at lambda/20
    np.testing.assert_allclose(sd["fwd"].flux.values, 1.0, rtol=2e-2)
    assert abs(sd["bwd"].flux.values[0]) < 1e-6          # one-way injection


def test_mie_scattering_cross_section_small():
    """Dielectric sphere (eps = 4, r = 0.25 um) in a TFSF box; scattered power through a box in the
    scattered-field region / (1 W/um^2) vs the Mie series.  dl = lambda/24 with staircasing:
    5 % (the GPU test runs the finer, BASELINE-sized version)."""
    dl = LAM / 24
    r, eps = 0.25, 4.0
    pulse = td.GaussianPulse(freq0=F0, fwidth=F0 / 5)
    freqs = [0.8 * F0, 0.9 * F0, F0, 1.1 * F0, 1.2 * F0]
    sim = td.Simulation(
        size=(1.4, 1.4, 1.4), grid_spec=td.GridSpec.uniform(dl=dl), run_time=60 / F0,
        structures=[td.Structure(geometry=td.Sphere(radius=r), medium=td.Medium(permittivity=eps))],
        sources=[td.TFSF(center=(0, 0, 0), size=(0.8, 0.8, 0.8), source_time=pulse, injection_axis=2, direction="+")],
        monitors=[td.FluxMonitor(center=(0, 0, 0), size=(1.0, 1.0, 1.0), freqs=freqs, name="sca"),
                  td.FluxMonitor(center=(0, 0, 0), size=(0.6, 0.6, 0.6), freqs=freqs, name="abs")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=1e-5)
    sd, _ = solve(sim)
    _, ana = mie_cross_sections(r, eps, freqs)
    np.testing.assert_allclose(sd["sca"].flux.values, ana, rtol=0.05)
    assert np.max(np.abs(sd["abs"].flux.values)) < 5e-3 * ana.max()     # lossless sphere absorbs nothing


def test_mie_series_known_values():
    from tidy3d_amd.analytic import mie_efficiencies
    assert mie_efficiencies(1.33 + 1e-8j, 3.0)[0] == pytest.approx(1.7534, abs=2e-4)   # Bohren & Huffman
    assert mie_efficiencies(1.5, 10.0)[0] == pytest.approx(2.8820, abs=2e-4)
    m, x = 1.5, 0.05
    assert mie_efficiencies(m, x)[1] == pytest.approx(8 / 3 * x ** 4 * abs((m * m - 1) / (m * m + 2)) ** 2, rel=5e-3)
