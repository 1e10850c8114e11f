"""``Absorber`` boundary (ref boundary.py:427-476): schema, grid extension and the matched-conductivity
layers of tidy3d_amd/coeffs.py ``damping_tables``.  The reference's own discretisation of the absorber
is server-side (parity unpinned), so the behaviour is pinned physically on the oracle: reflection at
normal incidence, also with a dispersive medium running through the layers; the HIP kernels are held
to the oracle by the ``absorber_*`` parity cases (tests/cases.py)."""
import json
import os

import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd.coeffs import damping_tables
from tidy3d_amd.discretize import absorber_profiles, discretize, make_boundaries, num_pml_layers

from test_physics_oracle import solve

REF_SIM = "/root/reference/tests/sims/simulation_sample.json"


def _sim(L, edge, probe_z, dl=0.075, medium=None, run_time=6e-13):
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
    return td.Simulation(
        size=(0, 0, L), grid_spec=td.GridSpec.uniform(dl=dl), run_time=run_time, shutoff=0,
        medium=medium or td.Medium(),
        sources=[td.UniformCurrentSource(center=(0, 0, -L / 2 + 1.5), size=(td.inf, td.inf, 0),
                                         source_time=pulse, polarization="Ex")],
        monitors=[td.FieldTimeMonitor(center=(0, 0, probe_z), size=(0, 0, 0), name="probe", fields=["Ex"],
                                      colocate=False)],
        boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=edge))


def _reflection(edge, medium=None):
    """Probe 3 um behind the sheet with the absorber 3 um further on, against the same probe in a domain
    so long that nothing comes back within the window."""
    sd, disc, _ = solve(_sim(6.0, edge, 0.0, medium=medium))
    sd2, _, _ = solve(_sim(36.0, td.Boundary.pml(), -18 + 3, medium=medium), n_steps=disc.spec.n_steps)
    a, b = sd["probe"].Ex.values.reshape(-1), sd2["probe"].Ex.values.reshape(-1)
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def test_absorber_layers_extend_the_grid_like_pml():
    """ref simulation.py:1002-1017 counts Absorber layers like PML layers; ref test_grid.py:255-273
    (size 4, dl 1, 2 layers -> boundaries -4 .. 4)."""
    sim = td.Simulation(size=(4, 4, 4), grid_spec=td.GridSpec.uniform(dl=1.0), run_time=1e-12,
                        boundary_spec=td.BoundarySpec.all_sides(td.Absorber(num_layers=2)))
    assert num_pml_layers(sim) == [(2, 2)] * 3
    for b in make_boundaries(sim):
        np.testing.assert_allclose(b, np.arange(-4, 5), atol=1e-12)
    assert td.Boundary.absorber().plus.num_layers == 40                      # ref boundary.py:466
    assert td.Boundary.absorber().plus.parameters.sigma_max == 6.4           # ref boundary.py:232


def test_absorber_profile_and_tables():
    ab = td.Absorber(num_layers=4, parameters=td.AbsorberParams(sigma_order=2, sigma_min=0.1, sigma_max=2.1))
    sim = td.Simulation(size=(0.4, 0.4, 0.4), grid_spec=td.GridSpec.uniform(dl=0.1), run_time=1e-13,
                        sources=[td.PointDipole(source_time=td.GaussianPulse(freq0=2e14, fwidth=2e13),
                                                polarization="Ex")],
                        boundary_spec=td.BoundarySpec(x=td.Boundary(minus=ab, plus=td.PECBoundary()),
                                                      y=td.Boundary.periodic(),
                                                      z=td.Boundary(minus=td.PML(num_layers=3), plus=ab)))
    spec = discretize(sim, n_steps=2).spec
    assert spec.shape == (8, 4, 11)
    (sbx, scx, lx, hx), (sby, scy, ly, hy), (sbz, scz, lz, hz) = spec.absorber
    assert (lx, hx, ly, hy, lz, hz) == (4, 0, 0, 0, 0, 4)
    d = np.array([4, 3, 2, 1]) / 4                      # depth of the cell boundaries, wall at depth 1
    np.testing.assert_allclose(sbx[:4], 0.1 + 2.0 * d ** 2)
    np.testing.assert_allclose(scx[:4], 0.1 + 2.0 * (d - 0.125) ** 2)
    assert not sbx[4:].any() and not scx[4:].any() and not sby.any()
    np.testing.assert_allclose(scz[-4:], 0.1 + 2.0 * ((np.arange(4) + 0.5) / 4) ** 2)
    np.testing.assert_allclose(sbz[-3:], 0.1 + 2.0 * (np.arange(1, 4) / 4) ** 2)
    assert sbz[-4] == 0                                  # the entrance plane belongs to the interior
    dm = damping_tables(spec)
    np.testing.assert_allclose(dm[0].fb, np.exp(-2 * sbx))
    assert dm[0].n_lo == 4 and dm[2].n_hi == 4 and dm[1].n_lo + dm[1].n_hi == 0
    assert spec.bc[0][0] == 0 and spec.pml[0][0].num_layers == 0      # PEC wall behind, no CPML tables


@pytest.mark.parametrize("layers,bound", [(40, 2.5e-2), (80, 2e-3)])
def test_absorber_reflection_normal_incidence(layers, bound):
    """Default parameters at 20 cells per wavelength: -36 dB with 40 layers, -60 dB with 80 (ref
    boundary.py:441-449: more layers -> slower ramp -> less reflection)."""
    r = _reflection(td.Boundary.absorber(num_layers=layers))
    assert r < bound, r


def test_absorber_with_dispersive_medium_through_it():
    """The case the reference recommends the absorber for (boundary.py:433): a dispersive medium that
    intersects the absorbing edge.  Stable, and as quiet as in vacuum."""
    med = td.Lorentz(eps_inf=2.0, coeffs=[(1.0, 5e14, 2e13)])
    r = _reflection(td.Boundary.absorber(num_layers=80), medium=med)
    assert r < 4e-3, r


@pytest.mark.skipif(not os.path.exists(REF_SIM), reason="reference checkout not present")
def test_reference_sample_absorber_edge_parses():
    sim = td.parse(json.load(open(REF_SIM)))
    edges = [e for pair in sim.boundary_spec.to_list for e in pair]
    ab = [e for e in edges if isinstance(e, td.Absorber)]
    assert len(ab) == 1 and ab[0].num_layers == 100 and ab[0].parameters.sigma_max == 6.4
