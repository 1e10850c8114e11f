"""DiffractionMonitor -> DiffractionData (ref monitor.py:1353, monitor_data.py:2672-2900).  The order
bookkeeping of the reference is server-side (parity unpinned); pinned physically on the oracle: the
orders' powers add up to the flux through the same plane (both sides of a lossless grating, all
frequencies), the grating equation gives the angles, a symmetric grating splits symmetrically, and an
empty cell holds everything in order (0, 0) with the source's polarisation."""
import json

import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import hdf5io
from tidy3d_amd.constants import C_0
from tidy3d_amd.exceptions import SetupError
from tidy3d_amd.web import load, save

from test_physics_oracle import solve

FREQS = [1.9e14, 2e14, 2.1e14]
PULSE = td.GaussianPulse(freq0=2e14, fwidth=2e13)


def _sim(structures, pol_angle=0.3, Lx=2.4, axis=2, medium=None):
    size, cen = [Lx, 0.3, 0.3], lambda v: tuple(v if a == axis else 0.0 for a in range(3))
    size[axis] = 4.0
    if axis != 2:
        size[2] = 0.3 if axis == 0 else 0.3
        size[0 if axis == 1 else 1] = Lx
    plane = tuple(0 if a == axis else td.inf for a in range(3))
    per = {"xyz"[a]: (td.Boundary.pml() if a == axis else td.Boundary.periodic()) for a in range(3)}
    return td.Simulation(
        size=tuple(size), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=2.5e-13, structures=structures, shutoff=0,
        medium=medium or td.Medium(),
        sources=[td.PlaneWave(center=cen(-1.5), size=plane, source_time=PULSE, direction="+", pol_angle=pol_angle)],
        monitors=[td.DiffractionMonitor(center=cen(1.2), size=plane, freqs=FREQS, name="t"),
                  td.FluxMonitor(center=cen(1.2), size=plane, freqs=FREQS, name="t_flux"),
                  td.DiffractionMonitor(center=cen(-1.8), size=plane, freqs=FREQS, name="r", normal_dir="-"),
                  td.FluxMonitor(center=cen(-1.8), size=plane, freqs=FREQS, name="r_flux")],
        boundary_spec=td.BoundarySpec(**per))


GRATING = [td.Structure(geometry=td.Box(center=(0.3, 0, 0), size=(1.0, td.inf, 0.6)), medium=td.Medium(permittivity=4.0))]


@pytest.fixture(scope="module")
def grating():
    return solve(_sim(GRATING))[0]


def test_empty_cell_keeps_everything_in_the_zero_order():
    sd, _, _ = solve(_sim([]))
    d = sd["t"]
    assert list(d.orders_x) == [-1, 0, 1] and list(d.orders_y) == [0]
    P = d.power.values
    np.testing.assert_allclose(P[1, 0], sd["t_flux"].flux.values, rtol=1e-4)
    assert np.all(P[[0, 2]] < 1e-12 * P[1, 0].max())
    # polarisation: E = cos(pol) x' + sin(pol) y'  ->  p = E_theta = x' part, s = E_phi = y' part (ref :2678-2681)
    a = d.amps.values[1, 0, 1]
    assert abs(a[0]) / abs(a[1]) == pytest.approx(np.tan(0.3), rel=1e-3)
    assert np.max(sd["r"].power.values) < 1e-6 * P[1, 0].max()


def test_orders_add_up_to_the_flux_on_both_sides(grating):
    sd = grating
    T, R = sd["t"].power.values.sum(axis=(0, 1)), sd["r"].power.values.sum(axis=(0, 1))
    np.testing.assert_allclose(T, sd["t_flux"].flux.values, rtol=2e-3)
    np.testing.assert_allclose(R, -sd["r_flux"].flux.values, rtol=3e-3)
    assert np.all(sd["t"].power.values[[0, 2], 0, :] > 0.05)            # the first orders carry real power
    np.testing.assert_allclose(sd["t"].power.values[0], sd["t"].power.values[2], rtol=5e-3)   # mirror-symmetric bar


def test_angles_follow_the_grating_equation(grating):
    d = grating["t"]
    theta = d.angles[0].values
    for i_f, f in enumerate(FREQS):
        np.testing.assert_allclose(np.sin(theta[:, 0, i_f]), np.abs(d.orders_x) * C_0 / f / 2.4, rtol=1e-12)
    assert d.sim_size == (2.4, 0.3) and d.bloch_vecs == (0.0, 0.0)
    np.testing.assert_allclose(d.ux[2], C_0 / np.array(FREQS) / 2.4)


def test_orders_are_counted_in_the_medium_of_the_monitor_plane():
    """A substrate of index 2 under the transmission monitor admits twice as many orders (ref
    monitor_data.py:2758-2767: u = order * lambda / (n * size))."""
    sub = td.Structure(geometry=td.Box(center=(0, 0, 1.5), size=(td.inf, td.inf, 2.0)), medium=td.Medium(permittivity=4.0))
    sd, _, _ = solve(_sim(GRATING + [sub]))
    d = sd["t"]
    assert list(d.orders_x) == [-3, -2, -1, 0, 1, 2, 3] and d.structure_index == 1
    assert list(sd["r"].orders_x) == [-1, 0, 1] and sd["r"].structure_index == -1
    np.testing.assert_allclose(d.power.values.sum(axis=(0, 1)), sd["t_flux"].flux.values, rtol=1e-2)
    assert np.degrees(d.angles[0].values[0, 0, 0]) > 75                              # +-3 is glancing at 190 THz


def test_monitor_normal_to_x():
    st = [td.Structure(geometry=td.Box(center=(0, 0.3, 0), size=(0.6, 1.0, td.inf)), medium=td.Medium(permittivity=4.0))]
    sd, _, _ = solve(_sim(st, axis=0))
    assert list(sd["t"].orders_x) == [-1, 0, 1] and list(sd["t"].orders_y) == [0]      # local x' = y, y' = z
    np.testing.assert_allclose(sd["t"].power.values.sum(axis=(0, 1)), sd["t_flux"].flux.values, rtol=2e-3)
    np.testing.assert_allclose(sd["r"].power.values.sum(axis=(0, 1)), -sd["r_flux"].flux.values, rtol=3e-3)


def test_validators():
    with pytest.raises(SetupError, match="inf"):
        td.DiffractionMonitor(center=(0, 0, 0), size=(1, td.inf, 0), freqs=[2e14], name="d")
    sim = _sim([])
    import dataclasses
    bad = dataclasses.replace(sim, boundary_spec=td.BoundarySpec(x=td.Boundary.pml(), y=td.Boundary.periodic(),
                                                                 z=td.Boundary.pml()))
    from tidy3d_amd.discretize import discretize
    with pytest.raises(SetupError, match="periodic"):
        discretize(bad, n_steps=2)


def test_hdf5_round_trip(grating, tmp_path):
    path = str(tmp_path / "diff.hdf5")
    save(grating, path)
    model = json.loads(hdf5io.read_tree(path)["/JSON_STRING"])
    e = model["data"][0]
    assert e["type"] == "DiffractionData" and e["Etheta"] == "DiffractionDataArray" and e["sim_size"] == [2.4, 0.3]
    back = load(path)
    assert back["t"].Etheta.dims == ("orders_x", "orders_y", "f")
    assert np.array_equal(back["t"].power.values, grating["t"].power.values)


@pytest.mark.parametrize("symmetry", [(1, 0, 0), (1, -1, 0)])
def test_diffraction_monitor_on_a_symmetric_unit_cell(symmetry):
    """DiffractionMonitor together with ``Simulation.symmetry`` on the periodic axes (round 3: symmetric unit cells are
    closed by a wall on the plus face; the near fields are expanded to the whole period before the order integrals): the
    amplitudes of a mirror-symmetric grating under a y-polarised plane wave (E_y tangential to the x plane: +1, normal to
    the y plane: -1) equal the full-cell run."""
    plane = (td.inf, td.inf, 0)

    def sim(sym):
        return td.Simulation(
            size=(2.4, 0.3, 4.0), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=2.5e-13, shutoff=0, symmetry=sym,
            structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(1.0, td.inf, 0.6)), medium=td.Medium(permittivity=4.0))],
            sources=[td.PlaneWave(center=(0, 0, -1.5), size=plane, source_time=PULSE, direction="+", pol_angle=np.pi / 2)],
            monitors=[td.DiffractionMonitor(center=(0, 0, 1.2), size=plane, freqs=FREQS, name="t"),
                      td.DiffractionMonitor(center=(0, 0, -1.8), size=plane, freqs=FREQS, name="r", normal_dir="-")],
            boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml()))
    full = solve(sim((0, 0, 0)))[0]
    half = solve(sim(symmetry))[0]
    for name in ("t", "r"):
        a, b = np.asarray(full[name].amps.values), np.asarray(half[name].amps.values)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-9 * np.abs(a).max(), name
    assert np.asarray(full["t"].power.values).sum() > 1.0
