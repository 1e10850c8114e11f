"""Medium2D (ref medium.py:6090): sheet materials on zero-thickness geometries.  The raster gives the tangential E nodes of the
sheet's plane the reference's volumetric equivalent (ref medium.py:6170-6238) — held to the LIVE reference's
``Medium2D.volumetric_equivalent`` for the neighbours and cell sizes of each node — and a conductive / Drude sheet transmits and
reflects what the sheet boundary condition says."""
import os

import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd.constants import C_0, EPSILON_0, ETA_0
from tidy3d_amd.data import medium_eps_table
from tidy3d_amd.discretize import discretize

PULSE = td.GaussianPulse(freq0=2e14, fwidth=4e13)


def _sim(sheet, dl=0.05, grid=None, z_sheet=0.012, z_glass_top=0.0):
    return td.Simulation(
        size=(1.0, 1.0, 2.0), grid_spec=grid or td.GridSpec.uniform(dl=dl), run_time=1e-13, subpixel=False,
        structures=[td.Structure(geometry=td.Box.from_bounds((-td.inf, -td.inf, -1.0), (td.inf, td.inf, z_glass_top)), medium=td.Medium(permittivity=2.25)),
                    td.Structure(geometry=td.Box(center=(0.2, 0, 0.3 + 0.5 * z_glass_top), size=(0.4, td.inf, 0.6 - z_glass_top)),
                                 medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 3.2e14, 2e13)])),
                    td.Structure(geometry=td.Box(center=(0, 0, z_sheet), size=(0.8, td.inf, 0)), medium=sheet)],
        sources=[td.PointDipole(center=(0, 0, 0.5), source_time=PULSE, polarization="Ex")],
        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))


@pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")
@pytest.mark.parametrize("z_sheet,z_glass_top", [(0.0, 0.0), (0.012, 0.0), (0.013, 0.013)])
def test_sheet_nodes_hold_the_reference_volumetric_equivalent(z_sheet, z_glass_top):
    """The sheet's neighbours are found at the geometry's OWN plane, before it is snapped to the grid (ref simulation.py:1317
    `subdivide(geometry, ...)`, geometry/utils_2d.py:61-70 get_neighbors): a sheet ON the glass surface sees glass below and air /
    the Lorentz block above — also when no grid line coincides with that surface (third case: the surface and the sheet at
    z = 0.013, the nearest grid plane, z = 0, inside the glass); a sheet 12 nm ABOVE the surface sees the upper medium on both
    sides although it is snapped onto the surface's grid plane."""
    from oracle.tidy3d_ref_loader import load_tidy3d
    tdr = load_tidy3d()
    sheet = td.Medium2D(ss=td.Medium(permittivity=1.0, conductivity=2e-3), tt=td.Drude(eps_inf=1.0, coeffs=[(3e14, 2e13)]))
    sheet_ref = tdr.Medium2D(ss=tdr.Medium(conductivity=2e-3), tt=tdr.Drude(eps_inf=1.0, coeffs=[(3e14, 2e13)]))
    glass, lor = tdr.Medium(permittivity=2.25), tdr.Lorentz(eps_inf=2.0, coeffs=[(1.5, 3.2e14, 2e13)])
    air = tdr.Medium()
    # a non-uniform grid along the normal: the two cell sizes next to the plane differ
    zb = np.concatenate([np.linspace(-1.0, 0.0, 21), 0.0 + np.cumsum(np.linspace(0.03, 0.07, 24))])
    zb = zb[zb <= 1.0 + 1e-9]
    grid = td.GridSpec(grid_x=td.UniformGrid(dl=0.05), grid_y=td.UniformGrid(dl=0.05), grid_z=td.CustomGridBoundaries(coords=tuple(zb)))
    spec = discretize(_sim(sheet, grid=grid, z_sheet=z_sheet, z_glass_top=z_glass_top), n_steps=2).spec
    b = np.asarray(spec.boundaries[2])
    k = int(np.argmin(np.abs(b - z_sheet)))
    assert b[k] == pytest.approx(0.0, abs=1e-12)
    dls = (b[k] - b[k - 1], b[k + 1] - b[k])
    assert dls[1] != pytest.approx(dls[0])
    freqs = np.array([1.5e14, 2e14, 2.7e14])
    tabs = [medium_eps_table(spec, f) for f in freqs]
    # (the reference's own pass, `Simulation.volumetric_structures`, needs shapely for its polygon subdivision — not installed here;
    #  its neighbour rule is applied by hand, the volumetric equivalent itself comes from the LIVE reference)
    for c, name in enumerate(("xx", "yy")):
        xs, ys, _ = spec.yee_coords(c)
        row = spec.mat_idx[c][k, len(ys) // 2, :]
        for x_probe in (-0.3, 0.2):                 # air above the glass / the Lorentz block above it
            i = int(np.argmin(np.abs(np.asarray(xs) - x_probe)))
            below = glass if z_sheet == z_glass_top else (air if x_probe < 0 else lor)
            above = air if x_probe < 0 else lor
            ve = sheet_ref.volumetric_equivalent(axis=2, adjacent_media=(below, above), adjacent_dls=dls)
            want = np.asarray(getattr(ve, name).eps_model(freqs))
            got = np.array([t[row[i]] for t in tabs])
            np.testing.assert_allclose(got, want, rtol=1e-9, err_msg=f"{name} at x = {x_probe}")
        # outside the sheet the plane's nodes are plain raster
        i_out = int(np.argmin(np.abs(np.asarray(xs) + 0.47)))
        assert spec.media[row[i_out]].name != "" and not spec.media[row[i_out]].name.startswith("medium2d_")
    # the normal component has no node on the plane
    assert not any(spec.media[i].name.startswith("medium2d_") for i in np.unique(spec.mat_idx[2]))


def _plane_wave_sim(structures, monitors, dl=0.02):
    per = td.Boundary.periodic()
    return td.Simulation(
        size=(4 * dl, 4 * dl, 6.0), grid_spec=td.GridSpec.uniform(dl=dl), run_time=4e-13, structures=structures, subpixel=False,
        sources=[td.UniformCurrentSource(center=(0, 0, -2.0), size=(td.inf, td.inf, 0), source_time=PULSE, polarization="Ex")],
        monitors=monitors, boundary_spec=td.BoundarySpec(x=per, y=per, z=td.Boundary.pml(num_layers=12)), shutoff=0)


@pytest.mark.parametrize("kind", ["conductive", "drude"])
def test_sheet_transmission_matches_the_sheet_boundary_condition(kind):
    """A sheet of conductance sigma_s(f) in vacuum at normal incidence: t = 1 / (1 + sigma_s eta0 / 2).  Ohmic sheet
    (2 mS: T = 0.528) and a Drude sheet (graphene-like: sigma_s(f) = eps0 wp^2 d / (gamma - i w)) on the fp64 oracle."""
    from test_physics_oracle import solve
    freqs = np.linspace(1.6e14, 2.4e14, 7)
    if kind == "conductive":
        comp = td.Medium(permittivity=1.0, conductivity=2e-3)
        sig_s = np.full(freqs.shape, 2e-3, complex)
    else:
        fp, gam = 1.2e14, 1.5e13                                           # (2 pi)^2 fp^2 d with d folded into the sheet quantity
        comp = td.Drude(eps_inf=1.0, coeffs=[(fp, gam)])
        w = 2 * np.pi * freqs
        sig_s = EPSILON_0 * (2 * np.pi * fp) ** 2 / (2 * np.pi * gam - 1j * w)
    sheet = td.Structure(geometry=td.Box(center=(0, 0, 0.0), size=(td.inf, td.inf, 0)), medium=td.Medium2D(ss=comp, tt=comp))
    mon = [td.FluxMonitor(center=(0, 0, 1.5), size=(td.inf, td.inf, 0), freqs=list(freqs), name="T")]
    sd0, _, _ = solve(_plane_wave_sim([], mon))
    sd1, _, _ = solve(_plane_wave_sim([sheet], mon))
    T = sd1["T"].flux.values / sd0["T"].flux.values
    want = 1.0 / np.abs(1.0 + 0.5 * sig_s * ETA_0) ** 2
    assert 0.3 < want.min() < 0.95
    np.testing.assert_allclose(T, want, rtol=0.01)


def test_lumped_resistor_enters_as_a_conductive_sheet():
    """LumpedResistor (ref lumped_element.py:72-168): a planar box -> a Medium2D sheet of conductance L_voltage / (L_lateral R)
    behind the user's structures; other lumped elements fail loudly instead of being dropped."""
    from tidy3d_amd.exceptions import SetupError, Tidy3dNotImplementedError
    r = td.LumpedResistor(center=(0, 0, 0), size=(0.4, 0.2, 0), resistance=75.0, voltage_axis=0, name="R1")
    assert r.sheet_conductance == pytest.approx(0.4 / 0.2 / 75.0)
    with pytest.raises(SetupError):
        td.LumpedResistor(center=(0, 0, 0), size=(0.4, 0.2, 0), resistance=75.0, voltage_axis=2, name="bad")
    sim = td.Simulation(size=(1.0, 1.0, 1.0), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=1e-13, subpixel=False,
                        sources=[td.PointDipole(center=(0, 0, 0.3), source_time=PULSE, polarization="Ex")],
                        lumped_elements=[r], boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    spec = discretize(sim, n_steps=2).spec
    sheets = [m for m in spec.media if m.name.startswith("medium2d_")]
    assert len(sheets) == 1 and sheets[0].sigma == pytest.approx(r.sheet_conductance / 0.05) and sheets[0].eps_inf == pytest.approx(1.0)
    k = int(np.argmin(np.abs(np.asarray(spec.boundaries[2]))))
    idx = spec.media.index(sheets[0])
    xs, ys, _ = spec.yee_coords(0)
    on = spec.mat_idx[0][k] == idx
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    assert np.array_equal(on, (np.abs(X) <= 0.2) & (np.abs(Y) <= 0.1))
    assert not (spec.mat_idx[0][k + 1] == idx).any() and not (spec.mat_idx[2] == idx).any()
    # the JSON form of the reference parses into the same thing; an element type this front end does not know raises when used
    d = sim.dict()
    assert d["lumped_elements"][0]["type"] == "LumpedResistor"
    back = td.Simulation.from_dict(d)
    assert back.lumped_elements[0].sheet_conductance == pytest.approx(r.sheet_conductance)
    d["lumped_elements"] = [{"type": "LumpedPort", "impedance": 50.0}]
    with pytest.raises(Tidy3dNotImplementedError):
        discretize(td.Simulation.from_dict(d), n_steps=2)
    # the coaxial form (ref lumped_element.py:170): an annulus of conductance ln(r_out / r_in) / (2 pi R)
    cx = td.CoaxialLumpedResistor(center=(0, 0, 0), outer_diameter=0.6, inner_diameter=0.2, normal_axis=2, resistance=50.0, name="C")
    assert cx.sheet_conductance == pytest.approx(np.log(3.0) / (2 * np.pi * 50.0))
    spec = discretize(sim.copy(lumped_elements=(cx,)), n_steps=2).spec
    idx = [i for i, m in enumerate(spec.media) if m.name.startswith("medium2d_")]
    assert len(idx) == 1 and spec.media[idx[0]].sigma == pytest.approx(cx.sheet_conductance / 0.05)
    on = spec.mat_idx[1][k] == idx[0]
    xs, ys, _ = spec.yee_coords(1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    rr = np.hypot(X, Y)
    assert np.array_equal(on, (rr <= 0.3) & ~(rr <= 0.1)) and on.sum() > 20
