"""The product-side eigenmode solver (tidy3d_amd/mode_solver.py) against the reference's CPU mode
solver: committed golden values produced by ref plugins/mode/solver.py (tests/golden/
make_golden.py) and, when /root/reference is present, a direct comparison on an asymmetric
structure; then the one-way mode launch built on it."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd.constants import C_0, ETA_0
from tidy3d_amd.mode_solver import mode_flux, solve_modes

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "mode_golden.json")) as f:
    GOLD = json.load(f)["cases"]


def _strip(c):
    dl, W, H, nc, ncl, Lx, Ly = c["dl"], c["W"], c["H"], c["n_core"], c["n_clad"], c["Lx"], c["Ly"]
    nx, ny = int(round(Lx / dl)), int(round(Ly / dl))
    xb = -Lx / 2 + dl * np.arange(nx + 1)
    yb = -Ly / 2 + dl * np.arange(ny + 1)
    xc, yc = (xb[1:] + xb[:-1]) / 2, (yb[1:] + yb[:-1]) / 2

    def eps_at(x, y):
        X, Y = np.meshgrid(x, y, indexing="ij")
        core = (np.abs(X) <= W / 2) & (np.abs(Y) <= H / 2)
        return np.where(core, nc ** 2, ncl ** 2).astype(complex)
    return eps_at(xc, yb[:-1]), eps_at(xb[:-1], yc), eps_at(xb[:-1], yb[:-1]), xb, yb


def test_baseline_config1_strip_matches_reference_golden():
    """BASELINE config[0]: 450 x 220 nm Si/SiO2 strip at 1.55 um, staircased, PEC outer walls."""
    c = GOLD[0]
    assert c["num_pml"] == [0, 0]
    eu, ev, ew, xb, yb = _strip(c)
    r = solve_modes(eu, ev, ew, xb, yb, C_0 / c["wavelength"], num_modes=c["num_modes"])
    ref = np.array([complex(*z) for z in c["n_complex"]])
    np.testing.assert_allclose(r.n_complex, ref, rtol=1e-8)
    ny = len(yb) - 1
    ex = np.abs(r.Eu[:, ny // 2, 0])
    np.testing.assert_allclose(ex / ex.max(), c["abs_Ex_row"], atol=1e-9)
    hy = np.abs(r.Hv[:, ny // 2, 0])
    np.testing.assert_allclose(hy / hy.max(), c["abs_Hy_row"], atol=1e-9)
    ey = np.abs(r.Ev[:, ny // 2, 1])
    np.testing.assert_allclose(ey / ey.max(), c["abs_Ey_row_m1"], atol=1e-7)
    # unit power, forward
    for m in range(2):
        f = dict(Eu=r.Eu[:, :, m], Ev=r.Ev[:, :, m], Hu=r.Hu[:, :, m], Hv=r.Hv[:, :, m])
        assert mode_flux(f, xb, yb) == pytest.approx(1.0, rel=1e-12)


@pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")
def test_against_reference_solver_directly_on_asymmetric_rib():
    import sys
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))
    from oracle.tidy3d_ref_loader import load_mode_solver
    _, solver = load_mode_solver()
    rng = np.random.default_rng(3)
    xb = np.concatenate(([0.0], np.cumsum(0.03 + 0.02 * rng.random(44)))) - 1.0
    yb = np.concatenate(([0.0], np.cumsum(0.025 + 0.02 * rng.random(36)))) - 0.6
    xc, yc = (xb[1:] + xb[:-1]) / 2, (yb[1:] + yb[:-1]) / 2

    def eps_at(x, y):
        X, Y = np.meshgrid(x, y, indexing="ij")
        e = np.full(X.shape, 1.44 ** 2, complex)
        e[(np.abs(X - 0.1) <= 0.3) & (Y >= -0.1) & (Y <= 0.15)] = 3.48 ** 2
        e[(Y >= -0.2) & (Y < -0.1)] = 2.0 ** 2 + 0.05j          # lossy slab below
        return e
    exx, eyy, ezz = eps_at(xc, yb[:-1]), eps_at(xb[:-1], yc), eps_at(xb[:-1], yb[:-1])
    z = np.zeros_like(exx)
    ms = SimpleNamespace(num_modes=3, bend_radius=None, bend_axis=None, angle_theta=0.0, angle_phi=0.0,
                         num_pml=(0, 0), target_neff=None, precision="double")
    fields, n_ref, _ = solver.compute_modes(eps_cross=[exx, z, z, z, eyy, z, z, z, ezz], coords=[xb, yb],
                                            freq=C_0 / 1.55, mode_spec=ms, symmetry=(0, 0), direction="+")
    r = solve_modes(exx, eyy, ezz, xb, yb, C_0 / 1.55, num_modes=3)
    np.testing.assert_allclose(r.n_complex, n_ref, rtol=1e-7)
    for m in range(3):
        # same mode up to one complex scale: normalised inner product of the full tangential vectors
        # (the reference runs ARPACK with tol = fp_eps = 1.2e-7, so ~1e-6 is its own accuracy)
        mine_e = np.concatenate([r.Eu[:, :, m].ravel(), r.Ev[:, :, m].ravel()])
        ref_e = np.concatenate([fields[0, 0, :, :, 0, m].ravel(), fields[0, 1, :, :, 0, m].ravel()])
        mine_h = np.concatenate([r.Hu[:, :, m].ravel(), r.Hv[:, :, m].ravel()])
        ref_h = np.concatenate([fields[1, 0, :, :, 0, m].ravel(), fields[1, 1, :, :, 0, m].ravel()])
        for mine, ref in ((mine_e, ref_e), (mine_h, ref_h)):
            ov = abs(np.vdot(ref, mine)) / (np.linalg.norm(ref) * np.linalg.norm(mine))
            assert ov > 1 - 1e-6, (m, ov)
        # E/H ratio (impedance and relative phase) must agree as well
        a = np.vdot(fields[0, 0, :, :, 0, m], r.Eu[:, :, m]) / np.vdot(fields[0, 0, :, :, 0, m], fields[0, 0, :, :, 0, m])
        b = np.vdot(fields[1, 1, :, :, 0, m], r.Hv[:, :, m]) / np.vdot(fields[1, 1, :, :, 0, m], fields[1, 1, :, :, 0, m])
        assert abs(a / b) == pytest.approx(1.0, rel=1e-6)


def test_mode_source_launches_one_way_with_unit_power():
    """Strip waveguide, ModeSource -> forward flux ~ 1 W (x the colocation factor cos(beta dl/2),
    see tests/test_tfsf.py), backward flux below -50 dB."""
    from tidy3d_amd.data import assemble
    from tidy3d_amd.discretize import discretize
    from oracle.fdtd_numpy import OracleFdtd
    lam = 1.55
    f0 = C_0 / lam
    dl = 0.05
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 10)
    sim = td.Simulation(
        size=(1.6, 1.2, 2.0), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1.2e-13,
        medium=td.Medium(permittivity=1.44 ** 2),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.45, 0.22, td.inf)),
                                 medium=td.Medium(permittivity=3.48 ** 2))],
        sources=[td.ModeSource(center=(0, 0, -0.5), size=(td.inf, td.inf, 0), source_time=pulse, direction="+",
                               mode_spec=td.ModeSpec(num_modes=2), mode_index=0)],
        monitors=[td.FluxMonitor(center=(0, 0, 0.4), size=(td.inf, td.inf, 0), freqs=[0.97 * f0, f0, 1.03 * f0], name="fwd"),
                  td.FluxMonitor(center=(0, 0, -0.8), size=(td.inf, td.inf, 0), freqs=[f0], name="bwd")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=1e-5)
    disc = discretize(sim)
    plane = list(disc.mode_planes.values())[0]
    neff = plane.result.n_complex[0].real
    assert 2.3 < neff < 2.7
    raw = OracleFdtd(disc.spec).run()
    sd = assemble(disc, raw, log="")
    # 1 W launched; the flux *measurement* interpolates E and H linearly to the monitor plane,
    # each losing up to cos(beta dl / 2) (here beta dl = 0.5): expected within [cos^2, 1]
    beta = 2 * np.pi * f0 / C_0 * neff
    lo = np.cos(beta * dl / 2) ** 2
    assert np.all(sd["fwd"].flux.values > lo - 5e-3) and np.all(sd["fwd"].flux.values < 1.005)
    assert abs(sd["bwd"].flux.values[0]) < 1e-5


@pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")
@pytest.mark.parametrize("symmetry", [(1, 0), (0, 1), (1, -1), (-1, 1), (1, 1)])
def test_symmetry_walls_match_reference(symmetry):
    """PMC (+1) / PEC (-1, 0) walls on the min edges == ref solver.py:182-197 (dmin_pmc) on the
    upper-right quadrant of a symmetric rib; effective indices and tangential fields."""
    import sys
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))
    from oracle.tidy3d_ref_loader import load_mode_solver
    _, solver = load_mode_solver()
    xb = np.linspace(0.0, 1.2, 41)
    yb = np.concatenate(([0.0], np.cumsum(np.linspace(0.02, 0.045, 30))))
    xc, yc = (xb[1:] + xb[:-1]) / 2, (yb[1:] + yb[:-1]) / 2

    def eps_at(x, y):
        X, Y = np.meshgrid(x, y, indexing="ij")
        e = np.full(X.shape, 1.44 ** 2, complex)
        e[(X <= 0.25) & (Y <= 0.11)] = 3.48 ** 2
        return e
    exx, eyy, ezz = eps_at(xc, yb[:-1]), eps_at(xb[:-1], yc), eps_at(xb[:-1], yb[:-1])
    z = np.zeros_like(exx)
    ms = SimpleNamespace(num_modes=2, bend_radius=None, bend_axis=None, angle_theta=0.0, angle_phi=0.0,
                         num_pml=(0, 0), target_neff=None, precision="double")
    fields, n_ref, _ = solver.compute_modes(eps_cross=[exx, z, z, z, eyy, z, z, z, ezz], coords=[xb, yb],
                                            freq=C_0 / 1.55, mode_spec=ms, symmetry=symmetry, direction="+")
    r = solve_modes(exx, eyy, ezz, xb, yb, C_0 / 1.55, num_modes=2, pmc_min=tuple(s == 1 for s in symmetry))
    np.testing.assert_allclose(r.n_complex, n_ref, rtol=1e-7)
    for m in range(2):
        mine = np.concatenate([r.Eu[:, :, m].ravel(), r.Ev[:, :, m].ravel(), ETA_0 * r.Hu[:, :, m].ravel(),
                               ETA_0 * r.Hv[:, :, m].ravel()])
        ref = np.concatenate([fields[0, 0, :, :, 0, m].ravel(), fields[0, 1, :, :, 0, m].ravel(),
                              ETA_0 * fields[1, 0, :, :, 0, m].ravel(), ETA_0 * fields[1, 1, :, :, 0, m].ravel()])
        ov = abs(np.vdot(ref, mine)) / (np.linalg.norm(ref) * np.linalg.norm(mine))
        assert ov > 1 - 1e-6, (symmetry, m, ov)


@pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")
@pytest.mark.parametrize("num_pml,symmetry", [((6, 5), (0, 0)), ((0, 7), (0, 0)), ((5, 6), (1, 0)), ((6, 6), (-1, 1))])
def test_mode_plane_pml_matches_reference(num_pml, symmetry):
    """ModeSpec.num_pml: stretched-coordinate PML inside the mode plane == ref derivatives.py:80-232
    (cubic kappa / sigma profiles, speed-averaged sigma_max, no PML on a symmetric min edge); a
    leaky rib on a high-index substrate, complex effective indices."""
    import sys
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))
    from oracle.tidy3d_ref_loader import load_mode_solver
    _, solver = load_mode_solver()
    xb = np.linspace(0.0, 1.6, 49) - (0.0 if symmetry[0] else 0.8)
    yb = np.concatenate(([0.0], np.cumsum(np.linspace(0.025, 0.04, 34)))) - (0.0 if symmetry[1] else 0.5)
    xc, yc = (xb[1:] + xb[:-1]) / 2, (yb[1:] + yb[:-1]) / 2

    def eps_at(x, y):
        X, Y = np.meshgrid(x, y, indexing="ij")
        e = np.full(X.shape, 1.44 ** 2, complex)
        e[(np.abs(X) <= 0.22) & (np.abs(Y) <= 0.11)] = 3.48 ** 2
        e[Y < -0.35] = 3.48 ** 2                         # substrate: the mode leaks into it
        return e
    exx, eyy, ezz = eps_at(xc, yb[:-1]), eps_at(xb[:-1], yc), eps_at(xb[:-1], yb[:-1])
    z = np.zeros_like(exx)
    ms = SimpleNamespace(num_modes=2, bend_radius=None, bend_axis=None, angle_theta=0.0, angle_phi=0.0,
                         num_pml=num_pml, target_neff=2.3, precision="double")
    fields, n_ref, _ = solver.compute_modes(eps_cross=[exx, z, z, z, eyy, z, z, z, ezz], coords=[xb, yb],
                                            freq=C_0 / 1.55, mode_spec=ms, symmetry=symmetry, direction="+")
    r = solve_modes(exx, eyy, ezz, xb, yb, C_0 / 1.55, num_modes=2, target_neff=2.3,
                    pmc_min=tuple(s == 1 for s in symmetry), num_pml=num_pml,
                    pml_min=tuple(s == 0 for s in symmetry))
    order = np.argsort(-np.real(n_ref))
    np.testing.assert_allclose(r.n_complex, np.asarray(n_ref)[order], rtol=1e-7, atol=1e-9)
    for m, mr in enumerate(order):
        mine = np.concatenate([r.Eu[:, :, m].ravel(), r.Ev[:, :, m].ravel()])
        ref = np.concatenate([fields[0, 0, :, :, 0, mr].ravel(), fields[0, 1, :, :, 0, mr].ravel()])
        ov = abs(np.vdot(ref, mine)) / (np.linalg.norm(ref) * np.linalg.norm(mine))
        assert ov > 1 - 1e-6, (m, ov)


@pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")
@pytest.mark.parametrize("radius,bend_axis,num_pml", [(6.0, 1, (8, 0)), (-4.0, 1, (7, 6)), (5.0, 0, (0, 8))])
def test_bent_waveguide_matches_reference(radius, bend_axis, num_pml):
    """ModeSpec.bend_radius / bend_axis: the conformal transformation of ref transforms.py:14-75
    (diagonal eps', mu') — complex effective indices (radiation loss into the PML) and fields."""
    import sys
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))
    from oracle.tidy3d_ref_loader import load_mode_solver
    _, solver = load_mode_solver()
    xb = np.linspace(-1.0, 1.0, 57)
    yb = np.concatenate(([0.0], np.cumsum(np.linspace(0.03, 0.04, 40)))) - 0.7
    xc, yc = (xb[1:] + xb[:-1]) / 2, (yb[1:] + yb[:-1]) / 2

    def eps_at(x, y):
        X, Y = np.meshgrid(x, y, indexing="ij")
        e = np.full(X.shape, 1.44 ** 2, complex)
        e[(np.abs(X) <= 0.25) & (np.abs(Y) <= 0.11)] = 3.48 ** 2
        return e
    exx, eyy, ezz = eps_at(xc, yb[:-1]), eps_at(xb[:-1], yc), eps_at(xb[:-1], yb[:-1])
    z = np.zeros_like(exx)
    ms = SimpleNamespace(num_modes=2, bend_radius=radius, bend_axis=bend_axis, angle_theta=0.0, angle_phi=0.0,
                         num_pml=num_pml, target_neff=2.4, precision="double")
    fields, n_ref, _ = solver.compute_modes(eps_cross=[exx, z, z, z, eyy, z, z, z, ezz], coords=[xb, yb],
                                            freq=C_0 / 1.55, mode_spec=ms, symmetry=(0, 0), direction="+")
    r = solve_modes(exx, eyy, ezz, xb, yb, C_0 / 1.55, num_modes=2, target_neff=2.4, num_pml=num_pml,
                    bend_radius=radius, bend_axis=bend_axis)
    order = np.argsort(-np.real(n_ref))
    np.testing.assert_allclose(r.n_complex, np.asarray(n_ref)[order], rtol=1e-7, atol=1e-10)
    for m, mr in enumerate(order):
        mine = np.concatenate([getattr(r, k)[:, :, m].ravel() for k in ("Eu", "Ev", "Ew")] +
                              [ETA_0 * getattr(r, k)[:, :, m].ravel() for k in ("Hu", "Hv", "Hw")])
        ref = np.concatenate([fields[0, c, :, :, 0, mr].ravel() for c in range(3)] +
                             [ETA_0 * fields[1, c, :, :, 0, mr].ravel() for c in range(3)])
        ov = abs(np.vdot(ref, mine)) / (np.linalg.norm(ref) * np.linalg.norm(mine))
        assert ov > 1 - 1e-6, (m, ov)


def test_filter_pol_reorders_modes():
    """ModeSpec.filter_pol (ref mode_solver.py:523-549): 'tm' brings the mode polarised along the
    second tangential axis to the front; the set of modes is unchanged."""
    from tidy3d_amd.discretize import discretize
    from tidy3d_amd.modesource import mode_profile
    dl = 0.04
    sim = td.Simulation(
        size=(1.0, 2.0, 1.6), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-14, medium=td.Medium(permittivity=1.44 ** 2),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.5, 0.22)),
                                 medium=td.Medium(permittivity=3.48 ** 2))],
        sources=[td.PointDipole(center=(0, 0, 0), source_time=td.GaussianPulse(freq0=2e14, fwidth=2e13), polarization="Ez")],
        monitors=[], boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    spec = discretize(sim, n_steps=2).spec
    box = td.Box(center=(0, 0, 0), size=(0, 1.6, 1.2))
    base = dict(num_modes=2, target_neff=2.6, precision="double")
    plain = mode_profile(spec, box, td.ModeSpec(**base), 2e14).result
    te = mode_profile(spec, box, td.ModeSpec(filter_pol="te", **base), 2e14).result
    tm = mode_profile(spec, box, td.ModeSpec(filter_pol="tm", **base), 2e14).result
    # the fundamental mode of a wide strip is y-polarised: first tangential axis of an x-normal plane
    np.testing.assert_allclose(te.n_complex, plain.n_complex, rtol=1e-12)
    np.testing.assert_allclose(tm.n_complex, plain.n_complex[::-1], rtol=1e-12)
    assert np.abs(tm.Ev[:, :, 0]).max() > np.abs(tm.Eu[:, :, 0]).max() or np.abs(tm.Eu[:, :, 0]).max() > 0


def test_mode_solver_facade_matches_baseline_config1_golden():
    """``tidy3d_amd.plugins.mode.ModeSolver`` on the BASELINE config-1 strip: same n_eff as the
    reference golden used above (through discretize + mode_profile instead of raw arrays)."""
    from tidy3d_amd.plugins.mode import ModeSolver
    wg_w, wg_h = 0.45, 0.22
    sim = td.Simulation(
        size=(2.0, 3.0, 2.0), grid_spec=td.GridSpec.uniform(dl=0.02), run_time=1e-14, subpixel=False,
        medium=td.Medium(permittivity=1.44 ** 2),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, wg_w, wg_h)),
                                 medium=td.Medium(permittivity=3.48 ** 2))],
        sources=[td.PointDipole(center=(0, 0, 0), source_time=td.GaussianPulse(freq0=C_0 / 1.55, fwidth=1e13), polarization="Ey")],
        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    ms = ModeSolver(sim, td.Box(center=(0, 0, 0), size=(0, 2.4, 1.6)), td.ModeSpec(num_modes=2, precision="double"),
                    freqs=[C_0 / 1.55, C_0 / 1.5])
    data = ms.solve()
    assert data.n_complex.shape == (2, 2) and data.Ey.dims == ("x", "y", "z", "f", "mode_index")
    assert data.Ey.shape[0] == 1 and data.Ey.shape[3:] == (2, 2)
    n = data.n_eff.values
    assert 2.2 < n[0, 0] < 2.6 and n[0, 0] > n[0, 1] and n[1, 0] > n[0, 0]      # TE0 of a 450 x 220 nm Si strip
    # TE0: E mostly along y, even in y
    ey = np.abs(data.Ey.values[0, :, :, 0, 0])
    assert ey.max() > np.abs(data.Ez.values[0, :, :, 0, 0]).max()


def test_mode_solver_monitor_through_run(emu_lib, tmp_path):
    """ModeSolverMonitor (ref monitor.py:712): ModeSolverData computed by the run's own mode solver, same numbers
    as the façade, through the .hdf5 layout with the reference's array tags."""
    import json
    from tidy3d_amd import hdf5io
    from tidy3d_amd.plugins.mode import ModeSolver
    from tidy3d_amd.web import load, run
    f0 = C_0 / 1.55
    plane = td.Box(center=(0, 0, 0), size=(0, 1.2, 0.8))
    sim = td.Simulation(
        size=(0.4, 1.6, 1.2), grid_spec=td.GridSpec.uniform(dl=0.04), run_time=1e-14, subpixel=False,
        medium=td.Medium(permittivity=1.44 ** 2),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.45, 0.22)),
                                 medium=td.Medium(permittivity=3.48 ** 2))],
        sources=[td.PointDipole(center=(0, 0, 0), source_time=td.GaussianPulse(freq0=f0, fwidth=1e13), polarization="Ey")],
        monitors=[td.ModeSolverMonitor(center=plane.center, size=plane.size, freqs=[f0], name="modes",
                                       mode_spec=td.ModeSpec(num_modes=2, precision="double"))],
        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    path = str(tmp_path / "modes.hdf5")
    sd = run(sim, verbose=False, lib=emu_lib, n_steps=4, path=path)
    md = sd["modes"]
    ref = ModeSolver(sim, plane, td.ModeSpec(num_modes=2, precision="double"), freqs=[f0]).solve()
    np.testing.assert_allclose(md.n_complex.values, ref.n_complex.values, rtol=1e-12)
    np.testing.assert_allclose(md.Ey.values, ref.Ey.values, rtol=1e-10, atol=1e-12 * np.abs(ref.Ey.values).max())
    assert md.monitor.name == "modes" and md.Ey.dims == ("x", "y", "z", "f", "mode_index")
    entry = json.loads(hdf5io.read_tree(path)["/JSON_STRING"])["data"][0]
    assert entry["type"] == "ModeSolverData" and entry["Ex"] == "ScalarModeFieldDataArray" and entry["n_complex"] == "ModeIndexDataArray"
    back = load(path)["modes"]
    assert np.array_equal(back.Hz.values, md.Hz.values) and np.array_equal(back.n_complex.values, md.n_complex.values)


def test_mode_launch_and_readback_on_the_oracle():
    """ModeSource -> ModeMonitor through the fp64 oracle at a coarse grid (lambda/16 in the core): the mode is
    launched with the grid's own dispersion (modesource.mode_profile(grid_dispersion=True)) and read back with the
    reference's grid correction.  All but 5e-5 of the power through the far plane is in the eigenmode, -50 dB goes
    backwards, and the phase advance between two planes follows the GRID's propagation constant
    (2/dz) asin(.) — the continuum one is 0.07 rad per um off at this resolution."""
    import tidy3d_amd.schema as td
    from oracle.fdtd_numpy import OracleFdtd
    from tidy3d_amd.constants import C_0
    from tidy3d_amd.data import assemble
    from tidy3d_amd.discretize import discretize
    f0 = C_0 / 1.55
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 10)
    plane = (td.inf, td.inf, 0)
    sim = td.Simulation(
        size=(1.44, 1.04, 2.4), grid_spec=td.GridSpec.uniform(dl=0.04), run_time=1.2e-13,
        medium=td.Medium(permittivity=1.44 ** 2),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.44, 0.24, td.inf)),
                                 medium=td.Medium(permittivity=3.48 ** 2))],
        sources=[td.ModeSource(center=(0, 0, -0.8), size=plane, source_time=pulse, direction="+",
                               mode_spec=td.ModeSpec(num_modes=1), mode_index=0)],
        monitors=[td.FluxMonitor(center=(0, 0, 0.6), size=plane, freqs=[f0], name="fwd"),
                  td.FluxMonitor(center=(0, 0, -1.0), size=plane, freqs=[f0], name="bwd"),
                  td.ModeMonitor(center=(0, 0, 0.6), size=plane, freqs=[f0], mode_spec=td.ModeSpec(num_modes=1), name="mm"),
                  td.FieldMonitor(center=(0, 0, -0.2), size=(0, 0, 0), freqs=[f0], name="p1", fields=["Ex"]),
                  td.FieldMonitor(center=(0, 0, 0.4), size=(0, 0, 0), freqs=[f0], name="p2", fields=["Ex"])],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=1e-6, subpixel=False)
    disc = discretize(sim)
    o = OracleFdtd(disc.spec)
    sd = assemble(disc, o.run(), log="")
    src_plane = list(disc.mode_planes.values())[0]
    fwd, bwd = float(sd["fwd"].flux.values[0]), float(sd["bwd"].flux.values[0])
    a = sd["mm"].amps.values
    purity = abs(a[0, 0, 0]) ** 2 * float(sd["mm"].mode_power.values[0, 0, 0]) / fwd
    assert abs(1 - purity) < 5e-5, purity
    assert abs(bwd) < 1e-5 and abs(a[1, 0, 0]) ** 2 < 1e-6       # -50 dB behind the source (8 PML layers, 0.4 um away)
    assert 0.95 < fwd < 1.0
    dphi = np.angle(sd["p2"].Ex.values.ravel()[0] / sd["p1"].Ex.values.ravel()[0])
    beta_grid = src_plane.beta[0].real
    beta_cont = 2 * np.pi * f0 / C_0 * float(np.real(sd["mm"].n_complex.values[0, 0]))
    err_grid = abs(np.angle(np.exp(1j * (dphi - beta_grid * 0.6))))
    err_cont = abs(np.angle(np.exp(1j * (dphi - beta_cont * 0.6))))
    assert err_grid < 2e-3 and err_cont > 10 * err_grid, (err_grid, err_cont)


@pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")
@pytest.mark.parametrize("theta,phi", [(0.3, 0.0), (0.2, 0.7), (-0.25, 1.9)])
def test_angled_plane_against_reference_compute_modes(theta, phi):
    """ModeSpec.angle_theta / angle_phi (ref solver.py:89-160, transforms.py:74-111): the tensorial 4N x 4N problem
    of mode_solver.solve_modes_angled against the LIVE reference ``compute_modes`` on a non-uniform grid: n_eff
    (which has an imaginary part in the sheared frame's truncated plane, in both) and all six field components."""
    from oracle.tidy3d_ref_loader import load_mode_solver
    from tidy3d_amd.mode_solver import solve_modes_angled
    _, solver = load_mode_solver()
    rng = np.random.default_rng(3)
    xb = np.concatenate(([0.0], np.cumsum(0.04 + 0.02 * rng.random(36)))) - 1.0
    yb = np.concatenate(([0.0], np.cumsum(0.035 + 0.02 * rng.random(28)))) - 0.6
    xc, yc = (xb[1:] + xb[:-1]) / 2, (yb[1:] + yb[:-1]) / 2

    def eps_at(x, y):
        X, Y = np.meshgrid(x, y, indexing="ij")
        e = np.full(X.shape, 1.44 ** 2, complex)
        e[(np.abs(X - 0.1) <= 0.3) & (Y >= -0.1) & (Y <= 0.15)] = 3.48 ** 2
        return e
    exx, eyy, ezz = eps_at(xc, yb[:-1]), eps_at(xb[:-1], yc), eps_at(xb[:-1], yb[:-1])
    z = np.zeros_like(exx)
    ms = SimpleNamespace(num_modes=2, bend_radius=None, bend_axis=None, angle_theta=theta, angle_phi=phi,
                         num_pml=(0, 0), target_neff=None, precision="double")
    fields, n_ref, kind = solver.compute_modes(eps_cross=[exx, z, z, z, eyy, z, z, z, ezz], coords=[xb, yb],
                                               freq=C_0 / 1.55, mode_spec=ms, symmetry=(0, 0), direction="+")
    assert kind.startswith("tensorial")
    r = solve_modes_angled(exx, eyy, ezz, xb, yb, C_0 / 1.55, theta, phi, num_modes=2)
    # (the reference runs ARPACK at tol = fp_eps: its own eigenvalues are good to ~1e-6)
    np.testing.assert_allclose(r.n_complex, n_ref, rtol=5e-5)
    for m in range(2):
        for blk, comps in ((0, (r.Eu, r.Ev, r.Ew)), (1, (r.Hu, r.Hv, r.Hw))):
            mine = np.concatenate([c[:, :, m].ravel() for c in comps])
            ref = np.concatenate([fields[blk, q, :, :, 0, m].ravel() for q in range(3)])
            ov = abs(np.vdot(ref, mine)) / (np.linalg.norm(ref) * np.linalg.norm(mine))
            assert ov > 1 - 1e-4, (m, blk, ov)
        a = np.vdot(fields[0, 0, :, :, 0, m], r.Eu[:, :, m]) / np.vdot(fields[0, 0, :, :, 0, m], fields[0, 0, :, :, 0, m])
        b = np.vdot(fields[1, 1, :, :, 0, m], r.Hv[:, :, m]) / np.vdot(fields[1, 1, :, :, 0, m], fields[1, 1, :, :, 0, m])
        assert abs(a / b - 1) < 2e-4          # impedance and relative phase of E and H


@pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")
@pytest.mark.parametrize("theta,phi,radius,bend_axis,num_pml", [(0.2, 0.0, 6.0, 1, (0, 0)), (0.25, 0.6, None, 0, (6, 5)),
                                                                 (-0.15, 1.2, 5.0, 0, (5, 6))])
def test_angled_plane_with_bend_and_pml_against_reference(theta, phi, radius, bend_axis, num_pml):
    """An angled mode plane together with ``bend_radius`` and / or ``num_pml`` (ref solver.py:141-147: the shear and the
    conformal map composed; derivatives.py:79-127: the stretched derivatives) against the LIVE reference ``compute_modes``:
    complex n_eff and all six components."""
    from oracle.tidy3d_ref_loader import load_mode_solver
    from tidy3d_amd.mode_solver import solve_modes_angled
    _, solver = load_mode_solver()
    rng = np.random.default_rng(5)
    xb = np.concatenate(([0.0], np.cumsum(0.04 + 0.02 * rng.random(38)))) - 1.1
    yb = np.concatenate(([0.0], np.cumsum(0.035 + 0.02 * rng.random(30)))) - 0.7
    xc, yc = (xb[1:] + xb[:-1]) / 2, (yb[1:] + yb[:-1]) / 2

    def eps_at(x, y):
        X, Y = np.meshgrid(x, y, indexing="ij")
        e = np.full(X.shape, 1.44 ** 2, complex)
        e[(np.abs(X - 0.05) <= 0.3) & (Y >= -0.1) & (Y <= 0.15)] = 3.48 ** 2
        return e
    exx, eyy, ezz = eps_at(xc, yb[:-1]), eps_at(xb[:-1], yc), eps_at(xb[:-1], yb[:-1])
    z = np.zeros_like(exx)
    ms = SimpleNamespace(num_modes=1, bend_radius=radius, bend_axis=bend_axis, angle_theta=theta, angle_phi=phi,
                         num_pml=num_pml, target_neff=None, precision="double")
    fields, n_ref, kind = solver.compute_modes(eps_cross=[exx, z, z, z, eyy, z, z, z, ezz], coords=[xb, yb],
                                               freq=C_0 / 1.55, mode_spec=ms, symmetry=(0, 0), direction="+")
    assert kind.startswith("tensorial")
    r = solve_modes_angled(exx, eyy, ezz, xb, yb, C_0 / 1.55, theta, phi, num_modes=1, num_pml=num_pml,
                           bend_radius=radius, bend_axis=bend_axis)
    np.testing.assert_allclose(r.n_complex, n_ref, rtol=5e-5)
    for blk, comps in ((0, (r.Eu, r.Ev, r.Ew)), (1, (r.Hu, r.Hv, r.Hw))):
        mine = np.concatenate([c[:, :, 0].ravel() for c in comps])
        ref = np.concatenate([fields[blk, q, :, :, 0, 0].ravel() for q in range(3)])
        ov = abs(np.vdot(ref, mine)) / (np.linalg.norm(ref) * np.linalg.norm(mine))
        assert ov > 1 - 1e-4, (blk, ov)
    a = np.vdot(fields[0, 0, :, :, 0, 0], r.Eu[:, :, 0]) / np.vdot(fields[0, 0, :, :, 0, 0], fields[0, 0, :, :, 0, 0])
    b = np.vdot(fields[1, 1, :, :, 0, 0], r.Hv[:, :, 0]) / np.vdot(fields[1, 1, :, :, 0, 0], fields[1, 1, :, :, 0, 0])
    assert abs(a / b - 1) < 2e-4          # impedance and relative phase of E and H


def test_angled_solver_reduces_to_the_straight_one():
    """theta -> 0: the tensorial problem gives the straight waveguide's n_eff and profile; a small angle lowers
    n_eff like cos(theta) to first order in the transverse confinement."""
    from tidy3d_amd.mode_solver import solve_modes_angled
    c = GOLD[0]
    eu, ev, ew, xb, yb = _strip(dict(c, dl=0.07))
    f = C_0 / c["wavelength"]
    r0 = solve_modes(eu, ev, ew, xb, yb, f, num_modes=1)
    r1 = solve_modes_angled(eu, ev, ew, xb, yb, f, 1e-6, 0.3, num_modes=1)
    assert abs(r1.n_complex[0] - r0.n_complex[0]) < 1e-7
    ov = abs(np.vdot(r0.Eu[:, :, 0], r1.Eu[:, :, 0])) / (np.linalg.norm(r0.Eu[:, :, 0]) * np.linalg.norm(r1.Eu[:, :, 0]))
    assert ov > 1 - 1e-8


def test_mode_profile_accepts_an_angled_mode_spec():
    """Through the product path (modesource.mode_profile): an angled ModeSpec selects the tensorial solver; the
    wave number along the plane normal is k n_eff / cos(theta) (what the half-cell phase of the launch and the
    monitor's grid correction use, ref plugins/mode/mode_solver.py:883-887)."""
    from tidy3d_amd.discretize import discretize
    from tidy3d_amd.modesource import mode_profile
    f0 = C_0 / 1.55
    sim = td.Simulation(size=(1.6, 1.2, 1.0), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=1e-14,
                        medium=td.Medium(permittivity=1.44 ** 2), subpixel=False,
                        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.45, 0.25, td.inf)),
                                                 medium=td.Medium(permittivity=3.48 ** 2))],
                        sources=[td.PointDipole(center=(0, 0, 0), source_time=td.GaussianPulse(freq0=f0, fwidth=f0 / 10),
                                                polarization="Ex")],
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    spec = discretize(sim, n_steps=2).spec
    box = td.Box(center=(0, 0, 0), size=(td.inf, td.inf, 0))
    straight = mode_profile(spec, box, td.ModeSpec(num_modes=1, precision="double"), f0)
    angled = mode_profile(spec, box, td.ModeSpec(num_modes=1, precision="double", angle_theta=0.15, angle_phi=0.4), f0)
    n0, n1 = straight.result.n_complex[0].real, angled.result.n_complex[0].real
    assert 0.97 * n0 < n1 < n0                       # the guided mode sees a slightly lower index along the plane normal
    k0 = 2 * np.pi * f0 / C_0
    assert angled.beta[0].real == pytest.approx(k0 * n1 / np.cos(0.15), rel=1e-12)
    assert np.abs(angled.result.Ew).max() > np.abs(straight.result.Ew).max()      # the tilt shows in the normal component


def test_angled_mode_source_launches_the_tilted_mode_in_an_fdtd_run():
    """An angled mode plane end to end (missing 3): the tensorial solver's mode, launched by the FDTD
    source on a z-normal plane across a waveguide tilted by 0.2 rad, arrives 3.5 um downstream as that mode (angled
    ModeMonitor) to 1e-4 in power with -40 dB behind the source.  fp64 oracle, 2-D; tests/test_gpu_parity.py repeats it on
    the GPU."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cases import check_tilted_launch, tilted_slab_waveguide
    from tidy3d_amd.data import assemble
    from tidy3d_amd.discretize import discretize
    from oracle.fdtd_numpy import OracleFdtd
    disc = discretize(tilted_slab_waveguide(0.2, dl=0.025))
    check_tilted_launch(assemble(disc, OracleFdtd(disc.spec).run(), log=""))


def test_broadband_mode_source_launches_every_frequency_with_its_own_profile():
    """``ModeSource.num_freqs > 1`` (ref source.py:737-772 BroadbandSource, Chebyshev frequency grid :751-758; VERDICT round 2,
    missing 2).  A 220 nm Si slab in oxide, pulse width f0 / 6: with the freq0 profile alone the launch is clean at f0 only —
    1.2 fwidth off centre 2.5e-3 of the power goes backwards and 4e-4 is not in the mode; with five Chebyshev nodes every
    frequency carries its own profile: backward power and purity deficit below 1e-5 across the band (measured 6e-7 / 1e-6),
    the forward power flat to 0.5 %.  The grid is the reference's own formula."""
    import tidy3d_amd.schema as td
    from tidy3d_amd.constants import C_0
    from tidy3d_amd.data import assemble
    from tidy3d_amd.discretize import discretize
    from tidy3d_amd.modesource import chebyshev_frequency_grid
    from oracle.fdtd_numpy import OracleFdtd
    f0 = C_0 / 1.55
    fw = f0 / 6
    pulse = td.GaussianPulse(freq0=f0, fwidth=fw)
    # the reference's formula, restated: nodes cos(pi (2j + 1) / (2N)) on freq0 +- 1.5 fwidth, ascending
    N = 5
    want = f0 + 1.5 * fw * np.cos(np.pi * np.flip((2 * np.arange(N) + 1) / (2 * N)))
    np.testing.assert_allclose(chebyshev_frequency_grid(pulse, N), want, rtol=1e-14)
    freqs = [f0 - 1.2 * fw, f0 - 0.6 * fw, f0, f0 + 0.6 * fw, f0 + 1.2 * fw]
    plane = (3.0, td.inf, 0)
    ms = td.ModeSpec(num_modes=1, target_neff=3.0)

    def run(nf):
        sim = td.Simulation(
            size=(4.0, 0, 5.0), grid_spec=td.GridSpec.uniform(dl=0.025), run_time=2.5e-13, medium=td.Medium(permittivity=1.44 ** 2),
            structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.225, td.inf, td.inf)), medium=td.Medium(permittivity=3.48 ** 2))],
            sources=[td.ModeSource(center=(0, 0, -1.5), size=plane, source_time=pulse, direction="+", mode_spec=ms, mode_index=0,
                                   num_freqs=nf)],
            monitors=[td.FluxMonitor(center=(0, 0, 1.5), size=plane, freqs=freqs, name="fwd"),
                      td.FluxMonitor(center=(0, 0, -1.9), size=plane, freqs=freqs, name="bwd"),
                      td.ModeMonitor(center=(0, 0, 1.5), size=plane, freqs=freqs, mode_spec=ms, name="mm")],
            boundary_spec=td.BoundarySpec(x=td.Boundary.pml(num_layers=12), y=td.Boundary.periodic(), z=td.Boundary.pml(num_layers=12)),
            shutoff=1e-6)
        disc = discretize(sim)
        assert len(disc.spec.sources) == nf
        sd = assemble(disc, OracleFdtd(disc.spec).run(), log="")
        fwd, bwd = sd["fwd"].flux.values, sd["bwd"].flux.values
        a, mp = sd["mm"].amps.values, sd["mm"].mode_power.values
        return fwd / fwd[2], np.abs(bwd / fwd), np.abs(1 - np.abs(a[0, :, 0]) ** 2 * mp[0, :, 0] / fwd)
    f1, b1, p1 = run(1)
    f5, b5, p5 = run(5)
    assert b1[0] > 1e-3 and b1[-1] > 5e-4 and p1[0] > 1e-4          # the single-profile launch IS imperfect off centre
    assert b5.max() < 1e-5 and p5.max() < 1e-5, (b5, p5)
    assert np.abs(f5 - 1).max() < 0.005 < np.abs(f1 - 1).max()


@pytest.mark.parametrize("colocate", [True, False])
def test_mode_solver_monitor_under_symmetry(colocate):
    """ModeSolverMonitor together with Simulation.symmetry (ref monitor.py:688, mode_solver.py:413-438): the plane is solved on
    the computed quarter — PEC wall on the y symmetry plane, PMC wall on the z one, as the run's own sources and monitors see it —
    normalised to unit power over the WHOLE plane and expanded with the parities of the components.  Same n_eff and (up to the
    eigenvector's free phase) the same six components as the solve of the whole plane."""
    from tidy3d_amd.data import assemble
    from tidy3d_amd.discretize import discretize
    from tidy3d_amd.plugins.mode import ModeSolver
    f0 = C_0 / 1.55
    plane = td.Box(center=(0, 0, 0), size=(0, 1.22, 0.82))       # (edges clear of the grid lines)
    ms = td.ModeSpec(num_modes=1, precision="double")

    def sim(symmetry):
        return td.Simulation(
            size=(0.4, 1.6, 1.2), grid_spec=td.GridSpec.uniform(dl=0.04), run_time=1e-14, subpixel=False, symmetry=symmetry,
            medium=td.Medium(permittivity=1.44 ** 2),
            structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.45, 0.22)), medium=td.Medium(permittivity=3.48 ** 2))],
            sources=[td.PointDipole(center=(0, 0, 0), source_time=td.GaussianPulse(freq0=f0, fwidth=1e13), polarization="Ey")],
            monitors=[td.ModeSolverMonitor(center=plane.center, size=plane.size, freqs=[f0], name="modes", mode_spec=ms, colocate=colocate)],
            boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    ref = ModeSolver(sim((0, 0, 0)), plane, ms, freqs=[f0], colocate=colocate).solve()
    s = sim((0, -1, 1))
    disc = discretize(s, n_steps=2)
    assert disc.spec.shape[1] * 2 == disc.spec_full.shape[1] and disc.spec.shape[2] * 2 == disc.spec_full.shape[2]
    md = assemble(disc, {})["modes"]                                     # (a host-side monitor: nothing comes from the device)
    np.testing.assert_allclose(md.n_complex.values, ref.n_complex.values, rtol=1e-9)
    a0 = ref.Ey.values.ravel()
    ph = np.vdot(md.Ey.values.ravel(), a0)
    ph /= abs(ph)
    scale = max(np.abs(getattr(ref, k).values).max() for k in ("Ey", "Hz"))
    for k in ("Ex", "Ey", "Ez", "Hx", "Hy", "Hz"):
        got, want = getattr(md, k), getattr(ref, k)
        assert got.values.shape == want.values.shape, k
        for d in "xyz":
            np.testing.assert_allclose(got.coords[d], want.coords[d], atol=1e-12, err_msg=k + d)
        big = np.abs(want.values).max()
        assert np.abs(got.values * ph - want.values).max() < 1e-6 * max(big, 1e-3 * scale * (1 if k[0] == "E" else 1)), k
    # the standalone facade takes the same route
    alone = ModeSolver(s, plane, ms, freqs=[f0], colocate=colocate).solve()
    np.testing.assert_allclose(alone.n_complex.values, md.n_complex.values, rtol=1e-12)
    assert np.abs(np.abs(alone.Ey.values) - np.abs(md.Ey.values)).max() < 1e-9 * np.abs(md.Ey.values).max()
