"""FieldProjectionAngleMonitor (far-field approximation; tidy3d_amd/projection.py restating ref
components/field_projection.py).  The reference's projector needs xarray and cannot run here, so the
pins are analytic: a Hertzian dipole radiates E_theta ~ sin(theta) with no E_phi, and the power
integrated over the far sphere equals the flux through the near-field box."""
import numpy as np

import tidy3d_amd.schema as td
from tidy3d_amd.data import assemble
from tidy3d_amd.discretize import discretize
from oracle.fdtd_numpy import OracleFdtd


def _dipole_sim():
    dl = 1.0 / 20
    f0 = 3e14                       # lambda = 1 um
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 6)
    theta = np.linspace(0, np.pi, 19)
    phi = np.linspace(0, 2 * np.pi, 13)
    sim = td.Simulation(
        size=(1.6, 1.6, 1.6), grid_spec=td.GridSpec.uniform(dl=dl), run_time=40 / f0,
        sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ez")],
        monitors=[td.FieldProjectionAngleMonitor(center=(0, 0, 0), size=(1.0, 1.0, 1.0), freqs=[f0], theta=theta, phi=phi,
                                                 proj_distance=1e4, name="far"),
                  td.FluxMonitor(center=(0, 0, 0), size=(1.0, 1.0, 1.0), freqs=[f0], name="flux")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=1e-5)
    return sim, theta, phi


def check_dipole_far_field(sd, theta, phi):
    far = sd["far"]
    e_t = np.abs(far.Etheta.values[0, :, :, 0])
    e_p = np.abs(far.Ephi.values[0, :, :, 0])
    assert far.Etheta.dims == ("r", "theta", "phi", "f") and far.Etheta.shape == (1, 19, 13, 1)
    # pattern: |E_theta| = A sin(theta), independent of phi; E_phi vanishes
    amp = e_t[9].mean()
    np.testing.assert_allclose(e_t, amp * np.sin(theta)[:, None] * np.ones((1, 13)), atol=0.03 * amp)
    assert e_p.max() < 0.02 * amp
    # power through the far sphere == flux through the near box
    p = far.power.values[0, :, :, 0]                       # W / um^2 at r
    r = 1e4
    integrand = p * np.sin(theta)[:, None] * r ** 2
    trap = getattr(np, "trapezoid", None) or np.trapz
    total = trap(trap(integrand, phi, axis=1), theta)
    np.testing.assert_allclose(total, float(sd["flux"].flux.values[0]), rtol=0.03)
    # impedance of free space in the far zone
    ratio = far.Etheta.values[0, 9, 0, 0] / far.Hphi.values[0, 9, 0, 0]
    assert abs(abs(ratio) - 376.73) < 0.5


def test_dipole_pattern_and_power():
    sim, theta, phi = _dipole_sim()
    disc = discretize(sim)
    sd = assemble(disc, OracleFdtd(disc.spec).run())
    check_dipole_far_field(sd, theta, phi)


def test_cartesian_and_kspace_agree_with_the_angular_projection():
    """The three monitor types are three parametrisations of the same far field (ref
    field_projection.py:584-829): a Cartesian point and a k-space direction must reproduce the
    angular projection at the (r, theta, phi) they map to."""
    dl = 1.0 / 16
    f0 = 3e14
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 6)
    th, ph = 0.6, 0.9
    rr = 5e3
    ux, uy = np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph)
    common = dict(center=(0, 0, 0.3), size=(1.2, 1.2, 0), freqs=[f0], normal_dir="+")
    sim = td.Simulation(
        size=(1.6, 1.6, 1.2), grid_spec=td.GridSpec.uniform(dl=dl), run_time=30 / f0,
        sources=[td.PointDipole(center=(0.1, 0, -0.1), source_time=pulse, polarization="Ex")],
        monitors=[td.FieldProjectionAngleMonitor(theta=[th], phi=[ph], proj_distance=rr, name="a", **common),
                  td.FieldProjectionCartesianMonitor(x=[rr * ux], y=[rr * uy], proj_axis=2, proj_distance=rr * np.cos(th),
                                                     name="c", **common),
                  td.FieldProjectionKSpaceMonitor(ux=[ux, 2.0], uy=[uy], proj_axis=2, proj_distance=rr, name="k", **common)],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=1e-5)
    disc = discretize(sim)
    sd = assemble(disc, OracleFdtd(disc.spec).run())
    a, c, k = sd["a"], sd["c"], sd["k"]
    assert c.Etheta.dims == ("x", "y", "z", "f") and k.Etheta.dims == ("ux", "uy", "r", "f")
    for name in ("Etheta", "Ephi", "Htheta", "Hphi"):
        va = getattr(a, name).values[0, 0, 0, 0]
        np.testing.assert_allclose(getattr(c, name).values[0, 0, 0, 0], va, rtol=1e-9)
        np.testing.assert_allclose(getattr(k, name).values[0, 0, 0, 0], va, rtol=1e-9)
    assert np.isnan(k.Etheta.values[1, 0, 0, 0])              # ux = 2: not a propagating direction
    assert abs(a.Etheta.values[0, 0, 0, 0]) > 0


def test_exact_projection_reproduces_the_simulated_field_and_tends_to_the_far_field():
    """far_field_approx=False (ref field_projection.py:831-1010, exact Green's function): projected to a
    point that lies INSIDE the simulation domain (outside the near-field box) the result must be the field
    the FDTD run itself has there; at a large distance it must agree with the far-field approximation."""
    dl = 1.0 / 20
    f0 = 3e14
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 6)
    # observation points in Cartesian form: x = 0.3, 0.45 at y = 0.2 on the plane z = 0.8 (local origin = centre)
    sim = td.Simulation(
        size=(2.2, 2.2, 2.2), grid_spec=td.GridSpec.uniform(dl=dl), run_time=30 / f0,
        sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ex"),
                 td.PointDipole(center=(0.1, -0.1, 0.05), source_time=pulse, polarization="Ez")],
        monitors=[td.FieldProjectionCartesianMonitor(center=(0, 0, 0), size=(1.0, 1.0, 1.0), freqs=[f0], x=[0.3, 0.45],
                                                     y=[0.2], proj_distance=0.8, proj_axis=2, far_field_approx=False,
                                                     name="near"),
                  td.FieldMonitor(center=(0.375, 0.2, 0.8), size=(0.15, 0, 0), freqs=[f0], name="probe"),
                  td.FieldProjectionAngleMonitor(center=(0, 0, 0), size=(1.0, 1.0, 1.0), freqs=[f0], theta=[0.7, 1.9],
                                                 phi=[0.4], proj_distance=3e3, far_field_approx=False, name="exact"),
                  td.FieldProjectionAngleMonitor(center=(0, 0, 0), size=(1.0, 1.0, 1.0), freqs=[f0], theta=[0.7, 1.9],
                                                 phi=[0.4], proj_distance=3e3, name="far")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=1e-5)
    disc = discretize(sim)
    sd = assemble(disc, OracleFdtd(disc.spec).run())
    near, probe = sd["near"], sd["probe"]
    assert near.Etheta.dims == ("x", "y", "z", "f")
    for ix, x in enumerate((0.3, 0.45)):
        pt = np.array([x, 0.2, 0.8])
        r = np.linalg.norm(pt)
        th, ph = np.arccos(pt[2] / r), np.arctan2(pt[1], pt[0])
        unit = {"r": (np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)),
                "theta": (np.cos(th) * np.cos(ph), np.cos(th) * np.sin(ph), -np.sin(th)),
                "phi": (-np.sin(ph), np.cos(ph), 0.0)}
        for F in "EH":
            cart = np.array([np.interp(x, np.asarray(probe[F + c].coords["x"]), probe[F + c].values[:, 0, 0, 0].real)
                             + 1j * np.interp(x, np.asarray(probe[F + c].coords["x"]), probe[F + c].values[:, 0, 0, 0].imag)
                             for c in "xyz"])
            scale = np.linalg.norm(cart)
            for name, uvec in unit.items():
                got = getattr(near, F + name).values[ix, 0, 0, 0]
                assert abs(got - np.dot(uvec, cart)) < 0.03 * scale, (F, name, got, np.dot(uvec, cart))
    ex, fa = sd["exact"], sd["far"]
    for comp in ("Etheta", "Ephi", "Htheta", "Hphi"):
        a, b = getattr(ex, comp).values, getattr(fa, comp).values
        assert np.max(np.abs(a - b)) < 2e-3 * np.max(np.abs(b)), comp
    assert np.max(np.abs(ex.Er.values)) < 1e-2 * np.max(np.abs(ex.Etheta.values))


def test_device_integration_equals_the_numpy_sums(emu_lib):
    """``fdtd_far_field`` (kernel K9, here compiled for the host by tests/hipemu) takes the same surface integrals as
    the NumPy path: every projected component agrees to rounding, for a lossless and for a lossy projection medium,
    for the angular and the k-space monitor."""
    sim, theta, phi = _dipole_sim()
    lossy = td.Medium(permittivity=2.0, conductivity=0.01)
    mons = list(sim.monitors) + [
        td.FieldProjectionAngleMonitor(center=(0, 0, 0), size=(1.0, 1.0, 1.0), freqs=[2.7e14, 3e14], theta=theta[::3], phi=phi[::4],
                                       proj_distance=50.0, medium=lossy, name="far_lossy"),
        td.FieldProjectionKSpaceMonitor(center=(0, 0, 0.5), size=(1.0, 1.0, 0), freqs=[3e14], ux=np.linspace(-0.6, 0.6, 5),
                                        uy=np.linspace(-0.5, 0.5, 4), proj_axis=2, proj_distance=1e3, name="far_k")]
    sim = sim.copy(monitors=mons, run_time=12 / 3e14, shutoff=0)
    disc = discretize(sim)
    raw = OracleFdtd(disc.spec).run()
    host = assemble(disc, raw)
    dev = assemble(disc, raw, device_lib=emu_lib)
    for name in ("far", "far_lossy", "far_k"):
        for comp, arr in host[name].field_components.items():
            a, b = np.asarray(arr.values), np.asarray(dev[name].field_components[comp].values)
            scale = max(float(np.nanmax(np.abs(a))), 1e-300)
            assert np.nanmax(np.abs(a - b)) <= 1e-11 * scale + 1e-300, (name, comp)
            assert np.array_equal(np.isnan(a), np.isnan(b))


def test_window_function_equals_the_reference():
    """``window_size`` of a surface projection monitor (ref monitor.py:898-951): the factor projection._window applies to
    the equivalent currents == the reference's own ``window_parameters`` + ``window_function`` on the same sample points,
    for a finite surface and for an infinite one clipped to the sampled range (``custom_bounds``, ref field_projection.py:
    533-541)."""
    import os
    import pytest
    if not os.path.isdir("/root/reference/tidy3d"):
        pytest.skip("reference checkout not present")
    from oracle.tidy3d_ref_loader import load_tidy3d
    from tidy3d_amd.projection import _window
    tdr = load_tidy3d()
    for size, ws in (((1.2, 0.8, 0), (0.3, 0.6)), ((td.inf, 0.9, 0), (1.0, 0.2)), ((0.7, td.inf, 0), (0.0, 0.5))):
        kw = dict(center=(0.1, -0.2, 0.3), size=size, freqs=[3e14], theta=[0.1], phi=[0.0], name="far", window_size=ws)
        mine, ref = td.FieldProjectionAngleMonitor(**kw), tdr.FieldProjectionAngleMonitor(**kw)
        pts = [np.linspace(-0.9, 1.0, 41), np.linspace(-0.8, 0.5, 33), np.array([0.3])]
        pts = [np.clip(p, c - s / 2, c + s / 2) if np.isfinite(s) else p for p, c, s in zip(pts, kw["center"], size)]
        pts = [np.unique(p) for p in pts]
        got = _window(mine, (0, 1), pts)[:, :, 0]
        bounds = [[p[0] for p in pts], [p[-1] for p in pts]]
        wsz, wmin, wplus = ref.window_parameters(custom_bounds=bounds)
        want = np.ones((pts[0].size, pts[1].size))
        for dim in (0, 1):
            if wsz[dim] > 0:
                f = ref.window_function(points=pts[dim], window_size=wsz, window_minus=wmin, window_plus=wplus, dim=dim)
                want = want * (f[:, None] if dim == 0 else f[None, :])
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-300)


def test_projection_under_symmetry_equals_the_full_run():
    """Projection monitors together with ``Simulation.symmetry`` (ref monitor.py / monitor_data.py:238-284 symmetry
    expansion): the surfaces are recorded on their images in the computed quarter and expanded with the parity rules; a
    symmetric problem gives the same far field with and without the flag.  Also a windowed surface monitor."""
    DL = 0.0625
    f0 = 3e14
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 6)
    theta = np.linspace(0.05, 3.0, 9)
    phi = np.linspace(0, 2 * np.pi, 7)

    def sim(symmetry):
        return td.Simulation(
            size=(24 * DL, 20 * DL, 22 * DL), grid_spec=td.GridSpec.uniform(dl=DL), run_time=30 / f0, symmetry=symmetry,
            structures=[td.Structure(geometry=td.Sphere(radius=0.2), medium=td.Medium(permittivity=3.0))],
            sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ez")],
            monitors=[td.FieldProjectionAngleMonitor(center=(0, 0, 0), size=(0.83, 0.71, 0.77), freqs=[f0], theta=theta, phi=phi,
                                                     proj_distance=1e3, name="box"),
                      td.FieldProjectionAngleMonitor(center=(0, 0, 0.41), size=(td.inf, td.inf, 0), freqs=[f0], theta=theta[:4], phi=phi,
                                                     proj_distance=1e3, name="top", window_size=(0.4, 0.7)),
                      td.FieldProjectionCartesianMonitor(center=(0, 0, 0.41), size=(0.9, 0.8, 0), freqs=[f0], x=[-3.0, 0.5, 2.0], y=[0.0, 1.5],
                                                         proj_axis=2, proj_distance=50.0, name="cart", far_field_approx=False)],
            boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=6)), shutoff=0)

    def run(symmetry):
        disc = discretize(sim(symmetry), n_steps=260)
        return assemble(disc, OracleFdtd(disc.spec).run())
    full = run((0, 0, 0))
    half = run((1, 1, -1))        # E_z of a z dipole: even in x and y (PMC), normal to the z plane (PEC)
    for name in ("box", "top", "cart"):
        for comp, arr in full[name].field_components.items():
            a, b = np.asarray(arr.values), np.asarray(half[name].field_components[comp].values)
            scale = max(np.abs(np.asarray(v.values)).max() for v in full[name].field_components.values())
            assert np.abs(a - b).max() <= 2e-6 * scale, (name, comp, np.abs(a - b).max() / scale)
    assert np.abs(full["top"].Etheta.values).max() > 0
