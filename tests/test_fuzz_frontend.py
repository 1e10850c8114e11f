"""The front end against the LIVE reference on random simulations: a ``tidy3d.Simulation`` built from the reference's own classes
(oracle/tidy3d_ref_loader.py) with random size / centre, uniform, custom or AUTOMATIC grids, random bodies (boxes, spheres,
cylinders; dielectric, lossy, dispersive, PEC), walls (PML / StablePML / absorber of random thickness, periodic, PEC, PMC), symmetry,
monitors (field, flux, time with start / stop / interval) goes through the entry the product uses (``web._as_mirror`` ->
``Simulation.json()`` -> mirror) — cell counts, grid lines (1e-13), time step, number of steps, DFT stride, monitor index spans
and time indices must be the reference's.  Skipped where the reference checkout is absent (the GPU box)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")


@pytest.fixture(scope="module")
def td_ref():
    from oracle.tidy3d_ref_loader import load_tidy3d
    return load_tidy3d()


def _random_sim(td, rng):
    size = tuple(float(rng.uniform(1.0, 3.0)) for _ in range(3))
    center = tuple(float(rng.uniform(-0.5, 0.5)) for _ in range(3))
    f0 = float(rng.uniform(1.5e14, 3.5e14))
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / float(rng.uniform(5, 15)))

    def pos(frac=0.4):
        return tuple(float(c + rng.uniform(-frac, frac) * s) for c, s in zip(center, size))
    media = [td.Medium(permittivity=float(rng.uniform(1.5, 12))), td.Medium(permittivity=2.5, conductivity=0.02), td.PEC,
             td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)]), td.Drude(eps_inf=1.5, coeffs=[(6e14, 5e13)]),
             td.Sellmeier(coeffs=[(1.04, 0.006), (0.23, 0.02)])]
    structures = []
    for _ in range(int(rng.integers(0, 5))):
        kind = int(rng.integers(0, 3))
        if kind == 0:
            geo = td.Box(center=pos(), size=tuple(float(rng.uniform(0.1, 0.8) * s) for s in size))
        elif kind == 1:
            geo = td.Sphere(center=pos(), radius=float(rng.uniform(0.1, 0.3) * min(size)))
        else:
            geo = td.Cylinder(center=pos(), radius=float(rng.uniform(0.1, 0.3) * min(size)), length=float(rng.uniform(0.2, 0.9) * min(size)),
                              axis=int(rng.integers(0, 3)))
        structures.append(td.Structure(geometry=geo, medium=media[int(rng.integers(0, len(media)))]))
    gk = int(rng.integers(0, 4))
    if gk == 0:
        grid = td.GridSpec.uniform(dl=float(rng.uniform(0.03, 0.08)))
    elif gk in (1, 2):
        grid = td.GridSpec.auto(min_steps_per_wvl=float(rng.uniform(8, 16)), wavelength=float(td.C_0 / f0) if rng.integers(0, 2) else None,
                                max_scale=float(rng.choice([1.2, 1.4, 1.6])))
    else:
        grid = td.GridSpec(grid_x=td.UniformGrid(dl=float(rng.uniform(0.04, 0.08))), grid_y=td.AutoGrid(min_steps_per_wvl=float(rng.uniform(8, 14))),
                           grid_z=td.CustomGrid(dl=tuple(float(v) for v in rng.uniform(0.03, 0.07, int(rng.integers(12, 40))))),
                           wavelength=float(td.C_0 / f0))
    symmetry = [0, 0, 0]
    sym_on = rng.integers(0, 4) == 0

    def boundary(a):
        r = int(rng.integers(0, 6))
        if r == 0 and not (sym_on and a == 0):
            return td.Boundary.periodic()
        faces = []
        for _ in range(2):
            q = int(rng.integers(0, 6))
            faces.append(td.PML(num_layers=int(rng.integers(4, 16))) if q < 2 else td.StablePML(num_layers=int(rng.integers(10, 40))) if q == 2
                         else td.Absorber(num_layers=int(rng.integers(10, 40))) if q == 3 else td.PECBoundary() if q == 4 else td.PMCBoundary())
        return td.Boundary(minus=faces[0], plus=faces[1])
    bspec = td.BoundarySpec(x=boundary(0), y=boundary(1), z=boundary(2))
    if sym_on:
        symmetry[0] = int(rng.choice([-1, 1]))
    mons = [td.FieldMonitor(center=pos(0.3), size=tuple(float(rng.choice([0.0, rng.uniform(0.1, 0.6) * s])) for s in size), freqs=[f0, 1.1 * f0], name="f"),
            td.FluxMonitor(center=pos(0.3), size=(td.inf, td.inf, 0), freqs=[f0], name="flux"),
            td.FieldTimeMonitor(center=pos(0.3), size=(0, 0, 0), name="t", interval=int(rng.integers(1, 7)), start=float(rng.uniform(0, 2e-14)),
                                stop=float(rng.choice([0, 1]) * rng.uniform(3e-14, 8e-14)) or None),
            td.FieldTimeMonitor(center=pos(0.3), size=tuple(float(rng.uniform(0.1, 0.5) * s) for s in size), name="tv", interval=int(rng.integers(2, 9)))]
    return td.Simulation(size=size, center=center, grid_spec=grid, run_time=float(rng.uniform(5e-14, 2e-13)), medium=td.Medium(permittivity=float(rng.choice([1.0, 1.44 ** 2]))),
                         structures=structures, sources=[td.PointDipole(center=pos(0.2), source_time=pulse, polarization="Ey")], monitors=mons,
                         boundary_spec=bspec, symmetry=tuple(symmetry), courant=float(rng.uniform(0.7, 0.99)), shutoff=1e-5)


@pytest.mark.parametrize("seed", range(12))
def test_random_reference_simulations_discretise_as_the_reference_says(td_ref, seed):
    from tidy3d_amd import discretize as D
    from tidy3d_amd.web import _as_mirror
    rng = np.random.default_rng(1000 + seed)
    done = 0
    for _ in range(40):
        try:
            sim = _random_sim(td_ref, rng)
        except Exception:                                 # the reference's own validators refuse the draw (e.g. a monitor outside the domain)
            continue
        mirror, was_td = _as_mirror(sim)
        assert was_td
        b = D.make_boundaries(mirror)
        ref_b = [np.asarray(getattr(sim.grid.boundaries, d)) for d in "xyz"]
        assert [len(x) - 1 for x in b] == [len(x) - 1 for x in ref_b], (seed, done, sim.grid_spec)
        for a in range(3):
            np.testing.assert_allclose(b[a], ref_b[a], rtol=1e-12, atol=1e-12)
        dt = D.compute_dt(mirror, b)
        assert dt == pytest.approx(sim.dt, rel=1e-12)
        tmesh = D.make_tmesh(D.run_time(mirror), dt)
        assert len(tmesh) == sim.num_time_steps
        assert D.nyquist_step(mirror, dt) == sim.nyquist_step
        t = tmesh[:: max(1, len(tmesh) // 40)]
        np.testing.assert_allclose(mirror.sources[0].source_time.amp_time(t), sim.sources[0].source_time.amp_time(t), rtol=1e-12, atol=1e-300)
        assert list(D.frequency_range(mirror)) == pytest.approx(list(sim.frequency_range), rel=1e-12)
        for m_ref, m in zip(sim.monitors, mirror.monitors):
            assert D.discretize_inds_monitor(b, m).tolist() == np.asarray(sim._discretize_inds_monitor(m_ref)).tolist(), m.name
            if hasattr(m_ref, "time_inds"):
                assert list(m.time_inds(tmesh)) == list(m_ref.time_inds(sim.tmesh)), m.name
        if done % 5 == 0 and int(np.prod([len(x) - 1 for x in b])) < 400_000:
            # the staircase raster at the Ex / Ez Yee nodes: geometry.inside of the reference's own objects, later structures on top
            spec = D.discretize(mirror.copy(subpixel=False, monitors=()), n_steps=2).spec
            if any(mirror.symmetry) or any(w >= 0 for w in (getattr(spec, "mirror_plus", None) or ())):
                done += 1
                continue                                  # (the raster covers the computed half only / carries image cells beyond a PMC plus wall)
            for c in (0, 2):
                xs, ys, zs = spec.yee_coords(c)
                X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
                expect = np.ones(X.shape, int)
                for idx, st_ref in enumerate(sim.structures):
                    expect[st_ref.geometry.inside(X, Y, Z)] = idx + 2
                got = spec.mat_idx[c].transpose(2, 1, 0) if spec.mat_idx is not None else np.ones(X.shape, int)
                # table indices follow first use, PEC is entry 0: compare the partition of the nodes, medium by medium
                for idx, st_ref in enumerate(sim.structures):
                    sel = expect == idx + 2
                    if sel.any():
                        vals = np.unique(got[sel])
                        assert len(vals) == 1, (seed, done, idx)
                        med = spec.media[int(vals[0])]
                        assert med.pec == isinstance(st_ref.medium, td_ref.PECMedium), (seed, done, idx)
                        if not med.pec:
                            assert med.eps_inf == pytest.approx(float(np.real(st_ref.medium.eps_model(np.inf) if hasattr(st_ref.medium, "eps_model") else 1.0)), rel=1e-9)
                assert (got[expect == 1] == 1).all()
        done += 1
    assert done >= 10, done


def test_mirror_classes_cover_every_field_of_the_reference_classes(td_ref):
    """A JSON key the mirror does not know is ignored on parsing — so every field of the reference's pydantic classes must either be
    a field of the mirror class, or be without effect on the fields (plotting, adjoint bookkeeping), or make the object a placeholder
    that raises when used (``nonlinear_spec``, ``modulation_spec``).  A new reference field shows up here."""
    import dataclasses
    import tidy3d_amd.schema as td
    harmless = {"attrs", "type", "name", "frequency_range", "allow_gain", "viz_spec", "heat_spec", "plot_params", "version",
                "background_permittivity", "simulation_type"}
    loud = {"nonlinear_spec", "modulation_spec"}
    unknown = {}
    for name, cls in td._REGISTRY.items():
        ref = getattr(td_ref, name, None)
        if ref is None or not hasattr(ref, "__fields__"):
            continue
        mine = {f.name for f in dataclasses.fields(cls)}
        extra = [k for k in ref.__fields__ if k not in mine and k not in harmless and k not in loud]
        if extra:
            unknown[name] = extra
    assert unknown == {}, unknown
    m = td.parse({"type": "Lorentz", "eps_inf": 2.0, "coeffs": [[1.0, 3e14, 1e13]], "modulation_spec": {"type": "ModulationSpec"}})
    assert isinstance(m, td.Unsupported) and "modulation_spec" in m.type
