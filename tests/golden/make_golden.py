"""Generate tests/golden/*.json by RUNNING THE REFERENCE'S OWN CODE (imported from
/root/reference through oracle/tidy3d_ref_loader.py).  Run once in the build container:

    python tests/golden/make_golden.py

The fixtures are committed; tests/test_golden_schema.py and tests/test_golden_modes.py read them
everywhere (the GPU box has no /root/reference).  Nothing here is solver logic: it only
constructs reference objects and records what the reference computes for them.
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle.tidy3d_ref_loader import load_mode_solver, load_tidy3d  # noqa: E402

td = load_tidy3d()


def c2l(z):
    z = complex(z)
    return [z.real, z.imag]


def simulations():
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
    pulse2 = td.GaussianPulse(freq0=3e14, fwidth=6e13, offset=4.0, phase=0.3, amplitude=2.0)
    sims = {}
    sims["uniform_pml"] = td.Simulation(
        size=(4, 3, 2), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=1e-13,
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(1, 1, 1)),
                                 medium=td.Medium(permittivity=4.0))],
        sources=[td.PointDipole(center=(0.3, 0, 0), source_time=pulse, polarization="Ez")],
        monitors=[td.FieldMonitor(center=(0, 0, 0), size=(2, 2, 0), freqs=[2e14], name="f"),
                  td.FluxMonitor(center=(0, 0, 0.5), size=(2, 2, 0), freqs=[2e14, 2.2e14], name="fl"),
                  td.FieldTimeMonitor(center=(0.1, 0, 0), size=(0, 0, 0), name="t", interval=3),
                  td.FieldMonitor(center=(0.2, 0.1, 0), size=(1.03, 0.77, 0.4), freqs=[1.9e14],
                                  name="nc", colocate=False),
                  td.FluxMonitor(center=(0, 0, 0), size=(1.5, 1.2, 0.8), freqs=[2e14], name="box")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML()))
    sims["mixed_bc_dispersive"] = td.Simulation(
        size=(2.03, 1.51, 1.0), center=(0.1, -0.2, 0.3),
        grid_spec=td.GridSpec(grid_x=td.UniformGrid(dl=0.04), grid_y=td.UniformGrid(dl=0.05),
                              grid_z=td.CustomGrid(dl=tuple(0.03 + 0.002 * np.arange(30)))),
        run_time=2.5e-13, courant=0.9,
        medium=td.Medium(permittivity=1.5),
        structures=[td.Structure(geometry=td.Sphere(center=(0.1, -0.2, 0.3), radius=0.3),
                                 medium=td.Lorentz(eps_inf=0.16, coeffs=[(1.5, 3e14, 1e13)])),
                    td.Structure(geometry=td.Box(center=(0.5, 0, 0.3), size=(0.2, 0.2, 0.2)),
                                 medium=td.PEC)],
        sources=[td.PointDipole(center=(0.1, -0.2, 0.3), source_time=pulse2, polarization="Ex"),
                 td.UniformCurrentSource(center=(0.3, -0.2, 0.3), size=(0.2, 0, 0.1),
                                         source_time=pulse, polarization="Hy")],
        monitors=[td.FieldTimeMonitor(center=(0.1, -0.2, 0.3), size=(0.5, 0.5, 0), name="tm",
                                      start=2e-14, stop=9e-14, interval=5),
                  td.FieldMonitor(center=(0.1, -0.2, 0.3), size=(td.inf, 0, td.inf), freqs=[3.3e14, 2.5e14],
                                  name="xz")],
        boundary_spec=td.BoundarySpec(x=td.Boundary.pml(num_layers=7), y=td.Boundary.periodic(),
                                      z=td.Boundary(minus=td.PECBoundary(), plus=td.StablePML(num_layers=9))),
        subpixel=True)
    sims["vacuum_pec_200"] = td.Simulation(
        size=(10, 10, 10), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=1e-12,
        sources=[td.PointDipole(center=(1.3, -0.7, 2.1), source_time=td.GaussianPulse(freq0=3.5e13, fwidth=1.2e13),
                                polarization="Ez")],
        monitors=[td.FieldTimeMonitor(center=(-2.1, 1.2, -0.6), size=(0, 0, 0), name="t", fields=["Ez"],
                                      colocate=False, interval=1)],
        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    sims["runtime_spec"] = td.Simulation(
        size=(3, 2, 2), grid_spec=td.GridSpec.uniform(dl=0.1), run_time=td.RunTimeSpec(quality_factor=3.0),
        structures=[td.Structure(geometry=td.Box(size=(1, 1, 1)), medium=td.Medium(permittivity=9.0))],
        sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ey")],
        monitors=[td.FluxTimeMonitor(center=(1, 0, 0), size=(0, 1, 1), name="ft", interval=2)],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=6)), subpixel=False)
    return sims


def record_sim(sim):
    exp = {
        "num_cells": [int(n) for n in sim.grid.num_cells],
        "boundaries": {d: np.asarray(getattr(sim.grid.boundaries, d)).tolist() for d in "xyz"},
        "dt": float(sim.dt),
        "num_time_steps": int(sim.num_time_steps),
        "tmesh_last": float(sim.tmesh[-1]),
        "run_time": float(sim._run_time),
        "num_pml_layers": [[int(a), int(b)] for a, b in sim.num_pml_layers],
        "nyquist_step": int(sim.nyquist_step),
        "frequency_range": [float(x) for x in sim.frequency_range],
        "monitors": {},
        "mediums_n_cfl": sorted(float(m.n_cfl) for m in sim.scene.mediums),
    }
    for m in sim.monitors:
        rec = {"span": np.asarray(sim._discretize_inds_monitor(m)).tolist()}
        if hasattr(m, "time_inds"):
            rec["time_inds"] = [int(x) for x in m.time_inds(sim.tmesh)]
            rec["num_steps"] = int(m.num_steps(sim.tmesh))
        exp["monitors"][m.name] = rec
    # inside() of every structure geometry at the Ex Yee points of a coarse probe lattice
    return {"json": json.loads(sim.json()), "expected": exp}


def source_times():
    t = np.linspace(0, 2e-13, 41)
    out = []
    for st in [td.GaussianPulse(freq0=2e14, fwidth=2e13),
               td.GaussianPulse(freq0=3e14, fwidth=6e13, offset=4.0, phase=0.3, amplitude=2.0),
               td.GaussianPulse(freq0=1e14, fwidth=3e13, remove_dc_component=False),
               td.ContinuousWave(freq0=2e14, fwidth=4e13, phase=-0.7)]:
        amp = st.amp_time(t)
        tm = np.arange(0, 3e-13, 2.1e-16)
        freqs = np.array([0.8, 1.0, 1.2]) * st.freq0
        sp = st.spectrum(tm, freqs, 2.1e-16)
        out.append({"json": json.loads(st.json()), "t": t.tolist(), "amp": [c2l(a) for a in amp],
                    "end_time": st.end_time(), "frequency_range": list(st.frequency_range()),
                    "spec_dt": 2.1e-16, "spec_n": len(tm), "spec_freqs": freqs.tolist(),
                    "spectrum": [c2l(s) for s in sp]})
    return out


def media():
    freqs = np.linspace(1e14, 6e14, 6)
    out = []
    for med in [td.Medium(permittivity=2.5, conductivity=0.03),
                td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 3e14, 1e13), (0.5, 5e14, 2e13)]),
                td.Lorentz(eps_inf=1.2, coeffs=[(0.8, 2e14, 5e14)]),
                td.Drude(eps_inf=1.5, coeffs=[(1.2e15, 8e13)]),
                td.Sellmeier(coeffs=[(1.03961212, 0.00600069867), (0.231792344, 0.0200179144)]),
                td.Debye(eps_inf=2.0, coeffs=[(1.0, 2e-15), (0.4, 7e-16)]),
                td.PoleResidue(eps_inf=1.1, poles=[((-1e13 - 2e15j), (1e13 + 3e15j)), ((-5e14 + 0j), (2e14 + 0j))]),
                td.material_library["Au"]["JohnsonChristy1972"]]:
        pr = med.pole_residue if hasattr(med, "pole_residue") else None
        rec = {"json": json.loads(med.json()), "freqs": freqs.tolist(),
               "eps_model": [c2l(e) for e in np.atleast_1d(med.eps_model(freqs))],
               "n_cfl": float(med.n_cfl)}
        if pr is not None:
            rec["eps_inf"] = float(pr.eps_inf)
            rec["poles"] = [[c2l(a), c2l(c)] for a, c in pr.poles]
        out.append(rec)
    return out


def geometry_inside():
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (400, 3))
    # include points exactly on faces
    pts[:20, 0] = 0.5
    pts[20:40, 1] = -0.3
    out = []
    for g in [td.Box(center=(0.1, 0, -0.1), size=(0.8, 0.6, 1.0)),
              td.Sphere(center=(0.1, 0.2, 0), radius=0.55),
              td.Cylinder(center=(0, 0.1, 0), radius=0.4, length=0.9, axis=1),
              td.Cylinder(center=(-0.2, 0, 0.1), radius=0.3, length=1.1, axis=2)]:
        ins = g.inside(pts[:, 0], pts[:, 1], pts[:, 2])
        out.append({"json": json.loads(g.json()), "points": pts.tolist(),
                    "inside": [bool(b) for b in ins], "bounds": [list(map(float, b)) for b in g.bounds]})
    return out


def apodization():
    t = np.linspace(0, 1e-12, 51)
    out = []
    for spec in [dict(start=2e-13, width=5e-14), dict(end=8e-13, width=1e-13),
                 dict(start=1e-13, end=7e-13, width=6e-14)]:
        # ApodizationSpec.plot's arithmetic (ref apodization.py:87-94) evaluated directly
        amp = np.ones_like(t)
        if spec.get("start") is not None:
            m = t < spec["start"]
            amp[m] *= np.exp(-0.5 * ((t[m] - spec["start"]) / spec["width"]) ** 2)
        if spec.get("end") is not None:
            m = t > spec["end"]
            amp[m] *= np.exp(-0.5 * ((t[m] - spec["end"]) / spec["width"]) ** 2)
        out.append({"spec": spec, "t": t.tolist(), "window": amp.tolist()})
    return out


def mode_solver_cases():
    """Reference EigSolver.compute_modes on the BASELINE config-1 strip (450 x 220 nm Si in SiO2,
    lambda = 1.55 um), staircased, PEC outer walls; plus one case with PML and one TM-like."""
    _, solver = load_mode_solver()
    from tidy3d.constants import C_0
    out = []
    for (dl, W, H, n_core, n_clad, npml, nm) in [(0.04, 0.45, 0.22, 3.48, 1.44, (0, 0), 2),
                                                  (0.05, 0.5, 0.22, 3.48, 1.45, (6, 6), 2)]:
        Lx, Ly = 3.0, 2.5
        nx, ny = int(round(Lx / dl)), int(round(Ly / dl))
        xb = -Lx / 2 + dl * np.arange(nx + 1)
        yb = -Ly / 2 + dl * np.arange(ny + 1)
        xc, yc = (xb[1:] + xb[:-1]) / 2, (yb[1:] + yb[:-1]) / 2
        xm, ym = xb[:-1], yb[:-1]

        def eps_at(x, y):
            X, Y = np.meshgrid(x, y, indexing="ij")
            core = (np.abs(X) <= W / 2) & (np.abs(Y) <= H / 2)
            return np.where(core, n_core ** 2, n_clad ** 2).astype(complex)
        exx, eyy, ezz = eps_at(xc, ym), eps_at(xm, yc), eps_at(xm, ym)
        z = np.zeros_like(exx)
        ms = SimpleNamespace(num_modes=nm, bend_radius=None, bend_axis=None, angle_theta=0.0,
                             angle_phi=0.0, num_pml=npml, target_neff=None, precision="double")
        fields, n_complex, _ = solver.compute_modes(
            eps_cross=[exx, z, z, z, eyy, z, z, z, ezz], coords=[xb, yb], freq=C_0 / 1.55,
            mode_spec=ms, symmetry=(0, 0), direction="+")
        E, Hf = fields[0], fields[1]          # (3, Nx, Ny, 1, M)
        rec = {"dl": dl, "W": W, "H": H, "n_core": n_core, "n_clad": n_clad, "Lx": Lx, "Ly": Ly,
               "num_pml": list(npml), "num_modes": nm, "wavelength": 1.55,
               "n_complex": [c2l(n) for n in n_complex],
               # dominant-component profiles through the core centre (gauge-free: |.| normalised)
               "abs_Ex_row": (np.abs(E[0, :, ny // 2, 0, 0]) / np.abs(E[0, :, ny // 2, 0, 0]).max()).tolist(),
               "abs_Hy_row": (np.abs(Hf[1, :, ny // 2, 0, 0]) / np.abs(Hf[1, :, ny // 2, 0, 0]).max()).tolist(),
               "abs_Ey_row_m1": (np.abs(E[1, :, ny // 2, 0, 1]) / max(np.abs(E[1, :, ny // 2, 0, 1]).max(), 1e-300)).tolist()}
        out.append(rec)
    return out


def autogrid_cases():
    """Simulations meshed by the reference's OWN GradedMesher (components/grid/mesher.py; its three
    third-party calls — a 2-D box, an R-tree query, a bracketed root find — are stood in for by
    oracle/tidy3d_ref_loader.py).  Stored: the simulation in the reference's JSON form and the cell
    boundaries ``sim.grid.boundaries`` the reference produces."""
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)

    def base(**kw):
        d = dict(size=(4, 3, 2), run_time=1e-13, grid_spec=td.GridSpec.auto(min_steps_per_wvl=12),
                 structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(1, 0.5, 0.22)),
                                          medium=td.Medium(permittivity=12.0)),
                             td.Structure(geometry=td.Sphere(center=(1.2, 0.3, 0.1), radius=0.3),
                                          medium=td.Medium(permittivity=4.0))],
                 sources=[td.PointDipole(center=(0.3, 0, 0), source_time=pulse, polarization="Ez")],
                 boundary_spec=td.BoundarySpec.all_sides(td.PML()))
        d.update(kw)
        return td.Simulation(**d)

    rng = np.random.default_rng(11)

    def random_boxes(n):
        out = []
        for _ in range(n):
            c = rng.uniform(-1.2, 1.2, 3) * (1.0, 0.8, 0.5)
            sz = rng.uniform(0.02, 1.5, 3) * (1.0, 0.8, 0.5)
            med = [td.Medium(permittivity=float(rng.uniform(1.5, 13))), td.PEC,
                   td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)]),
                   td.Medium(permittivity=2.0, conductivity=float(rng.uniform(0, 2)))][int(rng.integers(0, 4))]
            geo = [td.Box(center=tuple(c), size=tuple(sz)), td.Sphere(center=tuple(c), radius=float(sz[0] / 2)),
                   td.Cylinder(center=tuple(c), radius=float(sz[1] / 2), length=float(sz[2]), axis=2)][int(rng.integers(0, 3))]
            out.append(td.Structure(geometry=geo, medium=med))
        return out

    sims = {
        "default_grid_spec": base(grid_spec=td.GridSpec()),
        "base": base(),
        "fine_gentle": base(grid_spec=td.GridSpec.auto(min_steps_per_wvl=20, max_scale=1.2)),
        "symmetry": base(symmetry=(1, -1, 0)),
        "periodic": base(boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(),
                                                       z=td.Boundary.pml())),
        "mixed_axes": base(grid_spec=td.GridSpec(grid_x=td.AutoGrid(min_steps_per_wvl=15), grid_y=td.UniformGrid(dl=0.05),
                                                 grid_z=td.AutoGrid(), wavelength=1.3)),
        "waveguide_on_substrate": base(medium=td.Medium(permittivity=2.1), structures=[
            td.Structure(geometry=td.Box(center=(0, 0, -0.5), size=(td.inf, td.inf, 1.0)), medium=td.Medium(permittivity=2.1)),
            td.Structure(geometry=td.Box(center=(0, 0, 0.11), size=(td.inf, 0.45, 0.22)), medium=td.Medium(permittivity=12.1)),
            td.Structure(geometry=td.Cylinder(center=(0.7, 0.2, 0.11), radius=0.25, length=0.22, axis=2),
                         medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])),
            td.Structure(geometry=td.Box(center=(-1.0, 0.5, 0.05), size=(0.3, 0.02, 0.1)), medium=td.PEC)]),
        "two_d": base(size=(4, 0, 2), boundary_spec=td.BoundarySpec(x=td.Boundary.pml(), y=td.Boundary.periodic(),
                                                                    z=td.Boundary.pml())),
        "overrides_snapping_dlmin": base(grid_spec=td.GridSpec.auto(min_steps_per_wvl=12, override_structures=[
            td.MeshOverrideStructure(geometry=td.Box(center=(0, 0, 0), size=(1.4, 0.8, 0.5)), dl=(0.02, None, 0.015)),
            td.MeshOverrideStructure(geometry=td.Box(center=(1.2, 0.3, 0.1), size=(0.5, 0.5, 0.5)), dl=(0.04, 0.04, 0.04),
                                     enforce=True)],
            snapping_points=[(0.33, 0.41, -0.27), (1.9, 0, 0)], dl_min=0.012)),
        "empty_domain": base(structures=[]),
    }
    for k in range(6):
        sims[f"random_{k}"] = base(structures=random_boxes(3 + 2 * k),
                                   grid_spec=td.GridSpec.auto(min_steps_per_wvl=float(rng.uniform(8, 18)),
                                                              max_scale=float(rng.uniform(1.2, 1.6))))
    out = {}
    for name, sim in sims.items():
        out[name] = {"simulation": json.loads(sim.json()),
                     "boundaries": {d: np.asarray(getattr(sim.grid.boundaries, d)).tolist() for d in "xyz"}}
    return out


if __name__ == "__main__":
    gold = {"_generator": "tests/golden/make_golden.py (reference tidy3d v%s, numpy-1 fp_eps semantics)" % td.__version__,
            "simulations": {k: record_sim(s) for k, s in simulations().items()},
            "source_times": source_times(), "media": media(), "geometry": geometry_inside(),
            "apodization": apodization()}
    with open(os.path.join(HERE, "schema_golden.json"), "w") as f:
        json.dump(gold, f)
    with open(os.path.join(HERE, "mode_golden.json"), "w") as f:
        json.dump({"_generator": gold["_generator"], "cases": mode_solver_cases()}, f)
    with open(os.path.join(HERE, "autogrid_golden.json"), "w") as f:
        json.dump({"_generator": gold["_generator"], "cases": autogrid_cases()}, f)
    print("wrote", os.path.getsize(os.path.join(HERE, "schema_golden.json")), "+",
          os.path.getsize(os.path.join(HERE, "mode_golden.json")), "bytes")
