"""Lorentz reciprocity, discretely: on the Yee grid with diagonal media the field of component b at node B driven by a unit
current of component a at node A equals the field of component a at A driven by the same current of component b at B — at
every time step, through CPML, walls of either kind, periodic axes, graded cells, lossy and dispersive bodies.  The fp64 oracle holds
it to 1e-12 (a pin of the oracle that needs no reference data: a wrong index, stagger, coefficient placement or an
asymmetric CPML term breaks it); the HIP path to fp32 rounding — on the emulator here, on the device through step pairs
(tests/test_gpu_shell_pairs.py)."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd.discretize import discretize

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)
NAMES = ["Ex", "Ey", "Ez"]


def reciprocity_sims(N, bspec, structures, A, ca, B, cb, n_steps, graded=False, seed=0):
    """-> ((disc of run 1, index of the probe node in its record), (same for run 2))"""
    size = tuple(n * DL for n in N)
    grid = td.GridSpec.uniform(dl=DL)
    if graded:
        rng = np.random.default_rng(seed)

        def coords(n, s_):
            d = rng.uniform(0.75, 1.25, n)
            return tuple(np.concatenate(([0.0], np.cumsum(d))) * (s_ / d.sum()) - 0.5 * s_)
        grid = td.GridSpec(grid_x=td.CustomGridBoundaries(coords=coords(N[0], size[0])), grid_y=td.CustomGridBoundaries(coords=coords(N[1], size[1])),
                           grid_z=td.CustomGridBoundaries(coords=coords(N[2], size[2])))

    def base(srcs, mons):
        return td.Simulation(size=size, grid_spec=grid, run_time=1e-12, structures=structures, sources=srcs, monitors=mons, boundary_spec=bspec,
                             shutoff=0, subpixel=False)
    spec0 = discretize(base([td.PointDipole(center=(0, 0, 0), source_time=PULSE, polarization="Ex")], []), n_steps=2).spec

    def node(comp, ijk):
        xs, ys, zs = spec0.yee_coords(comp)
        return (float(xs[ijk[0]]), float(ys[ijk[1]]), float(zs[ijk[2]]))
    out = []
    for (src, sc), (mon, mc) in (((A, ca), (B, cb)), ((B, cb), (A, ca))):
        sim = base([td.PointDipole(center=node(sc, src), source_time=PULSE, polarization=NAMES[sc])],
                   [td.FieldTimeMonitor(center=node(mc, mon), size=(0, 0, 0), name="p", fields=[NAMES[mc]], colocate=False)])
        disc = discretize(sim, n_steps=n_steps)
        assert disc.spec.sources[0].ijk.tolist() == [list(src)], "the dipole must sit on one Yee node"
        m = disc.spec.monitors[0]
        out.append((disc, (mon[2] - m.lo[2], mon[1] - m.lo[1], mon[0] - m.lo[0])))
    return out


def series(raw, at):
    a = np.asarray(raw["p"])
    return a[:, 0, at[0], at[1], at[2]]


CONFIGS = {
    "cpml_pec_periodic_media": dict(
        N=(22, 18, 20), A=(8, 6, 7), ca=0, B=(15, 11, 13), cb=1,
        bspec=td.BoundarySpec(x=td.Boundary.pml(num_layers=5), y=td.Boundary(minus=td.PECBoundary(), plus=td.PML(num_layers=4)), z=td.Boundary.periodic()),
        structures=[td.Structure(geometry=td.Box(center=(0.1, 0, -0.1), size=(0.4, 0.5, 0.3)), medium=td.Medium(permittivity=4.0, conductivity=0.02)),
                    td.Structure(geometry=td.Sphere(center=(-0.2, 0.1, 0.15), radius=0.2), medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)]))]),
    "graded_stable_pml_pmc_anisotropic": dict(
        N=(20, 20, 18), A=(5, 12, 6), ca=2, B=(13, 7, 11), cb=0, graded=True,
        bspec=td.BoundarySpec(x=td.Boundary(minus=td.PMCBoundary(), plus=td.StablePML(num_layers=6)), y=td.Boundary.pml(num_layers=4),
                              z=td.Boundary(minus=td.PML(num_layers=3), plus=td.PECBoundary())),
        structures=[td.Structure(geometry=td.Cylinder(center=(0, 0, 0), radius=0.3, length=td.inf, axis=1),
                                 medium=td.AnisotropicMedium(xx=td.Medium(permittivity=2.0), yy=td.Medium(permittivity=3.5, conductivity=0.01), zz=td.Drude(eps_inf=1.5, coeffs=[(6e14, 5e13)]))),
                    td.Structure(geometry=td.Box(center=(0.2, 0.2, 0.1), size=(0.2, 0.2, 0.2)), medium=td.PEC)]),
    "absorber_layers": dict(
        N=(18, 18, 18), A=(6, 6, 6), ca=1, B=(11, 12, 10), cb=2,
        bspec=td.BoundarySpec.all_sides(td.Absorber(num_layers=5)),
        structures=[td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=0.25), medium=td.Medium(permittivity=6.0))]),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_is_reciprocal_to_rounding(name):
    from oracle.fdtd_numpy import OracleFdtd
    cfg = dict(CONFIGS[name])
    (d1, at1), (d2, at2) = reciprocity_sims(n_steps=260, **cfg)
    e1, e2 = series(OracleFdtd(d1.spec).run(), at1), series(OracleFdtd(d2.spec).run(), at2)
    assert np.abs(e1).max() > 0
    assert np.abs(e1 - e2).max() < 1e-11 * np.abs(e1).max(), float(np.abs(e1 - e2).max() / np.abs(e1).max())


@pytest.mark.parametrize("name,variant", [("cpml_pec_periodic_media", "fused"), ("graded_stable_pml_pmc_anisotropic", "two_pass")])
def test_hip_kernels_are_reciprocal_on_the_emulator(name, variant, emu_lib):
    from tidy3d_amd import lib as L
    from tidy3d_amd.engine import HipEngine
    cfg = dict(CONFIGS[name])
    (d1, at1), (d2, at2) = reciprocity_sims(n_steps=160, **cfg)
    out = []
    for d, at in ((d1, at1), (d2, at2)):
        with HipEngine(d.spec, lib=emu_lib, variant=L.VARIANT_FUSED if variant == "fused" else L.VARIANT_ZMARCH, axis_shift=0) as e:
            e.run()
            out.append(series(e.results(), at))
    assert np.abs(out[0]).max() > 0
    assert np.abs(out[0] - out[1]).max() < 2e-5 * np.abs(out[0]).max(), float(np.abs(out[0] - out[1]).max() / np.abs(out[0]).max())
