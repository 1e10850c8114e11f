"""Randomised self-check of round 6's step pairs on the device (or, with a library handed in, on the emulator): dispersive bodies of
any shape — through the CPML layers, on walls, across tile seams — advanced inside the pairs (FDTD_OPT_DISP), and big source lists —
TFSF boxes with and without polarisation / incidence angles, current sheets of any orientation through the layers, dipole crowds —
added as paged source terms while they inject (FDTD_OPT_SRC_PAGED); walls of every kind the pairs cover (PEC, PMC min faces,
absorber layers, CPML of random thickness, a periodic x), probes and DFT monitors, random tile shapes, runs cut in two.
Pairs == single steps, bit for bit, fields and records.
    python scripts/fuzz_round6.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tidy3d_amd.schema as td  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

DL = 0.05
MEDIA = [td.Lorentz(eps_inf=2.0, coeffs=[(2.0, 4e14, 2e13)]), td.Drude(eps_inf=1.5, coeffs=[(3e14, 1e13), (5e14, 3e13)]),
         td.Lorentz(eps_inf=1.8, coeffs=[(1.0, 3e14, 2e13), (0.7, 5e14, 4e13), (0.4, 7e14, 3e13)]), td.Debye(eps_inf=2.0, coeffs=[(1.0, 2e-15)])]


def case(rng):
    walls = str(rng.choice(["pec", "pmc", "abs", "cpml", "cpml", "cpml", "cpml_perx"]))
    nx = int(rng.choice([36, 60, 120, 260, 300, 516]))
    ny, nz = int(rng.integers(18, 44)), int(rng.integers(18, 44))
    if walls == "cpml_perx":
        nx = int(rng.choice([36, 100, 260, 300]))
    N = (nx, ny, nz)
    size = tuple((n - 1e-6) * DL for n in N)
    pulse = td.GaussianPulse(freq0=3e14, fwidth=float(rng.choice([1.5e14, 3e14])))

    def face(minus):
        if walls == "abs":
            return td.Absorber(num_layers=int(rng.integers(2, 6))) if rng.integers(0, 3) else td.PECBoundary()
        if walls.startswith("cpml"):
            return td.PML(num_layers=int(rng.integers(2, 7))) if rng.integers(0, 5) else td.PECBoundary()
        return td.PMCBoundary() if (walls == "pmc" and minus and rng.integers(0, 2)) else td.PECBoundary()
    bspec = td.BoundarySpec(**{ax: td.Boundary(minus=face(True), plus=face(False)) for ax in "xyz"})
    if walls == "cpml_perx":
        bspec = td.BoundarySpec(x=td.Boundary.periodic(), y=bspec.y, z=bspec.z)

    def pos(margin=1.0, frac=1.0):
        return tuple(float(rng.uniform(-0.5 * frac * s + min(margin * DL, 0.4 * s), 0.5 * frac * s - min(margin * DL, 0.4 * s))) for s in size)
    structures = []
    for _ in range(int(rng.integers(1, 4))):
        med = MEDIA[int(rng.integers(0, len(MEDIA)))] if rng.integers(0, 4) else td.Medium(permittivity=float(rng.uniform(1.5, 5)), conductivity=float(rng.choice([0, 0.03])))
        shape = int(rng.integers(0, 4))
        if shape == 0:
            geo = td.Sphere(center=pos(), radius=float(rng.uniform(0.08, 0.3) * min(size)))
        elif shape == 1:
            sz = [float(rng.uniform(0.1, 0.7) * s) for s in size]
            if rng.integers(0, 2):
                sz[int(rng.integers(0, 3))] = td.inf          # a bar / slab through the layers
            geo = td.Box(center=pos(), size=tuple(sz))
        elif shape == 2:
            geo = td.Cylinder(center=pos(), radius=float(rng.uniform(0.08, 0.25) * min(size)), length=float(rng.uniform(0.2, 1.2) * min(size)), axis=int(rng.integers(0, 3)))
        else:
            geo = td.Box(center=(0.5 * size[0] * float(rng.choice([-1, 1])), pos()[1], pos()[2]), size=(0.4, 0.4, 0.3))     # on an x face
        structures.append(td.Structure(geometry=geo, medium=med))
    srcs = []
    kind = int(rng.integers(0, 5))
    magnetic = walls != "abs"
    if kind == 0 and walls != "cpml_perx":        # a TFSF box
        bx = [float(rng.uniform(0.3, 0.55) * s) for s in size]
        kw = {}
        if rng.integers(0, 2):
            kw["pol_angle"] = float(rng.uniform(0, 1.5))
        if rng.integers(0, 3) == 0:
            kw.update(angle_theta=float(rng.uniform(0.1, 0.6)), angle_phi=float(rng.uniform(0, 1.5)))
        if magnetic:
            srcs.append(td.TFSF(center=pos(0, 0.2), size=tuple(bx), source_time=pulse, injection_axis=int(rng.integers(0, 3)), direction=str(rng.choice(["+", "-"])), **kw))
    elif kind == 1:                               # a current sheet through the layers (what a mode plane is to the engine)
        ax = int(rng.integers(0, 3))
        sz = [td.inf, td.inf, td.inf]
        sz[ax] = 0
        pols = ["Ex", "Ey", "Ez"] + (["Hx", "Hy", "Hz"] if magnetic else [])
        for pol in rng.choice(pols, size=int(rng.integers(1, 3)), replace=False):
            c = list(pos(3, 0.6))
            srcs.append(td.UniformCurrentSource(center=tuple(c), size=tuple(sz), source_time=pulse, polarization=str(pol)))
    elif kind == 2:                               # a crowd of dipoles (more nodes than the sweep's node table takes)
        for _ in range(int(rng.integers(30, 50))):
            srcs.append(td.PointDipole(center=pos(2.0), source_time=pulse, polarization=str(rng.choice(["Ex", "Ey", "Ez"]))))
    for _ in range(int(rng.integers(0 if srcs else 1, 3))):
        pol = str(rng.choice(["Ex", "Ey", "Ez"] + (["Hx", "Hy", "Hz"] if magnetic else [])))
        srcs.append(td.PointDipole(center=pos(2.0), source_time=td.GaussianPulse(freq0=3e14, fwidth=float(rng.choice([1.5e14, 4e14]))), polarization=pol))
    mons = []
    for q in range(int(rng.integers(0, 3))):
        fields = [str(f) for f in rng.choice(["Ex", "Ey", "Ez", "Hx", "Hy", "Hz"], size=int(rng.integers(1, 4)), replace=False)]
        mons.append(td.FieldTimeMonitor(center=pos(2.5), size=(0, 0, 0), name=f"m{q}", interval=int(rng.integers(1, 4)), fields=fields, colocate=False))
    for q in range(int(rng.integers(0, 3))):
        ax = int(rng.integers(0, 3))
        sz = [td.inf, td.inf, td.inf]
        sz[ax] = 0
        if rng.integers(0, 2):
            mons.append(td.FluxMonitor(center=pos(2.5), size=tuple(sz), freqs=[3e14, 3.2e14], name=f"fl{q}"))
        else:
            mons.append(td.FieldMonitor(center=pos(2.5), size=tuple(sz), freqs=[3e14], name=f"d{q}", colocate=False))
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, sources=srcs, monitors=mons, structures=structures,
                        boundary_spec=bspec, shutoff=0, subpixel=bool(rng.integers(0, 2)))
    steps = int(rng.integers(12, 44))
    disc = discretize(sim, n_steps=steps + 1)
    disc.spec.decay_every = int(rng.choice([0, 0, 8, 16]))
    return disc, steps, int(rng.integers(4, 17)), int(rng.integers(2, 40)), f"{walls} sources={kind}"


def run(disc, steps, twostep, split, lib=None, seed=0, disp=-1, paged=-1):
    with HipEngine(disc.spec, lib=lib, variant=L.VARIANT_FUSED, axis_shift=0) as e:
        e.set_option(L.OPT_TWOSTEP, twostep)
        e.set_option(L.OPT_SHELL_PAIRS, 1)
        e.set_option(L.OPT_SHELL2, 1)
        if disp >= 0:
            e.set_option(L.OPT_DISP, disp)
        if paged >= 0:
            e.set_option(L.OPT_SRC_PAGED, paged)
        rng = np.random.default_rng(seed)
        for c in range(6):
            f = e.get_field(c)
            e.set_field(c, ((1e-3 if c < 3 else 1e-3 / 376.73) * rng.uniform(-1, 1, size=f.shape)).astype(np.float32))
        tot = [0, 0, 0]
        for r in (split, steps - split):
            if r > 0:
                st = e.run(r)
                tot[0] += int(st.fused2_pairs); tot[1] += int(st.disp_pairs); tot[2] += int(st.src_paged_pairs)
        return [e.get_field(c) for c in range(6)], e.results(), tot


def run_cases(n_cases, seed=1, lib=None, quiet=False):
    """-> (cases that differ, cases whose pairs advanced dispersive cells, cases whose pairs carried paged source terms)"""
    rng = np.random.default_rng(seed)
    bad = n_disp = n_paged = 0
    for q in range(n_cases):
        disc, steps, w, zc, what = case(rng)
        split = int(rng.integers(0, steps))
        ref = run(disc, steps, 0, split, lib, seed=q)
        got = run(disc, steps, w + 64 * zc, split, lib, seed=q)
        ok = ref[2][0] == 0 and all(np.array_equal(a, b) for a, b in zip(ref[0], got[0])) and all(np.array_equal(ref[1][k], got[1][k]) for k in ref[1])
        if q % 4 == 0:          # the round-5 schedules are what they were
            old = run(disc, steps, w + 64 * zc, split, lib, seed=q, disp=0, paged=0)
            ok = ok and old[2][1] == 0 and old[2][2] == 0 and all(np.array_equal(a, b) for a, b in zip(ref[0], old[0]))
        if not quiet or not ok:
            print(f"case {q}: N={disc.spec.shape} {what} steps={steps} split={split} W={w} zc={zc} media={len(disc.spec.media)} monitors={len(ref[1])} "
                  f"pairs={got[2][0]} disp={got[2][1]} paged={got[2][2]} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
        n_disp += got[2][1] > 0
        n_paged += got[2][2] > 0
    return bad, n_disp, n_paged


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    bad, n_disp, n_paged = run_cases(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("fuzz:", n_cases - bad, "of", n_cases, "cases bit-identical;", n_disp, "advanced dispersive cells inside pairs,", n_paged, "carried paged source terms")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
