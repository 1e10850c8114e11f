"""Randomised cross-check of the hot path's variants against each other and against the oracle: random small simulations —
walls of every kind per face (PEC, PMC on min AND plus faces, CPML / StablePML of random thickness, absorber layers, periodic
axes), uniform or varying cell sizes, dielectric / lossy / PEC / Lorentz / Drude / diagonally anisotropic bodies through the layers, electric and magnetic dipoles anywhere (next to
walls too), a plane wave across a periodic cell now and then, time / DFT / flux monitors, decay checks, runs cut in two —
    fused sweep  ==  two-pass kernels  ==  a z-slab rank exchanging with itself (periodic z)      bit for bit
    fused sweep  vs  the fp64 oracle (oracle/fdtd_numpy.py)                                      <= 2e-5
    python scripts/fuzz_variants.py [n_cases] [seed]
Also run by the suites (tests/test_fuzz_variants.py on the CPU emulator, tests/test_gpu_production_path.py on the device).
Found in round 4: the fused step of a grid with a PMC plus wall across a periodic z took the image cells of the wrapped ghost
planes from before their refresh; and, on the device only, a race of the two-pass z-slab schedule (a monitor record on the main
stream against the comm stream's update of the slab's top plane, tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tidy3d_amd.schema as td  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402
from tidy3d_amd.exceptions import SetupError, Tidy3dNotImplementedError  # noqa: E402

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)


def case(rng, big=False):
    N = [int(rng.integers(10, 40 if big else 22)) for _ in range(3)]
    if big and rng.integers(0, 3) == 0:
        N[0] = int(rng.choice([250, 258, 300]))          # a second x tile
    size = tuple(n * DL for n in N)
    periodic = [bool(rng.integers(0, 4) == 0) for _ in range(3)]
    kind = int(rng.integers(0, 3))                        # 0: CPML world, 1: absorber world, 2: walls only
    wave = bool(rng.integers(0, 6) == 0)                  # a plane wave along x across a cell periodic in y and z (CPML on x): the
    if wave:                                              # TFSF corrections and the 1-D incident grid, with its comm-stream replica on slab ranks
        periodic, kind = [False, True, True], 0
        N[0] = max(N[0], 16)
        size = tuple(n * DL for n in N)

    faces = []

    def face(a):
        r = int(rng.integers(0, 6))
        f = td.PECBoundary()
        if wave:
            f = td.PML(num_layers=int(rng.integers(3, 6)))
        elif kind == 0 and r < 3:
            n = int(rng.integers(2, 6))
            f = td.StablePML(num_layers=n) if r == 0 else td.PML(num_layers=n)
        elif kind == 1 and r < 3 and N[a] >= 12:
            f = td.Absorber(num_layers=int(rng.integers(2, 5)))
        elif r >= 3 and rng.integers(0, 2):
            f = td.PMCBoundary()
        faces.append(f)
        return f
    edges = []
    for a in range(3):
        if periodic[a]:
            edges.append(td.Boundary.periodic())
            faces.extend(["per", "per"])
        else:
            edges.append(td.Boundary(minus=face(a), plus=face(a)))
    bspec = td.BoundarySpec(x=edges[0], y=edges[1], z=edges[2])

    def pos(margin=1.2):
        return tuple(float(rng.uniform(-0.5 * s + margin * DL, 0.5 * s - margin * DL)) for s in size)
    has_absorber = any(isinstance(f, td.Absorber) for f in faces)
    srcs = []
    for _ in range(int(rng.integers(1, 4))):
        pols = ["Ex", "Ey", "Ez"] if has_absorber else ["Ex", "Ey", "Ez", "Hx", "Hy", "Hz"]
        srcs.append(td.PointDipole(center=pos(), source_time=PULSE, polarization=str(rng.choice(pols))))
    if wave:
        srcs = srcs[:1] + [td.PlaneWave(center=(0.5 * size[0] - float(rng.uniform(2.5, 4.5)) * DL, 0, 0), size=(0, td.inf, td.inf), source_time=PULSE,
                                        direction="-", pol_angle=float(rng.uniform(0, 1.5)))]
    structures = []
    if rng.integers(0, 3) > 0:
        meds = [td.Medium(permittivity=float(rng.uniform(1.5, 5)), conductivity=float(rng.choice([0, 0.02]))), td.PEC,
                td.AnisotropicMedium(xx=td.Medium(permittivity=2.0), yy=td.Medium(permittivity=3.5, conductivity=0.01), zz=td.PEC),
                td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)]), td.Drude(eps_inf=1.5, coeffs=[(6e14, 5e13)]),
                td.AnisotropicMedium(xx=td.Drude(eps_inf=1.5, coeffs=[(6e14, 5e13)]), yy=td.Medium(permittivity=2.2), zz=td.Medium(permittivity=4.0))]
        for _ in range(int(rng.integers(1, 4))):
            med = meds[int(rng.integers(0, 3 if has_absorber else 6))]
            if rng.integers(0, 2):
                geo = td.Box(center=pos(0.0), size=tuple(float(rng.uniform(0.15, 0.7) * s) for s in size))
            else:
                geo = td.Sphere(center=pos(0.0), radius=float(rng.uniform(0.1, 0.35) * min(size)))
            structures.append(td.Structure(geometry=geo, medium=med))
    mons = [td.FieldTimeMonitor(center=pos(2.0), size=(0, 0, 0), name="probe", interval=int(rng.integers(1, 5)), colocate=False)]
    if rng.integers(0, 2):
        sz = [td.inf, td.inf, td.inf]
        sz[int(rng.integers(0, 3))] = 0
        mons.append(td.FieldMonitor(center=pos(2.0), size=tuple(sz), freqs=[2.8e14, 3.1e14], name="plane"))
    if rng.integers(0, 3) == 0:
        mons.append(td.FieldTimeMonitor(center=pos(2.0), size=tuple(float(rng.uniform(0.1, 0.4) * s) for s in size), name="vol",
                                        interval=int(rng.integers(2, 9)), colocate=False))
    if rng.integers(0, 3) == 0:
        sz = [td.inf, td.inf, td.inf]
        sz[int(rng.integers(0, 3))] = 0
        mons.append(td.FluxMonitor(center=pos(2.0), size=tuple(sz), freqs=[3e14], name="flux"))
    grid = td.GridSpec.uniform(dl=DL)
    if rng.integers(0, 3) == 0:                           # a third of the cases: cell sizes that vary along every axis (0.7 ... 1.3 dl)
        def coords(n, s_):
            d = rng.uniform(0.7, 1.3, n)
            return tuple(np.concatenate(([0.0], np.cumsum(d))) * (s_ / d.sum()) - 0.5 * s_)
        grid = td.GridSpec(grid_x=td.CustomGridBoundaries(coords=coords(N[0], size[0])), grid_y=td.CustomGridBoundaries(coords=coords(N[1], size[1])),
                           grid_z=td.CustomGridBoundaries(coords=coords(N[2], size[2])))
    sim = td.Simulation(size=size, grid_spec=grid, run_time=1e-12, sources=srcs, monitors=mons,
                        structures=structures, boundary_spec=bspec, shutoff=0)
    steps = int(rng.integers(20, 60))
    disc = discretize(sim, n_steps=steps)
    disc.spec.decay_every = int(rng.choice([0, 0, 8, 16]))
    desc = f"N={disc.spec.shape} bc={[f if isinstance(f, str) else type(f).__name__[:4] for f in faces]} " \
           f"src={[getattr(s, 'polarization', 'wave') for s in srcs]} media={[type(s.medium).__name__[:4] for s in structures]} mon={[m.name for m in mons]} steps={steps} " \
           f"decay={disc.spec.decay_every}"
    return disc, steps, periodic[2], desc


def run(disc, steps, split, variant, lib, comm=False, rows=0, zc=0, opts=None):
    with HipEngine(disc.spec, lib=lib, variant=variant, axis_shift=0, force_comm=comm, z_chunk=zc) as e:
        if comm:
            e.comm_init(e.unique_id())
        if rows:
            e.set_option(L.OPT_ROWS, rows)
        for k, v in (opts or {}).items():
            e.set_option(k, v)
        for r in (split, steps - split):
            if r > 0:
                e.run(r)
        return [e.get_field(c) for c in range(6)], e.results()


def inside(disc):
    """index of the cells inside every PMC plus wall (what lies beyond is refreshed before it is read)"""
    sl = [slice(None)] * 3
    for a, w in enumerate(getattr(disc.spec, "mirror_plus", None) or ()):
        if w >= 0:
            sl[2 - a] = slice(0, w)
    return tuple(sl)


def draw(rng, big):
    """everything random about one case (scripts/repro_fuzz_variants.py replays the same draws)"""
    while True:
        try:
            disc, steps, per_z, desc = case(rng, big)
            break
        except (Tidy3dNotImplementedError, SetupError):       # a combination the front end refuses: draw again
            continue
    split = int(rng.integers(0, steps))
    rows, zc = int(rng.choice([0, 3, 4, 7])), int(rng.choice([0, 2, 5, 16]))
    so = {L.OPT_BND_PLANES: int(rng.choice([0, 1, 3])), L.OPT_PML_FUSED: int(rng.choice([-1, 7]))}
    po = {L.OPT_TWOSTEP: int(rng.integers(4, 17)) + 64 * int(rng.integers(2, 12))}
    return disc, steps, per_z, desc, split, rows, zc, so, po


def run_cases(n_cases, seed=1, lib=None, quiet=False, oracle=True, big=False):
    """-> (cases whose variants differ, cases beyond 2e-5 from the oracle, worst oracle error)"""
    from oracle.fdtd_numpy import OracleFdtd
    rng = np.random.default_rng(seed)
    bad = far = 0
    worst = 0.0
    for q in range(n_cases):
        disc, steps, per_z, desc, split, rows, zc, so, po = draw(rng, big)
        sl = inside(disc)
        ref_f, ref_m = run(disc, steps, split, L.VARIANT_FUSED, lib, rows=rows, zc=zc)
        outs = {"two_pass": run(disc, steps, split, L.VARIANT_ZMARCH, lib, zc=zc)}
        # the three-launch CPML step of large grids (edge tiles on the second stream beside the interior launch), forced
        outs["fused_split"] = run(disc, steps, split, L.VARIANT_FUSED, lib, rows=3, zc=zc, opts={L.OPT_PML_SPLIT: 1})
        if per_z and disc.spec.shape[2] >= 8:
            # a z-slab rank exchanging with itself: boundary chunks of random thickness, CPML as slab kernels or inside the sweeps,
            # step pairs on request (taken where the rank has nothing that keeps single steps)
            outs["fused_slab"] = run(disc, steps, split, L.VARIANT_FUSED, lib, comm=True, opts=so)
            outs["fused_slab_pairs"] = run(disc, steps, split, L.VARIANT_FUSED, lib, comm=True, opts=po)
            outs["two_pass_slab"] = run(disc, steps, split, L.VARIANT_ZMARCH, lib, comm=True)
        diff = [k for k, (f, m) in outs.items()
                if not (all(np.array_equal(a[sl], b[sl]) for a, b in zip(ref_f, f)) and all(np.array_equal(ref_m[n], m[n]) for n in ref_m))]
        err = 0.0
        amp = max(float(np.abs(f).max()) for f in ref_f)
        if oracle:
            o = OracleFdtd(disc.spec).run()
            # (records the wave has not reached yet hold fp32 denormals — 1e-42 of a field of 1e4, case 136 of seed 901: judged on
            #  the scale of the largest record, and no finer than 1e-9 of the field's amplitude)
            scale = max(max(np.linalg.norm(v) / np.sqrt(v.size) for v in o.values()), 1e-9 * amp)
            for k, v in o.items():
                den = max(np.linalg.norm(v), 0.5 * scale * np.sqrt(v.size), 1e-300)
                err = max(err, float(np.linalg.norm(np.asarray(ref_m[k]) - v) / den))
        ok = not diff and err < 2e-5 and np.isfinite(amp)
        if not quiet or not ok:
            print(f"case {q}: {desc} split={split} rows={rows} zc={zc} variants={['fused'] + list(outs)} max|F|={amp:.3g} oracle={err:.2e} -> "
                  f"{'ok' if ok else 'DIFFERS: ' + str(diff) if diff else 'ORACLE'}", flush=True)
        bad += bool(diff)
        far += err >= 2e-5
        worst = max(worst, err)
    return bad, far, worst


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    bad, far, worst = run_cases(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1, big=bool(int(os.environ.get("FUZZ_BIG", "0"))))
    print(f"fuzz_variants: {n_cases - bad} of {n_cases} cases bit-identical across the variants; {far} beyond 2e-5 from the oracle (worst {worst:.2e})")
    sys.exit(1 if (bad or far) else 0)


if __name__ == "__main__":
    main()
