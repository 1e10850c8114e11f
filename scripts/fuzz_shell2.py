#!/usr/bin/env python
"""Randomised check of shell2 pairs (fdtd_shell2.hpp: the CPML shell advanced two steps per sweep with psi carried) against single
steps, bit for bit: grids of 1 - 3 x tiles of the bulk, walls of every kind per face (CPML / StablePML of random thickness, PEC,
PMC on min faces), dielectric / lossy / PEC bodies through the layers, random initial fields (the layers work from the first
step), dipoles of both kinds deep inside the bulk (applied by the bulk sweep) or none, probes / time monitors / DFT planes inside
the bulk, runs cut in two, the form of the launches (one launch, one per instantiation, one per box) and the tile shapes of the
boxes (lanes per row, waves per workgroup, planes per chunk).
  python scripts/fuzz_shell2.py [cases] [seed] [periodic]     (tests/test_emu_shell2.py runs a dozen on the emulator, the GPU suite 40;
  `periodic`: x and / or y periodic — the boxes wrap x through halo lanes, the rows next to a y wrap take single steps)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tidy3d_amd.schema as td  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)


def case(rng, small=False, periodic=False):
    nx = int(rng.choice([int(rng.integers(40, 120)), int(rng.integers(250, 330)), int(rng.integers(500, 560))] if not small else [int(rng.integers(40, 90)), int(rng.integers(250, 280))]))
    ny, nz = int(rng.integers(18, 40 if small else 56)), int(rng.integers(18, 36 if small else 60))
    N = (nx, ny, nz)

    def face(minus):
        k = int(rng.integers(0, 6))
        if k <= 2:
            return td.PML(num_layers=int(rng.integers(2, 8)))
        if k == 3:
            return td.StablePML(num_layers=int(rng.integers(3, 9)))
        if k == 4 and minus:
            return td.PMCBoundary()
        return td.PECBoundary()
    faces = [[face(True), face(False)] for _ in range(3)]
    if not any(isinstance(f, (td.PML, td.StablePML)) for pair in faces for f in pair):
        faces[int(rng.integers(0, 3))][1] = td.PML(num_layers=4)
    bounds = [td.Boundary(minus=faces[a][0], plus=faces[a][1]) for a in range(3)]
    if periodic:            # (round 5, last part) x and / or y periodic — a metasurface / grating cell; z then carries layers
        which = int(rng.integers(0, 3))
        if which in (0, 2):
            bounds[0] = td.Boundary.periodic()
            N = (max(40, N[0] // 4 * 4), N[1], N[2])      # (rows of a multiple of four cells: the fused sweep)
        if which in (1, 2):
            bounds[1] = td.Boundary.periodic()
        if not any(isinstance(f, (td.PML, td.StablePML)) for f in faces[2]):
            bounds[2] = td.Boundary.pml(num_layers=int(rng.integers(2, 8)))
    bspec = td.BoundarySpec(x=bounds[0], y=bounds[1], z=bounds[2])
    size = tuple((n - 1e-6) * DL for n in N)
    h = [0.5 * v for v in size]
    structures = []
    if rng.integers(0, 3):
        structures.append(td.Structure(geometry=td.Box(center=(0, 0.1 * h[1], 0), size=(td.inf, 0.6 * h[1], 0.5 * h[2])),
                                       medium=td.Medium(permittivity=float(rng.uniform(1.5, 4)), conductivity=float(rng.choice([0.0, 0.02])))))
        if rng.integers(0, 2):
            structures.append(td.Structure(geometry=td.Sphere(center=(0.2 * h[0], 0.0, 0.1 * h[2]), radius=0.5 * min(h[1], h[2])), medium=td.Medium(permittivity=2.0)))
        if rng.integers(0, 2):
            structures.append(td.Structure(geometry=td.Box(center=(-h[0] + 2 * DL, 0, 0), size=(8 * DL, 0.3 * h[1], 0.4 * h[2])), medium=td.PEC))
    srcs = []
    if rng.integers(0, 3):          # deep inside the bulk (|c| < 0.1 of the half extent: more than three cells from any collar)
        for pol in rng.choice(["Ex", "Ey", "Ez", "Hx", "Hy", "Hz"], size=int(rng.integers(1, 4)), replace=False):
            srcs.append(td.PointDipole(center=tuple(float(rng.uniform(-0.08, 0.08) * v) for v in h), source_time=PULSE, polarization=str(pol)))
    if not srcs:
        srcs = [td.PointDipole(center=(0, 0, 0), source_time=PULSE, polarization="Ez")]
    mons = []
    for q in range(int(rng.integers(0, 3))):
        mons.append(td.FieldTimeMonitor(center=tuple(float(rng.uniform(-0.1, 0.1) * v) for v in h), size=(0, 0, 0), name=f"p{q}", interval=int(rng.integers(1, 4))))
    if rng.integers(0, 2):
        mons.append(td.FieldMonitor(center=(0, 0, 0), size=(0.2 * h[0], 0.2 * h[1], 0), freqs=[2.5e14, 3e14], name="f"))
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, sources=srcs, monitors=mons, structures=structures,
                        boundary_spec=bspec, shutoff=0)
    steps = int(rng.integers(8, 30))
    disc = discretize(sim, n_steps=steps + 1)
    disc.spec.decay_every = int(rng.choice([0, 0, 8]))
    return disc, steps


def run(disc, steps, split, lib, twostep, shell2, shape, seed):
    with HipEngine(disc.spec, lib=lib, variant=L.VARIANT_FUSED, axis_shift=0) as e:
        e.set_option(L.OPT_TWOSTEP, twostep)
        e.set_option(L.OPT_SHELL_PAIRS, 1)
        e.set_option(L.OPT_SHELL2, shell2)
        if shape:
            e.set_option(L.OPT_SHELL2_SHAPE, shape)
        rng = np.random.default_rng(seed)
        for c in range(6):
            f = e.get_field(c)
            e.set_field(c, ((1e-3 if c < 3 else 1e-3 / 376.73) * rng.uniform(-1, 1, size=f.shape)).astype(np.float32))
        pairs = s2 = why = 0
        for r in (split, steps - split):
            if r > 0:
                st = e.run(r)
                pairs += int(st.fused2_pairs)
                s2 += int(st.shell2_pairs)
                why = int(st.fused2_off_reason)
        return [e.get_field(c) for c in range(6)], e.results(), pairs, s2, why


def run_cases(n_cases, seed=1, lib=None, quiet=False, small=False, periodic=False):
    """-> (cases that differ, cases that took shell2 pairs)"""
    rng = np.random.default_rng(seed)
    bad = taken = 0
    for q in range(n_cases):
        disc, steps = case(rng, small, periodic)
        split = int(rng.integers(0, steps))
        w, zc = int(rng.integers(4, 17)), int(rng.integers(2, 40))
        shell2 = int(rng.integers(1, 4))
        if periodic:
            shell2 = 1          # (periodic faces: the one-launch form only)
        qw = int(rng.choice([0, 0, int(rng.integers(3, 65))]))
        shape = qw + 128 * int(rng.integers(1, 8)) + 1024 * int(rng.choice([0, int(rng.integers(1, 40))])) + (int(rng.integers(1, 9)) << 17) + (int(rng.choice([0, int(rng.integers(1, 40))])) << 21)
        ref_f, ref_m, p0, _, _ = run(disc, steps, split, lib, 0, 0, 0, q)
        got_f, got_m, p1, s2, why = run(disc, steps, split, lib, w + 64 * zc, shell2, shape, q)
        ok = p0 == 0 and all(np.array_equal(a, b) for a, b in zip(ref_f, got_f)) and all(np.array_equal(ref_m[k], got_m[k]) for k in ref_m)
        if not quiet or not ok:
            print(f"case {q}: N={disc.spec.shape} steps={steps} split={split} W={w} zc={zc} form={shell2} shape={shape} "
                  f"monitors={len(ref_m)} pairs={p1} shell2={s2}{'' if p1 else ' (reason %d)' % why} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
        taken += s2 > 0
    return bad, taken


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    bad, taken = run_cases(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1, periodic=len(sys.argv) > 3 and sys.argv[3] == "periodic")
    print("fuzz:", n_cases - bad, "of", n_cases, "cases bit-identical;", taken, "took shell2 pairs")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
