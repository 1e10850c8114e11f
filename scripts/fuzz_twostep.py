"""Randomised self-check of the two-step sweep on the device: random grid shapes (1-3 x tiles, ragged rows / chunks), tile shapes,
wall types (PEC, PMC, absorber layers, CPML of random thickness -> shell pairs), media, dispersive bodies and plane waves across
a unit cell (-> z holes in the bulk), electric / magnetic dipoles, probes and DFT monitors; two steps per sweep == single
sweeps, bit for bit.
    python scripts/fuzz_twostep.py [n_cases] [seed]
Also run by the GPU suite (tests/test_gpu_production_path.py, run_cases)."""
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
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)


def case(rng):
    nx = int(rng.choice([4, 8, 36, 120, 252, 256, 260, 300, 508, 512, 516, 600]))
    ny, nz = int(rng.integers(2, 48)), int(rng.integers(2, 56))
    N = (nx, ny, nz)
    size = tuple(n * DL for n in N)
    pmc = [bool(rng.integers(0, 2)) for _ in range(3)]
    kind = int(rng.integers(0, 4))
    absorb = kind == 0          # a quarter of the cases: absorber layers on some faces (then no magnetic dipoles)
    cpml = kind in (1, 3)       # half: CPML on some faces (step pairs with a shell of single steps when the grid has a bulk left)
    holes = kind == 3           # half of those: z holes — a dispersive body thin along z, or (periodic x / y) a plane wave's plane
    wave = holes and bool(rng.integers(0, 2))
    if cpml:
        ny, nz = max(ny, 14), max(nz, 14)
        if holes:
            nz = int(rng.integers(40, 64))
            nx = int(rng.choice([36, 120, 260, 300]))
        N = (nx, ny, nz)
        size = tuple(n * DL for n in N)

    def face(ax_n, minus, p):
        if absorb and rng.integers(0, 2) and ax_n >= 8:
            return td.Absorber(num_layers=int(rng.integers(2, min(7, ax_n // 2))))
        if cpml and rng.integers(0, 4) and ax_n >= 8:
            return td.PML(num_layers=int(rng.integers(2, 7)))
        return td.PMCBoundary() if (minus and p) else td.PECBoundary()
    bspec = td.BoundarySpec(**{ax: td.Boundary(minus=face(n_, True, p), plus=face(n_, False, p))
                               for ax, p, n_ in zip("xyz", pmc, N)})
    if wave:                    # a metasurface's unit cell: periodic x / y, CPML z, a plane wave coming down
        bspec = td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml(num_layers=int(rng.integers(3, 7))))

    def pos(margin=0.8):
        return tuple(float(rng.uniform(-0.5 * s + min(margin * DL, 0.45 * s), 0.5 * s - min(margin * DL, 0.45 * s))) for s in size)
    srcs = []
    for _ in range(int(rng.integers(1, 5))):
        pol = str(rng.choice(["Ex", "Ey", "Ez"] if absorb else ["Ex", "Ey", "Ez", "Hx", "Hy", "Hz"]))
        c = list(pos())
        if pol in ("Hy", "Hz") and nx > 256:          # not in the columns next to a seam (that case keeps single steps)
            i = (c[0] + 0.5 * size[0]) / DL
            if min(abs(i - 256), abs(i - 512)) < 3:
                c[0] += 5 * DL if c[0] + 5 * DL < 0.5 * size[0] - DL else -5 * DL
        srcs.append(td.PointDipole(center=tuple(c), source_time=PULSE, polarization=pol))
    if wave:
        srcs = srcs[:1] + [td.PlaneWave(center=(0, 0, 0.5 * size[2] - float(rng.uniform(4, 9)) * DL), size=(td.inf, td.inf, 0),
                                        source_time=td.GaussianPulse(freq0=3e14, fwidth=2.4e14), direction="-",
                                        pol_angle=float(rng.uniform(0, 1.5)))]
    mons = []
    for q in range(int(rng.integers(0, 4))):
        fields = [str(f) for f in rng.choice(["Ex", "Ey", "Ez", "Hx", "Hy", "Hz"], size=int(rng.integers(1, 4)), replace=False)]
        mons.append(td.FieldTimeMonitor(center=pos(2.5), size=(0, 0, 0), name=f"m{q}", interval=int(rng.integers(1, 4)), fields=fields,
                                        colocate=False))
    for q in range(int(rng.integers(0, 3))):           # DFT monitors: planes of any orientation, volumes, flux planes
        kind = int(rng.integers(0, 4))
        c = pos(2.5)
        if kind < 3:
            sz = [td.inf, td.inf, td.inf]
            sz[kind] = 0
            fields = [str(f) for f in rng.choice(["Ex", "Ey", "Ez", "Hx", "Hy", "Hz"], size=int(rng.integers(1, 5)), replace=False)]
            if rng.integers(0, 3) == 0:
                mons.append(td.FluxMonitor(center=c, size=tuple(sz), freqs=[3e14, 3.2e14], name=f"fl{q}"))
            else:
                mons.append(td.FieldMonitor(center=c, size=tuple(sz), freqs=[3e14], name=f"d{q}", fields=fields))
        else:
            mons.append(td.FieldMonitor(center=c, size=tuple(float(rng.uniform(0.1, 0.5) * s_) for s_ in size), freqs=[2.9e14, 3e14], name=f"v{q}"))
    structures = []
    if rng.integers(0, 2):
        structures = [td.Structure(geometry=td.Box(center=pos(), size=tuple(float(rng.uniform(0.1, 0.6) * s) for s in size)),
                                   medium=td.Medium(permittivity=float(rng.uniform(1.5, 6)), conductivity=float(rng.choice([0, 0.03])))),
                      td.Structure(geometry=td.Sphere(center=pos(), radius=float(rng.uniform(0.05, 0.3) * min(size))),
                                   medium=td.Medium(permittivity=2.5)),
                      td.Structure(geometry=td.Box(center=pos(), size=(0.1, 0.1, 0.1)), medium=td.PEC)]
    if holes:                   # dispersive bodies: their planes (+- 2) take single steps with the shell
        zc_ = float(rng.uniform(-0.2, 0.2)) * size[2]
        body = [td.Structure(geometry=td.Box(center=(0, 0, zc_), size=(td.inf, td.inf, float(rng.uniform(1.5, 5)) * DL)),
                             medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])),
                td.Structure(geometry=td.Sphere(center=(pos()[0], pos()[1], zc_ + float(rng.uniform(-3, 3)) * DL), radius=float(rng.uniform(2, 4)) * DL),
                             medium=td.Drude(eps_inf=1.5, coeffs=[(6e14, 5e13)]))]
        structures = structures + body[:int(rng.integers(1, 3))] if not wave or rng.integers(0, 2) else structures
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, sources=srcs, monitors=mons,
                        structures=structures, boundary_spec=bspec, shutoff=0)
    steps = int(rng.integers(9, 40))
    disc = discretize(sim, n_steps=steps + 1)
    disc.spec.decay_every = int(rng.choice([0, 0, 7, 16]))
    w, zc = int(rng.integers(4, 17)), int(rng.integers(2, 40))
    return N, disc, steps, w, zc, (pmc, "abs" if absorb else (("wave" if wave else "holes") if holes else ("cpml" if cpml else ""))), bool(structures)


def run(disc, steps, twostep, split, lib=None, shell2=1, shape=0):
    with HipEngine(disc.spec, lib=lib, variant=L.VARIANT_FUSED, axis_shift=0) as e:
        e.set_option(L.OPT_TWOSTEP, twostep)
        e.set_option(L.OPT_SHELL_PAIRS, 1)          # CPML grids: shell pairs whatever the cost model says of these small grids
        # the form of a shell pair: 0 = the shell as two single steps (round 4), 1 = as shell2_step_kernel launches (two steps per sweep, psi
        # carried) wherever that applies, 2 / 3 = the same with one launch per instantiation / per box; and the tile shapes of its boxes
        e.set_option(L.OPT_SHELL2, shell2)
        if shape:
            e.set_option(L.OPT_SHELL2_SHAPE, shape)
        pairs = why = s2 = 0
        for r in (split, steps - split):
            if r > 0:
                st = e.run(r)
                pairs += int(st.fused2_pairs)
                s2 += int(st.shell2_pairs)
                why = int(st.fused2_off_reason)
        S2[0] = s2
        return [e.get_field(c) for c in range(6)], e.results(), pairs, why


S2 = [0]


def run_cases(n_cases, seed=1, lib=None, quiet=False):
    """-> (cases that differ, cases that took step pairs)"""
    rng = np.random.default_rng(seed)
    bad = taken = 0
    for q in range(n_cases):
        N, disc, steps, w, zc, pmc, mat = case(rng)
        split = int(rng.integers(0, steps))
        # (drawn from a generator of their own: the simulations of a seed stay those of round 4)
        r2 = np.random.default_rng([seed, q, 5])
        shell2 = int(r2.integers(0, 4))
        qw = int(r2.choice([0, 0, int(r2.integers(3, 65))]))
        shape = qw + 128 * int(r2.integers(1, 8)) + 1024 * int(r2.choice([0, int(r2.integers(1, 40))])) + (int(r2.integers(1, 9)) << 17) + (int(r2.choice([0, int(r2.integers(1, 40))])) << 21)
        ref_f, ref_m, p0, _ = run(disc, steps, 0, split, lib)
        got_f, got_m, p1, why = run(disc, steps, w + 64 * zc, split, lib, shell2, shape)
        ok = p0 == 0 and all(np.array_equal(a, b) for a, b in zip(ref_f, got_f)) and all(np.array_equal(ref_m[k], got_m[k]) for k in ref_m)
        amp = max(float(np.abs(f).max()) for f in ref_f)
        if not quiet or not ok:
            print(f"case {q}: N={disc.spec.shape} steps={steps} split={split} W={w} zc={zc} pmc={pmc} media={mat} monitors={len(ref_m)} "
                  f"pairs={p1}{'' if p1 else ' (reason %d)' % why} shell2={shell2}:{S2[0]} max|F|={amp:.3g} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
        taken += p1 > 0
    return bad, taken


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    bad, taken = run_cases(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("fuzz:", n_cases - bad, "of", n_cases, "cases bit-identical;", taken, "took step pairs")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
