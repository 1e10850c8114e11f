"""Do the z-slab ranks of a CPML-walled 512^3 problem take the same time per step under the split `balanced_slabs` chooses?
One GPU cannot loop an END rank's exchange back to itself (its other z face carries layers), so the ranks are timed as standalone
problems on one GPU, in the step pairs a rank takes: the end rank = its planes with CPML on x / y / z-min and a PEC wall where the cut
would be; a middle rank = its planes with CPML on x / y and PEC walls on z (beside it: the same planes as a real slab rank, periodic z
with the RCCL exchange looped back — what the cut's hole and exchanges add).  Prints one JSON line per (ranks, cost model).
Every slab is timed in a process of its own: in the first form of this script (one process) every engine created after the first
looped-back RCCL engine ran 18 - 35 % slower than the same slab in a fresh process (profiles/r6/r6er_end_rank_split_one_process_artifact.jsonl;
cause not established).  A rank process holds one engine.
    python scripts/probe_end_rank.py [--ranks 8,4] [--steps 200]"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch  # noqa: F401  (before the solver library: one HIP runtime per process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tidy3d_amd.schema as td  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.discretize import discretize  # noqa: E402
from tidy3d_amd.engine import HipEngine, balanced_slabs  # noqa: E402

DL, LAYERS = 0.05, 12


def sim_for(n, nz, z_minus, z_plus, steps):
    """n x n x nz cells (layers inside the count), a dielectric cylinder along z (material words everywhere, no ADE)"""
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
    zl = LAYERS * (z_minus == "pml") + LAYERS * (z_plus == "pml")
    face = {"pml": td.PML(num_layers=LAYERS), "pec": td.PECBoundary(), "periodic": td.Periodic()}
    b = td.BoundarySpec(x=td.Boundary.pml(num_layers=LAYERS), y=td.Boundary.pml(num_layers=LAYERS),
                        z=td.Boundary(minus=face[z_minus], plus=face[z_plus]))
    sim = td.Simulation(size=((n - 2 * LAYERS) * DL, (n - 2 * LAYERS) * DL, (nz - zl) * DL), grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12,
                        structures=[td.Structure(geometry=td.Cylinder(center=(0, 0, 0), radius=100 * DL, length=td.inf, axis=2), medium=td.Medium(permittivity=4.0))],
                        sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ez")], monitors=[], boundary_spec=b, shutoff=0)
    sp = discretize(sim, n_steps=steps).spec
    sp.decay_every = 0
    return sp


def time_slab(sp, steps, warm, loop_back=False):
    kw = dict(variant=L.VARIANT_FUSED)
    if loop_back:
        kw["force_comm"] = True
    with HipEngine(sp, **kw) as e:
        if loop_back:
            e.comm_init(e.unique_id())
            e.set_option(L.OPT_PML_FUSED, 7)
        rng = np.random.default_rng(0)
        for c in range(6):
            e.set_field(c, rng.uniform(-1e-3, 1e-3, tuple(reversed(sp.shape))).astype(np.float32))
        e.run(warm)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            st = e.run(steps)
            dt = (time.perf_counter() - t0) / steps * 1e3
            best = dt if best is None else min(best, dt)
        return best, int(st.fused2_pairs), int(st.shell2_pairs)


def fresh(n, nz, zm, zp, steps, warm, loop_back=False):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--slab", f"{n},{nz},{zm},{zp},{int(loop_back)}", "--steps", str(steps)],
                       capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("[")]
    if not lines:
        raise RuntimeError(r.stderr[-2000:])
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="8,4")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--slab", default="", help="(internal) n,nz,z_minus,z_plus,loop_back: time this one slab and print [ms, pairs, shell2 pairs]")
    args = ap.parse_args()
    n, steps, warm = args.n, args.steps, 30
    total = steps * 3 + warm + 8
    if args.slab:
        n_, nz, zm, zp, lb = args.slab.split(",")
        print(json.dumps(list(time_slab(sim_for(int(n_), int(nz), zm, zp, total), steps, warm, loop_back=bool(int(lb))))), flush=True)
        return
    whole = sim_for(n, n, "pml", "pml", 4)
    for world in [int(x) for x in args.ranks.split(",")]:
        for model in ("single_steps", "pairs"):
            slabs = balanced_slabs(whole, world, pairs=model == "pairs")
            n_end, n_mid = slabs[0][1] - slabs[0][0], slabs[1][1] - slabs[1][0]
            t_end, p_end, s_end = fresh(n, n_end, "pml", "pec", steps, warm)
            t_mid, p_mid, s_mid = fresh(n, n_mid, "pec", "pec", steps, warm)
            rec = {"ranks": world, "cost_model": model, "planes": [b - a for a, b in slabs], "end_rank_ms_per_step": t_end, "middle_rank_ms_per_step": t_mid,
                   "end_over_middle": t_end / t_mid, "pairs": [p_end, p_mid], "shell2_pairs": [s_end, s_mid]}
            if world > 2:
                t_loop, p_loop, _ = fresh(n, n_mid, "periodic", "periodic", steps, warm, loop_back=True)
                rec.update(middle_rank_looped_back_ms_per_step=t_loop, middle_rank_looped_back_pairs=p_loop)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
