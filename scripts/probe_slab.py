"""Per-rank timeline proxy for the N-GPU strong-scaling run: one 512 x 512 x (512/N) slab on ONE GPU,
periodic in z, with the RCCL ghost exchange looped back to the same rank (HipEngine force_comm) —
every kernel, event and Send/Recv of the multi-rank schedule runs, only the wire is missing.
Prints one JSON line per (N, mode)."""
import json
import sys
import time

import numpy as np
import torch  # noqa: F401  (before the solver library: one HIP runtime per process)

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine


def spec_for(n, nz, steps, pml=False):
    dl = 0.05
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
    b = td.BoundarySpec(x=td.Boundary(minus=td.PECBoundary(), plus=td.PECBoundary()),
                        y=td.Boundary(minus=td.PECBoundary(), plus=td.PECBoundary()),
                        z=td.Boundary.periodic())
    structures = []
    nxy = n
    if pml:          # V2/V3-like slab: CPML on x and y (inside the cell count), a Lorentz cylinder along z
        b = td.BoundarySpec(x=td.Boundary.pml(num_layers=12), y=td.Boundary.pml(num_layers=12), z=td.Boundary.periodic())
        nxy = n - 24
        structures = [td.Structure(geometry=td.Cylinder(center=(0, 0, 0), radius=100 * dl, length=td.inf, axis=2),
                                   medium=td.Lorentz(eps_inf=2.0, coeffs=[(2.0, 4e14, 2e13)]) if int(pml) == 1
                                   else td.Medium(permittivity=4.0))]             # --pml 2: V2-like (no ADE)
    sim = td.Simulation(size=(nxy * dl, nxy * dl, nz * dl), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                        structures=structures, sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ez")],
                        monitors=[], boundary_spec=b, shutoff=0)
    sp = discretize(sim, n_steps=steps).spec
    sp.decay_every = 0
    return sp


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--slabs", default="8,4,2")
    ap.add_argument("--modes", default="single_fused,comm_fused,comm_two_pass")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--zchunk", type=int, default=0)
    ap.add_argument("--bnd", type=int, default=0)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--autotune", type=int, default=0)
    ap.add_argument("--pml", type=int, default=0)
    ap.add_argument("--pml-fused", type=int, default=-1)
    ap.add_argument("--warm", type=int, default=30)
    ap.add_argument("--opt", action="append", default=[], help="NAME=VALUE engine options (tidy3d_amd.lib.OPT_*)")
    ap.add_argument("--twostep", default="0,-1", help="FDTD_OPT_TWOSTEP values tried inside each engine (0 = single steps, -1 = default shape, W + 64 zc)")
    ap.add_argument("--ref512", type=int, default=1, help="time the one-GPU 512^3 V0 run (two steps per sweep) first: the denominator of implied_speedup_vs_1gpu")
    args = ap.parse_args()
    n = 512
    steps, warm = args.steps, args.warm
    ref_ms = 1.10
    if args.ref512 and not args.pml:
        sp = spec_for(n, n, 200, 0)
        import dataclasses
        from tidy3d_amd.spec import BC_PEC
        sp = dataclasses.replace(sp, bc=(sp.bc[0], sp.bc[1], (BC_PEC, BC_PEC)))
        with HipEngine(sp, variant=L.VARIANT_FUSED) as e:
            rng = np.random.default_rng(0)
            for c in range(6):
                e.set_field(c, rng.uniform(-1e-3, 1e-3, (n, n, n)).astype(np.float32))
            e.run(20)
            t0 = time.perf_counter()
            st = e.run(100)
            ref_ms = (time.perf_counter() - t0) / 100 * 1e3
            print(json.dumps({"ref": "512^3 V0 on one GPU", "ms_per_step": ref_ms, "fused2_pairs": int(st.fused2_pairs)}), flush=True)
    all_modes = {"single_fused": dict(variant=L.VARIANT_FUSED),
                 "comm_fused": dict(variant=L.VARIANT_FUSED, force_comm=True),
                 "comm_two_pass": dict(variant=L.VARIANT_ZMARCH, force_comm=True)}
    for ngpu in [int(x) for x in args.slabs.split(",")]:
        nz = n // ngpu
        sp = spec_for(n, nz, steps + warm + 8, args.pml)
        for mode in args.modes.split(","):
            kw = dict(all_modes[mode], z_chunk=args.zchunk)
            with HipEngine(sp, **kw) as e:
                if kw.get("force_comm"):
                    e.comm_init(e.unique_id())
                if args.bnd:
                    e.set_option(L.OPT_BND_PLANES, args.bnd)
                if args.rows:
                    e.set_option(L.OPT_ROWS, args.rows)
                e.set_option(L.OPT_AUTOTUNE, args.autotune)
                if args.pml_fused >= 0:
                    e.set_option(L.OPT_PML_FUSED, args.pml_fused)
                for kv in args.opt:
                    e.set_option(getattr(L, kv.split("=")[0]), int(kv.split("=")[1]))
                rng = np.random.default_rng(0)
                for c in range(6):
                    e.set_field(c, rng.uniform(-1e-3, 1e-3, (nz, n, n)).astype(np.float32))
                for ts in [int(x) for x in args.twostep.split(",")]:
                    e.set_option(L.OPT_TWOSTEP, ts)
                    e.run(warm)
                    t0 = time.perf_counter()
                    stt = e.run(steps)
                    dt = time.perf_counter() - t0
                    print(json.dumps({"slab_of": ngpu, "nz": nz, "mode": mode, "twostep": ts, "fused2_pairs": int(stt.fused2_pairs), "shape": [int(stt.fused2_shape) & 63, int(stt.fused2_shape) >> 6],
                                      "why": int(stt.fused2_off_reason), "zchunk": args.zchunk, "bnd": args.bnd, "rows": args.rows,
                                      "tile": [int(stt.tile_rows), int(stt.tile_zchunk)], "autotune": args.autotune, "pml": args.pml, "pml_fused": args.pml_fused, "opt": args.opt,
                                      "ms_per_step": dt / steps * 1e3, "one_gpu_ms_per_step": ref_ms, "ideal_ms": ref_ms / ngpu,
                                      "implied_speedup_vs_1gpu": ref_ms / (dt / steps * 1e3)}), flush=True)


if __name__ == "__main__":
    main()
