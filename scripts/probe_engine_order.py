"""Does an engine run slower when an earlier engine of the same process held an RCCL communicator?  (scripts/probe_end_rank.py's first form
and the `slab` stage of scripts/gpu_visit.sh time several looped-back slab ranks in ONE process; the later ones read 10 - 35 % slow.)
One process per sequence; the LAST engine of a sequence is the one reported.
    python scripts/probe_engine_order.py            (runs the sequences, a process each)
    python scripts/probe_engine_order.py --seq c64,c128"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEQS = ["c128", "c64,c128", "p64,c128", "v512,c128", "c64,c64,c64,c128", "p128", "c64,p128", "p64,p128"]
SEQS_TORCH = ["v512", "t0,v512", "c64,v512", "t0,c128", "t0,p128", "c128", "p128"]       # t0: torch.distributed's own RCCL (world size 1) initialised first


def run_seq(names):
    import numpy as np
    import torch  # noqa: F401
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import probe_slab
    from tidy3d_amd import lib as L
    from tidy3d_amd.engine import HipEngine
    out = None
    for nm in names:
        if nm[0] == "t":
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
            torch.cuda.set_device(0)
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            x = torch.ones(1024, device="cuda"); dist.all_reduce(x); dist.barrier(); torch.cuda.synchronize()
            continue
        kind, nz = nm[0], int(nm[1:])                 # c: looped-back slab rank (periodic z, RCCL), p: the same slab without a communicator, v: 512^3-like cube,
        #                                               q: looped-back slab rank that carries CPML on x / y ($BOXES: FDTD_OPT_SLAB_BOXES_FIRST)
        sp = probe_slab.spec_for(512, nz, 400, 2 if kind == "q" else 0)
        if kind == "v":                                # the headline's walls
            import dataclasses
            from tidy3d_amd.spec import BC_PEC
            sp = dataclasses.replace(sp, bc=(sp.bc[0], sp.bc[1], (BC_PEC, BC_PEC)))
        kw = dict(variant=L.VARIANT_FUSED)
        if kind in "cq":
            kw["force_comm"] = True
        with HipEngine(sp, **kw) as e:
            if kind in "cq":
                e.comm_init(e.unique_id())
            if kind == "q":
                e.set_option(L.OPT_PML_FUSED, 7)
                if os.environ.get("BOXES"):
                    e.set_option(L.OPT_SLAB_BOXES_FIRST, int(os.environ["BOXES"]))
            rng = np.random.default_rng(0)
            for c in range(6):
                e.set_field(c, rng.uniform(-1e-3, 1e-3, tuple(reversed(sp.shape))).astype(np.float32))
            e.run(30)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                st = e.run(100)
                best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
            out = {"sequence": ",".join(names), "last": nm, "ms_per_step": best, "pairs": int(st.fused2_pairs), "stream_overlap": int(st.stream_overlap),
                   "stream_retries": int(st.stream_retries), "boxes": os.environ.get("BOXES", ""), "shell2_pairs": int(st.shell2_pairs), "placement": [int(st.placement) >> 8, int(st.placement) & 255]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--seq" in sys.argv:
        run_seq(sys.argv[sys.argv.index("--seq") + 1].split(","))
    else:
        for s in (SEQS_TORCH if "--torch" in sys.argv else SEQS):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--seq", s], capture_output=True, text=True, timeout=600)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(lines[-1] if lines else json.dumps({"sequence": s, "error": r.stderr[-400:]}), flush=True)
