"""One rank of a multi-GPU solve started by ``tidy3d_amd.web.run(..., devices=[...])`` (or by hand under torchrun):

    python -m tidy3d_amd.dist_main --sim sim.pkl --out data.pkl [--backend nccl] [--n-steps N]

Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT from the environment (torch.distributed.run's
contract), runs ``tidy3d_amd.dist.run`` — one process per GPU, z-slab decomposition, RCCL ghost-plane exchange inside
libfdtd_hip.so — and has rank 0 pickle the SimulationData to ``--out``.  ``--lib`` / ``--hook`` exist for the CPU
test of this module (the HIP sources under the emulator, gloo): never needed on a GPU box."""
from __future__ import annotations

import argparse
import importlib
import os
import pickle
import sys


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--sim", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--n-steps", type=int, default=-1)
    ap.add_argument("--lib", default=None, help="explicit path of the solver library (tests)")
    ap.add_argument("--hook", default=None, help="module:function called with the loaded library before the run (tests)")
    args = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    from . import dist as tdist
    from .lib import load_library
    import datetime
    rank, world, local = tdist.env_ranks()
    if not args.lib:
        load_library()            # before the first HIP call of this process: see lib._prefer_hw_queues
    # a rank whose siblings never arrive (one died at import, ran out of memory, ...) gives up instead of waiting for ever
    tmo = datetime.timedelta(seconds=int(os.environ.get("TIDY3D_AMD_RDV_TIMEOUT", "300")))
    if args.backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local), timeout=tmo)
    else:
        dist.init_process_group(backend="gloo", timeout=tmo)
    try:
        lib = load_library(args.lib) if args.lib else load_library()
        if args.hook:
            mod, fn = args.hook.split(":")
            getattr(importlib.import_module(mod), fn)(lib)
        with open(args.sim, "rb") as f:
            sim = pickle.load(f)
        sd = tdist.run(sim, verbose=False, n_steps=None if args.n_steps < 0 else args.n_steps, lib=lib,
                       device=local if args.backend == "nccl" else 0)
        if rank == 0:
            with open(args.out, "wb") as f:
                pickle.dump(sd, f, protocol=pickle.HIGHEST_PROTOCOL)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
