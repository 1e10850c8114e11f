#!/usr/bin/env python
"""Tile classes of the two-step sweep (FDTD_OPT_TILE_SPLIT) off / on / off / on inside ONE engine per workload (same placement of
the arrays): the bench workloads with bodies at n^3.
  python scripts/probe_tile_split.py [n] [workloads, comma separated] [steps]   -> one JSON line per (workload, mode)"""
import json
import os
import sys
import time

import numpy as np
import torch  # before the solver library: it then binds to the HIP runtime torch ships (one runtime per process)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import build_spec  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wls = (sys.argv[2] if len(sys.argv) > 2 else "v1,v2,va").split(",")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
L.load_library()

for wl in wls:
    spec = build_spec(n, 2000, wl)
    with HipEngine(spec, device=0) as e:
        for c in range(6):
            arr = np.empty((n, n, n), dtype=np.float32)
            for k in range(n):
                arr[k] = np.random.default_rng(c * 100003 + k).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
            e.set_field(c, arr)
        for name, v in [("never", 0), ("always", 1), ("never", 0), ("always", 1), ("default", -1)]:
            e.set_option(L.OPT_TILE_SPLIT, v)
            e.run(10)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st = e.run(steps)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / steps * 1e3)
            ms = sorted(ts)[1]
            print(json.dumps({"workload": wl, "n": n, "tile_split": name, "ms_per_step": ms, "samples": ts,
                              "gcells_per_s": n ** 3 / ms * 1e-6, "fused2_pairs": int(st.fused2_pairs),
                              "shell2_pairs": int(st.shell2_pairs)}), flush=True)
