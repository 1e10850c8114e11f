#!/usr/bin/env python
"""What a kernel that holds MORE tile bodies costs by itself (round 6): the bench workloads advanced by their own instantiation and —
FDTD_OPT_WHATIF = 9 — by the one that also carries the lines for paged source terms, over a map without any source segment: every
tile dispatches to the same bodies as before, results unchanged.  Alternated inside one engine.
    python scripts/probe_bodies.py [n] [rounds]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for wl in ("v4", "v2"):
    spec = bench.build_spec(n, 64, wl)
    with HipEngine(spec, variant=L.VARIANT_FUSED) as e:
        for c in range(6):
            arr = np.empty((n, n, n), dtype=np.float32)
            for k in range(n):
                arr[k] = bench.init_plane(c, k, n)
            e.set_field(c, arr)
        e.run(80)                 # past the dipole's pulse: no source term of any kind
        t = {0: [], 9: []}
        k = {0: [], 9: []}
        for r in range(rounds):
            for w in ((0, 9) if r % 2 == 0 else (9, 0)):
                e.set_option(L.OPT_WHATIF, w)
                e.set_option(L.OPT_FLAGS, 0)
                e.run(4)
                t0 = time.perf_counter()
                e.run(40)
                t[w].append((time.perf_counter() - t0) / 40 * 1e3)
                e.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
                st = e.run(10)
                k[w].append(st.fused_kernel_ms / max(1, st.fused_kernel_launches))
        e.set_option(L.OPT_WHATIF, 0)
        print(json.dumps({"workload": wl, "n": n, "ms_per_step": float(np.median(t[0])), "ms_per_step_larger_kernel": float(np.median(t[9])),
                          "bulk_ms_per_launch": float(np.median(k[0])), "bulk_ms_per_launch_larger_kernel": float(np.median(k[9])),
                          "ratio": float(np.median(k[9]) / np.median(k[0])), "spread": [float(np.ptp(k[0])), float(np.ptp(k[9]))]}), flush=True)
