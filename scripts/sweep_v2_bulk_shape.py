#!/usr/bin/env python
"""Planes per chunk of the clipped bulk sweep of shell2 pairs (bench workload v2 / v3 at n^3) inside ONE engine: the library's choice
(16 planes beside the shell's boxes up to 512^3) against forced FDTD_OPT_TWOSTEP words, alternating, median of `rounds` samples.
    python scripts/sweep_v2_bulk_shape.py [n] [workload] [rounds] [steps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402
from tidy3d_amd.engine import HipEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wl = sys.argv[2] if len(sys.argv) > 2 else "v2"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
shapes = [-1] + [16 + 64 * zc for zc in (12, 16, 20, 22, 24, 28, 32, 44)]
spec = bench.build_spec(n, 4000, wl)
with HipEngine(spec, device=0) as e:
    for c in range(6):
        e.set_field(c, np.stack([bench.init_plane(c, k, n) for k in range(n)]).astype(np.float32))
    e.run(20)
    t = {s: [] for s in shapes}
    info = {}
    for r in range(rounds):
        for s in (shapes if r % 2 == 0 else shapes[::-1]):
            e.set_option(L.OPT_TWOSTEP, s)
            e.run(6)
            t0 = time.perf_counter()
            st = e.run(steps)
            t[s].append((time.perf_counter() - t0) / steps * 1e3)
            info[s] = (int(st.fused2_shape) & 63, int(st.fused2_shape) >> 6, int(st.fused2_pairs), int(st.shell2_pairs))
    for s in shapes:
        a = np.array(t[s])
        print(json.dumps({"n": n, "workload": wl, "twostep": s, "shape": info[s][:2], "pairs": info[s][2], "shell2_pairs": info[s][3],
                          "ms_per_step_median": float(np.median(a)), "min": float(a.min()), "max": float(a.max()),
                          "gcells_per_s": n ** 3 / float(np.median(a)) * 1e-6}), flush=True)
