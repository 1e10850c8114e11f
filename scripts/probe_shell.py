#!/usr/bin/env python
"""Shell pairs against single steps inside ONE engine (same placement of the arrays): bench workload (default v2) at n^3.
  python scripts/probe_shell.py [n] [workload] [steps]   -> one JSON line per mode"""
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
wl = sys.argv[2] if len(sys.argv) > 2 else "v2"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
L.load_library()

spec = build_spec(n, 2000, wl)
with HipEngine(spec, device=0) as e:
    for c in range(6):
        arr = np.empty((n, n, n), dtype=np.float32)
        for k in range(n):
            arr[k] = np.random.default_rng(c * 100003 + k).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
        e.set_field(c, arr)
    P = {L.OPT_TWOSTEP: -1, L.OPT_SHELL_PAIRS: -1, L.OPT_STRIP: 8 + 64 * 3, L.OPT_EDGE_ZCHUNK: -1}
    modes = [("single", {L.OPT_TWOSTEP: 0}), ("pairs", P),
             ("pairs_one_stream", {**P, L.OPT_SHELL_PAIRS: 2}),
             ("pairs_strip8x4", {**P, L.OPT_STRIP: 8 + 64 * 4}),
             ("pairs_strip8x4_one_stream", {**P, L.OPT_STRIP: 8 + 64 * 4, L.OPT_SHELL_PAIRS: 2}),
             ("pairs_strip16x3", {**P, L.OPT_STRIP: 16 + 64 * 3}),
             ("pairs_strip4x3", {**P, L.OPT_STRIP: 4 + 64 * 3}),
             ("pairs_edge8", {**P, L.OPT_EDGE_ZCHUNK: 8}), ("pairs_edge5", {**P, L.OPT_EDGE_ZCHUNK: 5}),
             ("pairs_edge4", {**P, L.OPT_EDGE_ZCHUNK: 4}), ("pairs_edge8_one_stream", {**P, L.OPT_EDGE_ZCHUNK: 8, L.OPT_SHELL_PAIRS: 2}),
             ("pairs_16x16_edge8", {**P, L.OPT_EDGE_ZCHUNK: 8, L.OPT_TWOSTEP: 16 + 64 * 16}),
             ("pairs_16x24_edge8", {**P, L.OPT_EDGE_ZCHUNK: 8, L.OPT_TWOSTEP: 16 + 64 * 24}),
             ("pairs_16x16", {**P, L.OPT_TWOSTEP: 16 + 64 * 16}), ("pairs_16x32", {**P, L.OPT_TWOSTEP: 16 + 64 * 32}),
             ("pairs_16x64", {**P, L.OPT_TWOSTEP: 16 + 64 * 64}),
             ("pairs_16x16_strip8x4", {**P, L.OPT_TWOSTEP: 16 + 64 * 16, L.OPT_STRIP: 8 + 64 * 4}),
             ("single", {L.OPT_TWOSTEP: 0}), ("pairs", P)]
    if os.environ.get("PROBE_MODES"):
        keep = os.environ["PROBE_MODES"].split(",")
        modes = [m for m in modes if m[0] in keep]
    for name, opts in modes:
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_option(L.OPT_FLAGS, 0)
        e.run(10)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.run(steps)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps * 1e3)
        e.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
        st = e.run(10)
        print(json.dumps({"mode": name, "n": n, "workload": wl, "ms_per_step": float(np.median(ts)), "samples": ts,
                          "gcells_per_s": n ** 3 / float(np.median(ts)) / 1e6,
                          "shell_pairs": int(st.shell_pairs), "fused2_pairs": int(st.fused2_pairs), "shape": int(st.fused2_shape),
                          "why": int(st.fused2_off_reason),
                          "bulk_ms_per_launch": st.fused_kernel_ms / max(1, st.fused_kernel_launches), "bulk_launches": int(st.fused_kernel_launches),
                          "shell_ms_sum_per_pair": st.shell_kernel_ms / max(1, int(st.shell_pairs)), "shell_launches": int(st.shell_kernel_launches),
                          "stream_overlap": int(st.stream_overlap)}), flush=True)
