#!/usr/bin/env python
"""shell2 pairs (shell2_step_kernel: the CPML shell two steps per sweep) against the round-4 shell pairs and single steps inside ONE
engine (same placement of the arrays), with the tile shapes of the boxes varied: bench workload (default v2) at n^3.
  python scripts/probe_shell2.py [n] [workload] [steps]   -> one JSON line per mode; fields compared with the single-step run"""
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


def shape_word(qw=32, ww=8, zcw=0, ws=4, zcs=0):
    return qw + 128 * (ww % 8) + 1024 * zcw + (ws << 17) + (zcs << 21)


spec = build_spec(n, 2000, wl)
with HipEngine(spec, device=0) as e:
    def fill():
        for c in range(6):
            arr = np.empty((n, n, n), dtype=np.float32)
            for k in range(n):
                arr[k] = np.random.default_rng(c * 100003 + k).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
            e.set_field(c, arr)
    P = {L.OPT_TWOSTEP: -1, L.OPT_SHELL_PAIRS: -1, L.OPT_SHELL2: 0, L.OPT_SHELL2_SHAPE: 0}
    S2 = {**P, L.OPT_SHELL2: 1}
    def sw(qw=0, ww=8, zcw=0, ws=0, zcs=0):
        return qw + 128 * (ww % 8) + 1024 * zcw + (ws << 17) + (zcs << 21)
    modes = [("single", {L.OPT_TWOSTEP: 0}), ("shell_pairs_r4", P), ("shell2", S2),
             ("shell2_one_stream", {**S2, L.OPT_SHELL_PAIRS: 2}),
             ("shell2_ws4", {**S2, L.OPT_SHELL2_SHAPE: sw(ws=4)}),
             ("shell2_ws4_one_stream", {**S2, L.OPT_SHELL2_SHAPE: sw(ws=4), L.OPT_SHELL_PAIRS: 2}),
             ("shell2_ws8", {**S2, L.OPT_SHELL2_SHAPE: sw(ws=8)}),
             ("shell2_ws8_one_stream", {**S2, L.OPT_SHELL2_SHAPE: sw(ws=8), L.OPT_SHELL_PAIRS: 2}),
             ("shell2_ws3", {**S2, L.OPT_SHELL2_SHAPE: sw(ws=3)}),
             ("shell2_ww4", {**S2, L.OPT_SHELL2_SHAPE: sw(ww=4)}),
             ("shell2_q32", {**S2, L.OPT_SHELL2_SHAPE: sw(qw=32)}),
             ("shell2_zc16", {**S2, L.OPT_SHELL2_SHAPE: sw(zcw=16, zcs=16)}),
             ("shell2_zc64", {**S2, L.OPT_SHELL2_SHAPE: sw(zcw=64, zcs=64)}),
             ("shell2_16x32", {**S2, L.OPT_TWOSTEP: 16 + 64 * 32}),
             ("shell2_by_axes", {**P, L.OPT_SHELL2: 2}), ("shell2_by_axes_ws4", {**P, L.OPT_SHELL2: 2, L.OPT_SHELL2_SHAPE: sw(ws=4)}),
             ("shell2_by_axes_ws8", {**P, L.OPT_SHELL2: 2, L.OPT_SHELL2_SHAPE: sw(ws=8)}), ("shell2_by_box", {**P, L.OPT_SHELL2: 3}),
             ("single", {L.OPT_TWOSTEP: 0}), ("shell2", S2)]
    if os.environ.get("PROBE_MODES"):
        keep = os.environ["PROBE_MODES"].split(",")
        modes = [m for m in modes if m[0] in keep]
    # the check: 12 steps from the same fields, shell2 pairs against single steps, bit for bit
    if not os.environ.get("PROBE_NO_CHECK"):
        ref = None
        for name, opts in [("single", {L.OPT_TWOSTEP: 0}), ("shell2", S2)]:
            e.reset()
            fill()
            for k, v in opts.items():
                e.set_option(k, v)
            st = e.run(12)
            got = [e.get_field(c) for c in range(6)]
            if ref is None:
                ref = got
            else:
                same = all(np.array_equal(a, b) for a, b in zip(got, ref))
                worst = max(float(np.abs(a - b).max()) for a, b in zip(got, ref))
                print(json.dumps({"check": "shell2 == single steps", "same_bits": bool(same), "worst": worst, "shell2_pairs": int(st.shell2_pairs),
                                  "field_max": max(float(np.abs(a).max()) for a in ref)}), flush=True)
        del ref, got
    fill()
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
                          "shell_pairs": int(st.shell_pairs), "shell2_pairs": int(st.shell2_pairs), "fused2_pairs": int(st.fused2_pairs), "shape": int(st.fused2_shape),
                          "why": int(st.fused2_off_reason),
                          "bulk_ms_per_launch": st.fused_kernel_ms / max(1, st.fused_kernel_launches), "bulk_launches": int(st.fused_kernel_launches),
                          "shell_ms_sum_per_pair": st.shell_kernel_ms / max(1, int(st.shell_pairs)), "shell_launches": int(st.shell_kernel_launches),
                          "stream_overlap": int(st.stream_overlap)}), flush=True)
