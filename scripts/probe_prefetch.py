#!/usr/bin/env python
"""The PREFETCH instantiations of the two-step sweep (FDTD_OPT_WHATIF = 10 ... 12, csrc/fdtd_kernels2.hpp: part of the next plane
travels global memory -> LDS by LDS-DMA while this plane is computed) against the normal sweep INSIDE ONE ENGINE: first the bits
(same start fields, `check_steps` steps, all six arrays compared), then the time — alternating, `rounds` samples of `steps` steps.
    python scripts/probe_prefetch.py [n] [rounds] [steps]"""
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

NAMES = {0: "normal", 10: "H_x H_y H_z through LDS", 11: "E_y H_y through LDS", 12: "E_x E_y E_z through LDS",
         13: "E_x of the row above from the wave above (ninth exchange array), no DMA", 14: "the sweep without it (round-6 start)",
         15: "13 + E_z through LDS (one global load per row and plane, half a plane ahead; E1 single-buffered)"}
if os.environ.get("PF_VARIANTS"):
    NAMES = {0: "normal", **{int(v): NAMES.get(int(v), f"variant {v}") for v in os.environ["PF_VARIANTS"].split(",")}}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    check_steps = 24
    spec = bench.build_spec(n, 64, "v0")
    start = [np.stack([bench.init_plane(c, k, n) for k in range(n)]).astype(np.float32) for c in range(6)]
    with HipEngine(spec, variant=L.VARIANT_FUSED) as e:
        for c in range(6):
            e.set_field(c, start[c])
        if os.environ.get("PF_TWOSTEP"):             # W + 64 * planes per chunk
            e.set_option(L.OPT_TWOSTEP, int(os.environ["PF_TWOSTEP"]))
        e.run(20)                                   # placement probe, tile shape
        ref = None
        for w in NAMES:
            e.reset()
            for c in range(6):
                e.set_field(c, start[c])
            e.set_option(L.OPT_WHATIF, w)
            st = e.run(check_steps)
            got = [e.get_field(c) for c in range(6)]
            if ref is None:
                ref = got
                print(json.dumps({"n": n, "whatif": w, "pairs": int(st.fused2_pairs), "max_abs": float(max(np.abs(f).max() for f in got))}), flush=True)
            else:
                same = [bool(np.array_equal(a, b)) for a, b in zip(got, ref)]
                print(json.dumps({"n": n, "whatif": w, "what": NAMES[w], "pairs": int(st.fused2_pairs), "bit_identical_to_normal": all(same),
                                  "arrays": same}), flush=True)
        t_run = {w: [] for w in NAMES}
        t_ker = {w: [] for w in NAMES}
        for r in range(rounds):
            order = list(NAMES) if r % 2 == 0 else list(NAMES)[::-1]
            for w in order:
                e.set_option(L.OPT_WHATIF, w)
                e.set_option(L.OPT_FLAGS, 0)
                e.run(4)
                t0 = time.perf_counter()
                e.run(steps)
                t_run[w].append((time.perf_counter() - t0) / steps * 1e3)
                e.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
                st = e.run(10)
                t_ker[w].append(st.fused_kernel_ms / max(1, st.fused_kernel_launches) / 2)
        e.set_option(L.OPT_WHATIF, 0)
        base = float(np.median(t_run[0]))
        for w in NAMES:
            a, k = np.array(t_run[w]), np.array(t_ker[w])
            print(json.dumps({"n": n, "whatif": w, "what": NAMES[w], "ms_per_step_median": float(np.median(a)), "min": float(a.min()),
                              "max": float(a.max()), "kernel_ms_per_step_median": float(np.median(k)), "kernel_min": float(k.min()),
                              "kernel_max": float(k.max()), "vs_normal": float(np.median(a)) / base, "samples": rounds,
                              "steps_per_sample": steps}), flush=True)


if __name__ == "__main__":
    main()
