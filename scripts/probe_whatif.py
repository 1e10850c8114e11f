#!/usr/bin/env python
"""Where the time of the two-step sweep goes: the WHAT-IF instantiations of fused2_step_kernel
(FDTD_OPT_WHATIF, csrc/fdtd_kernels2.hpp) timed INSIDE ONE ENGINE — same allocations, same clocks — alternating with the normal
sweep, `rounds` samples each of `steps` steps (20 pairs); prints one JSON line per variant: median / min / max ms per step of the
whole run and of the kernel (hipEvents around the launch), relative to the normal sweep of the same rounds.
    python scripts/probe_whatif.py [n] [rounds] [steps]"""
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

NAMES = {0: "normal", 1: "E_y / H_y not loaded", 2: "no second barrier", 3: "halo rows load nothing",
         4: "all plane loads hit one cached row (no HBM reads)", 5: "no field stores", 6: "no barriers, no LDS exchange",
         7: "loads + stores only (copy floor of this tiling)", 8: "no barriers (LDS traffic kept)"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    spec = bench.build_spec(n, 64, "v0")
    with HipEngine(spec, variant=L.VARIANT_FUSED) as e:
        for c in range(6):
            arr = np.empty((n, n, n), dtype=np.float32)
            for k in range(n):
                arr[k] = bench.init_plane(c, k, n)
            e.set_field(c, arr)
        e.run(20)                                   # placement probe, tile shape
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
