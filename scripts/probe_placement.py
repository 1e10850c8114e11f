"""GPU probe: what the placement probe of the library buys (fdtd_capi.hip probe_placement): engines created one after the
other and all kept alive; each logs the time of the probe sweeps on its first allocations and on the set it kept, and
its measured step time.   python scripts/probe_placement.py <n> <workload> <engines> <tries>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine

n = int(sys.argv[1]); wl = sys.argv[2]; n_eng = int(sys.argv[3]); tries = int(sys.argv[4])
rng = np.random.default_rng(1)
arr = np.empty((n, n, n), dtype=np.float32)
pl = [rng.uniform(-1e-3, 1e-3, (n, n)).astype(np.float32) for _ in range(8)]
for k in range(n):
    arr[k] = pl[k % 8]
spec = bench.build_spec(n, 100000, wl)
held = []
for i in range(n_eng):
    eng = HipEngine(spec)
    held.append(eng)
    eng.set_option(L.OPT_PLACEMENT_TRIES, tries)
    for c in range(6):
        eng.set_field(c, np.roll(arr, c, axis=0))
    eng.set_option(L.OPT_FLAGS, 0)
    st = eng.run(10)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.run(40); ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print(json.dumps({"wl": wl, "engine": i, "tries": tries, "tried": int(st.placement) >> 8, "kept": int(st.placement) & 255,
                      "probe_ms_per_sweep_first": st.placement_ms_first / 3, "probe_ms_per_sweep_kept": st.placement_ms_kept / 3,
                      "ms_per_step": round(sorted(ts)[1], 4)}), flush=True)
