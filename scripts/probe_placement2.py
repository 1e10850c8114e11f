"""GPU probe: does the placement probe (fdtd_capi.hip probe_placement: pairs of sweeps on equal data, round 3) pick well?
For each of <engines> engines created one after the other (each in the state the previous ones left the allocator in):
step time on the FIRST allocations (probe off), then the probe runs on the same engine, then the step time again.
    python scripts/probe_placement2.py <n> <workload> <engines>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine

n = int(sys.argv[1]); wl = sys.argv[2]; n_eng = int(sys.argv[3])
rng = np.random.default_rng(1)
arr = np.empty((n, n, n), dtype=np.float32)
pl = [rng.uniform(-1e-3, 1e-3, (n, n)).astype(np.float32) for _ in range(8)]
for k in range(n):
    arr[k] = pl[k % 8]
spec = bench.build_spec(n, 100000, wl)
held = []


def steady(eng):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.run(40); ts.append((time.perf_counter() - t0) / 40 * 1e3)
    return round(sorted(ts)[1], 4)


for i in range(n_eng):
    eng = HipEngine(spec)
    held.append(eng)
    eng.set_option(L.OPT_PLACEMENT_TRIES, 0)
    for c in range(6):
        eng.set_field(c, np.roll(arr, c, axis=0))
    eng.run(10)
    before = steady(eng)
    eng.set_option(L.OPT_PLACEMENT_TRIES, 3)
    st = eng.run(10)
    after = steady(eng)
    print(json.dumps({"wl": wl, "engine": i, "ms_per_step_first_allocations": before, "ms_per_step_after_probe": after,
                      "tried": int(st.placement) >> 8, "kept": int(st.placement) & 255,
                      "probe_ms_per_sweep_first": st.placement_ms_first / 4, "probe_ms_per_sweep_kept": st.placement_ms_kept / 4}), flush=True)
    if i % 2 == 1:
        held.pop(0).close()          # let the allocator's state vary
