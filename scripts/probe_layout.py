"""GPU probe: the plain sweep (V0, 512^3) against the placement of the twelve field arrays in device memory
($FDTD_FIELD_LAYOUT = 3: one allocation, programmable distances — fdtd_capi.hip alloc_field_set).  One engine per
placement, created and closed in turn in ONE process (the allocation is released in between, so every placement
starts from the same free memory).  Needs a library built with the placements compiled in:
    FDTD_EXTRA_HIPCC_FLAGS=-DFDTD_PLACEMENT_PROBE python -m tidy3d_amd.build --force
Lines: {"s1": bytes added to the array stride, "s2": bytes between the sets, ...}"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wl = sys.argv[2] if len(sys.argv) > 2 else "v0"
cases = json.loads(os.environ["LAYOUTS"])
spec = bench.build_spec(n, 4000, wl)
rng = np.random.default_rng(1)
plane = [rng.uniform(-1e-3, 1e-3, (n, n)).astype(np.float32) for _ in range(4)]
arr = np.empty((n, n, n), dtype=np.float32)
for k in range(n):
    arr[k] = plane[k % 4]
held = []
for case in cases:
    for k_, v_ in (("FDTD_FIELD_LAYOUT", case.get("layout", 3)), ("FDTD_FIELD_S1", case.get("s1", 0)), ("FDTD_FIELD_S2", case.get("s2", 0)), ("FDTD_FIELD_S0", case.get("s0", 0)), ("FDTD_FIELD_ROUND", case.get("round", 1 << 30))):
        os.environ[k_] = str(v_)
    try:
        eng = HipEngine(spec)
        for c in range(6):
            eng.set_field(c, arr)
        eng.set_option(L.OPT_FLAGS, 0)
        eng.run(5)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); eng.run(20); ts.append(time.perf_counter() - t0)
        print(json.dumps({"wl": wl, **case, "ms_per_step": sorted(ts)[1] / 20 * 1e3}), flush=True)
        if case.get("hold"):
            held.append(eng)         # keep the memory: the next engine lands elsewhere
        else:
            eng.close()
    except Exception as e:       # noqa: BLE001
        print(json.dumps({"wl": wl, **case, "error": str(e)[:200]}), flush=True)
