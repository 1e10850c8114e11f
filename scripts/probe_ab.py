"""GPU probe: A/B of run-time options INSIDE one engine — same arrays, same physical pages, same code object — the only
comparison that is free of the placement effect.  Settings are cycled `rounds` times.

    python scripts/probe_ab.py <n> <workloads> <option name> <v1,v2,...> [rounds]     e.g.  512 v0,v1,v2 OPT_MEM_HINTS 0,1
    python scripts/probe_ab.py <n> <workloads> SETS "OPT_ROWS=3,OPT_MEM_HINTS=1;OPT_ROWS=7,OPT_MEM_HINTS=1" [rounds]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine

n = int(sys.argv[1]); wls = sys.argv[2].split(",")
if sys.argv[3] == "SETS":
    values = [s_ for s_ in sys.argv[4].split(";")]
    def apply(eng, v):
        for kv in v.split(","):
            k_, x_ = kv.split("=")
            eng.set_option(getattr(L, k_), int(x_))
else:
    values = [int(v) for v in sys.argv[4].split(",")]
    def apply(eng, v):
        eng.set_option(getattr(L, sys.argv[3]), v)
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 4
rng = np.random.default_rng(1)
pl = [rng.uniform(-1e-3, 1e-3, (n, n)).astype(np.float32) for _ in range(4)]
arr = np.empty((n, n, n), dtype=np.float32)
for k in range(n):
    arr[k] = pl[k % 4]
for wl in wls:
    spec = bench.build_spec(n, 100000, wl)
    eng = HipEngine(spec)
    for c in range(6):
        if os.environ.get("PROBE_AB_UNIQUE"):       # every plane of every component its own noise (as scripts/probe_r02.py)
            for k in range(n):
                arr[k] = np.random.default_rng(c * 100003 + k).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
        eng.set_field(c, arr)
    eng.set_option(L.OPT_FLAGS, 0)
    eng.run(10)
    res = {v: [] for v in values}
    for _ in range(rounds):
        for v in values:
            apply(eng, v)
            eng.run(4)
            t0 = time.perf_counter(); eng.run(40); res[v].append(round((time.perf_counter() - t0) / 40 * 1e3, 4))
    print(json.dumps({"wl": wl, "n": n, "option": sys.argv[3], "ms_per_step": {str(v): res[v] for v in values}}), flush=True)
    eng.close()
