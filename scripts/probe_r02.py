"""GPU probe (round 2): the fused sweep across workloads (V0 plain, V1 materials, V2 materials + CPML),
tile shapes and CPML placements, one process per workload so a launch failure cannot poison the next.
Prints one JSON line per configuration:  python scripts/probe_r02.py [n] [workloads] [quick]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time
sys.path.insert(0, %r)
import numpy as np
import bench
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine
n = int(sys.argv[1]); wl = sys.argv[2]; cfgs = json.loads(sys.argv[3])
spec = bench.build_spec(n, 4000, wl)
eng = HipEngine(spec)
for c in range(6):
    arr = np.empty((n, n, n), dtype=np.float32)
    for k in range(n):
        arr[k] = np.random.default_rng(c * 100003 + k).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
    eng.set_field(c, arr)
for cfg in cfgs:
    try:
        eng.set_option(L.OPT_ROWS, cfg.get("rows", 3))
        eng.set_option(L.OPT_ZCHUNK, cfg.get("zc", 16))
        eng.set_option(L.OPT_PML_FUSED, cfg.get("pml", -1))
        eng.set_option(L.OPT_XCD_REMAP, cfg.get("remap", 1))
        eng.set_option(L.OPT_FLAGS, 0)
        eng.run(5)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); eng.run(20); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[1]
        eng.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
        st = eng.run(10)
        print(json.dumps({"wl": wl, "n": n, **cfg, "ms_per_step": t / 20 * 1e3, "gcells": n**3 * 20 / t / 1e9,
                          "fused_ms": st.fused_kernel_ms / max(1, st.fused_kernel_launches)}), flush=True)
    except Exception as e:
        print(json.dumps({"wl": wl, **cfg, "error": str(e)[:200]}), flush=True)
eng.close()
''' % ROOT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wls = sys.argv[2].split(",") if len(sys.argv) > 2 else ["v0", "v1", "v2"]
quick = len(sys.argv) > 3
shapes = [dict(rows=3, zc=16), dict(rows=7, zc=16), dict(rows=3, zc=32), dict(rows=7, zc=32), dict(rows=15, zc=16),
          dict(rows=5, zc=16), dict(rows=3, zc=16, remap=0)]
if quick:
    shapes = shapes[:2]
custom = json.loads(os.environ["PROBE_CFGS"]) if os.environ.get("PROBE_CFGS") else None
for wl in wls:
    cfgs = list(shapes)
    if wl in ("v2", "v3", "v4"):
        cfgs = [dict(c, pml=p) for p in (-1, 0) for c in shapes[:4]] + [dict(rows=3, zc=16, pml=6)]
    if custom is not None:
        cfgs = custom.get(wl, custom.get("*", cfgs))
    r = subprocess.run([sys.executable, "-c", CHILD, str(n), wl, json.dumps(cfgs)], capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    if r.returncode != 0:
        print(json.dumps({"wl": wl, "error": r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else "rc %d" % r.returncode}))
    sys.stdout.flush()
