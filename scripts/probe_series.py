"""GPU probe: is the step time of the plain sweep a property of the ENGINE (placement of its arrays) or of TIME (clock /
power state)?  Three engines in turn, 40 blocks of 20 steps each, the series printed; rocm-smi clocks in between."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine

n = 512
spec = bench.build_spec(n, 4000, "v0")
rng = np.random.default_rng(1)
arr = np.empty((n, n, n), dtype=np.float32)
pl = [rng.uniform(-1e-3, 1e-3, (n, n)).astype(np.float32) for _ in range(4)]
for k in range(n):
    arr[k] = pl[k % 4]


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        c = d[sorted(d)[0]]
        return {k: v for k, v in c.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "socclk", "power"))}
    except Exception as e:      # noqa: BLE001
        return {"error": str(e)[:100]}


print(json.dumps({"smi_idle": smi()}), flush=True)
held = []
for i in range(4):
    eng = HipEngine(spec)
    for c in range(6):
        eng.set_field(c, arr)
    eng.set_option(L.OPT_FLAGS, 0)
    eng.run(5)
    series = []
    for b in range(40):
        t0 = time.perf_counter(); eng.run(20); series.append(round((time.perf_counter() - t0) / 20 * 1e3, 3))
        if b == 20:
            p = subprocess.Popen(["rocm-smi", "--showclocks", "--showpower", "--json"], stdout=subprocess.PIPE, text=True)
            for _ in range(30):
                eng.run(20)
            out = p.communicate(timeout=30)[0]
            try:
                d = json.loads(out); cc = d[sorted(d)[0]]
                print(json.dumps({"smi_busy": {k: v for k, v in cc.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "socclk", "power"))}}), flush=True)
            except Exception as e:      # noqa: BLE001
                print(json.dumps({"smi_busy_error": str(e)[:100]}), flush=True)
    print(json.dumps({"engine": i, "series": series}), flush=True)
    if i % 2 == 0:
        held.append(eng)         # keep this one's memory: the next engine lands elsewhere
    else:
        eng.close()
