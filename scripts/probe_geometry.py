"""GPU probe: which (rows, zchunk) launch geometries run, and how fast (each in a subprocess so a
launch failure cannot poison the next one)."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time
sys.path.insert(0, %r)
import numpy as np
import bench
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine
n, rows, zc, variant, remap = map(int, sys.argv[1:6]); lb = int(sys.argv[6])
spec = bench.build_spec(n, 64, "v0")
eng = HipEngine(spec, variant=variant, z_chunk=zc)
eng.set_option(L.OPT_ROWS, rows)
eng.set_option(L.OPT_XCD_REMAP, 1)
eng.set_option(L.OPT_FUSED_LB, lb)
rng = np.random.default_rng(0)
for c in range(6):
    eng.set_field(c, rng.uniform(-1e-3, 1e-3, (n, n, n)).astype(np.float32))
eng.run(3)
t0 = time.perf_counter(); st = eng.run(20); t = time.perf_counter() - t0
print(json.dumps({"n": n, "rows": rows, "zchunk": zc, "variant": variant, "lb": lb, "mcells": n**3*20/t/1e6, "run_ms": st.run_ms}))
''' % ROOT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for variant in (3,):
  for lb in (0,):
    for rows in (2, 3, 4, 5):
      for zc in (8, 12, 16, 24):
            remap = 1
            r = subprocess.run([sys.executable, "-c", CHILD, str(n), str(rows), str(zc), str(variant), str(remap), str(lb)],
                               capture_output=True, text=True)
            out = r.stdout.strip().splitlines()
            print(out[-1] if out else json.dumps({"rows": rows, "zchunk": zc, "error": r.stderr.strip().splitlines()[-1][:200]}), flush=True)
