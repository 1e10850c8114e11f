"""GPU probe: A/B of option sets inside each of several engines that are all kept alive (each has its own placement in
device memory): does the better setting depend on the placement?
    python scripts/probe_ab_held.py <n> <workload> "<set>;<set>;..." <engines>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine

n = int(sys.argv[1]); wl = sys.argv[2]; sets = sys.argv[3].split(";"); n_eng = int(sys.argv[4])
rng = np.random.default_rng(1)
arr = np.empty((n, n, n), dtype=np.float32)
pl = [rng.uniform(-1e-3, 1e-3, (n, n)).astype(np.float32) for _ in range(8)]
for k in range(n):
    arr[k] = pl[k % 8]
nz = int(os.environ.get("PROBE_SLAB_NZ", n))                    # a z-slab of the cube: what one rank of an N-GPU run holds
if nz != n:
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import probe_slab
    spec = probe_slab.spec_for(n, nz, 100000, wl == "v2")
    arr = arr[:nz]
else:
    spec = bench.build_spec(n, 100000, wl)
held = []
for i in range(n_eng):
    if os.environ.get("PROBE_COMM"):                         # the z-slab schedule with RCCL looped back onto the one rank
        eng = HipEngine(spec, axis_shift=0, variant=L.VARIANT_FUSED, force_comm=True)
        eng.comm_init(eng.unique_id())
    else:
        eng = HipEngine(spec, axis_shift=0) if nz != n else HipEngine(spec)
    held.append(eng)
    for c in range(6):
        eng.set_field(c, np.roll(arr, c, axis=0))
    eng.set_option(L.OPT_FLAGS, 0)
    eng.run(10)
    res = {s_: [] for s_ in sets}
    for _ in range(2):
        for s_ in sets:
            for kv in s_.split(","):
                k_, x_ = kv.split("=")
                eng.set_option(getattr(L, k_), int(x_))
            eng.run(4)
            reps = 40 * max(1, n // nz)
            t0 = time.perf_counter(); eng.run(reps); res[s_].append(round((time.perf_counter() - t0) / reps * 1e3, 4))
    print(json.dumps({"wl": wl, "engine": i, "ms_per_step": res}), flush=True)
