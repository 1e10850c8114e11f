import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine
n = 512; V = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
spec = bench.build_spec(n, 64, "v0")
start = [np.stack([bench.init_plane(c, k, n) for k in range(n)]).astype(np.float32) for c in range(6)]
with HipEngine(spec, variant=L.VARIANT_FUSED) as e:
    for c in range(6): e.set_field(c, start[c])
    e.run(20)
    ref = None
    for w in [0] + [V] * reps + [0]:
        e.reset()
        for c in range(6): e.set_field(c, start[c])
        e.set_option(L.OPT_WHATIF, w)
        st = e.run(steps)
        got = [e.get_field(c) for c in range(6)]
        if ref is None:
            ref = got; continue
        for c in range(6):
            d = np.argwhere(got[c] != ref[c])
            print("variant", w, "comp", c, "mismatches", len(d), flush=True)
            if len(d):
                ks, js, is_ = d[:, 0], d[:, 1], d[:, 2]
                print("  k range", ks.min(), ks.max(), "k%32 hist", np.bincount(ks % 32, minlength=32).tolist())
                print("  j range", js.min(), js.max(), "j%13 hist", np.bincount(js % 13, minlength=13).tolist())
                print("  i range", is_.min(), is_.max())
                print("  first", d[:3].tolist(), "maxdiff", float(np.abs(got[c] - ref[c]).max()))
