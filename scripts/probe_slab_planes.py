import sys, json
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import numpy as np, time, torch
import probe_end_rank as p
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine
for nz in [int(x) for x in sys.argv[1].split(",")]:
    sp = p.sim_for(512, nz, "pec", "pec", 700)
    with HipEngine(sp, variant=L.VARIANT_FUSED) as e:
        rng = np.random.default_rng(0)
        for c in range(6):
            e.set_field(c, rng.uniform(-1e-3, 1e-3, tuple(reversed(sp.shape))).astype(np.float32))
        e.run(30)
        e.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
        st = e.run(100)
        e.set_option(L.OPT_FLAGS, 0)
        t0 = time.perf_counter(); st2 = e.run(200); dt = (time.perf_counter() - t0) / 200 * 1e3
        d = {k: (getattr(st, k) if not hasattr(getattr(st, k), "__len__") else list(getattr(st, k))) for k, _ in st._fields_ if k in ("fused2_pairs", "shell2_pairs", "fused2_shape", "fused_kernel_ms", "seam_kernel_ms", "tile_rows", "tile_zchunk", "fused2_off_reason", "shell2_shape")}
        print(json.dumps({"nz": nz, "ms_per_step": dt, "shape": [int(st2.fused2_shape) & 63, int(st2.fused2_shape) >> 6], **{k: (float(v) if not isinstance(v, list) else v) for k, v in d.items()}}), flush=True)
