"""GPU probe: do extra HIP streams in the process slow the two-stream z-slab schedule down (streams share hardware
queues beyond $GPU_MAX_HW_QUEUES, default 4)?  Creates k extra streams (each used once), then times the 64-plane slab
with the RCCL exchange looped back.   python scripts/probe_hw_queues.py <k extra streams>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
import probe_slab
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine

k = int(sys.argv[1])
extra = []
for i in range(k):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.zeros(16, device="cuda").add_(1)
    extra.append(s)
torch.cuda.synchronize()
n, nz = 512, 64
spec = probe_slab.spec_for(n, nz, 100000, True)
eng = HipEngine(spec, axis_shift=0, variant=L.VARIANT_FUSED, force_comm=True)
eng.comm_init(eng.unique_id())
rng = np.random.default_rng(0)
for c in range(6):
    eng.set_field(c, rng.uniform(-1e-3, 1e-3, (nz, n, n)).astype(np.float32))
eng.run(30)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.run(300); ts.append((time.perf_counter() - t0) / 300 * 1e3)
print(json.dumps({"extra_streams": k, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "ms_per_step": round(sorted(ts)[1], 4)}), flush=True)
eng.close()
