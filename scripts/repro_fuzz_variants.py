"""Re-run ONE case of scripts/fuzz_variants.py (case index, seed, FUZZ_BIG as in the failing run) and say what differs between
the plain fused run and the slab-rank runs: which field components, which monitors, where.
    FUZZ_BIG=1 python scripts/repro_fuzz_variants.py 71 77 [repeats]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_variants as F  # noqa: E402
from tidy3d_amd import lib as L  # noqa: E402

target, seed = int(sys.argv[1]), int(sys.argv[2])
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
big = bool(int(os.environ.get("FUZZ_BIG", "0")))
rng = np.random.default_rng(seed)
for q in range(target + 1):
    disc, steps, per_z, desc, split, rows, zc, so, po = F.draw(rng, big)
print(desc, "split", split, "rows", rows, "zc", zc, flush=True)
for m in disc.spec.monitors:
    print("  monitor", m.name, m.kind, "lo", m.lo, "hi", m.hi, "steps", list(m.steps[:6]), "...", len(m.steps))
sl = F.inside(disc)
ref_f, ref_m = F.run(disc, steps, split, L.VARIANT_FUSED, None, rows=rows, zc=zc)
for rep in range(repeats):
    for name, var, comm, opts in (("fused", L.VARIANT_FUSED, False, None), ("fused_split", L.VARIANT_FUSED, False, {L.OPT_PML_SPLIT: 1}),
                                  ("fused_slab", L.VARIANT_FUSED, True, so), ("fused_slab_pairs", L.VARIANT_FUSED, True, po),
                                  ("two_pass_slab", L.VARIANT_ZMARCH, True, None)):
        if comm and not per_z:
            continue
        f, m = F.run(disc, steps, split, var, None, comm=comm, opts=opts)
        bad_f = [c for c, (a, b) in enumerate(zip(ref_f, f)) if not np.array_equal(a[sl], b[sl])]
        bad_m = [k for k in ref_m if not np.array_equal(ref_m[k], m[k])]
        print(rep, name, "fields differ:", bad_f, "monitors differ:", bad_m, flush=True)
        for k in bad_m:
            d = np.argwhere(ref_m[k] != m[k])
            print("    ", k, ref_m[k].shape, "n diff", len(d), "first", d[:4].tolist(), "last", d[-2:].tolist(),
                  "max |diff|", float(np.abs(ref_m[k] - m[k]).max()), "of", float(np.abs(ref_m[k]).max()))
        for c in bad_f:
            d = np.argwhere(ref_f[c] != f[c])
            print("     field", c, "n diff", len(d), "z", d[:, 0].min(), d[:, 0].max(), "y", d[:, 1].min(), d[:, 1].max(), "x", d[:, 2].min(), d[:, 2].max())
