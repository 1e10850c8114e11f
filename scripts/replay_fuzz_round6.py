#!/usr/bin/env python
"""One case of scripts/fuzz_round6.py again (same seed, same draw), with where the arrays differ:
    python scripts/replay_fuzz_round6.py <seed> <case index> [disp] [paged]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_round6 as F  # noqa: E402

seed, q_want = int(sys.argv[1]), int(sys.argv[2])
disp = int(sys.argv[3]) if len(sys.argv) > 3 else -1
paged = int(sys.argv[4]) if len(sys.argv) > 4 else -1
rng = np.random.default_rng(seed)
for q in range(q_want + 1):
    disc, steps, w, zc, what = F.case(rng)
    split = int(rng.integers(0, steps))
print(f"case {q}: N={disc.spec.shape} {what} steps={steps} split={split} W={w} zc={zc} media={len(disc.spec.media)}", flush=True)
ref = F.run(disc, steps, 0, split, None, seed=q)
for rep in range(2):
    got = F.run(disc, steps, w + 64 * zc, split, None, seed=q, disp=disp, paged=paged)
    print("pairs / disp / paged", got[2])
    for c in range(6):
        d = np.argwhere(ref[0][c] != got[0][c])
        if len(d):
            print("  comp", c, "mismatches", len(d), "k", d[:, 0].min(), d[:, 0].max(), "j", d[:, 1].min(), d[:, 1].max(), "i", d[:, 2].min(), d[:, 2].max(),
                  "first", d[0].tolist(), "maxdiff", float(np.abs(ref[0][c] - got[0][c]).max()))
    for k in ref[1]:
        if not np.array_equal(ref[1][k], got[1][k]):
            print("  monitor", k, "differs")
sp = disc.spec
for m in sp.monitors:
    if getattr(m, "name", "") in ref[1] and not np.array_equal(ref[1][m.name], got[1][m.name]):
        a, b = np.asarray(ref[1][m.name]), np.asarray(got[1][m.name])
        d = np.argwhere(a != b)
        print("monitor", m.name, type(m).__name__, {k: getattr(m, k) for k in ("lo", "hi", "box", "comps", "components", "steps", "interval", "kind") if hasattr(m, k)})
        print("  shape", a.shape, "differing entries", len(d), "first", d[:6].tolist(), "ref", a[tuple(d[0])], "got", b[tuple(d[0])], "maxdiff", float(np.abs(a - b).max()), "max", float(np.abs(a).max()))
        print("  differing time indices", sorted(set(d[:, 0].tolist()))[:40] if a.ndim > 1 else d[:10].tolist())
print("sources:", [(type(s).__name__, getattr(s, "name", None)) for s in getattr(sp, "sources", [])][:6], "n_media", len(sp.media), "shape", sp.shape)
