#!/usr/bin/env python
"""Kernel timeline out of a rocprofv3 results .db (rocpd schema):  python scripts/trace_db.py FILE.db [last N dispatches]
Prints per-kernel statistics and the last N dispatches (start relative to the first of them, duration, stream / queue, grid)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select * from kernels order by start").fetchall()
ix = {c: i for i, c in enumerate(cols)}


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("fdtd::", "")[-60:]


stats = {}
for r in rows:
    n = short(r[ix["name"]])
    d = (r[ix["end"]] - r[ix["start"]]) / 1e3
    s = stats.setdefault(n, [0, 0.0, 1e30, 0.0])
    s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
print(f"{'kernel':60s} {'calls':>6s} {'total us':>10s} {'avg us':>9s} {'min':>8s} {'max':>8s}")
for n, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:60s} {s[0]:6d} {s[1]:10.1f} {s[1] / s[0]:9.2f} {s[2]:8.2f} {s[3]:8.2f}")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if N:
    tail = rows[-N:]
    t0 = tail[0][ix["start"]]
    qk = "queue_id" if "queue_id" in ix else None
    sk = "stream_id" if "stream_id" in ix else None
    for r in tail:
        g = [r[ix[k]] for k in ("grid_x", "workgroup_x", "workgroup_y") if k in ix]
        print(f"{(r[ix['start']] - t0) / 1e3:9.1f} +{(r[ix['end']] - r[ix['start']]) / 1e3:8.1f} us  q={r[ix[qk]] if qk else '-'} s={r[ix[sk]] if sk else '-'} "
              f"grid={g}  {short(r[ix['name']])}")
