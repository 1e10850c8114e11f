"""Per-kernel averages of the SQ counter passes of scripts/gpu_visit.sh stage `sq` (rocprofv3 --pmc, one csv per pass)."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for w in sorted(glob.glob(os.path.join(root, "sq_*"))):
    if not os.path.isdir(w):
        continue
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(w, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].replace("void fdtd::", "").replace("fdtd::", "").split("<")[0]
                a = acc[name][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    res = {}
    for name, cs in acc.items():
        d = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        d["launches"] = max(v[1] for v in cs.values())
        if d.get("SQ_WAVE_CYCLES"):
            d["wait_frac"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        if d.get("SQ_BUSY_CYCLES") and d.get("SQ_ACTIVE_INST_VALU"):
            d["valu_busy_frac"] = d["SQ_ACTIVE_INST_VALU"] / (4.0 * d["SQ_BUSY_CYCLES"])
        res[name] = d
    out[os.path.basename(w)] = res
print(json.dumps(out, indent=1))
