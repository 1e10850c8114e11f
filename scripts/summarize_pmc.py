"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) into per-kernel HBM
bytes per launch.  Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md
section HBM: the counters are in KiB; FETCH_SIZE under-reports wide (16 B/lane) coalesced reads by
exactly 2x on gfx950, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is uncalibrated (taken
at face value)."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for counter in [c for c in sys.argv[2:]] or ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void fdtd::", "").replace("fdtd::", "")
                if not name.startswith("fused_step_kernel"):       # keep <MAT, launch bounds, CPML axes> of the sweep
                    name = name.split("<")[0]
                acc[name][0] += float(row["Counter_Value"])
                acc[name][1] += 1
    for name, (tot, n) in acc.items():
        out.setdefault(name, {})[counter + "_KiB_per_launch"] = tot / max(n, 1)
        out[name]["launches_" + counter] = n
for name, d in out.items():
    fs, ws = d.get("FETCH_SIZE_KiB_per_launch"), d.get("WRITE_SIZE_KiB_per_launch")
    if fs is not None and ws is not None:
        d["hbm_bytes_per_launch"] = (2.0 * fs + ws) * 1024.0
        d["read_bytes_per_launch"] = 2.0 * fs * 1024.0
        d["write_bytes_per_launch"] = ws * 1024.0
print(json.dumps(out, indent=1))
