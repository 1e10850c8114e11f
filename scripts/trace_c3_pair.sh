cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6q_trace -o c3 -- python $GRAFT_REPO_ROOT/scripts/probe_c3_paged.py 40 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r6q_trace/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find a window in the paged phase: look for src_fill kernels
idx = [i for i, r in enumerate(rows) if "src_fill_points" in r["Kernel_Name"]]
i0 = idx[len(idx) // 2]
# go back to the point_source kernel before
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0 - 2:i0 + 14]:
    n = r["Kernel_Name"].split("(")[0][-60:]
    print("%-62s q%s start %8.1f us  dur %8.1f us  grid %s" % (n, r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size")))
PY
rm -rf gpurun_out/r6q_trace
