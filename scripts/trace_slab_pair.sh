#!/bin/bash
# kernel timeline of ONE step pair of the 8-GPU per-rank proxies (scripts/probe_slab.py, 512 x 512 x 64, RCCL looped back):
# plain slab and slab with CPML on x / y — which stream sets the pace of a pair
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r6tr}; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for P in 0 2; do
  cd /tmp
  A=""; [ $P = 2 ] && A="--pml 2 --pml-fused 7"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slab_p$P -o trace -- python $R/scripts/probe_slab.py --slabs 8 --modes comm_fused --ref512 0 --steps 40 --warm 10 --twostep=-1 $A > $O/slab_p$P.jsonl 2> $O/slab_p$P.err
  cd $R
  python - <<PY
import csv, glob
f = glob.glob("$O/prof_slab_p$P/**/trace_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full pair: from the second-to-last fused2 launch to the last
idx = [i for i, r in enumerate(rows) if "fused2_step_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
out = open("$O/timeline_p$P.txt", "w")
for r in rows[a:b + 1]:
    line = f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f} us  q={r.get("Queue_Id","?")} grid={r.get("Grid_Size_X", r.get("Grid_Size","?"))} wg={r.get("Workgroup_Size_X","?")}x{r.get("Workgroup_Size_Y","?")} {r["Kernel_Name"][:70]}'
    print(line); out.write(line + "\n")
PY
  cut -c1-150 $O/prof_slab_p$P/trace_kernel_stats.csv | head -12 | tee $O/kernel_stats_p$P.txt
  find $O/prof_slab_p$P -name '*kernel_trace*' -delete
done
