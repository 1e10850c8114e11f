#!/bin/bash
# kernel timeline of the 8-GPU per-rank proxy (scripts/probe_slab.py): durations of the RCCL
# Send/Recv kernels, boundary / interior sweeps and the gaps between them
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_slab -o trace -- python $R/scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 60 > $R/gpurun_out/prof_slab.jsonl 2> $R/gpurun_out/prof_slab.err
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_slab/**/trace_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# last 4 steps worth of kernels
tail = rows[-40:]
for r in tail:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:12.1f} us  +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  q={r.get("Queue_Id","?")} grid={r.get("Grid_Size_X", r.get("Grid_Size","?"))} {r["Kernel_Name"][:60]}')
PY
cut -c1-120 gpurun_out/prof_slab/trace_kernel_stats.csv | head
find gpurun_out/prof_slab -name '*kernel_trace*' -delete
for zc in 4 8; do python scripts/probe_slab.py --slabs 8 --modes single_fused,comm_fused --zchunk $zc; done
