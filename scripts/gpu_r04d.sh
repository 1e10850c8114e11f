#!/bin/bash
# r04d: kernel stats and PMC traffic of the CPML-carrying step (V2) on the final defaults
cd /root/repo; mkdir -p gpurun_out; R=/root/repo
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04d_prof_v2 -o trace -- python $R/scripts/probe_r02.py --child 512 v2 '[{}]' > /dev/null 2> $R/gpurun_out/r04d.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/r04d_pmc_v2/pmc_$C -o pmc -- python $R/scripts/probe_r02.py --child 512 v2 '[{}]' > /dev/null 2>> $R/gpurun_out/r04d.err
done
python $R/scripts/summarize_pmc.py $R/gpurun_out/r04d_pmc_v2 > $R/gpurun_out/r04d_pmc_v2_summary.json
cd $R
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04d_pmc_v2_summary.json'))
for k, v in d.items():
    if 'fused' in k: print(k, round(v.get('hbm_bytes_per_launch', 0) / 1e9, 3), 'GB', v.get('launches_FETCH_SIZE'))
PY
head -4 gpurun_out/r04d_prof_v2/trace_kernel_stats.csv | cut -c1-60,230-330
find gpurun_out -name '*kernel_trace*' -size +8M -delete
find gpurun_out -name '*counter_collection*' -size +4M -delete
