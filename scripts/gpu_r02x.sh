#!/bin/bash
# r02x: non-temporal stores only in the plain instantiations ("ntst0"), plus non-temporal loads of E_y / H_y ("ntld"),
# against HEAD ("prev"): one process per variant, alternated; HBM traffic of ntld (PMC)
cd /root/repo; mkdir -p gpurun_out; R=/root/repo
export TMPDIR=/tmp
export PROBE_CFGS='{"*":[{"lib":"prev"},{"lib":"ntst0"},{"lib":"ntld"},{"lib":"prev"},{"lib":"ntst0"},{"lib":"ntld"},{"lib":"prev"},{"lib":"ntst0"},{"lib":"ntld"}]}'
timeout 1200 python scripts/probe_r02.py 512 v0,v1,v2 > gpurun_out/probe_r02x.jsonl 2> gpurun_out/probe_r02x.err
cut -c1-130 gpurun_out/probe_r02x.jsonl
cd /tmp
for V in ntld; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_x_$V/pmc_$C -o pmc -- python $R/scripts/probe_r02.py --child 512 v0 "[{\"lib\":\"$V\"}]" > /dev/null 2> $R/gpurun_out/pmc_x_${V}_$C.err
  done
  python $R/scripts/summarize_pmc.py $R/gpurun_out/pmc_x_$V > $R/gpurun_out/pmc_x_${V}_summary.json
  grep -A8 fused_step $R/gpurun_out/pmc_x_${V}_summary.json
done
cd $R
find gpurun_out -name '*counter_collection*' -size +4M -delete
