#!/bin/bash
# r02w: non-temporal stores in the sweep (lib "ntst") against HEAD (lib "prev"): one process per variant, alternated;
# then HBM traffic of both (PMC)
cd /root/repo; mkdir -p gpurun_out; R=/root/repo
export TMPDIR=/tmp
export PROBE_CFGS='{"*":[{"lib":"prev"},{"lib":"ntst"},{"lib":"prev"},{"lib":"ntst"},{"lib":"prev"},{"lib":"ntst"}]}'
timeout 900 python scripts/probe_r02.py 512 v0,v1,v2 > gpurun_out/probe_r02w.jsonl 2> gpurun_out/probe_r02w.err
cut -c1-150 gpurun_out/probe_r02w.jsonl
cd /tmp
for V in prev ntst; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_w_$V/pmc_$C -o pmc -- python $R/scripts/probe_r02.py --child 512 v0 "[{\"lib\":\"$V\"}]" > /dev/null 2> $R/gpurun_out/pmc_w_${V}_$C.err
  done
  python $R/scripts/summarize_pmc.py $R/gpurun_out/pmc_w_$V > $R/gpurun_out/pmc_w_${V}_summary.json
  cat $R/gpurun_out/pmc_w_${V}_summary.json
done
cd $R
find gpurun_out -name '*counter_collection*' -size +4M -delete
