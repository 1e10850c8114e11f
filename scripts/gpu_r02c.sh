#!/bin/bash
# GPU visit r02c: where does the CPML-carrying sweep spend its time?  kernel trace of V2 / V1 / V0, PMC traffic of
# V0 and V2, SQ counters of V2; tile shapes for the split launch
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PROBE_CFGS='{"v2": [{"pml": -1}, {"pml": -1, "rows": 7}, {"pml": -1, "remap": 1}, {"pml": 0}], "v0": [{"rows": 3}, {"rows": 7}, {"rows": 7, "zc": 32}, {"rows": 15}], "v1": [{"rows": 3}, {"rows": 7}]}'
(timeout 300 python scripts/probe_r02.py 512 v2,v0,v1) > gpurun_out/probe_r02c.jsonl 2> gpurun_out/probe_c.err
cat gpurun_out/probe_r02c.jsonl
cd /tmp
for W in v2 v0; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -o trace -- python $R/bench.py --workload $W --steps 20 --warmup 3 --repeats 1 --no-cpu --no-workloads > $R/gpurun_out/prof_${W}_bench.json 2> $R/gpurun_out/prof_$W.err
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$W/pmc_$C -o pmc -- python $R/bench.py --workload $W --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads > /dev/null 2> $R/gpurun_out/pmc_${W}_$C.err
  done
  python $R/scripts/summarize_pmc.py $R/gpurun_out/pmc_$W > $R/gpurun_out/pmc_${W}_summary.json
done
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_v2/pmc_SQ -o pmc -- python $R/bench.py --workload v2 --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads > /dev/null 2> $R/gpurun_out/pmc_v2_SQ.err
cd $R
find gpurun_out -name '*kernel_trace*' -size +8M -delete
find gpurun_out -name '*counter_collection*' -size +8M -delete
cat gpurun_out/pmc_v0_summary.json gpurun_out/pmc_v2_summary.json
head -8 gpurun_out/prof_v2/trace_kernel_stats.csv | cut -c1-70,180-320
tail -2 gpurun_out/probe_c.err gpurun_out/pmc_v2_SQ.err
