#!/bin/bash
# GPU visit r02g: BASELINE configs 3 / 4 / 5 with their printed numbers, config-3-like and small-grid throughput,
# kernel trace of the small grid (launch-bound regime)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "config" 2>&1) > gpurun_out/configs.log
grep -E "^\[|passed|failed|best agreement" gpurun_out/configs.log
(timeout 200 python scripts/probe_c3.py 200) > gpurun_out/probe_c3.json 2> gpurun_out/probe_c3.err
cat gpurun_out/probe_c3.json
(timeout 300 python scripts/probe_small.py 200,128,64 2000) > gpurun_out/probe_small.jsonl 2> gpurun_out/probe_small.err
(timeout 300 python scripts/probe_small.py 200,128 2000 pml) >> gpurun_out/probe_small.jsonl 2>> gpurun_out/probe_small.err
cat gpurun_out/probe_small.jsonl
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -o trace -- python $R/scripts/probe_small.py 200 500 > /dev/null 2> $R/gpurun_out/prof_small.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small_pml -o trace -- python $R/scripts/probe_small.py 200 500 pml > /dev/null 2>> $R/gpurun_out/prof_small.err
cd $R
find gpurun_out -name '*kernel_trace*' -size +8M -delete
head -12 gpurun_out/prof_small/trace_kernel_stats.csv | cut -c1-60,150-280
head -12 gpurun_out/prof_small_pml/trace_kernel_stats.csv | cut -c1-60,150-280
