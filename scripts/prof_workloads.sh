#!/bin/bash
# rocprofv3 kernel stats of the PML / material / ADE workloads
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for W in v1 v2 v3; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -o trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --workload $W > $R/gpurun_out/bench_$W.json 2> $R/gpurun_out/bench_$W.err
  find $R/gpurun_out/prof_$W -name '*kernel_trace*' -delete
  echo "== $W"; cut -c1-110 $R/gpurun_out/prof_$W/trace_kernel_stats.csv | head -12
  python -c "import json; d=json.load(open('$R/gpurun_out/bench_$W.json')); print(d['value'], d['ms_per_step'])"
done
