#!/bin/bash
# r04t: final tree: GPU suite, bench, kernel trace, PMC traffic
set -x
cd /root/repo; mkdir -p gpurun_out; R=/root/repo
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/r04t_pytest_gpu.log
tail -6 gpurun_out/r04t_pytest_gpu.log
(timeout 400 python bench.py --steps 100 --warmup 10) > gpurun_out/r04t_bench.json 2> gpurun_out/r04t_bench.err
cat gpurun_out/r04t_bench.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04t_prof_v0 -o trace -- python $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu --no-workloads > $R/gpurun_out/r04t_prof_v0_bench.json 2> $R/gpurun_out/r04t_prof_v0.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/r04t_pmc_v0/pmc_$C -o pmc -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads > /dev/null 2> $R/gpurun_out/r04t_pmc_v0_$C.err
done
python $R/scripts/summarize_pmc.py $R/gpurun_out/r04t_pmc_v0 > $R/gpurun_out/r04t_pmc_v0_summary.json
grep -A8 fused_step $R/gpurun_out/r04t_pmc_v0_summary.json
cd $R
find gpurun_out -name '*kernel_trace*' -size +8M -delete
find gpurun_out -name '*counter_collection*' -size +4M -delete
tail -3 gpurun_out/r04t_bench.err
