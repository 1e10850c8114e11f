#!/bin/bash
# GPU visit r02f: full GPU test suite on the current library (decay parity, config 3 purity 1e-5, Mie 2 %), bench line,
# kernel trace + PMC traffic of the headline workload
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "from tidy3d_amd import build; import sys; sys.exit(1 if build.needs_build() else 0)" || echo "WARNING: libfdtd_hip.so is stale"
(timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | tail -60) > gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
(timeout 400 python bench.py --steps 100 --warmup 10) > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v0 -o trace -- python $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu --no-workloads > $R/gpurun_out/prof_v0_bench.json 2> $R/gpurun_out/prof_v0.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_v0/pmc_$C -o pmc -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads > /dev/null 2> $R/gpurun_out/pmc_v0_$C.err
done
python $R/scripts/summarize_pmc.py $R/gpurun_out/pmc_v0 > $R/gpurun_out/pmc_v0_summary.json
cd $R
find gpurun_out -name '*kernel_trace*' -size +8M -delete
find gpurun_out -name '*counter_collection*' -size +8M -delete
tail -3 gpurun_out/bench.err
