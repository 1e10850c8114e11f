#!/bin/bash
# GPU visit r02a: parity tests on the rewritten sweep, tile-shape / CPML-placement probe, bench, kernel trace of V2
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "from tidy3d_amd import build; import sys; sys.exit(1 if build.needs_build() else 0)" || echo "WARNING: libfdtd_hip.so is stale"
(timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40) > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
(timeout 400 python scripts/probe_r02.py 512 v0,v1,v2) > gpurun_out/probe_r02.jsonl 2> gpurun_out/probe.err
cat gpurun_out/probe_r02.jsonl
(timeout 300 python bench.py --steps 100 --warmup 10) > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v2 -o trace -- python $R/bench.py --workload v2 --steps 20 --warmup 3 --repeats 1 --no-cpu > $R/gpurun_out/prof_v2_bench.json 2> $R/gpurun_out/prof_v2.err
cd $R
find gpurun_out -name '*kernel_trace*' -size +8M -delete
tail -3 gpurun_out/bench.err gpurun_out/probe.err
