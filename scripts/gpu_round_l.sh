#!/bin/bash
# GPU-box visit (rounds l, m): parity tests, bench, rocprofv3 kernel stats of the bench and of the feature probe.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "from tidy3d_amd import build; import sys; sys.exit(1 if build.needs_build() else 0)" || echo "WARNING: libfdtd_hip.so is stale"
(timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/pytest_gpu.log
(timeout 200 python bench.py --steps 100 --warmup 10) > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_feat -o trace -- python $R/scripts/probe_features.py 100 > $R/gpurun_out/probe_features.jsonl 2> $R/gpurun_out/prof_feat.err
cd $R
find gpurun_out -name '*kernel_trace*' -delete
cat gpurun_out/probe_features.jsonl
tail -5 gpurun_out/pytest_gpu.log
