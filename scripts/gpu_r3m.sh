#!/bin/bash
# round 3, visit m: after the memset / copy ordering fix in the placement probe — the 512^3 tests three times, GPU suite, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3m
O=gpurun_out/r3m
export TMPDIR=/tmp
for i in 1 2 3; do
(timeout 900 python -m pytest tests/test_gpu_production_path.py -m gpu -q -x -p no:cacheprovider -k "bench_v2 or bench_v0 or three_launch" 2>&1 | grep -E "passed|failed|^E   " | tail -4) >> $O/pytest_512_x3.log
done
cat $O/pytest_512_x3.log
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|^E  " | tail -10) > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), d['config']['tile']['placement'], 'V2', round(d['workloads']['v2']['value']), d['workloads']['v2']['ms_per_step'])"
