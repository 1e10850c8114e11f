#!/bin/bash
# round 3, visit l: current tree — GPU suite, bench, config 2 and smaller cubes (launch overhead), set-up times
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3l
O=gpurun_out/r3l
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "\[mie|\[config|\[tilted|\[graphs|stream_overlap|passed|failed|Error|^E  " | tail -40) > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'traffic', d['roofline']['traffic'], d['config']['tile']['placement'], 'V2', round(d['workloads']['v2']['value']), d['workloads']['v2']['ms_per_step'], d['workloads']['v2']['traffic'])"
timeout 300 python scripts/probe_small.py 200,128,64 4000 > $O/small.jsonl 2> $O/small.err
cat $O/small.jsonl
timeout 300 python scripts/probe_small.py 152,224 2000 pml >> $O/small_pml.jsonl 2>> $O/small.err
cat $O/small_pml.jsonl
