#!/bin/bash
# round 3, visit n: GPU suite with the new parity cases (angled TFSF, PMC plus faces), bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3n
O=gpurun_out/r3n
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | tail -12) > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), d['config']['tile']['placement'], 'V2', round(d['workloads']['v2']['value']), d['workloads']['v2']['ms_per_step'])"
