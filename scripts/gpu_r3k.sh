#!/bin/bash
# round 3, visit k: placement probe on sweep PAIRS with equal data — does it pick well? device-only store paths parity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3k
O=gpurun_out/r3k
export TMPDIR=/tmp
timeout 900 python scripts/probe_placement2.py 512 v0 8 > $O/probe_placement_pairs.jsonl 2> $O/probe_placement_pairs.err
cat $O/probe_placement_pairs.jsonl
(timeout 900 python -m pytest tests/test_gpu_production_path.py -m gpu -q -p no:cacheprovider -k "store_hints or captured" 2>&1 | grep -E "passed|failed|Error|^E  " | tail -8) > $O/pytest_sel.log
cat $O/pytest_sel.log
for i in 1 2 3; do timeout 300 python bench.py --no-cpu --no-workloads --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench', round(d['value']), d['ms_per_step'], d['config']['tile']['placement'])"; done | tee $O/bench_3_processes.txt
