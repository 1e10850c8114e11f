#!/bin/bash
# round 3, visit c: pooled x-CPML with prefetch (A/B inside one engine), Mie with the predicted shift, Au film vs Airy
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3c
O=gpurun_out/r3c
export TMPDIR=/tmp
timeout 600 python scripts/probe_ab.py 512 v2 OPT_PML_POOL 0,1 4 > $O/probe_pml_pool_v2.jsonl 2> $O/probe_pml_pool_v2.err
cat $O/probe_pml_pool_v2.jsonl
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_production_path.py -m gpu -q -x -s -p no:cacheprovider -k "mie or film or three_launch or bench_v2" 2>&1 | grep -E "mie|config5|passed|failed|Error|assert" | tail -20) > $O/pytest_sel.log
cat $O/pytest_sel.log
