#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3i
O=gpurun_out/r3i
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_production_path.py tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider -k "captured or tilted or film" 2>&1 | grep -E "graphs\]|tilted|config5|passed|failed|Error|assert " | tail -12) > $O/pytest_sel.log
cat $O/pytest_sel.log
