#!/bin/bash
# round 3, visit g: captured step pairs (hipGraph) — parity and small-grid speed; the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
O=$R/gpurun_out/r3g
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_production_path.py -m gpu -q -x -s -p no:cacheprovider -k "captured" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30) > $O/pytest_graph.log
cat $O/pytest_graph.log
(timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15) > $O/pytest_gpu.log
cat $O/pytest_gpu.log
