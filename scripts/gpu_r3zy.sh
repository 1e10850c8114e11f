#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zy; O=$R/gpurun_out/r3zy; cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -s -m gpu -k "everything_at_once" > $O/pytest.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" $O/pytest.log | tail -12
