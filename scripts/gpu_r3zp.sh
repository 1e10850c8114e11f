#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zp; O=$R/gpurun_out/r3zp; cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -s -m gpu -k "absorber_layers or everything_at_once" > $O/pytest.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" $O/pytest.log | tail -6
timeout 600 python scripts/probe_ab.py 512 v4a SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=-1" 3 > $O/ab.jsonl 2> $O/ab.err; cut -c1-300 $O/ab.jsonl
bash scripts/gpu_r3zt.sh
