#!/bin/bash
# r3z: materials in the two-step sweep: bench V1 (dielectric sphere, 93 media) bit-identical to single sweeps, speed
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3z; O=$R/gpurun_out/r3z; cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -s -m gpu -k "two_steps_per_sweep" > $O/pytest.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" $O/pytest.log | tail -8
timeout 600 python scripts/probe_ab.py 512 v1 SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=-1;OPT_TWOSTEP=2064;OPT_TWOSTEP=2056" 3 > $O/ab_v1.jsonl 2> $O/ab.err; cut -c1-400 $O/ab_v1.jsonl
