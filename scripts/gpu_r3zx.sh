#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zx; O=$R/gpurun_out/r3zx; cd $R
timeout 600 python scripts/probe_ab.py 512 v1 SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=2064;OPT_TWOSTEP=2056" 3 > $O/ab_v1.jsonl 2> $O/ab_v1.err; cut -c1-300 $O/ab_v1.jsonl
timeout 600 python -m pytest tests/test_gpu_production_path.py -q -m gpu -k "materials or everything_at_once" 2>&1 | tail -2
