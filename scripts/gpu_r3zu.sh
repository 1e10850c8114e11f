#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zu; O=$R/gpurun_out/r3zu; cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py tests/test_gpu_parity.py -q -m gpu -k "two_steps_per_sweep or bench_v0 or config2" 2>&1 | tail -2
timeout 600 python scripts/probe_ab.py 512 v0,v1 SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=-1" 3 > $O/ab.jsonl 2> $O/ab.err; cut -c1-300 $O/ab.jsonl
