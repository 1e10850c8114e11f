#!/bin/bash
# r3zq: DFT monitors recording on first steps of pairs (H terms from the sweep's copy of H^{n+1/2}): parity, a complete open scattering problem
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zq; O=$R/gpurun_out/r3zq; cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -s -m gpu -k "absorber_layers or everything_at_once or two_steps_per_sweep_bit_identical_bench_v0" > $O/pytest.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" $O/pytest.log | tail -7
timeout 600 python scripts/probe_ab.py 512 v4a,v0 SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=-1" 3 > $O/ab.jsonl 2> $O/ab.err; cut -c1-300 $O/ab.jsonl
