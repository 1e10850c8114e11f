#!/bin/bash
# r3zs: absorber layers inside the two-step sweep: parity on the device, speed on 512^3 open problems (40 layers x 6 faces)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zs; O=$R/gpurun_out/r3zs; cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -s -m gpu -k "absorber_layers or everything_at_once" > $O/pytest.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" $O/pytest.log | tail -5
timeout 600 python scripts/probe_ab.py 512 va,v1a SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=2064;OPT_TWOSTEP=2056" 3 > $O/ab_abs.jsonl 2> $O/ab.err; cut -c1-330 $O/ab_abs.jsonl
timeout 600 python scripts/probe_ab.py 512 v0 SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=-1" 3 >> $O/ab_abs.jsonl 2>> $O/ab.err; tail -1 $O/ab_abs.jsonl | cut -c1-300
