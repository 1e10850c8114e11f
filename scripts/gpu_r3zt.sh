#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zt; O=$R/gpurun_out/r3zt; cd $R
timeout 1200 python scripts/fuzz_twostep.py 60 1 > $O/fuzz.log 2> $O/fuzz.err; tail -4 $O/fuzz.log; grep -c MISMATCH $O/fuzz.log; tail -3 $O/fuzz.err
