#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zo; O=$R/gpurun_out/r3zo; cd $R
timeout 600 python scripts/long_run_check.py 256 10001 > $O/long.log 2> $O/long.err; tail -2 $O/long.log; tail -2 $O/long.err
timeout 600 python scripts/long_run_check.py 320 4001 >> $O/long.log 2>> $O/long.err; tail -1 $O/long.log
