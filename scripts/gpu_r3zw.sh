#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zw; O=$R/gpurun_out/r3zw; cd $R
timeout 600 python scripts/probe_ab.py 512 v0,v1 SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=2064;OPT_TWOSTEP=2056" 3 > $O/ab.jsonl 2> $O/ab.err; cut -c1-330 $O/ab.jsonl
timeout 600 python scripts/probe_ab.py 512 v0 SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=2064;OPT_TWOSTEP=2056" 3 >> $O/ab.jsonl 2>> $O/ab.err; tail -1 $O/ab.jsonl | cut -c1-330
