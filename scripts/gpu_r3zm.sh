#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zm; O=$R/gpurun_out/r3zm; cd $R
timeout 600 python scripts/probe_ab.py 512 v0 SETS "OPT_XCD_REMAP=8;OPT_XCD_REMAP=16;OPT_XCD_REMAP=32;OPT_XCD_REMAP=64;OPT_XCD_REMAP=0;OPT_XCD_REMAP=40" 4 > $O/ab_order.jsonl 2> $O/ab.err; cut -c1-600 $O/ab_order.jsonl
timeout 600 python scripts/probe_ab.py 512 v0 SETS "OPT_XCD_REMAP=8;OPT_XCD_REMAP=32;OPT_XCD_REMAP=40" 4 >> $O/ab_order.jsonl 2>> $O/ab.err; tail -1 $O/ab_order.jsonl | cut -c1-400
