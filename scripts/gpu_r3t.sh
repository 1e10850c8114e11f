#!/bin/bash
# r3t: x neighbours by DPP wave shifts in the single-step sweep too (FDTD_OPT_MEM_HINTS = 2), A/B inside engines on V0 (two-step off),
# V1 (materials) and V2 (materials + CPML); same-bits check through the parity tests of the hint paths
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3t; O=$R/gpurun_out/r3t; cd $R
timeout 600 python scripts/probe_ab.py 512 v0,v1,v2 SETS "OPT_TWOSTEP=0,OPT_MEM_HINTS=1;OPT_TWOSTEP=0,OPT_MEM_HINTS=2" 4 > $O/ab_dpp.jsonl 2> $O/ab_dpp.err
cat $O/ab_dpp.jsonl | cut -c1-400
timeout 600 python scripts/probe_ab.py 512 v0,v2 SETS "OPT_TWOSTEP=0,OPT_MEM_HINTS=1;OPT_TWOSTEP=0,OPT_MEM_HINTS=2" 4 > $O/ab_dpp2.jsonl 2>> $O/ab_dpp.err
cat $O/ab_dpp2.jsonl | cut -c1-400
tail -n 3 $O/ab_dpp.err
