#!/bin/bash
# r03e: grouped tile order on the CPML-carrying step (three launches) and with materials; finer G on the plain sweep
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03e.jsonl
S="OPT_XCD_REMAP=0;OPT_XCD_REMAP=1;OPT_XCD_REMAP=8;OPT_XCD_REMAP=16;OPT_XCD_REMAP=32"
timeout 600 python scripts/probe_ab_held.py 512 v2 "$S" 4 >> gpurun_out/probe_r03e.jsonl 2> gpurun_out/probe_r03e.err
timeout 600 python scripts/probe_ab_held.py 512 v1 "$S" 4 >> gpurun_out/probe_r03e.jsonl 2>> gpurun_out/probe_r03e.err
S="OPT_XCD_REMAP=6;OPT_XCD_REMAP=8;OPT_XCD_REMAP=12;OPT_XCD_REMAP=16;OPT_XCD_REMAP=19;OPT_XCD_REMAP=24;OPT_XCD_REMAP=32"
timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 4 >> gpurun_out/probe_r03e.jsonl 2>> gpurun_out/probe_r03e.err
cat gpurun_out/probe_r03e.jsonl; tail -2 gpurun_out/probe_r03e.err
