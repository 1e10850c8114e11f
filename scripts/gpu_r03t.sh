#!/bin/bash
# r03t: x tiles fastest in the tile order (MEM_HINTS=2 is the probe switch), inside engines
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03t.jsonl
for w in v0 v1 v2; do timeout 600 python scripts/probe_ab_held.py 512 $w "OPT_MEM_HINTS=1;OPT_MEM_HINTS=2;OPT_MEM_HINTS=2,OPT_XCD_REMAP=16;OPT_MEM_HINTS=2,OPT_XCD_REMAP=4;OPT_MEM_HINTS=1,OPT_XCD_REMAP=-1" 3 >> gpurun_out/probe_r03t.jsonl 2>> gpurun_out/probe_r03t.err; done
cat gpurun_out/probe_r03t.jsonl; tail -2 gpurun_out/probe_r03t.err
