#!/bin/bash
# r03l: occupancy sensitivity inside engines: LDS padding lowers the workgroups per CU (4 -> 3 -> 2)
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03l.jsonl
S="OPT_LDS_PAD=0;OPT_LDS_PAD=30000;OPT_LDS_PAD=60000"
for w in v0 v1 v2; do timeout 600 python scripts/probe_ab_held.py 512 $w "$S" 2 >> gpurun_out/probe_r03l.jsonl 2>> gpurun_out/probe_r03l.err; done
cat gpurun_out/probe_r03l.jsonl; tail -2 gpurun_out/probe_r03l.err
