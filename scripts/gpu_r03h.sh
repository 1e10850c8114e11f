#!/bin/bash
# r03h: field loads through global-address-space pointers (saddr form where the compiler finds it) against flat loads
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03h.jsonl
S="OPT_MEM_HINTS=1;OPT_MEM_HINTS=5;OPT_MEM_HINTS=0"
timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 3 >> gpurun_out/probe_r03h.jsonl 2> gpurun_out/probe_r03h.err
timeout 600 python scripts/probe_ab_held.py 512 v1 "$S" 3 >> gpurun_out/probe_r03h.jsonl 2>> gpurun_out/probe_r03h.err
cat gpurun_out/probe_r03h.jsonl; tail -2 gpurun_out/probe_r03h.err
