#!/bin/bash
# r03k: H stores ahead of the exchange (8), with the non-temporal hint (9), raised priority while a plane's loads go out (16/17)
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03k.jsonl
timeout 600 python scripts/probe_ab_held.py 512 v0 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=9;OPT_MEM_HINTS=17" 3 >> gpurun_out/probe_r03k.jsonl 2> gpurun_out/probe_r03k.err
timeout 600 python scripts/probe_ab_held.py 512 v1 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=9;OPT_MEM_HINTS=17" 3 >> gpurun_out/probe_r03k.jsonl 2>> gpurun_out/probe_r03k.err
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=8;OPT_MEM_HINTS=9;OPT_MEM_HINTS=16" 3 >> gpurun_out/probe_r03k.jsonl 2>> gpurun_out/probe_r03k.err
cat gpurun_out/probe_r03k.jsonl; tail -2 gpurun_out/probe_r03k.err
