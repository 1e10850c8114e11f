#!/bin/bash
# tile order (XCD remap) against placement: 10 engines held, both orders inside each
cd /root/repo; mkdir -p gpurun_out
timeout 900 python scripts/probe_ab_held.py 512 v0 "OPT_XCD_REMAP=0;OPT_XCD_REMAP=1;OPT_XCD_REMAP=0,OPT_MEM_HINTS=0;OPT_XCD_REMAP=1,OPT_MEM_HINTS=0;OPT_MEM_HINTS=1" 10 > gpurun_out/probe_r03b.jsonl 2> gpurun_out/probe_r03b.err
cat gpurun_out/probe_r03b.jsonl; tail -2 gpurun_out/probe_r03b.err
