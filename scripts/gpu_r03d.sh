#!/bin/bash
# r03d: tile orders (plain, contiguous XCD split, runs of G tiles per XCD) inside each of 8 engines held side by side
cd /root/repo; mkdir -p gpurun_out
S="OPT_XCD_REMAP=0;OPT_XCD_REMAP=1;OPT_XCD_REMAP=2;OPT_XCD_REMAP=4;OPT_XCD_REMAP=8;OPT_XCD_REMAP=16;OPT_XCD_REMAP=43;OPT_XCD_REMAP=171"
timeout 900 python scripts/probe_ab_held.py 512 v0 "$S" 8 > gpurun_out/probe_r03d.jsonl 2> gpurun_out/probe_r03d.err
cat gpurun_out/probe_r03d.jsonl; tail -2 gpurun_out/probe_r03d.err
