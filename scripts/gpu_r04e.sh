#!/bin/bash
# r04e: finer z-chunks at 512^3 inside engines
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r04e.jsonl
S="OPT_ROWS=3,OPT_ZCHUNK=8;OPT_ROWS=3,OPT_ZCHUNK=4;OPT_ROWS=3,OPT_ZCHUNK=6;OPT_ROWS=3,OPT_ZCHUNK=12;OPT_ROWS=3,OPT_ZCHUNK=5"
for w in v0 v1; do timeout 600 python scripts/probe_ab_held.py 512 $w "$S" 3 >> gpurun_out/probe_r04e.jsonl 2>> gpurun_out/probe_r04e.err; done
grep "^{" gpurun_out/probe_r04e.jsonl
