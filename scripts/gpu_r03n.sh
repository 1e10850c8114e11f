#!/bin/bash
# r03n: tile shapes on the slab one rank of an 8-GPU (and 4-GPU) run holds: 512 x 512 x 64 (128), inside engines
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03n.jsonl
S="OPT_ROWS=3,OPT_ZCHUNK=16;OPT_ROWS=3,OPT_ZCHUNK=8;OPT_ROWS=3,OPT_ZCHUNK=32;OPT_ROWS=3,OPT_ZCHUNK=4;OPT_ROWS=2,OPT_ZCHUNK=16;OPT_ROWS=2,OPT_ZCHUNK=8;OPT_ROWS=7,OPT_ZCHUNK=16;OPT_ROWS=3,OPT_ZCHUNK=22;OPT_ROWS=3,OPT_ZCHUNK=11"
PROBE_SLAB_NZ=64 timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 2 >> gpurun_out/probe_r03n.jsonl 2> gpurun_out/probe_r03n.err
PROBE_SLAB_NZ=128 timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 2 >> gpurun_out/probe_r03n.jsonl 2>> gpurun_out/probe_r03n.err
cat gpurun_out/probe_r03n.jsonl; tail -2 gpurun_out/probe_r03n.err
