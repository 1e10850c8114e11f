#!/bin/bash
# r03z: tile shapes of the z-slab schedule (RCCL looped back, boundary chunks of 2 planes), inside engines
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03z.jsonl
S="OPT_ROWS=3,OPT_ZCHUNK=8;OPT_ROWS=3,OPT_ZCHUNK=4;OPT_ROWS=3,OPT_ZCHUNK=16;OPT_ROWS=3,OPT_ZCHUNK=10;OPT_ROWS=3,OPT_ZCHUNK=6;OPT_ROWS=2,OPT_ZCHUNK=8;OPT_ROWS=7,OPT_ZCHUNK=8"
PROBE_COMM=1 PROBE_SLAB_NZ=64 timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 2 >> gpurun_out/probe_r03z.jsonl 2> gpurun_out/probe_r03z.err
PROBE_COMM=1 PROBE_SLAB_NZ=64 timeout 600 python scripts/probe_ab_held.py 512 v2 "$S" 2 >> gpurun_out/probe_r03z.jsonl 2>> gpurun_out/probe_r03z.err
PROBE_COMM=1 PROBE_SLAB_NZ=128 timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 2 >> gpurun_out/probe_r03z.jsonl 2>> gpurun_out/probe_r03z.err
grep "^{" gpurun_out/probe_r03z.jsonl
