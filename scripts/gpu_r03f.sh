#!/bin/bash
# r03f: tile shapes under the new default order (runs of 8 tiles per XCD), inside engines held side by side
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03f.jsonl
S="OPT_ROWS=3,OPT_ZCHUNK=16;OPT_ROWS=7,OPT_ZCHUNK=16;OPT_ROWS=3,OPT_ZCHUNK=8;OPT_ROWS=3,OPT_ZCHUNK=32;OPT_ROWS=7,OPT_ZCHUNK=32;OPT_ROWS=2,OPT_ZCHUNK=16;OPT_ROWS=3,OPT_ZCHUNK=64"
timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 3 >> gpurun_out/probe_r03f.jsonl 2> gpurun_out/probe_r03f.err
timeout 600 python scripts/probe_ab_held.py 512 v1 "$S" 3 >> gpurun_out/probe_r03f.jsonl 2>> gpurun_out/probe_r03f.err
S="OPT_ROWS=3,OPT_ZCHUNK=16;OPT_ROWS=3,OPT_ZCHUNK=8;OPT_ROWS=3,OPT_ZCHUNK=32;OPT_ROWS=3,OPT_ZCHUNK=16,OPT_PML_SPLIT=0;OPT_ROWS=3,OPT_ZCHUNK=16,OPT_PML_SPLIT=1,OPT_PML_FUSED=6;OPT_ROWS=3,OPT_ZCHUNK=16,OPT_PML_FUSED=0;OPT_ROWS=3,OPT_ZCHUNK=16,OPT_PML_FUSED=7"
timeout 600 python scripts/probe_ab_held.py 512 v2 "$S" 3 >> gpurun_out/probe_r03f.jsonl 2>> gpurun_out/probe_r03f.err
cat gpurun_out/probe_r03f.jsonl; tail -2 gpurun_out/probe_r03f.err
