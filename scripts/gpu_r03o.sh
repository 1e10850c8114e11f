#!/bin/bash
# r03o: rows per workgroup once more (4 and 6 were never measured), inside engines, new defaults
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03o.jsonl
S="OPT_ROWS=3,OPT_ZCHUNK=8;OPT_ROWS=4,OPT_ZCHUNK=8;OPT_ROWS=6,OPT_ZCHUNK=8;OPT_ROWS=7,OPT_ZCHUNK=8;OPT_ROWS=1,OPT_ZCHUNK=8"
timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 2 >> gpurun_out/probe_r03o.jsonl 2> gpurun_out/probe_r03o.err
timeout 600 python scripts/probe_ab_held.py 512 v1 "$S" 2 >> gpurun_out/probe_r03o.jsonl 2>> gpurun_out/probe_r03o.err
cat gpurun_out/probe_r03o.jsonl; tail -2 gpurun_out/probe_r03o.err
