#!/bin/bash
# tile shapes with the non-temporal stores, inside one engine
cd /root/repo; mkdir -p gpurun_out
S="OPT_ROWS=3,OPT_ZCHUNK=16;OPT_ROWS=7,OPT_ZCHUNK=16;OPT_ROWS=5,OPT_ZCHUNK=16;OPT_ROWS=3,OPT_ZCHUNK=32;OPT_ROWS=7,OPT_ZCHUNK=32;OPT_ROWS=3,OPT_ZCHUNK=16,OPT_XCD_REMAP=0;OPT_ROWS=3,OPT_ZCHUNK=16,OPT_XCD_REMAP=1"
PROBE_AB_UNIQUE=1 timeout 900 python scripts/probe_ab.py 512 v0,v1 SETS "$S" 3 > gpurun_out/probe_r03a.jsonl 2> gpurun_out/probe_r03a.err
cat gpurun_out/probe_r03a.jsonl; tail -2 gpurun_out/probe_r03a.err
