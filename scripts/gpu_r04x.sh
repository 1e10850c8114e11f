#!/bin/bash
# r04x: tile shapes of the CPML-carrying step once more on the final kernels, inside engines
cd /root/repo; mkdir -p gpurun_out
S="OPT_ROWS=3,OPT_ZCHUNK=16;OPT_ROWS=3,OPT_ZCHUNK=12;OPT_ROWS=3,OPT_ZCHUNK=20;OPT_ROWS=3,OPT_ZCHUNK=24;OPT_ROWS=2,OPT_ZCHUNK=16;OPT_ROWS=7,OPT_ZCHUNK=16;OPT_ROWS=3,OPT_ZCHUNK=10"
timeout 600 python scripts/probe_ab_held.py 512 v2 "$S" 3 > gpurun_out/probe_r04x.jsonl 2> gpurun_out/probe_r04x.err
grep "^{" gpurun_out/probe_r04x.jsonl
