#!/bin/bash
# r04o: y-edge tile rows on an x+y instantiation (2 waves per SIMD like the all-axes one, fewer instructions); LDS_PAD=4 is the probe switch
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_LDS_PAD=0;OPT_LDS_PAD=4" 3 > gpurun_out/probe_r04o.jsonl 2> gpurun_out/probe_r04o.err
grep "^{" gpurun_out/probe_r04o.jsonl
