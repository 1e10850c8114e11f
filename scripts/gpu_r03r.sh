#!/bin/bash
# r03r: CPML instantiations capped at 128 / 168 VGPRs (4 / 3 waves per SIMD, 104-128 bytes of scratch) vs 149 / 199 (3 / 2 waves)
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=64" 3 > gpurun_out/probe_r03r.jsonl 2> gpurun_out/probe_r03r.err
cat gpurun_out/probe_r03r.jsonl; tail -2 gpurun_out/probe_r03r.err
