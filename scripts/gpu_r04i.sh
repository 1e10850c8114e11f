#!/bin/bash
# r04i: H-side x psi stored behind the E update instead of in the H phase (x-CPML instantiation)
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=128" 3 > gpurun_out/probe_r04i.jsonl 2> gpurun_out/probe_r04i.err
grep "^{" gpurun_out/probe_r04i.jsonl
