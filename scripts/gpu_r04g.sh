#!/bin/bash
# r04g: x block of the CPML parameters in registers (no per-plane scalar re-loads of its pointers; SGPR spills instead)
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=16" 3 > gpurun_out/probe_r04g.jsonl 2> gpurun_out/probe_r04g.err
grep "^{" gpurun_out/probe_r04g.jsonl
