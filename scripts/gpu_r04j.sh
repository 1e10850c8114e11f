#!/bin/bash
# r04j: H-side psi stored behind the E update: x-CPML instantiation only (129), all CPML instantiations (128)
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=129;OPT_MEM_HINTS=128" 3 > gpurun_out/probe_r04j.jsonl 2> gpurun_out/probe_r04j.err
grep "^{" gpurun_out/probe_r04j.jsonl
