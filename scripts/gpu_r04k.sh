#!/bin/bash
# r04k: with the H-side psi stored late: non-temporal field stores in the CPML instantiations once more (129), inside engines; config-3-like
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=129;OPT_MEM_HINTS=0" 3 > gpurun_out/probe_r04k.jsonl 2> gpurun_out/probe_r04k.err
grep "^{" gpurun_out/probe_r04k.jsonl
timeout 300 python scripts/probe_c3.py 200 >> gpurun_out/probe_r04k.jsonl 2>> gpurun_out/probe_r04k.err; tail -1 gpurun_out/probe_r04k.jsonl
