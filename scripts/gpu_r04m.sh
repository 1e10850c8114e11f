#!/bin/bash
# r04m: deferred E stores in the CPML instantiations too (384 = late psi_H + deferred E stores)
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=384" 3 > gpurun_out/probe_r04m.jsonl 2> gpurun_out/probe_r04m.err
grep "^{" gpurun_out/probe_r04m.jsonl
