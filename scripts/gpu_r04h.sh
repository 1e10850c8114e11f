#!/bin/bash
# r04h: CPML instantiations with global-address-space field loads (32); x-CPML instantiation with its E-side psi fetched behind the H phase (64)
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_MEM_HINTS=1;OPT_MEM_HINTS=32;OPT_MEM_HINTS=64" 3 > gpurun_out/probe_r04h.jsonl 2> gpurun_out/probe_r04h.err
grep "^{" gpurun_out/probe_r04h.jsonl
