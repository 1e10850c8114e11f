#!/bin/bash
# r03q: edge tiles of the CPML step on the instantiation that carries just their axes (x+y / x+z, 3 waves per SIMD) vs the
# all-axes one (2 waves) on every edge tile, inside engines
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_ab_held.py 512 v2 "OPT_PML_SPLIT=1;OPT_PML_SPLIT=2" 3 > gpurun_out/probe_r03q.jsonl 2> gpurun_out/probe_r03q.err
cat gpurun_out/probe_r03q.jsonl; tail -2 gpurun_out/probe_r03q.err
timeout 300 python scripts/probe_c3.py 200 >> gpurun_out/probe_r03q.jsonl 2>> gpurun_out/probe_r03q.err
tail -1 gpurun_out/probe_r03q.jsonl
