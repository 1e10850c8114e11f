#!/bin/bash
# r03y: thickness of the boundary chunks of the z-slab schedule (RCCL looped back), inside engines: 64- and 128-plane slabs
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r03y.jsonl
S="OPT_BND_PLANES=16;OPT_BND_PLANES=8;OPT_BND_PLANES=4;OPT_BND_PLANES=2;OPT_BND_PLANES=1"
PROBE_COMM=1 PROBE_SLAB_NZ=64 timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 3 >> gpurun_out/probe_r03y.jsonl 2> gpurun_out/probe_r03y.err
PROBE_COMM=1 PROBE_SLAB_NZ=128 timeout 600 python scripts/probe_ab_held.py 512 v0 "$S" 2 >> gpurun_out/probe_r03y.jsonl 2>> gpurun_out/probe_r03y.err
PROBE_COMM=1 PROBE_SLAB_NZ=64 timeout 600 python scripts/probe_ab_held.py 512 v2 "$S" 2 >> gpurun_out/probe_r03y.jsonl 2>> gpurun_out/probe_r03y.err
grep "^{" gpurun_out/probe_r03y.jsonl; grep -v "version\|Hostname\|Librccl" gpurun_out/probe_r03y.err | tail -3
