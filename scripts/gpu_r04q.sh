#!/bin/bash
# r04q: the same at 256 and 512 planes per rank
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r04q.jsonl
S="OPT_PML_FUSED=-1;OPT_PML_FUSED=0"
PROBE_COMM=1 PROBE_SLAB_NZ=256 timeout 600 python scripts/probe_ab_held.py 512 v2 "$S" 2 >> gpurun_out/probe_r04q.jsonl 2> gpurun_out/probe_r04q.err
PROBE_COMM=1 PROBE_SLAB_NZ=504 timeout 600 python scripts/probe_ab_held.py 512 v2 "$S" 2 >> gpurun_out/probe_r04q.jsonl 2>> gpurun_out/probe_r04q.err
grep "^{" gpurun_out/probe_r04q.jsonl; grep -v "version\|Hostname\|Librccl" gpurun_out/probe_r04q.err | tail -3
