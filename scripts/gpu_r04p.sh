#!/bin/bash
# r04p: z-slab schedule with the CPML inside the sweeps (default) against slab kernels (PML_FUSED=0), inside engines, RCCL looped back
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r04p.jsonl
S="OPT_PML_FUSED=0;OPT_PML_FUSED=7;OPT_PML_FUSED=6"
PROBE_COMM=1 PROBE_SLAB_NZ=64 timeout 600 python scripts/probe_ab_held.py 512 v2 "$S" 3 >> gpurun_out/probe_r04p.jsonl 2> gpurun_out/probe_r04p.err
PROBE_COMM=1 PROBE_SLAB_NZ=128 timeout 600 python scripts/probe_ab_held.py 512 v2 "$S" 2 >> gpurun_out/probe_r04p.jsonl 2>> gpurun_out/probe_r04p.err
grep "^{" gpurun_out/probe_r04p.jsonl; grep -v "version\|Hostname\|Librccl" gpurun_out/probe_r04p.err | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "exchange or slab or rccl or comm" 2>&1 | tail -3
