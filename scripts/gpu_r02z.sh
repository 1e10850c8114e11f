#!/bin/bash
# r02z: cache hints as separately compiled instantiations, switched inside one engine (placement-free A/B)
cd /root/repo; mkdir -p gpurun_out
PROBE_AB_UNIQUE=1 timeout 900 python scripts/probe_ab.py 512 v0,v1,v2 OPT_MEM_HINTS 0,1,3 4 > gpurun_out/probe_r02z.jsonl 2> gpurun_out/probe_r02z.err
cat gpurun_out/probe_r02z.jsonl; tail -2 gpurun_out/probe_r02z.err
