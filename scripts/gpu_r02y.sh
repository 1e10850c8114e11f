#!/bin/bash
# r02y: cache hints of the sweep, switched inside one engine (placement-free A/B)
cd /root/repo; mkdir -p gpurun_out
timeout 900 python scripts/probe_ab.py 512 v0,v1,v2 OPT_MEM_HINTS 0,1,3,2 4 > gpurun_out/probe_r02y.jsonl 2> gpurun_out/probe_r02y.err
cat gpurun_out/probe_r02y.jsonl; tail -2 gpurun_out/probe_r02y.err
