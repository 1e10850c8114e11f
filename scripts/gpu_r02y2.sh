#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python scripts/probe_ab.py 512 v2 OPT_MEM_HINTS 0,2 3 > gpurun_out/probe_r02y2.jsonl 2> gpurun_out/probe_r02y2.err
PROBE_AB_UNIQUE=1 timeout 900 python scripts/probe_ab.py 512 v2,v0 OPT_MEM_HINTS 0,1,2,3 3 >> gpurun_out/probe_r02y2.jsonl 2>> gpurun_out/probe_r02y2.err
PROBE_CFGS='{"*":[{},{"lib":"prev"}]}' timeout 900 python scripts/probe_r02.py 512 v2 >> gpurun_out/probe_r02y2.jsonl 2>> gpurun_out/probe_r02y2.err
cat gpurun_out/probe_r02y2.jsonl; tail -2 gpurun_out/probe_r02y2.err
