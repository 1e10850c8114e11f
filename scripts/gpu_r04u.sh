#!/bin/bash
# r04u: hardware queues: extra streams in the process against the two-stream z-slab schedule; GPU_MAX_HW_QUEUES
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r04u.jsonl
for k in 0 1 2 3 4 6; do timeout 200 python scripts/probe_hw_queues.py $k >> gpurun_out/probe_r04u.jsonl 2>> gpurun_out/probe_r04u.err; done
for k in 0 2 3 4 6; do GPU_MAX_HW_QUEUES=8 timeout 200 python scripts/probe_hw_queues.py $k >> gpurun_out/probe_r04u.jsonl 2>> gpurun_out/probe_r04u.err; done
grep "^{" gpurun_out/probe_r04u.jsonl
