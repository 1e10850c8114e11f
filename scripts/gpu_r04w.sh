#!/bin/bash
# r04w: the bench line with 4 (runtime default) and 8 hardware queues, alternated on one box
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/r04w_hwq.jsonl
for i in 1 2 3; do for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 60 --warmup 10 --repeats 3 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'hw_queues': $q, 'v0_ms': round(d['ms_per_step'],4), 'v2_ms': round(d['workloads']['v2']['ms_per_step'],4), 'placement': d['config']['tile']['placement']['kept']}))" >> gpurun_out/r04w_hwq.jsonl
done; done
cat gpurun_out/r04w_hwq.jsonl
