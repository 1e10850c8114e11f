#!/bin/bash
# r02v: large arrays at size-aligned virtual addresses (aligned_device_alloc) against plain hipMalloc
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r02v.jsonl; : > gpurun_out/probe_r02v.err
A='{"layout":0,"hold":1}'; L="$A"; for i in $(seq 1 7); do L="$L,$A"; done
for al in 1 0; do
  echo "{\"FDTD_ALIGNED_ALLOC\":$al, \"part\":\"8 engines held, plain sweep\"}" >> gpurun_out/probe_r02v.jsonl
  FDTD_ALIGNED_ALLOC=$al LAYOUTS="[$L]" timeout 300 python scripts/probe_layout.py 512 v0 >> gpurun_out/probe_r02v.jsonl 2>> gpurun_out/probe_r02v.err
done
echo '{"part":"one process per line"}' >> gpurun_out/probe_r02v.jsonl
export PROBE_CFGS='{"*":[{"env":{"FDTD_ALIGNED_ALLOC":1}},{"env":{"FDTD_ALIGNED_ALLOC":0}},{"env":{"FDTD_ALIGNED_ALLOC":1}},{"env":{"FDTD_ALIGNED_ALLOC":0}},{"env":{"FDTD_ALIGNED_ALLOC":1}},{"env":{"FDTD_ALIGNED_ALLOC":0}}]}'
timeout 900 python scripts/probe_r02.py 512 v0,v1,v2 >> gpurun_out/probe_r02v.jsonl 2>> gpurun_out/probe_r02v.err
cut -c1-170 gpurun_out/probe_r02v.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
