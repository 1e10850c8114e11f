#!/bin/bash
# r02o: one process per variant.  prev = HEAD (91091dd kernels), product = uniform-row coefficients from the LDS table;
# field layouts ($FDTD_FIELD_LAYOUT, fdtd_capi.hip) on the plain sweep
cd /root/repo; mkdir -p gpurun_out
export PROBE_CFGS='{"v0":[{"lib":"prev"},{"env":{"FDTD_FIELD_LAYOUT":1}},{"env":{"FDTD_FIELD_LAYOUT":2}},{"lib":"prev"},{"env":{"FDTD_FIELD_LAYOUT":1}},{"env":{"FDTD_FIELD_LAYOUT":2}}], "*":[{"lib":"prev"},{},{"lib":"prev"},{},{"lib":"prev"},{}]}'
timeout 1500 python scripts/probe_r02.py 512 v0,v2,v1 > gpurun_out/probe_r02o.jsonl 2> gpurun_out/probe_r02o.err
cat gpurun_out/probe_r02o.jsonl
