#!/bin/bash
# r02n: same-box A/B with one process per variant; component skew of the field sets
cd /root/repo; mkdir -p gpurun_out
export PROBE_CFGS='{"*":[{"lib":"prev"},{},{"env":{"FDTD_FIELD_SKEW":1088}},{"env":{"FDTD_FIELD_SKEW":16448}},{"env":{"FDTD_FIELD_SKEW":263232}},{"lib":"prev"},{}]}'
timeout 1500 python scripts/probe_r02.py 512 v0,v2,v1 > gpurun_out/probe_r02n.jsonl 2> gpurun_out/probe_r02n.err
cat gpurun_out/probe_r02n.jsonl
