#!/bin/bash
# GPU visit r02d: effect of the instruction diet (clamped unconditional loads) on V0 / V1 / V2
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
export PROBE_CFGS='{"v2": [{"pml": -1}, {"pml": 0}, {"pml": 1}], "v0": [{"rows": 3}, {"rows": 7}], "v1": [{"rows": 3}]}'
(timeout 300 python scripts/probe_r02.py 512 v2,v0,v1) > gpurun_out/probe_r02d.jsonl 2> gpurun_out/probe_d.err
cat gpurun_out/probe_r02d.jsonl
tail -2 gpurun_out/probe_d.err
