#!/bin/bash
# GPU visit r02l: uniform material coefficients from the LDS table instead of dependent scalar loads (same-box A/B)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
export PROBE_CFGS='{"*": [{"lib": "prev"}, {}, {"lib": "prev"}, {}, {"lds_pad": 30000}]}'
(timeout 600 python scripts/probe_r02.py 512 v1,v2,v0) > gpurun_out/probe_r02l.jsonl 2> gpurun_out/probe_l.err
cat gpurun_out/probe_r02l.jsonl
tail -2 gpurun_out/probe_l.err
