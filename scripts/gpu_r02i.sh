#!/bin/bash
# GPU visit r02i: same-box A/B — prev (HEAD before), cur (explicit FMAs in the CPML arithmetic, wall zeroing after the
# corrections, scalar-base stores), oldst (the same with per-lane 64-bit store addresses)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
export PROBE_CFGS='{"*": [{"lib": "prev"}, {}, {"lib": "oldst"}, {"lib": "prev"}, {}, {"lib": "oldst"}]}'
(timeout 500 python scripts/probe_r02.py 512 v2,v1,v0) > gpurun_out/probe_r02i.jsonl 2> gpurun_out/probe_i.err
cat gpurun_out/probe_r02i.jsonl
tail -2 gpurun_out/probe_i.err
