#!/bin/bash
# GPU visit r02e: same-box A/B of two builds (prev = before the unconditional clamped loads), plain tile order
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
export PROBE_CFGS='{"*": [{"lib": "prev", "remap": 0}, {"remap": 0}, {"lib": "prev", "remap": 0}, {"remap": 0}, {"remap": 1}, {"lib": "prev", "remap": 1}]}'
(timeout 400 python scripts/probe_r02.py 512 v0,v1,v2) > gpurun_out/probe_r02e.jsonl 2> gpurun_out/probe_e.err
cat gpurun_out/probe_r02e.jsonl
tail -2 gpurun_out/probe_e.err
