#!/bin/bash
# GPU visit r02h: small grids — one launch vs three for the CPML-carrying step, wider tile-shape probing
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python scripts/probe_small.py 200,128,64 2000) > gpurun_out/probe_small_h.jsonl 2> gpurun_out/probe_small_h.err
for S in 0 1; do (timeout 300 python scripts/probe_small.py 296,200,128 1000 pml $S) >> gpurun_out/probe_small_h.jsonl 2>> gpurun_out/probe_small_h.err; done
cat gpurun_out/probe_small_h.jsonl
export PROBE_CFGS='{"v2": [{"pml": -1}, {"pml": -1, "split": 0}]}'
tail -2 gpurun_out/probe_small_h.err
