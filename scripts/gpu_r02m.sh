#!/bin/bash
# GPU visit r02m: same-box A/B: prev (HEAD), ldslut (uniform material coefficients from LDS), cur (+ CPML parameter
# block read once into locals)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
export PROBE_CFGS='{"*": [{"lib": "prev"}, {"lib": "ldslut"}, {}, {"lib": "prev"}, {"lib": "ldslut"}, {}]}'
(timeout 600 python scripts/probe_r02.py 512 v2,v1,v0) > gpurun_out/probe_r02m.jsonl 2> gpurun_out/probe_m.err
cat gpurun_out/probe_r02m.jsonl
tail -2 gpurun_out/probe_m.err
