#!/bin/bash
# GPU visit r02j: isolate the three changes of r02i on one box: fma (explicit FMAs in the CPML arithmetic), wall (wall
# zeroing after the corrections), unist (scalar-base stores); the product library is HEAD (= prev)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
export PROBE_CFGS='{"*": [{}, {"lib": "fma"}, {"lib": "wall"}, {"lib": "unist"}, {}, {"lib": "fma"}, {"lib": "wall"}, {"lib": "unist"}]}'
(timeout 600 python scripts/probe_r02.py 512 v2,v1,v0) > gpurun_out/probe_r02j.jsonl 2> gpurun_out/probe_j.err
cat gpurun_out/probe_r02j.jsonl
tail -2 gpurun_out/probe_j.err
