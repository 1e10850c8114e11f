#!/bin/bash
# GPU visit r02k: how much does the sweep care about occupancy?  extra LDS per workgroup lowers the workgroups per CU
# (4 waves each): 0 -> 4+, 30000 -> 3 (LDS-limited), 60000 -> 2, 100000 -> 1
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
export PROBE_CFGS='{"v0": [{"lds_pad": 0}, {"lds_pad": 30000}, {"lds_pad": 60000}, {"lds_pad": 100000}, {"lds_pad": 0}], "v1": [{"lds_pad": 0}, {"lds_pad": 30000}, {"lds_pad": 60000}], "v2": [{"lds_pad": 0}, {"lds_pad": 30000}, {"lds_pad": 60000}, {"pml": 1, "lds_pad": 0}, {"pml": 1, "lds_pad": 60000}]}'
(timeout 600 python scripts/probe_r02.py 512 v0,v1,v2) > gpurun_out/probe_r02k.jsonl 2> gpurun_out/probe_k.err
cat gpurun_out/probe_r02k.jsonl
tail -2 gpurun_out/probe_k.err
