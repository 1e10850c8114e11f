#!/bin/bash
# GPU visit r02b: CPML placement with psi loads issued at the top of the plane (x-only instantiation for interior
# tiles, all-axes one for edge tiles), XCD remap on / off
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
export PROBE_CFGS='{"v2": [{"pml": -1}, {"pml": 0}, {"pml": 1}, {"pml": 6}, {"pml": -1, "remap": 0}, {"pml": 0, "remap": 0}, {"pml": 1, "remap": 0}, {"pml": -1, "zc": 8}, {"pml": -1, "zc": 32}, {"pml": -1, "rows": 2}],
 "v1": [{"remap": 1}, {"remap": 0}, {"remap": 0, "zc": 32}, {"remap": 0, "rows": 7}],
 "v0": [{"remap": 1}, {"remap": 0}, {"remap": 0, "zc": 32}, {"remap": 0, "rows": 7}, {"remap": 0, "rows": 2}]}'
(timeout 600 python scripts/probe_r02.py 512 v2,v1,v0) > gpurun_out/probe_r02b.jsonl 2> gpurun_out/probe_b.err
cat gpurun_out/probe_r02b.jsonl
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pml or media or configs or absorber or drude" 2>&1 | tail -5) > gpurun_out/pytest_gpu_b.log
cat gpurun_out/pytest_gpu_b.log
tail -3 gpurun_out/probe_b.err
