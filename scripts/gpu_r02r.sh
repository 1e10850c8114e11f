#!/bin/bash
# r02r: placement sweep inside ONE pool allocation (same physical memory for every case)
cd /root/repo; mkdir -p gpurun_out
export LAYOUTS="$(cat scripts/layouts_b.json)"
timeout 900 python scripts/probe_layout.py 512 v0 > gpurun_out/probe_r02r.jsonl 2> gpurun_out/probe_r02r.err
cat gpurun_out/probe_r02r.jsonl; tail -3 gpurun_out/probe_r02r.err
