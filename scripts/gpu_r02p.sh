#!/bin/bash
# r02p: placement sweep of the field arrays (plain sweep, 512^3)
cd /root/repo; mkdir -p gpurun_out
export LAYOUTS="$(cat scripts/layouts_a.json)"
timeout 900 python scripts/probe_layout.py 512 v0 > gpurun_out/probe_r02p.jsonl 2> gpurun_out/probe_r02p.err
cat gpurun_out/probe_r02p.jsonl; tail -3 gpurun_out/probe_r02p.err
