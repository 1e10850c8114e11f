#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_series.py > gpurun_out/probe_r02q.jsonl 2> gpurun_out/probe_r02q.err
cat gpurun_out/probe_r02q.jsonl; tail -3 gpurun_out/probe_r02q.err
