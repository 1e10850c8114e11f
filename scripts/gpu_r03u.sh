#!/bin/bash
# r03u: placement probe of the library: 6 engines held, 3 candidates each; then bench
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_placement.py 512 v0 6 3 > gpurun_out/probe_r03u.jsonl 2> gpurun_out/probe_r03u.err
timeout 600 python scripts/probe_placement.py 512 v2 4 3 >> gpurun_out/probe_r03u.jsonl 2>> gpurun_out/probe_r03u.err
cat gpurun_out/probe_r03u.jsonl; tail -2 gpurun_out/probe_r03u.err
(timeout 400 python bench.py --steps 100 --warmup 10) > gpurun_out/r03u_bench.json 2> gpurun_out/r03u_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r03u_bench.json')); print(d['value'], d['ms_per_step'], d['config']['tile'], d['workloads']['v2']['value'])"
