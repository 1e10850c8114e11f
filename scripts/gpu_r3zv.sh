#!/bin/bash
# r3zv: the placement probe times the two-step sweep for the runs it covers: bench in five fresh processes (each its own placement lottery)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zv; O=$R/gpurun_out/r3zv; cd $R
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu --no-workloads > $O/bench_$i.json 2>> $O/err.log
  python -c "
import json; d=json.load(open('$O/bench_$i.json')); print($i, round(d['value']), round(d['ms_per_step'],4), d['config']['tile']['placement'], round(d['single_steps']['value']))"
done
timeout 300 python -m pytest tests/test_gpu_production_path.py -q -m gpu -k "two_steps_per_sweep_bit_identical_bench_v0" 2>&1 | tail -2
