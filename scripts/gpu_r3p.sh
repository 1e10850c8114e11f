#!/bin/bash
# r3p: two time steps per sweep — parity on the device, then tile shapes (waves per workgroup x planes per chunk) on the
# bench workload, each in its own process (placement probe on: every line is the engine's best of 4 placements)
mkdir -p gpurun_out/r3p; O=gpurun_out/r3p
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -m gpu -k "two_steps_per_sweep" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in 0 16:32 16:16 16:64 12:32 12:16 8:32 8:64 10:32 14:32; do
  if [ $cfg = 0 ]; then v=0; else w=${cfg%%:*}; zc=${cfg##*:}; v=$((w + 64 * zc)); fi
  timeout 300 python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu --no-workloads --opt OPT_TWOSTEP=$v > $O/bench_$cfg.json 2> $O/bench_$cfg.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$cfg.json").read().strip().splitlines()[-1]); print("$cfg", d["value"], d["ms_per_step"], d["roofline"].get("avg_launch_ms"))
except Exception as e: print("$cfg", "failed", e)
PY
done
