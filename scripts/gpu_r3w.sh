#!/bin/bash
# r3w: one seam kernel that reads only the scratch array (was: two kernels with strided reads of the field arrays, 6 % of a launch)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3w; O=$R/gpurun_out/r3w; cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -m gpu -k "two_steps_per_sweep or bench_v0" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python scripts/probe_twostep.py --steps 60 --rounds 3 0 auto $((16+64*32)) $((8+64*32)) > $O/ab512.jsonl 2> $O/ab.err
timeout 600 python scripts/probe_twostep.py --n 1024 --steps 20 --rounds 2 0 auto $((16+64*32)) $((8+64*32)) $((16+64*64)) > $O/ab1024.jsonl 2>> $O/ab.err
python - <<'PY'
import json
for f in ["ab512","ab1024"]:
    for l in open(f"gpurun_out/r3w/{f}.jsonl"):
        d=json.loads(l); print(f, d["twostep"], d["waves"], d["zchunk"], d["ms_per_step"], d["gcells_per_s"])
PY
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v0 -o trace -- python $R/bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu --no-workloads --placement-tries 0 > $O/prof_v0_bench.json 2> $O/prof_v0.err
cd $R; find gpurun_out/r3w -name '*kernel_trace*' -size +8M -delete
cut -d, -f1-4 $O/prof_v0/trace_kernel_stats.csv | cut -c1-50,220-300 | head -6
timeout 600 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), d['roofline'].get('two_steps_per_sweep'), d.get('single_steps'), 'V2', round(d['workloads']['v2']['value']), d['workloads']['v2']['ms_per_step'])"
