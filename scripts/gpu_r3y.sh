#!/bin/bash
# r3y: after the node-table rework (vote-based walk, sources of step n+1 and monitor records in / behind the sweep): bench + parity
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3y; O=$R/gpurun_out/r3y; cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py tests/test_gpu_parity.py -q -s -m gpu -k "two_steps_per_sweep or bench_v0 or config2" > $O/pytest.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" $O/pytest.log | tail -4
timeout 600 python scripts/probe_twostep.py --steps 60 --rounds 3 0 auto $((16+64*32)) > $O/ab512.jsonl 2> $O/ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r3y/ab512.jsonl"):
    d=json.loads(l); print(d["n"], d["twostep"], d["waves"], d["zchunk"], d["ms_per_step"], d["gcells_per_s"])
PY
timeout 600 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), d['roofline'].get('two_steps_per_sweep'), d.get('single_steps'), 'V2', round(d['workloads']['v2']['value']), d['workloads']['v2']['ms_per_step'])"
