#!/bin/bash
# r3x: small time monitors sampled inside the two-step sweep: config 2 (200^3 + probe recording every step) on pairs
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3x; O=$R/gpurun_out/r3x; cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py tests/test_gpu_parity.py -q -s -m gpu -k "two_steps_per_sweep or bench_v0 or config2" > $O/pytest.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" $O/pytest.log | tail -6
timeout 300 python scripts/probe_twostep.py --n 200 --steps 200 --rounds 3 0 auto $((8+64*8)) $((8+64*12)) $((8+64*16)) $((8+64*24)) $((16+64*12)) > $O/ab200.jsonl 2> $O/ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r3x/ab200.jsonl"):
    d=json.loads(l); print(d["n"], d["twostep"], d["waves"], d["zchunk"], d["ms_per_step"], d["gcells_per_s"])
PY
