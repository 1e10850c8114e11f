#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zk; O=$R/gpurun_out/r3zk; cd $R
for n in 192 256 384; do timeout 300 python scripts/probe_twostep.py --n $n --steps 60 --rounds 3 0 auto >> $O/sizes.jsonl 2>> $O/err.log; done
python - <<'PY'
import json
for l in open("gpurun_out/r3zk/sizes.jsonl"):
    d=json.loads(l); print(d["n"], d["twostep"], f"{d['waves']}x{d['zchunk']}", d["ms_per_step"], d["gcells_per_s"])
PY
timeout 300 python -m pytest tests/test_gpu_production_path.py -q -m gpu -k "bit_identical_bench_v0 or config2" 2>&1 | tail -1
