#!/bin/bash
# r3zl: the final two-step sweep (default shape) against single sweeps over grid sizes, one engine per size
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zl; O=$R/gpurun_out/r3zl; cd $R
for n in 128 192 200 256 320 384 448 512 640 768; do
  timeout 300 python scripts/probe_twostep.py --n $n --steps 60 --rounds 3 0 auto >> $O/sizes.jsonl 2>> $O/err.log
done
python - <<'PY'
import json
rows={}
for l in open("gpurun_out/r3zl/sizes.jsonl"):
    d=json.loads(l); rows.setdefault(d["n"],{})[d["twostep"]]=d
for n,r in rows.items():
    a,b=r[0],r[-1]; print(n, "single", a["gcells_per_s"], "two-step", b["gcells_per_s"], f"{b['waves']}x{b['zchunk']}", "x", round(a["ms_per_step"]/b["ms_per_step"],2))
PY
