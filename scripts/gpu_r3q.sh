#!/bin/bash
# r3q: two steps per sweep, variants inside one engine: default build vs -fno-slp-vectorize; waves x planes; prefetch
mkdir -p gpurun_out/r3q; O=gpurun_out/r3q
PF=65536
V="0 $((16+64*32)) $((16+64*24)) $((16+64*40)) $((14+64*32)) $((12+64*32)) $((12+64*32+PF)) $((8+64*32)) $((8+64*32+PF)) $((8+64*16+PF))"
timeout 600 python scripts/probe_twostep.py --steps 60 --rounds 3 $V > $O/ab_default.jsonl 2> $O/ab_default.err
timeout 600 python scripts/probe_twostep.py --lib tidy3d_amd/libfdtd_hip_noslp.so --steps 60 --rounds 3 $V > $O/ab_noslp.jsonl 2> $O/ab_noslp.err
python - <<'PY'
import json
for f in ["ab_default","ab_noslp"]:
    for l in open(f"gpurun_out/r3q/{f}.jsonl"):
        d=json.loads(l); print(f, d["waves"], d["zchunk"], d["prefetch"], d["ms_per_step"], d["gcells_per_s"], d["all"])
PY
tail -3 $O/*.err
