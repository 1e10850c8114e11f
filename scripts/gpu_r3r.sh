#!/bin/bash
# r3r: two steps per sweep built in its own translation unit (SLP vectorizer off), default on: parity tests, the auto
# shape on smaller grids, shapes / tile orders at 512^3 inside one engine, bench
mkdir -p gpurun_out/r3r; O=gpurun_out/r3r
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -m gpu -k "two_steps_per_sweep or bench_v0" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for n in 128 192 256 320 384; do
  timeout 300 python scripts/probe_twostep.py --n $n --steps 100 --rounds 3 0 auto $((16+64*8)) $((16+64*16)) $((16+64*32)) $((8+64*8)) $((8+64*16)) $((8+64*32)) >> $O/small.jsonl 2>> $O/small.err
done
timeout 600 python scripts/probe_twostep.py --steps 60 --rounds 3 0 auto $((16+64*28)) $((16+64*36)) $((15+64*32)) $((13+64*32)) $((11+64*32)) $((8+64*32)) $((8+64*24)) > $O/shapes512.jsonl 2>> $O/shapes.err
for G in 4 16 32; do
  timeout 300 python scripts/probe_twostep.py --steps 60 --rounds 2 --opt OPT_XCD_REMAP=$G 0 auto $((8+64*32)) | sed "s/^{/{\"G\": $G, /" >> $O/order512.jsonl 2>> $O/shapes.err
done
python - <<'PY'
import json
for f in ["small","shapes512","order512"]:
    for l in open(f"gpurun_out/r3r/{f}.jsonl"):
        d=json.loads(l); print(f, d.get("G",""), d["n"], d["twostep"], d["waves"], d["zchunk"], d["ms_per_step"], d["gcells_per_s"])
PY
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), d['roofline'].get('two_steps_per_sweep'), d.get('single_steps'), 'V2', round(d['workloads']['v2']['value']))"
tail -2 $O/*.err
