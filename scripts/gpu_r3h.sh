#!/bin/bash
# round 3, visit h: tile shapes on the 64-plane slab of an 8-GPU run (rows x z-chunk against the 1024 / 768 workgroup slots), graphs re-test
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
O=$R/gpurun_out/r3h
export TMPDIR=/tmp
for RZ in "0 0" "3 10" "3 6" "4 20" "4 10" "4 12" "4 8" "5 20" "5 15" "7 30" "7 20" "2 6" "2 8"; do
  set -- $RZ
  timeout 200 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 300 --rows $1 --zchunk $2 >> $O/slab_v0_shapes.jsonl 2>> $O/slab_v0_shapes.err
done
for RZ in "0 0" "4 20" "4 10" "4 12"; do
  set -- $RZ
  timeout 200 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 300 --rows $1 --zchunk $2 --pml 2 >> $O/slab_pml_shapes.jsonl 2>> $O/slab_pml_shapes.err
  timeout 200 python scripts/probe_slab.py --slabs 4 --modes comm_fused --steps 200 --rows $1 --zchunk $2 >> $O/slab4_v0_shapes.jsonl 2>> $O/slab4_v0_shapes.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3h/*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], "N", d["slab_of"], "rows", d["rows"], "zc", d["zchunk"], "pml", d["pml"], round(d["ms_per_step"], 4))
PY
(timeout 900 python -m pytest tests/test_gpu_production_path.py tests/test_gpu_parity.py -m gpu -q -x -s -p no:cacheprovider -k "captured or tilted or film" 2>&1 | grep -E "graphs|tilted|config5|passed|failed|Error|assert " | tail -12) > $O/pytest_sel.log
cat $O/pytest_sel.log
