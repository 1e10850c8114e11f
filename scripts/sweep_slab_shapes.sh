#!/bin/bash
# tile shapes of the two-step sweep on z-slab ranks (512 x 512 x 512/N proxies, RCCL looped back): the library's choice (-1) against
# forced W + 64 * planes-per-chunk words
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r6w}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
w() { echo $(( $1 + 64 * $2 )); }
(timeout 300 python scripts/probe_slab.py --slabs 2 --modes comm_fused --ref512 0 --steps 200 --twostep=-1,$(w 16 32),$(w 16 42),$(w 16 21),$(w 12 32)
 timeout 300 python scripts/probe_slab.py --slabs 4 --modes comm_fused --ref512 0 --steps 200 --twostep=-1,$(w 16 31),$(w 16 21),$(w 16 16),$(w 12 21)
 timeout 300 python scripts/probe_slab.py --slabs 8 --modes comm_fused --ref512 0 --steps 200 --twostep=-1,$(w 16 10),$(w 16 12),$(w 16 15),$(w 16 20),$(w 12 15),$(w 12 20)) 2> $O/slab_shapes.err | grep slab_of > $O/slab_shapes.jsonl
python - <<PY
import json
for l in open("$O/slab_shapes.jsonl"):
    d = json.loads(l); print(d["slab_of"], d["twostep"], d["shape"], d["fused2_pairs"], round(d["ms_per_step"], 4))
PY
tail -2 $O/slab_shapes.err
# ranks that carry CPML on x / y (V2-like: --pml 2): library's choice against single steps
(for N in 8 4 2; do timeout 300 python scripts/probe_slab.py --slabs $N --modes comm_fused --ref512 0 --steps 200 --pml 2 --pml-fused 7 --twostep=0,-1; done) 2>> $O/slab_shapes.err | grep slab_of > $O/slab_shapes_pml.jsonl
python - <<PY
import json
for l in open("$O/slab_shapes_pml.jsonl"):
    d = json.loads(l); print("pml", d["slab_of"], d["twostep"], d["shape"], d["fused2_pairs"], round(d["ms_per_step"], 4))
PY
