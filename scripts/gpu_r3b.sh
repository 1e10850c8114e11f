#!/bin/bash
# round 3, visit b: pooled x-CPML (A/B inside one engine), the GPU suite, V2 bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3b
O=gpurun_out/r3b
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25) > $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
timeout 600 python scripts/probe_ab.py 512 v2 OPT_PML_POOL 0,1 4 > $O/probe_pml_pool_v2.jsonl 2> $O/probe_pml_pool_v2.err
cat $O/probe_pml_pool_v2.jsonl
timeout 600 python scripts/probe_ab.py 512 v2 SETS "OPT_PML_POOL=1,OPT_MEM_HINTS=1;OPT_PML_POOL=1,OPT_MEM_HINTS=0;OPT_PML_POOL=0,OPT_MEM_HINTS=1;OPT_PML_POOL=1,OPT_MEM_HINTS=1,OPT_PML_SPLIT=0" 3 > $O/probe_pml_pool_hints.jsonl 2> $O/probe_pml_pool_hints.err
cat $O/probe_pml_pool_hints.jsonl
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3b/bench.json"))
print("V0", d["value"], d["ms_per_step"], "V2", d["workloads"]["v2"]["value"], d["workloads"]["v2"]["ms_per_step"])
PY
