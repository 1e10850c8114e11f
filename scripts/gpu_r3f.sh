#!/bin/bash
# round 3, visit f: z-slab rank proxies with in-sweep CPML and short edge chunks; ADE over plane sub-ranges; V2 edge chunks A/B
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
O=$R/gpurun_out/r3f
export TMPDIR=/tmp
for P in 1 2; do
  for F in 0 3; do
    timeout 300 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 200 --pml $P --pml-fused $F >> $O/slab.jsonl 2>> $O/slab.err
  done
  timeout 300 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 200 --pml $P --pml-fused 3 --opt OPT_EDGE_ZCHUNK=0 >> $O/slab.jsonl 2>> $O/slab.err
  timeout 300 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 200 --pml $P --pml-fused 3 --opt OPT_EDGE_ZCHUNK=4 >> $O/slab.jsonl 2>> $O/slab.err
done
timeout 300 python scripts/probe_slab.py --slabs 4 --modes comm_fused --steps 200 --pml 2 --pml-fused 0 >> $O/slab.jsonl 2>> $O/slab.err
timeout 300 python scripts/probe_slab.py --slabs 4 --modes comm_fused --steps 200 --pml 2 --pml-fused 3 >> $O/slab.jsonl 2>> $O/slab.err
grep slab_of $O/slab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['slab_of'], 'pml', d['pml'], 'fused', d['pml_fused'], d['opt'], round(d['ms_per_step'], 4))
"
timeout 600 python scripts/probe_ab.py 512 v2 OPT_EDGE_ZCHUNK 0,-1,4,2 3 > $O/probe_edge_zchunk_v2.jsonl 2> $O/probe_edge_zchunk_v2.err
cat $O/probe_edge_zchunk_v2.jsonl
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_production_path.py -m gpu -q -x -p no:cacheprovider -k "pipelined or three_launch or config5 or mie or film" 2>&1 | tail -5) > $O/pytest_sel.log
cat $O/pytest_sel.log
