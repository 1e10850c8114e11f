#!/bin/bash
# round 3, visit a: production-path parity tests, two-step slab schedule probe, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_production_path.py -x -q -s -m gpu -p no:cacheprovider > $O/pytest_production.log 2>&1
echo "pytest rc=$?" >> $O/pytest_production.log
tail -5 $O/pytest_production.log
timeout 600 python scripts/probe_ab.py 512 v0 SETS "OPT_TBLOCK=0;OPT_TBLOCK=8;OPT_TBLOCK=16;OPT_TBLOCK=24;OPT_TBLOCK=32;OPT_TBLOCK=64;OPT_TBLOCK=128;OPT_TBLOCK=4104;OPT_TBLOCK=4112;OPT_TBLOCK=4128" 3 > $O/probe_tblock_v0.jsonl 2> $O/probe_tblock_v0.err
cat $O/probe_tblock_v0.jsonl
timeout 600 python scripts/probe_ab.py 512 v0,v1 SETS "OPT_TBLOCK=0,OPT_MEM_HINTS=1;OPT_TBLOCK=16,OPT_MEM_HINTS=1;OPT_TBLOCK=0,OPT_MEM_HINTS=0;OPT_TBLOCK=8,OPT_MEM_HINTS=0;OPT_TBLOCK=16,OPT_MEM_HINTS=0;OPT_TBLOCK=32,OPT_MEM_HINTS=0;OPT_TBLOCK=4112,OPT_MEM_HINTS=0" 3 > $O/probe_tblock_hints.jsonl 2> $O/probe_tblock_hints.err
cat $O/probe_tblock_hints.jsonl
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
