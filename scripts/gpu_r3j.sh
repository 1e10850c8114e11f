#!/bin/bash
# round 3, visit j: final tree — GPU suite, bench, kernel trace with the probe off, PMC traffic V0 / V2, configs 3 / 4 / 5, slab proxies
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
O=$R/gpurun_out/r3j
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "\[mie|\[config|\[tilted|\[graphs|stream_overlap|passed|failed|Error|^E  " | tail -40) > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'V2', round(d['workloads']['v2']['value']), d['workloads']['v2']['ms_per_step'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v0 -o trace -- python $R/bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu --no-workloads --placement-tries 0 > $O/prof_v0_bench.json 2> $O/prof_v0.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v2 -o trace -- python $R/bench.py --workload v2 --steps 60 --warmup 10 --repeats 2 --no-cpu --no-workloads --placement-tries 0 > $O/prof_v2_bench.json 2> $O/prof_v2.err
for W in v0 v2; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$W/pmc_$C -o pmc -- python $R/bench.py --workload $W --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --placement-tries 0 > /dev/null 2> $O/pmc_${W}_$C.err
  done
  python $R/scripts/summarize_pmc.py $O/pmc_$W > $O/pmc_${W}_summary.json
done
cd $R
grep -A8 "fused_step" $O/pmc_v0_summary.json | head -12
find gpurun_out/r3j -name '*kernel_trace*' -size +8M -delete
find gpurun_out/r3j -name '*counter_collection*' -size +4M -delete
timeout 300 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 300 > $O/slab.jsonl 2> $O/slab.err
for F in 0 1 3; do
  timeout 300 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 300 --pml 2 --pml-fused $F >> $O/slab.jsonl 2>> $O/slab.err
done
grep slab_of $O/slab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['slab_of'], 'pml', d['pml'], 'fused', d['pml_fused'], round(d['ms_per_step'], 4))
"
