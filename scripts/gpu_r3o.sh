#!/bin/bash
# round 3, visit o: final kernel sources — PMC traffic V0 / V2 (re-stamped: the source hash changed with the mirror-fill kernel),
# kernel statistics with the probe off, 1024^3 line, bench
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
O=$R/gpurun_out/r3o
export TMPDIR=/tmp
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
find gpurun_out/r3o -name '*kernel_trace*' -size +8M -delete
find gpurun_out/r3o -name '*counter_collection*' -size +4M -delete
timeout 600 python bench.py --size 1024 --steps 20 --warmup 4 --repeats 3 --no-cpu --no-workloads > $O/bench_1024.json 2> $O/bench_1024.err
python -c "
import json; d=json.load(open('$O/bench_1024.json')); print('1024^3', round(d['value']), d['ms_per_step'], d['config']['tile']['placement'])"
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), d['config']['tile']['placement'], 'V2', round(d['workloads']['v2']['value']), d['workloads']['v2']['ms_per_step'])"
