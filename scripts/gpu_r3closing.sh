#!/bin/bash
# r3z: FINAL tree of round 3 (two-step sweep with materials, monitors, node table) — the whole GPU suite, kernel statistics (placement probe off) and PMC traffic of the two-step
# sweep, the single sweep and the V2 step, 1024^3, bench
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3closing; O=$R/gpurun_out/r3closing; cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v0 -o trace -- python $R/bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu --no-workloads --placement-tries 0 > $O/prof_v0_bench.json 2> $O/prof_v0.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v0s -o trace -- python $R/bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu --no-workloads --placement-tries 0 --opt OPT_TWOSTEP=0 > $O/prof_v0s_bench.json 2> $O/prof_v0s.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v2 -o trace -- python $R/bench.py --workload v2 --steps 60 --warmup 10 --repeats 2 --no-cpu --no-workloads --placement-tries 0 > $O/prof_v2_bench.json 2> $O/prof_v2.err
for W in v0 v0s v2; do
  case $W in v0) A="";; v0s) A="--opt OPT_TWOSTEP=0";; v2) A="--workload v2";; esac
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$W/pmc_$C -o pmc -- python $R/bench.py $A --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --placement-tries 0 > /dev/null 2> $O/pmc_${W}_$C.err
  done
  python $R/scripts/summarize_pmc.py $O/pmc_$W > $O/pmc_${W}_summary.json
done
cd $R
find gpurun_out/r3closing -name '*kernel_trace*' -size +8M -delete
find gpurun_out/r3closing -name '*counter_collection*' -size +4M -delete
timeout 600 python bench.py --size 1024 --steps 20 --warmup 4 --repeats 3 --no-cpu --no-workloads > $O/bench_1024.json 2> $O/bench_1024.err
python -c "
import json; d=json.load(open('$O/bench_1024.json')); print('1024^3', round(d['value']), d['ms_per_step'], d['roofline'].get('two_steps_per_sweep'), d.get('single_steps'))"
timeout 600 python scripts/probe_ab.py 512 v1 SETS "OPT_TWOSTEP=0;OPT_TWOSTEP=-1" 3 > $O/ab_v1.jsonl 2> $O/ab_v1.err; cut -c1-300 $O/ab_v1.jsonl
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), d['roofline'].get('two_steps_per_sweep'), d.get('single_steps'), 'V2', round(d['workloads']['v2']['value']), d['workloads']['v2']['ms_per_step'])"
python -c "
import json
for w in ['v0','v0s','v2']:
    d=json.load(open('$O/pmc_%s_summary.json'%w))
    for k,v in d.items():
        if 'fused' in k or 'seam' in k: print(w,k,round(v.get('hbm_bytes_per_launch',0)/1e9,3),'GB  read',round(v.get('read_bytes_per_launch',0)/1e9,3),'write',round(v.get('write_bytes_per_launch',0)/1e9,3))
"
head -4 $O/prof_v0/trace_kernel_stats.csv | cut -c1-200
