#!/bin/bash
# ONE launcher for every GPU visit (replaces the per-visit gpu_r*.sh of rounds 2 and 3):
#   gpurun --timeout 1500 -- 'bash scripts/gpu_visit.sh TAG stage [stage ...]'
# Output goes to gpurun_out/TAG/ (merged back by gpurun); what is to be judged is copied from there into profiles/<round>/.
# Stages:
#   tests        the whole GPU suite (pytest -m gpu)
#   bench        the driver's default bench line
#   stats        rocprofv3 --kernel-trace --stats of the bench workloads v0 (step pairs), v0 single steps, v2 (placement probe off)
#   pmc          HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of v0, v0 single steps, v1, va, v2 -> pmc_*_summary.json
#   shell        shell pairs against single steps inside one engine (scripts/probe_shell.py), 512^3 v2
#   variants     A/B of the library builds under variants/ (compiler flags), V0 bench line, two interleaved rounds
#   slab         per-rank proxy of the 8 / 4 / 2-GPU strong-scaling run (scripts/probe_slab.py)
#   mie          config-4 problem at lambda0 / 20, 30, 40 (scripts/probe_mie_refinement.py)
#   bench1024    1024^3 on one GPU
#   sq           SQ counters (two passes) of the V0 two-step sweep and of the single sweep -> sq_counters.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export TMPDIR=/tmp
for S in "$@"; do
  case $S in
    tests)
      timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log;;
    bench)
      timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
      python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("V0", round(d["value"]), round(d["ms_per_step"], 4), "frac", round(r["frac"], 3), "traffic_frac", r.get("traffic_frac"), "single", d.get("single_steps", {}).get("value"))
for k, w in d.get("workloads", {}).items():
    print(k, round(w["value"]), round(w["ms_per_step"], 4), "whole_step_frac", round(w["whole_step_frac"], 3), w.get("two_steps_per_sweep"))
PY
      ;;
    stats)
      cd /tmp
      for W in v0 v0s v2 v3; do
        case $W in v0) A="";; v0s) A="--opt OPT_TWOSTEP=0";; *) A="--workload $W";; esac
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -o trace -- python $R/bench.py $A --steps 100 --warmup 10 --repeats 2 --no-cpu --no-workloads --no-single-steps --placement-tries 0 > $O/prof_${W}_bench.json 2> $O/prof_$W.err
        head -6 $O/prof_$W/trace_kernel_stats.csv | cut -c1-160
      done
      find $O -name '*kernel_trace*' -size +8M -delete
      cd $R;;
    pmc)
      cd /tmp
      for W in v0 v0s v1 va v2 v3; do
        case $W in v0) A="";; v0s) A="--opt OPT_TWOSTEP=0";; *) A="--workload $W";; esac
        for C in FETCH_SIZE WRITE_SIZE; do
          timeout 240 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$W/pmc_$C -o pmc -- python $R/bench.py $A --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --no-single-steps --placement-tries 0 > /dev/null 2> $O/pmc_${W}_$C.err
        done
        python $R/scripts/summarize_pmc.py $O/pmc_$W > $O/pmc_${W}_summary.json
      done
      find $O -name '*counter_collection*' -size +4M -delete
      cd $R
      python - <<PY
import json
for w in ["v0", "v0s", "v1", "va", "v2", "v3"]:
    d = json.load(open("$O/pmc_%s_summary.json" % w))
    for k, v in d.items():
        if "hbm_bytes_per_launch" in v and ("fused" in k or "seam" in k or "strip" in k or "shell2" in k or "ade2" in k):
            print(w, k, round(v["hbm_bytes_per_launch"] / 1e9, 3), "GB  read", round(v["read_bytes_per_launch"] / 1e9, 3), "write", round(v["write_bytes_per_launch"] / 1e9, 3), "x", v.get("launches_FETCH_SIZE"))
PY
      ;;
    shell)
      timeout 600 python scripts/probe_shell.py 512 v2 40 > $O/probe_shell_v2_512.jsonl 2> $O/probe_shell.err; cut -c1-130 $O/probe_shell_v2_512.jsonl;;
    variants)    # A/B of builds of the library under variants/ (TIDY3D_AMD_LIBRARY): the V0 bench line of each, two rounds interleaved
      for ROUND in 1 2; do
        for F in $R/variants/libfdtd_hip_*.so; do
          T=$(basename $F .so | sed 's/libfdtd_hip_//')
          TIDY3D_AMD_LIBRARY=$F timeout 200 python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu --no-workloads --no-single-steps 2> /dev/null \
            | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'variant': '$T', 'round': $ROUND, 'ms_per_step': d['ms_per_step'], 'gcells_per_s': d['value']}))" | tee -a $O/variants.jsonl
        done
      done;;
    slab)
      # (a process per slab: engines created after an engine that held an RCCL communicator run 10 - 100 % slower, profiles/r6/r6ord_*)
      : > $O/slab.jsonl
      for N in 8 4 2; do timeout 300 python scripts/probe_slab.py --slabs $N --modes comm_fused --twostep 0,-1 --ref512 $((N == 8)) 2>> $O/slab.err | grep "^{" >> $O/slab.jsonl; done
      cut -c1-200 $O/slab.jsonl;;
    mie)
      RUN_PERIODS=400 NFREQ=25 timeout 900 python scripts/probe_mie_refinement.py 20 30 40 > $O/mie_converged_25f.jsonl 2> $O/mie.err; cut -c1-160 $O/mie_converged_25f.jsonl;;
    sq)
      cd /tmp
      for W in ${SQ_WORKLOADS:-v0 v0s}; do
        case $W in v0) A="";; v0s) A="--opt OPT_TWOSTEP=0";; *) A="--workload $W";; esac
        timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/sq_$W/pmc_sq1 -o pmc -- python $R/bench.py $A --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --no-single-steps --placement-tries 0 > /dev/null 2> $O/sq_${W}_1.err
        timeout 240 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq_$W/pmc_sq2 -o pmc -- python $R/bench.py $A --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --no-single-steps --placement-tries 0 > /dev/null 2> $O/sq_${W}_2.err
      done
      python $R/scripts/summarize_sq.py $O > $O/sq_counters.json
      find $O -name '*counter_collection*' -size +4M -delete
      cd $R; head -c 1500 $O/sq_counters.json;;
    bench1024)
      timeout 600 python bench.py --size 1024 --steps 20 --warmup 4 --repeats 3 --no-cpu --no-workloads > $O/bench_1024.json 2> $O/bench_1024.err; cut -c1-200 $O/bench_1024.json;;
    *) echo "unknown stage $S";;
  esac
done
