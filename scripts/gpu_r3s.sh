#!/bin/bash
# r3s: DPP wave shifts instead of ds_bpermute in the two-step kernel, shape model: parity, auto shapes on 128^3 ... 640^3,
# SQ counters of the two-step kernel (16 and 8 waves), bench
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3s; O=$R/gpurun_out/r3s
cd $R
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -m gpu -k "two_steps_per_sweep or bench_v0" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for n in 128 192 256 320 384 448 512 640; do
  timeout 300 python scripts/probe_twostep.py --n $n --steps 60 --rounds 3 0 auto $((16+64*32)) $((8+64*32)) >> $O/auto.jsonl 2>> $O/auto.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r3s/auto.jsonl"):
    d=json.loads(l); print(d["n"], d["twostep"], d["waves"], d["zchunk"], d["ms_per_step"], d["gcells_per_s"])
PY
export TMPDIR=/tmp; cd /tmp
for cfg in 0 $((16+64*32)) $((8+64*32)); do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sq1_$cfg -o pmc -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --placement-tries 0 --opt OPT_TWOSTEP=$cfg > /dev/null 2> $O/sq1_$cfg.err
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2_$cfg -o pmc -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --placement-tries 0 --opt OPT_TWOSTEP=$cfg > /dev/null 2> $O/sq2_$cfg.err
done
cd $R
python - <<'PY'
import csv, glob, json, collections
out={}
for d in sorted(glob.glob("gpurun_out/r3s/sq*_*")):
    if not __import__("os").path.isdir(d): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name=row["Kernel_Name"].split("(")[0].replace("void fdtd::","").split("<")[0]
            a=acc[name][row["Counter_Name"]]; a[0]+=float(row["Counter_Value"]); a[1]+=1
    out[d.split("/")[-1]]={k:{c:v[0]/max(v[1],1) for c,v in cs.items()} for k,cs in acc.items() if "fused" in k or "seam" in k}
json.dump(out, open("gpurun_out/r3s/sq_summary.json","w"), indent=1)
for tag,ks in out.items():
    for k,cs in ks.items(): print(tag,k,{c:round(v) for c,v in cs.items()})
PY
find gpurun_out/r3s -name '*counter_collection*' -size +2M -delete
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('V0', round(d['value']), d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), d['roofline'].get('two_steps_per_sweep'), d.get('single_steps'), 'V2', round(d['workloads']['v2']['value']))"
