#!/bin/bash
# r3zr: SQ counters of the final two-step sweep (16 waves x 32 planes) beside the single sweep
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zr; O=$R/gpurun_out/r3zr
export TMPDIR=/tmp; cd /tmp
for cfg in 0 -1; do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sq1_$cfg -o pmc -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --placement-tries 0 --opt OPT_TWOSTEP=$cfg > /dev/null 2> $O/sq1_$cfg.err
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2_$cfg -o pmc -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --placement-tries 0 --opt OPT_TWOSTEP=$cfg > /dev/null 2> $O/sq2_$cfg.err
done
cd $R
python - <<'PY'
import csv, glob, json, collections, os
out={}
for d in sorted(glob.glob("gpurun_out/r3zr/sq*_*")):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name=row["Kernel_Name"].split("(")[0].replace("void fdtd::","").replace("fdtd::","").split("<")[0]
            a=acc[name][row["Counter_Name"]]; a[0]+=float(row["Counter_Value"]); a[1]+=1
    out[d.split("/")[-1]]={k:{c:v[0]/max(v[1],1) for c,v in cs.items()} for k,cs in acc.items() if "fused" in k or "seam" in k}
json.dump(out, open("gpurun_out/r3zr/sq_summary.json","w"), indent=1)
for tag,ks in out.items():
    for k,cs in ks.items(): print(tag,k,{c:round(v/1e6,2) for c,v in cs.items()})
PY
find gpurun_out/r3zr -name '*counter_collection*' -size +2M -delete
