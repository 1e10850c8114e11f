#!/bin/bash
# round 3, visit d: pooled x-CPML (E side in front of the barrier): A/B inside one engine, SQ counters per form, slab proxies
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
O=$R/gpurun_out/r3d
export TMPDIR=/tmp
timeout 600 python scripts/probe_ab.py 512 v2 OPT_PML_POOL 0,1 4 > $O/probe_pml_pool_v2.jsonl 2> $O/probe_pml_pool_v2.err
cat $O/probe_pml_pool_v2.jsonl
cd /tmp
for P in 0 1; do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_pool$P -o pmc -- python $R/bench.py --workload v2 --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --placement-tries 0 --opt OPT_PML_POOL=$P > /dev/null 2> $O/pmc_pool$P.err
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc2_pool$P -o pmc -- python $R/bench.py --workload v2 --steps 6 --warmup 2 --repeats 1 --no-cpu --no-workloads --placement-tries 0 --opt OPT_PML_POOL=$P > /dev/null 2> $O/pmc2_pool$P.err
done
cd $R
python - <<'PY'
import csv, glob, json, collections
out = {}
for tag in ("pmc_pool0", "pmc_pool1", "pmc2_pool0", "pmc2_pool1"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"gpurun_out/r3d/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void fdtd::", "")
            if not name.startswith("fused_step_kernel"):
                continue
            a = acc[name][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    out[tag] = {k: {c: v[0] / max(v[1], 1) for c, v in d.items()} | {"launches": max(v[1] for v in d.values())} for k, d in acc.items()}
json.dump(out, open("gpurun_out/r3d/sq_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
find gpurun_out/r3d -name '*counter_collection*' -size +2M -delete
find gpurun_out/r3d -name '*.csv' -size +4M -delete
timeout 300 python scripts/probe_slab.py --slabs 8 --modes single_fused,comm_fused --steps 200 > $O/slab_v0.jsonl 2> $O/slab_v0.err
cat $O/slab_v0.jsonl
timeout 300 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 200 --pml 1 > $O/slab_pml.jsonl 2> $O/slab_pml.err
cat $O/slab_pml.jsonl
