#!/bin/bash
# A/B of the CPML placement on the PML workloads (V2, V3): slab kernels (0), y/z recursions inside
# the fused sweep (6), all three axes inside (7)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for W in ${WORKLOADS:-v2 v3}; do
  for P in ${MASKS:-0 6 7}; do
    timeout 300 python $R/bench.py --steps 40 --warmup 5 --no-cpu --workload $W --pml-fused $P ${AB_EXTRA:-} > $R/gpurun_out/ab_${W}_$P.json 2> $R/gpurun_out/ab_${W}_$P.err
    python -c "import json; d=json.load(open('$R/gpurun_out/ab_${W}_$P.json')); print('$W pml_fused=$P', round(d['value']), round(d['ms_per_step'],4), d['roofline']['per_step_ms'])"
  done
done
