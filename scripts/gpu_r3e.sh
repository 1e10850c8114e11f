#!/bin/bash
# round 3, visit e: timeline of a 64-plane slab step with CPML (slab kernels vs in-sweep), kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
O=$R/gpurun_out/r3e
export TMPDIR=/tmp
for F in 0 3; do
  timeout 300 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 200 --pml 1 --pml-fused $F >> $O/slab_pml.jsonl 2>> $O/slab_pml.err
done
for B in 1 4; do
  timeout 300 python scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 200 --pml 1 --pml-fused 0 --bnd $B >> $O/slab_pml.jsonl 2>> $O/slab_pml.err
done
grep slab_of $O/slab_pml.jsonl
cd /tmp
for F in 0 3; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_f$F -o trace -- python $R/scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 12 --warm 6 --pml 1 --pml-fused $F > /dev/null 2> $O/trace_f$F.err
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_v0 -o trace -- python $R/scripts/probe_slab.py --slabs 8 --modes comm_fused --steps 12 --warm 6 > /dev/null 2> $O/trace_v0.err
cd $R
ls -la $O/trace_f0 $O/trace_v0 | head
