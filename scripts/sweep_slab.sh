#!/bin/bash
# Tile-shape sweep of the pipelined fused z-slab schedule on the per-rank proxy (scripts/probe_slab.py).
mkdir -p gpurun_out
for zc in 8 16 24 32; do
  for bnd in 4 8 16; do
    timeout 120 python scripts/probe_slab.py --slabs 8,4,2 --modes comm_fused --steps 200 --zchunk $zc --bnd $bnd 2>/dev/null | grep slab_of
  done
done > gpurun_out/sweep_slab.jsonl
timeout 120 python scripts/probe_slab.py --slabs 8,4,2,1 --modes single_fused --steps 200 2>/dev/null | grep slab_of >> gpurun_out/sweep_slab.jsonl
cat gpurun_out/sweep_slab.jsonl
