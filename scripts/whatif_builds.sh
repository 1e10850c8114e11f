#!/bin/bash
# What-if builds of the two-step sweep: libraries whose fused2_step_kernel SKIPS the work an optimisation
# would hide or remove — their results are wrong, their times bound what the optimisation could gain:
#   whatif1  E_y / H_y of a plane are not loaded        -> the most an LDS-DMA prefetch of those two arrays (the 2 KB per wave left beside xch) can buy
#   whatif2  no second barrier per plane                -> the most a one-barrier (skewed) pipeline can buy
#   whatif3  the three halo rows of a workgroup load nothing -> the most sharing halo rows between y-neighbouring workgroups can buy
# Built into variants/ (git-ignored, travels with gpurun); timed by `scripts/gpu_visit.sh TAG variants` (V0 bench line, two rounds interleaved).
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
for V in 0 1 2 3; do
  FDTD_EXTRA_HIPCC_FLAGS="-DFDTD_WHATIF=$V" python - <<PY
import os, shutil, sys
sys.path.insert(0, os.getcwd())
import tidy3d_amd.build as b
keep = b.LIB + ".keep"
shutil.copy(b.LIB, keep)
b.build(force=True, verbose=False)
shutil.move(b.LIB, os.path.join("variants", "libfdtd_hip_whatif$V.so"))
shutil.move(keep, b.LIB)
PY
done
ls -la variants/
