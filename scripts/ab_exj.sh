#!/bin/bash
# A/B of the library with E_x of the row above handed down through LDS (the default build) against a build without it
# (variants/libfdtd_hip_noexj.so, -DFDTD_NO_EXJ=1): the whole default bench line of each, rounds interleaved on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r6v}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for ROUND in 1 2 3; do
  for V in ${AB_VARIANTS:-exj noexj}; do
    if [ $V = exj ] || [ $V = default ]; then unset TIDY3D_AMD_LIBRARY; else export TIDY3D_AMD_LIBRARY=$R/variants/libfdtd_hip_$V.so; fi
    timeout 400 python bench.py --no-cpu ${AB_BENCH_ARGS:-} > $O/bench_${V}_$ROUND.json 2> $O/bench_${V}_$ROUND.err
    python - <<PY
import json
d = json.load(open("$O/bench_${V}_$ROUND.json"))
print("$V", $ROUND, "v0", round(d["value"]), " ".join(f"{k} {round(w['value'])}" for k, w in d.get("workloads", {}).items()), "single", round(d.get("single_steps", {}).get("value", 0)))
PY
  done
done
