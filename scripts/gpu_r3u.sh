#!/bin/bash
# r3u: the whole library built with the SLP vectorizer off against the default build — V2 (materials + CPML, the
# instruction-heavy sweep), V1 and single-step V0, processes alternating (placement differs per process: medians of 4)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3u; O=$R/gpurun_out/r3u; cd $R
for i in 1 2 3 4; do
  for lib in default noslp; do
    if [ $lib = noslp ]; then export TIDY3D_AMD_LIBRARY=$R/tidy3d_amd/libfdtd_hip_noslp.so; else unset TIDY3D_AMD_LIBRARY; fi
    for wl in v2 v1 v0; do
      timeout 200 python bench.py --workload $wl --steps 60 --warmup 10 --repeats 3 --no-cpu --no-workloads --opt OPT_TWOSTEP=0 2>> $O/err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib':'$lib','wl':'$wl','i':$i,'ms':d['ms_per_step'],'gcells':d['value']/1e3}))" >> $O/ab_slp.jsonl
    done
  done
done
python - <<'PY'
import json, collections, statistics
r=collections.defaultdict(list)
for l in open("gpurun_out/r3u/ab_slp.jsonl"):
    d=json.loads(l); r[(d["wl"],d["lib"])].append(d["ms"])
for k in sorted(r): print(k, [round(x,4) for x in r[k]], "median", round(statistics.median(r[k]),4), "min", round(min(r[k]),4))
PY
tail -n 3 $O/err.log
