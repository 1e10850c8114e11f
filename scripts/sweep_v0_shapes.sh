#!/bin/bash
# tile shapes of the plain two-step sweep on cubes between 256^3 and 640^3: the library's choice (-1) against forced W + 64 * planes words
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r6sh}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
w() { echo $(( $1 + 64 * $2 )); }
for N in ${SIZES:-320 384 448 576 640}; do
  for T in -1 $(w 8 32) $(w 8 24) $(w 16 32) $(w 16 24) $(w 16 48) $(w 16 64) $(w 12 32); do
    timeout 200 python bench.py --size $N --steps 100 --warmup 10 --repeats 3 --no-cpu --no-workloads --no-single-steps --opt OPT_TWOSTEP=$T 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); t=d['roofline'].get('two_steps_per_sweep',{}); print(json.dumps({'n': $N, 'twostep': $T, 'shape': [t.get('waves_per_workgroup'), t.get('planes_per_chunk')], 'gcells_per_s': round(d['value']/1e3,1), 'ms_per_step': round(d['ms_per_step'],4)}))" | tee -a $O/v0_shapes.jsonl
  done
done
