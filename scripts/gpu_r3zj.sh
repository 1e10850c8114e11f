#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zj; O=$R/gpurun_out/r3zj; cd $R
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print('V0', round(d['value']), d['ms_per_step'], 'frac', round(r['frac'],3), 'whole_step_frac', round(r['whole_step_frac'],3), 'traffic', r['traffic'], d['single_steps']['value'], {k:round(v['value']) for k,v in d['workloads'].items()})"
