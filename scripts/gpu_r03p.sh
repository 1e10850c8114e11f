#!/bin/bash
# r03p: timeline of the three launches of a CPML-carrying step (start / end of each kernel, per step)
cd /root/repo; mkdir -p gpurun_out; R=/root/repo
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03p_trace -o trace -- python $R/scripts/probe_r02.py --child 512 v2 '[{}]' > /dev/null 2> $R/gpurun_out/r03p.err
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03p_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'fused_step_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# group into steps of three launches, print the last 6 steps
out = []
for i in range(0, len(rows) - 2, 3):
    g = rows[i:i + 3]
    t0 = min(int(r['Start_Timestamp']) for r in g)
    out.append([(r['Kernel_Name'].split('<')[1].split('>')[0], r.get('Grid_Size', r.get('Grid_Size_X', '?')), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3) for r in g])
for g in out[-6:]:
    print(' | '.join('%s grid %s: %.0f -> %.0f us' % x for x in g))
PY
find gpurun_out -name '*kernel_trace*' -size +8M -delete
