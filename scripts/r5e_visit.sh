mkdir -p gpurun_out/r5e; O=gpurun_out/r5e
PROBE_MODES=single,shell2,shell2_one_stream PROBE_NO_CHECK=1 timeout 300 python scripts/probe_shell2.py 512 v2 40 > $O/probe_shell2_v2_512.jsonl 2> $O/err1
# per-box times: one launch per box, one stream
python - > $O/per_box.jsonl 2> $O/err2 <<'PY'
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from bench import build_spec
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine
L.load_library()
spec = build_spec(512, 2000, "v2")
with HipEngine(spec, device=0) as e:
    e.set_option(L.OPT_SHELL2, 3); e.set_option(L.OPT_SHELL_PAIRS, 2); e.set_option(L.OPT_PLACEMENT_TRIES, 0)
    e.run(10)
    e.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
    st = e.run(20)
    print(json.dumps({"shell2_pairs": int(st.shell2_pairs), "shell_ms_per_pair": st.shell_kernel_ms / max(1, int(st.shell2_pairs)), "launches": int(st.shell_kernel_launches), "bulk_ms": st.fused_kernel_ms / max(1, st.fused_kernel_launches)}))
PY
cat $O/per_box.jsonl
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_box -o trace -- python - > /dev/null 2> $GRAFT_REPO_ROOT/$O/err3 <<'PY'
import json, sys, os
R = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, R); os.chdir(R)
import numpy as np, torch
from bench import build_spec
from tidy3d_amd import lib as L
from tidy3d_amd.engine import HipEngine
L.load_library()
spec = build_spec(512, 2000, "v2")
with HipEngine(spec, device=0) as e:
    e.set_option(L.OPT_SHELL2, 3); e.set_option(L.OPT_SHELL_PAIRS, 2); e.set_option(L.OPT_PLACEMENT_TRIES, 0)
    e.run(40)
PY
cd $GRAFT_REPO_ROOT
grep -i "shell2\|fused2\|seam" $O/prof_box/trace_kernel_stats.csv | cut -c1-200
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r5e/prof_box/*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    sh = [r for r in rows if "shell2" in r["Kernel_Name"]]
    # the six launches of a pair, in order: average duration per position
    d = collections.defaultdict(list)
    for i, r in enumerate(sh):
        d[i % 6].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(d): print("box", k, "us", round(sum(d[k]) / len(d[k]), 1), "grid", sh[k]["Grid_Size_X"] if "Grid_Size_X" in sh[k] else "")
PY
find $O -name '*kernel_trace*' -size +2M -delete
C3_SHELL2=1 timeout 300 python scripts/probe_c3.py 200 > $O/c3_shell2.jsonl 2> $O/err4; cat $O/c3_shell2.jsonl
C3_SHELL2=0 timeout 300 python scripts/probe_c3.py 200 > $O/c3_default_r4.jsonl 2> $O/err5; cat $O/c3_default_r4.jsonl
