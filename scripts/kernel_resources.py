#!/usr/bin/env python
"""Register / scratch / LDS use of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage),
one line per kernel:  python scripts/kernel_resources.py tidy3d_amd/csrc/fdtd_fused2c.hip [substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flags = ["-fno-slp-vectorize"] if "fused2" in src else []
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-disable-lsr",
       "-I/opt/rocm/include", *flags, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
        continue
    m = re.search(r"remark: .*?\s(\w[\w ]*\w): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][-70:]:70s} sgpr {r.get('TotalSGPRs', r.get('SGPRs', -1)):4d} vgpr {r.get('VGPRs', -1):4d} agpr {r.get('AGPRs', 0):3d} "
              f"spill s/v {r.get('SGPRs Spill', 0):4d}/{r.get('VGPRs Spill', 0):3d} scratch {r.get('ScratchSize [bytes/lane]', r.get('ScratchSize', 0)):5d} "
              f"occ {r.get('Occupancy [waves/SIMD]', r.get('Occupancy', -1)):2d} lds {r.get('LDS Size [bytes/block]', r.get('LDS Size', 0))}")
