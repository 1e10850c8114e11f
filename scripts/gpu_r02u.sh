#!/bin/bash
# r02u: is the placement effect the size of the page-table fragments (virtual alignment of the allocations)?
# libhsakmt aligns a device allocation's virtual address to at most 4 KiB << $HSA_MAX_VA_ALIGN (default 9 = 2 MiB).
cd /root/repo; mkdir -p gpurun_out
export FDTD_DEBUG_ADDR=1
: > gpurun_out/probe_r02u.jsonl; : > gpurun_out/probe_r02u.err
A='{"layout":0,"hold":1}'; L="$A"; for i in $(seq 1 7); do L="$L,$A"; done
for al in "" 18 13 21; do
  echo "{\"HSA_MAX_VA_ALIGN\":\"$al\"}" >> gpurun_out/probe_r02u.jsonl; echo "== align $al" >> gpurun_out/probe_r02u.err
  if [ -n "$al" ]; then export HSA_MAX_VA_ALIGN=$al; fi
  LAYOUTS="[$L]" timeout 300 python scripts/probe_layout.py 512 v0 >> gpurun_out/probe_r02u.jsonl 2>> gpurun_out/probe_r02u.err
done
echo '{"part":"pool 16 GiB, HSA_MAX_VA_ALIGN=18"}' >> gpurun_out/probe_r02u.jsonl
export HSA_MAX_VA_ALIGN=18
LAYOUTS='[{"layout":4},{"layout":4,"s1":2097152},{"layout":3},{"layout":1}]' timeout 300 python scripts/probe_layout.py 512 v0 >> gpurun_out/probe_r02u.jsonl 2>> gpurun_out/probe_r02u.err
cat gpurun_out/probe_r02u.jsonl | cut -c1-120
grep -v -i warn gpurun_out/probe_r02u.err | awk '/== align 18/{f=1} f' | head -30
