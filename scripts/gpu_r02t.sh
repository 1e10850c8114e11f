#!/bin/bash
# r02t: is the speed a property of WHERE in device memory the arrays lie?
#  A: 24 engines in one process, all held (each lands deeper in memory)    B: a 96 GiB pool, the arrays slid through it
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r02t.jsonl
A='{"layout":0,"hold":1}'; L="$A"; for i in $(seq 1 23); do L="$L,$A"; done
LAYOUTS="[$L]" timeout 600 python scripts/probe_layout.py 512 v0 >> gpurun_out/probe_r02t.jsonl 2> gpurun_out/probe_r02t.err
echo '{"part":"B"}' >> gpurun_out/probe_r02t.jsonl
G=1073741824
L=""; for i in $(seq 0 22); do L="$L{\"layout\":4,\"s0\":$((i*4*G))},"; done; L="${L%,}"
FDTD_FIELD_POOL=$((96*G)) LAYOUTS="[$L]" timeout 600 python scripts/probe_layout.py 512 v0 >> gpurun_out/probe_r02t.jsonl 2>> gpurun_out/probe_r02t.err
cat gpurun_out/probe_r02t.jsonl | cut -c1-120; tail -2 gpurun_out/probe_r02t.err
