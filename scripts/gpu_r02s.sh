#!/bin/bash
# r02s: one fresh process per placement (12 separate allocations of different rounded sizes; one block), addresses logged
cd /root/repo; mkdir -p gpurun_out
export FDTD_DEBUG_ADDR=1
M=1048576
: > gpurun_out/probe_r02s.jsonl; : > gpurun_out/probe_r02s.err
for c in '{"layout":0}' '{"layout":5,"round":1073741824}' '{"layout":5,"round":'$((516*M))'}' '{"layout":5,"round":'$((528*M))'}' '{"layout":5,"round":'$((576*M))'}' '{"layout":5,"round":'$((768*M))'}' '{"layout":3}' '{"layout":3,"s1":'$((510*M))'}' '{"layout":1}' '{"layout":0}' '{"layout":5,"round":1073741824}'; do
  echo "== $c" >> gpurun_out/probe_r02s.err
  LAYOUTS="[$c,$c]" timeout 300 python scripts/probe_layout.py 512 v0 >> gpurun_out/probe_r02s.jsonl 2>> gpurun_out/probe_r02s.err
done
cat gpurun_out/probe_r02s.jsonl; grep -c . gpurun_out/probe_r02s.err
