#!/bin/bash
# r03i: config-3-like (424 x 224 x 824, CPML on six faces) under the cyclic axis renamings; small grids; configs 3-5 rerun
cd /root/repo; mkdir -p gpurun_out
timeout 600 python scripts/probe_c3.py 150 0,1,2 > gpurun_out/probe_r03i_c3.jsonl 2> gpurun_out/probe_r03i.err
cat gpurun_out/probe_r03i_c3.jsonl
timeout 300 python scripts/probe_small.py > gpurun_out/probe_r03i_small.jsonl 2>> gpurun_out/probe_r03i.err
cat gpurun_out/probe_r03i_small.jsonl
tail -3 gpurun_out/probe_r03i.err
