#!/bin/bash
# r04a: BASELINE configs 3 / 4 / 5 on the final defaults (placement probe, z-chunk 8 without in-sweep CPML)
# cost-model axis renaming: config 3 runs with x = 224)
cd /root/repo; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "config" 2>&1) > gpurun_out/r04a_configs.log
grep -E "^\[|passed|failed|best agreement|Error|error" gpurun_out/r04a_configs.log | cut -c1-400
(timeout 200 python scripts/probe_c3.py 200) > gpurun_out/r04a_probe_c3.json 2> gpurun_out/r04a_probe_c3.err
cat gpurun_out/r04a_probe_c3.json
