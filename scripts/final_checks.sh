#!/bin/bash
# randomised bit checks on the device after the last kernel changes of round 6 (EXJ in every instantiation, the (Ca, Cb) table in the
# dynamic LDS, wall-start shell boxes, slab-rank tile shapes) + BASELINE configs 3 / 4 / 5 with their printed numbers
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r6zz}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python scripts/fuzz_round6.py 200 11 > $O/fuzz_round6.log 2>&1; tail -1 $O/fuzz_round6.log
timeout 600 python scripts/fuzz_twostep.py 150 7 > $O/fuzz_twostep.log 2>&1; tail -1 $O/fuzz_twostep.log
timeout 900 python scripts/fuzz_shell2.py 200 9 > $O/fuzz_shell2.log 2>&1; tail -1 $O/fuzz_shell2.log
timeout 900 python scripts/fuzz_shell2.py 150 10 periodic > $O/fuzz_shell2_periodic.log 2>&1; tail -1 $O/fuzz_shell2_periodic.log
timeout 900 python scripts/fuzz_slab_cpml_device.py 100 3 > $O/fuzz_slab_cpml_device.log 2>&1; tail -1 $O/fuzz_slab_cpml_device.log
timeout 900 python scripts/fuzz_cell.py 100 4 > $O/fuzz_cell.log 2>&1; tail -1 $O/fuzz_cell.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "config3 or config4 or config5" > $O/configs_3_4_5.log 2>&1; grep -i "gcells\|Mcells\|passed\|failed\|set-up\|setup" $O/configs_3_4_5.log | cut -c1-220 | tail -20
