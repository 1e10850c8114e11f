#!/bin/bash
# more seeds of the randomised device checks (final tree of round 6)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r6fz}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for S in 31 32 33; do timeout 900 python scripts/fuzz_round6.py 150 $S > $O/fuzz_round6_$S.log 2>&1; grep -h "MISMATCH" $O/fuzz_round6_$S.log | head -5; tail -1 $O/fuzz_round6_$S.log; done
for S in 41 42; do timeout 600 python scripts/fuzz_twostep.py 150 $S > $O/fuzz_twostep_$S.log 2>&1; grep -h "MISMATCH\|differ" $O/fuzz_twostep_$S.log | head -5; tail -1 $O/fuzz_twostep_$S.log; done
for S in 51 52; do timeout 900 python scripts/fuzz_shell2.py 150 $S > $O/fuzz_shell2_$S.log 2>&1; grep -h "MISMATCH\|differ" $O/fuzz_shell2_$S.log | head -5; tail -1 $O/fuzz_shell2_$S.log; done
timeout 900 python scripts/fuzz_shell2.py 150 53 periodic > $O/fuzz_shell2_53p.log 2>&1; tail -1 $O/fuzz_shell2_53p.log
timeout 900 python scripts/fuzz_cell.py 150 61 > $O/fuzz_cell_61.log 2>&1; tail -1 $O/fuzz_cell_61.log
timeout 900 python scripts/fuzz_slab_cpml_device.py 100 71 2>/dev/null | tail -1
timeout 900 python scripts/fuzz_variants.py 60 81 > $O/fuzz_variants_81.log 2>&1; tail -2 $O/fuzz_variants_81.log
