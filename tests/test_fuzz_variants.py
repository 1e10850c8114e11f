"""scripts/fuzz_variants.py on the CPU emulator: random small simulations through the fused sweep, the two-pass kernels and (periodic
z) a z-slab rank exchanging with itself — bit-identical — and against the fp64 oracle (<= 2e-5).  The device runs more and larger
cases (tests/test_gpu_production_path.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))


@pytest.mark.parametrize("seed", [11, 12])
def test_variants_agree_on_random_simulations(seed, emu_lib):
    import fuzz_variants
    bad, far, worst = fuzz_variants.run_cases(2, seed=seed, lib=emu_lib, quiet=True)
    assert bad == 0 and far == 0, (bad, far, worst)
