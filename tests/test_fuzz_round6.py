"""scripts/fuzz_round6.py: random simulations with dispersive bodies of any shape and big source lists (TFSF boxes, current sheets,
dipole crowds) inside every kind of wall the step pairs cover — pairs (dispersive cells advanced inside them, the lists as paged
source terms) == single steps, bit for bit.  A few cases on the CPU emulator here; the device runs more and larger ones."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_round6_pairs_on_random_simulations(seed, emu_lib):
    import fuzz_round6
    bad, n_disp, n_paged = fuzz_round6.run_cases(3, seed=seed, lib=emu_lib, quiet=True)
    assert bad == 0, (bad, n_disp, n_paged)


@pytest.mark.gpu
def test_round6_pairs_on_random_simulations_on_the_device(hip_lib):
    import fuzz_round6
    bad, n_disp, n_paged = fuzz_round6.run_cases(40, seed=5, lib=hip_lib, quiet=True)
    print(f"\n[fuzz round 6] 40 cases: {n_disp} advanced dispersive cells inside pairs, {n_paged} carried paged source terms")
    assert bad == 0 and n_disp >= 10 and n_paged >= 10, (bad, n_disp, n_paged)
