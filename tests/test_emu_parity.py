"""Kernel-logic parity on CPU: the unmodified HIP sources, compiled for the host against the
fiber emulator (tests/hipemu), driven through the same C ABI and compared with the oracle.
fp32 vs fp64 => tolerance 2e-5 rel-L2 over 60 steps (typical: 1e-7 .. 1e-6)."""
import numpy as np
import pytest

from cases import CASES, run_case

TOL = 2e-5


@pytest.mark.parametrize("name", sorted(CASES))
def test_emulated_kernels_match_oracle(name, emu_lib):
    worst, disc = run_case(name, emu_lib, n_steps=60)
    assert worst < TOL, (name, disc.spec.shape, worst)


@pytest.mark.parametrize("variant,zchunk,rows", [(1, 0, 4), (2, 1, 1), (2, 3, 2), (2, 64, 8)])
def test_launch_geometry_does_not_change_results(emu_lib, variant, zchunk, rows):
    from tidy3d_amd import lib as L
    from tidy3d_amd.discretize import discretize
    from tidy3d_amd.engine import HipEngine
    from cases import media_mix, pec_box_vec
    for fn in (media_mix, pec_box_vec):
        disc = discretize(fn(), n_steps=25)
        outs = []
        for (v, zc, r) in ((2, 32, 4), (variant, zchunk, rows)):
            with HipEngine(disc.spec, lib=emu_lib, variant=v, z_chunk=zc) as e:
                e.set_option(L.OPT_ROWS, r)
                e.run()
                outs.append([e.get_field(c) for c in range(6)])
        for a, b in zip(*outs):
            if variant == 2:
                assert np.array_equal(a, b)       # same arithmetic per cell -> bit identical
            else:
                assert np.allclose(a, b, rtol=0, atol=1e-6 * max(1e-30, np.abs(a).max()))
