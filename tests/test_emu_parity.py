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


def test_more_than_255_media(emu_lib):
    """300 structures of 300 distinct media: 16-bit host indices, fdtd_set_material16, three 10-bit indices per
    device word (the reference allows 65,530 structures, ref components/scene.py:52; 8-bit indices capped a
    simulation at 254 media)."""
    import tidy3d_amd.schema as td
    from cases import DL, PULSE, rel_err
    from oracle.fdtd_numpy import OracleFdtd
    from tidy3d_amd.discretize import discretize
    from tidy3d_amd.engine import HipEngine
    n = (20, 15, 8)
    boxes = []
    for q in range(300):
        i, j = q % 20, q // 20
        boxes.append(td.Structure(geometry=td.Box(center=((i + 0.5 - 10) * DL, (j + 0.5 - 7.5) * DL, 0.0), size=(DL, DL, 4 * DL)),
                                  medium=td.Medium(permittivity=1.5 + 0.01 * q, conductivity=0.001 * (q % 7))))
    sim = td.Simulation(size=tuple(v * DL for v in n), grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, structures=boxes,
                        sources=[td.PointDipole(center=(0.03, -0.07, 0.01), source_time=PULSE, polarization="Ez")],
                        monitors=[td.FieldMonitor(center=(0, 0, 0), size=(td.inf, td.inf, 0), freqs=[3e14], name="f")],
                        boundary_spec=td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.periodic(),
                                                      z=td.Boundary.pec()), shutoff=0, subpixel=False)
    spec = discretize(sim, n_steps=50).spec
    assert len(spec.media) == 302 and spec.mat_idx.dtype == np.uint16 and int(spec.mat_idx.max()) == 301
    ref = OracleFdtd(spec)
    want = ref.run()["f"]
    with HipEngine(spec, lib=emu_lib) as e:
        e.run()
        got = e.results()["f"]
        ez = e.get_field(2)
    assert rel_err(got, want) < 2e-5
    assert rel_err(ez, ref.E[2]) < 2e-5
