"""Cyclic renaming of the axes (tidy3d_amd/engine.py ``permute_spec``): Maxwell's curl keeps its form under
x -> y -> z -> x, so a narrow grid can be laid out with its best-filled axis along x (where a wavefront owns 256
consecutive cells) and the results renamed back.  The renamed run must reproduce the plain one."""
import numpy as np
import pytest

from cases import CASES
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine, best_axis_shift, lane_efficiency, permute_spec, unpermute_array


def test_policy():
    assert lane_efficiency(256) == 1.0 and lane_efficiency(60) == pytest.approx(60 / 256) and lane_efficiency(258) == pytest.approx(258 / 512)
    assert best_axis_shift((512, 512, 512)) == 0 and best_axis_shift((512, 512, 512), ((12, 12),) * 3) == 0
    assert best_axis_shift((1024, 1024, 256), ((0, 0), (0, 0), (12, 12))) == 0                  # config 5 stays
    # config 3: x = 224 fills its one row segment better than 424 fills two, and fewer tiles meet a y / z slab
    assert best_axis_shift((424, 224, 824), ((12, 12),) * 3) == 1
    # ... unless a long-lived source plane is normal to z (BASELINE config 3's mode plane): only there it can be a z hole of the step pairs
    assert best_axis_shift((424, 224, 824), ((12, 12),) * 3, sheet=(2, 0.71)) == 0
    assert best_axis_shift((424, 224, 824), ((12, 12),) * 3, sheet=(2, 0.10)) == 1            # a short pulse: the faster layout wins
    assert best_axis_shift((424, 224, 824), ((12, 12),) * 3, sheet=(1, 0.71)) == 2            # normal to y: the layout that brings y to z
    assert best_axis_shift((60, 60, 400)) == 2 and best_axis_shift((60, 400, 60)) == 1
    assert best_axis_shift((258, 250, 264)) == 1            # 258 spills into a second tile, 250 does not
    assert best_axis_shift((8, 12, 20)) == 0                # tiny grids: launch-bound, left alone


@pytest.mark.parametrize("s", [1, 2])
def test_permute_spec_is_a_renaming(s):
    spec = discretize(CASES["media_mix"](), n_steps=3).spec
    p = permute_spec(spec, s)
    sig = [(a + s) % 3 for a in range(3)]
    assert p.shape == tuple(spec.shape[sig[a]] for a in range(3)) and p.bc == tuple(spec.bc[sig[a]] for a in range(3))
    # component c' of the renamed problem is component sig[c'] of the original, transposed
    for c in range(3):
        assert np.array_equal(unpermute_array(p.mat_idx[c], s), spec.mat_idx[sig[c]])
    back = permute_spec(p, 3 - s)
    assert back.shape == spec.shape and np.array_equal(back.mat_idx, spec.mat_idx)
    assert all(np.array_equal(a.ijk, b.ijk) and np.array_equal(a.comp, b.comp) for a, b in zip(back.sources, spec.sources))
    assert [(m.comps, m.lo, m.hi) for m in back.monitors] == [(tuple(m.comps), tuple(m.lo), tuple(m.hi)) for m in spec.monitors]


@pytest.mark.parametrize("case", ["media_mix", "tfsf_box", "bloch_box", "absorber_mix"])
def test_renamed_run_equals_plain_run(case, emu_lib):
    disc = discretize(CASES[case](), n_steps=14)
    outs = []
    for s in (0, 1, 2):
        with HipEngine(disc.spec, lib=emu_lib, axis_shift=s) as e:
            assert e.axis_shift == s
            e.run()
            outs.append(([e.get_field(c) for c in range(6)], e.results()))
    for s in (1, 2):
        for c in range(6):      # the sums of the CPML / Bloch corrections run in a different order: fp32 rounding
            assert np.abs(outs[s][0][c] - outs[0][0][c]).max() <= 1e-6 * max(np.abs(outs[0][0][c]).max(), 1e-30), (s, c)
        for k, v in outs[0][1].items():
            assert outs[s][1][k].shape == v.shape
            assert np.abs(outs[s][1][k] - v).max() <= 1e-6 * max(np.abs(v).max(), 1e-30), (s, k)


def test_set_field_round_trip(emu_lib):
    disc = discretize(CASES["pec_box"](), n_steps=2)
    rng = np.random.default_rng(0)
    nx, ny, nz = disc.spec.shape
    with HipEngine(disc.spec, lib=emu_lib, axis_shift=2) as e:
        for c in range(6):
            a = rng.standard_normal((nz, ny, nx)).astype(np.float32)
            e.set_field(c, a)
            assert np.array_equal(e.get_field(c), a)


def test_long_lived_source_planes_are_found_for_the_layout_choice():
    """engine.source_sheet: the plane of a mode source / a plane wave (more nodes than the two-step sweep's table takes, one grid
    plane) with the fraction of the run it injects for — what lets best_axis_shift put the plane's normal along the device's z, where
    it can be a z hole of the step pairs (BASELINE config 3: 99.8 -> 106.1 Gcells/s, profiles/r5/r5zc_c3_full_layouts.jsonl).  A
    dipole, a grid below the pairs' threshold or a run without CPML give None."""
    import tidy3d_amd.schema as td
    from tidy3d_amd.constants import C_0
    from tidy3d_amd.discretize import discretize
    from tidy3d_amd.engine import source_sheet
    f0 = C_0 / 1.55
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 10)

    def sim(source, size=(1.6, 0.8, 1.2), pml=True):
        return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=0.01), run_time=4e-13, medium=td.Medium(permittivity=2.0),
                             structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.22, 0.4)), medium=td.Medium(permittivity=12.0))],
                             sources=[source], monitors=[],
                             boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8) if pml else td.PECBoundary()), shutoff=1e-5)
    mode = td.ModeSource(center=(-0.5, 0, 0), size=(0, td.inf, td.inf), source_time=pulse, direction="+", mode_spec=td.ModeSpec(num_modes=1))
    spec = discretize(sim(mode)).spec
    assert spec.n_cells >= 1 << 20
    axis, frac = source_sheet(spec)
    assert axis == 0 and 0.3 < frac <= 1.0
    wave = td.PlaneWave(center=(0, 0, 0.4), size=(td.inf, td.inf, 0), source_time=pulse, direction="-")
    sheet = source_sheet(discretize(sim(wave)).spec)
    assert sheet is not None and sheet[0] == 2
    dip = td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ey")
    assert source_sheet(discretize(sim(dip)).spec) is None
    assert source_sheet(discretize(sim(mode, pml=False)).spec) is None
    assert source_sheet(discretize(sim(mode, size=(0.8, 0.4, 0.6))).spec) is None       # below 2^20 cells: single steps anyway
