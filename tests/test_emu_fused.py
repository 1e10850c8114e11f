"""The fused single-sweep E+H kernel against the two-pass kernels (same library, same inputs):
tile edges in x (two 256-cell tiles -> the x-halo column), y (halo wave) and z (chunk prologue),
every boundary type, materials, ADE and CPML corrections around it."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)


def _sim(N, bspec, structures=()):
    size = tuple(n * DL for n in N)
    c = (0.3 * size[0], 0.1, 0.05)
    extra = []
    if N[0] >= 128:       # wide grids: dipoles next to both x faces too (the few steps of a test must reach the x slabs)
        extra = [td.PointDipole(center=(-0.5 * size[0] + 2.3 * DL, 0.03, 0.02), source_time=PULSE, polarization="Ez"),
                 td.PointDipole(center=(0.5 * size[0] - 1.6 * DL, -0.04, 0.06), source_time=PULSE, polarization="Hy"),
                 td.PointDipole(center=(0.5 * size[0] - 3.2 * DL, 0.1, -0.07), source_time=PULSE, polarization="Ey")]
    return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12,
                         structures=list(structures),
                         sources=extra + [td.PointDipole(center=c, source_time=PULSE, polarization="Ez"),
                                  td.PointDipole(center=(-0.2 * size[0], -0.1, 0), source_time=PULSE, polarization="Hy"),
                                  td.PointDipole(center=(0.01, 0.02, -0.1), source_time=PULSE, polarization="Ex")],
                         monitors=[td.FieldMonitor(center=(0, 0, 0), size=(td.inf, td.inf, 0), freqs=[3e14], name="f"),
                                   td.FieldTimeMonitor(center=(0, 0, 0), size=(0.2, 0.2, 0.2), name="t", interval=3,
                                                       colocate=False)],
                         boundary_spec=bspec, shutoff=0)


PER = td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.periodic())
PEC = td.BoundarySpec.all_sides(td.PECBoundary())
PMC = td.BoundarySpec(x=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                      y=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                      z=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()))
PML = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.pml(num_layers=3), z=td.Boundary.pml(num_layers=3))
MEDIA = [td.Structure(geometry=td.Sphere(center=(0.05, 0, 0), radius=0.2),
                      medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])),
         td.Structure(geometry=td.Box(center=(-0.3, 0, 0), size=(0.25, 0.3, 0.2)),
                      medium=td.Medium(permittivity=3.0, conductivity=0.02)),
         td.Structure(geometry=td.Box(center=(0.3, 0.1, 0), size=(0.1, 0.1, 0.1)), medium=td.PEC)]

MEDIA_WIDE = [td.Structure(geometry=td.Box(center=(-3.0, 0, 0), size=(8.0, 0.3, 0.2)),
                           medium=td.Medium(permittivity=3.0, conductivity=0.02)),
              td.Structure(geometry=td.Sphere(center=(6.3, 0, 0), radius=0.25),
                           medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])),
              td.Structure(geometry=td.Box(center=(0.1, 0.1, 0), size=(0.1, 0.1, 0.1)), medium=td.PEC)]

PML_ODD = td.BoundarySpec(x=td.Boundary(minus=td.PML(num_layers=5), plus=td.PML(num_layers=3)), y=td.Boundary.pml(num_layers=2),
                          z=td.Boundary(minus=td.PECBoundary(), plus=td.PML(num_layers=3)))
PML_MEET = td.BoundarySpec(x=td.Boundary.pml(num_layers=7), y=td.Boundary.pml(num_layers=3), z=td.Boundary.periodic())

ABS = td.BoundarySpec(x=td.Boundary.absorber(num_layers=4, parameters=td.AbsorberParams(sigma_max=1.5)),
                      y=td.Boundary(minus=td.PML(num_layers=3), plus=td.Absorber(num_layers=3)),
                      z=td.Boundary.absorber(num_layers=3))

CONFIGS = {
    "absorber_media": ((32, 14, 10), ABS, MEDIA),
    "periodic_two_x_tiles": ((264, 10, 9), PER, ()),
    "pec_two_x_tiles": ((260, 9, 8), PEC, ()),
    "pmc_min_faces": ((16, 10, 9), PMC, ()),
    "pml_media": ((32, 14, 10), PML, MEDIA),
    # x layer counts that are not multiples of 4, rows padded to a multiple of 4 (x slab ranges are rounded
    # outwards inside the sweep); a PEC z-min wall under the z recursion
    "pml_odd_layers_padded_rows": ((22, 9, 8), PML_ODD, MEDIA),
    # the rounded x ranges meet: the whole axis is a member
    "pml_x_ranges_meet": ((2, 6, 8), PML_MEET, ()),
    "periodic_media": ((32, 12, 10), PER, MEDIA),
    # x-CPML across x-tiles: the low-x slab lies in tile 0, the high-x slab in the LAST tile
    # (tile_x > 0: the LDS coefficient table is filled from offset tile_x * 256), odd layer counts, media through both
    "pml_two_x_tiles": ((256, 8, 7), PML_ODD, MEDIA_WIDE),
}


def _run(spec, lib, variant, rows, zc, pml_mask=None, split=1, order=None, hints=None):
    with HipEngine(spec, lib=lib, variant=variant, z_chunk=zc) as e:
        e.set_option(L.OPT_ROWS, rows)
        if order is not None:
            e.set_option(L.OPT_XCD_REMAP, order)
        if hints is not None:
            e.set_option(L.OPT_MEM_HINTS, hints)
        # (grids this small default to ONE launch of the all-axes instantiation: ask for the three-launch split,
        # which is what large grids run, unless a test wants the single launch)
        e.set_option(L.OPT_PML_SPLIT, split)
        if pml_mask is not None:
            e.set_option(L.OPT_PML_FUSED, pml_mask)
        e.run()
        return [e.get_field(c) for c in range(6)], e.results()


@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("rows,zc", [(4, 16), (3, 2), (1, 1), (7, 5)])
def test_fused_equals_two_pass(name, rows, zc, emu_lib):
    N, bspec, structures = CONFIGS[name]
    disc = discretize(_sim(N, bspec, structures), n_steps=24)
    ref_f, ref_m = _run(disc.spec, emu_lib, L.VARIANT_ZMARCH, 4, 2)
    got_f, got_m = _run(disc.spec, emu_lib, L.VARIANT_FUSED, rows, zc)
    # explicit fma's + -ffp-contract=off + identical summation order: bit-for-bit agreement
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.array_equal(got_m[k], ref_m[k]), k


@pytest.mark.parametrize("order", [0, 1, 2, 5, 64])
@pytest.mark.parametrize("name", ["pec_two_x_tiles", "pml_media"])
def test_tile_order_changes_nothing(order, name, emu_lib):
    """Plain order, the contiguous XCD split, runs of G tiles per XCD (incl. a run longer than the launch): every
    tile is visited exactly once whatever the order, so the fields are bit-identical."""
    N, bspec, structures = CONFIGS[name]
    disc = discretize(_sim(N, bspec, structures), n_steps=12)
    ref_f, ref_m = _run(disc.spec, emu_lib, L.VARIANT_ZMARCH, 4, 2)
    got_f, got_m = _run(disc.spec, emu_lib, L.VARIANT_FUSED, 3, 2, split=0, order=order)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.array_equal(got_m[k], ref_m[k]), k


@pytest.mark.parametrize("name", ["pec_two_x_tiles", "periodic_two_x_tiles", "absorber_media", "pml_media", "pml_odd_layers_padded_rows"])
def test_store_order_changes_nothing(name, emu_lib):
    """FDTD_OPT_MEM_HINTS = 0 selects the instantiations with the plain store placement (fields at the end of the plane,
    the H-side psi of the CPML in the H phase) instead of the measured one (H ahead of the row exchange, non-temporal on
    the device, without CPML; H-side psi behind the E update with CPML): same values to the same places."""
    N, bspec, structures = CONFIGS[name]
    disc = discretize(_sim(N, bspec, structures), n_steps=12)
    a_f, a_m = _run(disc.spec, emu_lib, L.VARIANT_FUSED, 3, 4, hints=1)
    b_f, b_m = _run(disc.spec, emu_lib, L.VARIANT_FUSED, 3, 4, hints=0)
    for c in range(6):
        assert np.array_equal(a_f[c], b_f[c]), c
    for k in a_m:
        assert np.array_equal(a_m[k], b_m[k]), k


@pytest.mark.parametrize("name", ["pml_media", "pml_two_x_tiles"])
@pytest.mark.parametrize("mask", [0, 6, 7, -7])
@pytest.mark.parametrize("rows,zc", [(3, 16), (4, 3)])
def test_fused_cpml_placement(mask, rows, zc, name, emu_lib):
    """The CPML recursions as slab kernels (0), y/z inside the sweep (6), all inside (7): identical
    arithmetic and summation order (H: x, y, z; E: y, z, x) -> bit-for-bit the two-pass result."""
    N, bspec, structures = CONFIGS[name]
    disc = discretize(_sim(N, bspec, structures), n_steps=24 if name == "pml_media" else 14)
    ref_f, ref_m = _run(disc.spec, emu_lib, L.VARIANT_ZMARCH, 4, 2)
    # mask < 0: all axes inside the sweep as ONE launch (the small-grid default) instead of the three-launch split
    got_f, got_m = _run(disc.spec, emu_lib, L.VARIANT_FUSED, rows, zc, abs(mask), split=0 if mask < 0 else 1)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.array_equal(got_m[k], ref_m[k]), k


@pytest.mark.parametrize("bnd", [0, 3])
def test_pipelined_slab_schedule_with_corrections(emu_lib, bnd):
    """Logic of the pipelined z-slab schedule (boundary chunks + their corrections on the comm
    stream, ONE exchange per step, joined tails on monitor / decay steps, re-priming across run
    calls) with CPML, ADE, dipoles, a plane wave and monitors: 1-rank self exchange == plain run.
    (tests/test_gpu_parity.py repeats it under real stream concurrency.)"""
    from cases import pipelined_slab_case
    disc = discretize(pipelined_slab_case((20, 16, 24)), n_steps=40)
    disc.spec.decay_every = 16
    assert disc.spec.tfsf
    with HipEngine(disc.spec, lib=emu_lib) as e:
        e.run()
        ref = [e.get_field(c) for c in range(6)]
        ref_m = e.results()
    with HipEngine(disc.spec, lib=emu_lib, force_comm=True) as e:
        assert e.variant == L.VARIANT_FUSED
        e.comm_init(e.unique_id())
        if bnd:
            e.set_option(L.OPT_BND_PLANES, bnd)
        e.run(17)
        e.run(23)
        got = [e.get_field(c) for c in range(6)]
        got_m = e.results()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    for k in ref_m:
        assert np.array_equal(ref_m[k], got_m[k]), k


def test_autotuned_tile_shape_changes_nothing_but_the_shape(emu_lib):
    """With FDTD_OPT_AUTOTUNE fdtd_run times a few (rows, z-chunk) shapes of the fused sweep on its
    first call (grids of >= 2^20 cells; forced here) and keeps the fastest: the sweep only reads set A and writes set B,
    so the probing has no side effect, and the arithmetic does not depend on the launch geometry."""
    N, bspec, structures = CONFIGS["pml_media"]
    disc = discretize(_sim(N, bspec, structures), n_steps=20)
    with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED) as e:
        e.set_option(L.OPT_AUTOTUNE, 0)
        e.run()
        ref = [e.get_field(c) for c in range(6)]
        ref_m = e.results()
        assert (e.stats().tile_rows, e.stats().tile_zchunk) == (3, 16)
    with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED) as e:
        e.set_option(L.OPT_AUTOTUNE, 2)
        e.run(7)
        e.run(13)
        st = e.stats()
        assert st.tile_rows in (2, 3) and st.tile_zchunk in (8, 16, 24, 32)
        got = [e.get_field(c) for c in range(6)]
        got_m = e.results()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    for k in ref_m:
        assert np.array_equal(ref_m[k], got_m[k]), k


@pytest.mark.parametrize("faces", ["y-", "y+", "z-", "z+", "y-z+", "x-y+z-"])
def test_split_cpml_launch_one_sided(emu_lib, faces):
    """The y/z recursions folded into the sweep run only in the tile rows / plane ranges that meet a
    slab (three launches: z-slab planes, y-slab tile rows, plain middle); absorbers on single faces
    move every range boundary.  Bit-for-bit the two-pass result."""
    def edge(axis, side):
        return td.PML(num_layers=4) if f"{axis}{side}" in faces else td.PECBoundary()
    bspec = td.BoundarySpec(x=td.Boundary(minus=edge("x", "-"), plus=edge("x", "+")),
                            y=td.Boundary(minus=edge("y", "-"), plus=edge("y", "+")),
                            z=td.Boundary(minus=edge("z", "-"), plus=edge("z", "+")))
    disc = discretize(_sim((24, 22, 21), bspec, MEDIA), n_steps=20)
    ref_f, ref_m = _run(disc.spec, emu_lib, L.VARIANT_ZMARCH, 4, 2)
    got_f, got_m = _run(disc.spec, emu_lib, L.VARIANT_FUSED, 3, 5, 6)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.array_equal(got_m[k], ref_m[k]), k


@pytest.mark.parametrize("case", ["bloch_box", "bloch_xy_pml_z"])
def test_bloch_fused_equals_two_pass(case, emu_lib):
    """Bloch boundaries (ghost-cell device layout, complex fields as a solver pair): the fused sweep and the
    two-pass kernels — vector and scalar — give the same bits."""
    import cases
    disc = discretize(cases.CASES[case](), n_steps=12)
    out = []
    for variant in (L.VARIANT_FUSED, L.VARIANT_ZMARCH, L.VARIANT_SIMPLE):
        with HipEngine(disc.spec, lib=emu_lib, variant=variant) as e:
            e.run()
            out.append(([e.get_field(c) for c in range(6)], e.results()))
    for f, m in out[1:]:
        for c in range(6):
            if case == "bloch_box":
                # a Bloch z: the fused sweep's chunk prologue updates the ROTATED plane below the slab, the two-pass
                # kernels rotate the UPDATED plane — the same number up to fp32 rounding
                assert np.abs(f[c] - out[0][0][c]).max() <= 2e-6 * np.abs(out[0][0][c]).max(), c
            else:
                assert np.array_equal(f[c], out[0][0][c]), c
        for k in m:
            if case != "bloch_box":
                assert np.array_equal(m[k], out[0][1][k]), k
    assert np.iscomplexobj(out[0][0][0]) and np.abs(out[0][0][0].imag).max() > 0


@pytest.mark.parametrize("name", ["pec_two_x_tiles", "pml_media"])
def test_placement_probe_changes_nothing(name, emu_lib):
    """The first large run may move the field arrays to other allocations (``probe_placement``: it times plain sweeps on
    up to three further sets and keeps the fastest).  Forced on here (option value >= 100 lifts the size threshold):
    initial fields set BEFORE the run must arrive in whatever set is kept, and everything after is bit-identical."""
    N, bspec, structures = CONFIGS[name]
    disc = discretize(_sim(N, bspec, structures), n_steps=12)
    rng = np.random.default_rng(3)
    nx, ny, nz = disc.spec.shape
    init = [rng.uniform(-1e-3, 1e-3, (nz, ny, nx)).astype(np.float32) for _ in range(6)]

    def run(tries):
        with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED) as e:
            e.set_option(L.OPT_PLACEMENT_TRIES, tries)
            for c in range(6):
                e.set_field(c, init[c])
            e.run(5)
            mid = [e.get_field(c) for c in range(6)]
            st = e.run()
            assert (int(st.placement) >> 8) == tries % 100, st.placement          # candidates timed
            return mid, [e.get_field(c) for c in range(6)], e.results()
    ref = run(0)
    got = run(103)
    for a, b in zip(ref[0] + ref[1], got[0] + got[1]):
        assert np.array_equal(a, b)
    for k in ref[2]:
        assert np.array_equal(ref[2][k], got[2][k]), k


@pytest.mark.parametrize("tb", [3, 4, 7, 4096 + 4])
@pytest.mark.parametrize("name", ["absorber_media", "pec_two_x_tiles", "pmc_min_faces", "pec_media_tall"])
def test_two_step_slab_schedule_changes_nothing(name, tb, emu_lib):
    """FDTD_OPT_TBLOCK: two time steps per pass over the grid, slab by slab (A(s+1) = step n on slab s+1, then B(s) = step
    n+1 on slab s, in place) with the corrections (dipoles on E and H, ADE, absorber layers, lossy and PEC media) applied
    per slab.  Same kernels on the same values -> the same bits as single steps; pairs give way to single steps around
    monitor records (the time monitor records every 3rd step) and field-decay checks, and across run() calls."""
    if name == "pec_media_tall":
        N, bspec, structures = (24, 10, 19), PEC, MEDIA
    elif name == "absorber_media":
        ab = td.Boundary.absorber(num_layers=3)
        N, bspec, structures = (24, 12, 10), td.BoundarySpec(x=ab, y=td.Boundary(minus=td.PECBoundary(), plus=td.Absorber(num_layers=4)), z=ab), MEDIA
    else:
        N, bspec, structures = CONFIGS[name]
    disc = discretize(_sim(N, bspec, structures), n_steps=26)
    disc.spec.decay_every = 8

    def run(tblock):
        with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
            e.set_option(L.OPT_ROWS, 3)
            e.set_option(L.OPT_TBLOCK, tblock)
            st = e.run(11)
            pairs = int(st.two_step_pairs)
            st = e.run(15)
            return [e.get_field(c) for c in range(6)], e.results(), pairs + int(st.two_step_pairs), int(st.tblock_planes)
    ref_f, ref_m, p0, t0 = run(0)
    got_f, got_m, p1, t1 = run(tb)
    assert p0 == 0 and t0 == 0
    nz = disc.spec.shape[2]
    if nz >= 2 * (tb % 4096):
        assert t1 == tb % 4096 and p1 >= 3, (p1, t1)          # pairs were really taken
    else:
        assert p1 == 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.array_equal(got_m[k], ref_m[k]), k
