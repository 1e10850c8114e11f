"""shell2 pairs (fdtd_shell2.hpp, fdtd_capi.hip): step pairs on grids walled by CPML with the shell advanced by
shell2_step_kernel — TWO steps per sweep with the CPML recursions (psi, both sides ping-ponged) carried through both — against
single steps of the same library on the CPU emulator: the same formulas in the same order -> the same bits.  Random initial
fields fill the layers from the first step on; layer counts that are odd, different per face or absent on a face (a PEC or PMC
wall instead), StablePML, one / two / three x tiles of the bulk, materials running through the layers, tile shapes of the boxes
(lanes per row, waves per workgroup, planes per chunk), sources deep inside the bulk (applied by the bulk sweep) and sources
inside the shell (the pair falls back to the single-step shell while they inject), monitors inside the bulk recorded from
pairs, runs cut in two (psi parities of both sides carried across runs and into single steps)."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)


def pml(n):
    return td.PML(num_layers=n)


B_ALL = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.pml(num_layers=3), z=td.Boundary.pml(num_layers=3))
B_ODD = td.BoundarySpec(x=td.Boundary(minus=pml(5), plus=pml(3)), y=td.Boundary(minus=pml(2), plus=pml(4)),
                        z=td.Boundary(minus=td.PECBoundary(), plus=pml(3)))
B_XZ = td.BoundarySpec(x=td.Boundary(minus=pml(4), plus=td.PECBoundary()), y=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                       z=td.Boundary.pml(num_layers=2))
B_YZ = td.BoundarySpec(x=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()), y=td.Boundary.pml(num_layers=3),
                       z=td.Boundary(minus=td.PMCBoundary(), plus=pml(4)))
B_STABLE = td.BoundarySpec(x=td.Boundary(minus=td.StablePML(num_layers=9), plus=td.StablePML(num_layers=6)), y=td.Boundary.pml(num_layers=3),
                           z=td.Boundary.pml(num_layers=3))

SHAPES = {
    "one_tile": (48, 22, 20),
    "one_tile_wide": (97, 23, 19),
    "two_x_tiles": (300, 20, 19),
    "seam_at_bulk_edge": (256, 18, 18),
    "three_x_tiles": (536, 18, 17),
    "one_tile_100": (100, 23, 19),      # (a periodic x needs rows of a multiple of four cells to stay on the fused sweep)
}

MEDIA = [td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.3, 0.25)), medium=td.Medium(permittivity=3.0, conductivity=0.02)),
         td.Structure(geometry=td.Sphere(center=(0.15, 0.1, 0), radius=0.22), medium=td.Medium(permittivity=2.0)),
         td.Structure(geometry=td.Box(center=(-0.3, -0.1, 0.1), size=(0.1, 0.1, 0.1)), medium=td.PEC)]


def _sim(N, bspec, structures=(), monitors=(), extra=(), deep_only=True):
    size = tuple((n - 1e-6) * DL for n in N)
    hx = 0.5 * size[0]
    srcs = [td.PointDipole(center=(0.02, 0.01, 0.03), source_time=PULSE, polarization="Ez"),
            td.PointDipole(center=(-0.11, 0.06, -0.05), source_time=PULSE, polarization="Ex"),
            td.PointDipole(center=(0.07, -0.04, 0.02), source_time=PULSE, polarization="Hy")]
    if not deep_only:      # inside the shell: in the x-min layers
        srcs.append(td.PointDipole(center=(-hx + 1.3 * DL, 0.03, 0.02), source_time=PULSE, polarization="Ey"))
    return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, sources=srcs + list(extra),
                         structures=list(structures), monitors=list(monitors), boundary_spec=bspec, shutoff=0)


def _run(spec, lib, twostep, shell2, shape=0, runs=(11, 15), seed=7, fields=True):
    # (shell2 = 1: one launch of the all-axes instantiation over the six boxes; 2: one launch per instantiation over finer boxes; 3: one per box)
    with HipEngine(spec, lib=lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
        e.set_option(L.OPT_ROWS, 3)
        e.set_option(L.OPT_PML_SPLIT, 1)
        e.set_option(L.OPT_TWOSTEP, twostep)
        e.set_option(L.OPT_SHELL_PAIRS, 1)
        e.set_option(L.OPT_SHELL2, shell2)
        if shape:
            e.set_option(L.OPT_SHELL2_SHAPE, shape)
        if fields:
            rng = np.random.default_rng(seed)
            for c in range(6):
                f = e.get_field(c)
                amp = 1e-3 if c < 3 else 1e-3 / 376.73
                e.set_field(c, (amp * rng.uniform(-1, 1, size=f.shape)).astype(np.float32))
        pairs = shell_pairs = shell2_pairs = 0
        why = 0
        for r in runs:
            st = e.run(r)
            pairs += int(st.fused2_pairs)
            shell_pairs += int(st.shell_pairs)
            shell2_pairs += int(st.shell2_pairs)
            why = int(st.fused2_off_reason)
        return [e.get_field(c) for c in range(6)], e.results(), pairs, shell_pairs, shell2_pairs, why


def shape_word(qw=0, ww=8, zcw=0, ws=4, zcs=0):
    return qw + 128 * (ww % 8) + 1024 * zcw + (ws << 17) + (zcs << 21)


CASES = [("one_tile", B_ALL, 5, 3, shape_word()), ("one_tile", B_ALL, 5, 3, shape_word(qw=9, ww=7)), ("one_tile_wide", B_ALL, 5, 3, shape_word(qw=21, ww=3)), ("one_tile", B_ODD, 16, 32, shape_word(qw=8, ww=4, zcw=3, ws=2, zcs=5)),
         ("one_tile", B_XZ, 4, 2, shape_word(qw=16, ww=2)), ("one_tile", B_YZ, 8, 5, shape_word(qw=5, ww=3, zcw=4)),
         ("one_tile_wide", B_STABLE, 6, 4, shape_word(qw=11, ww=4, ws=8, zcs=3)),
         ("two_x_tiles", B_ALL, 5, 3, shape_word(qw=64, ww=5)), ("two_x_tiles", B_ALL, 5, 3, shape_word(qw=62, ww=5)), ("two_x_tiles", B_STABLE, 7, 3, shape_word(qw=31, ww=6)), ("two_x_tiles", B_ODD, 8, 4, shape_word(qw=32, ww=2, zcw=6)),
         ("seam_at_bulk_edge", B_ODD, 6, 5, shape_word(qw=20, ww=4)),
         ("three_x_tiles", B_ALL, 6, 32, shape_word())]


@pytest.mark.parametrize("name,bspec,w,zc,shape", CASES)
def test_shell2_pairs_equal_single_steps(name, bspec, w, zc, shape, emu_lib):
    N = SHAPES[name]
    disc = discretize(_sim(N, bspec), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, _, p0, s0, q0, why0 = _run(disc.spec, emu_lib, 0, 0)
    got_f, _, p1, s1, q1, why1 = _run(disc.spec, emu_lib, w + 64 * zc, 1 + (w + zc) % 3, shape)
    assert p0 == 0 and s0 == 0 and q0 == 0 and why0 == 1            # switched off
    assert p1 == 5 + 7 and q1 == p1 and why1 == 0, (p1, s1, q1, why1)
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), (c, float(np.abs(got_f[c] - ref_f[c]).max()), np.argwhere(got_f[c] != ref_f[c])[:5])


@pytest.mark.parametrize("name,bspec,w,zc,shape", [("one_tile", B_ODD, 5, 3, shape_word(qw=12, ww=4)), ("two_x_tiles", B_ALL, 8, 4, shape_word(qw=32, ww=3, zcw=5, ws=3)),
                                                   ("one_tile_wide", B_STABLE, 6, 6, shape_word(qw=64, ww=8))])
def test_shell2_pairs_with_materials_and_monitors(name, bspec, w, zc, shape, emu_lib):
    """Dielectric / lossy / PEC bodies running through the layers (packed medium words: the MAT instantiation), a probe, a time
    monitor and a DFT flux plane inside the bulk recorded from pairs — fields and records equal those of single steps."""
    N = SHAPES[name]
    mons = [td.FieldTimeMonitor(center=(0.03, 0.02, 0.01), size=(0, 0, 0), name="probe", interval=1),
            td.FieldTimeMonitor(center=(0.0, 0.0, 0.0), size=(0.2, 0.1, 0), name="patch", interval=2, fields=("Ex", "Hz")),
            td.FluxMonitor(center=(0.05, 0, 0), size=(0, 0.2, 0.2), freqs=[2.5e14, 3e14], name="flux")]
    disc = discretize(_sim(N, bspec, structures=MEDIA, monitors=mons), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, ref_r, p0, _, q0, _ = _run(disc.spec, emu_lib, 0, 0)
    got_f, got_r, p1, s1, q1, why1 = _run(disc.spec, emu_lib, w + 64 * zc, 1, shape)
    assert p0 == 0 and q0 == 0
    assert q1 > 0 and q1 == p1 and why1 == 0, (p1, s1, q1, why1)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), (c, float(np.abs(got_f[c] - ref_f[c]).max()))
    assert set(got_r) == set(ref_r)
    for k in ref_r:
        assert np.array_equal(np.asarray(got_r[k]), np.asarray(ref_r[k])), k


def test_sources_inside_the_shell_fall_back_to_the_single_step_shell(emu_lib):
    """The boxes of a shell2 pair apply no sources: while a list with a node in (or within three cells of) the shell injects,
    the pair goes out in the round-4 form — the shell as two single steps through the third set — and the psi parities of both
    forms stay consistent; same bits as single steps."""
    N = SHAPES["one_tile"]
    disc = discretize(_sim(N, B_ALL, deep_only=False), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, _, p0, _, _, _ = _run(disc.spec, emu_lib, 0, 0)
    got_f, _, p1, s1, q1, why1 = _run(disc.spec, emu_lib, 5 + 64 * 3, 1)
    assert p0 == 0 and p1 == 12 and s1 == 12 and q1 == 0 and why1 == 0, (p1, s1, q1, why1)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c


def test_shell2_and_single_step_shell_alternate(emu_lib):
    """Runs that alternate between the two forms of a shell pair and single steps (the option is switched between runs): the E-side
    psi lives in either of its two sets, the parameter blocks of every kernel follow it."""
    N = SHAPES["one_tile"]
    disc = discretize(_sim(N, B_ODD, structures=MEDIA[:1]), n_steps=30)
    disc.spec.decay_every = 0
    ref_f, _, _, _, _, _ = _run(disc.spec, emu_lib, 0, 0, runs=(30,))
    with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
        e.set_option(L.OPT_ROWS, 3)
        e.set_option(L.OPT_PML_SPLIT, 1)
        e.set_option(L.OPT_SHELL_PAIRS, 1)
        rng = np.random.default_rng(7)
        for c in range(6):
            f = e.get_field(c)
            amp = 1e-3 if c < 3 else 1e-3 / 376.73
            e.set_field(c, (amp * rng.uniform(-1, 1, size=f.shape)).astype(np.float32))
        took = []
        for steps, twostep, s2 in [(3, 5 + 64 * 3, 1), (4, 5 + 64 * 3, 0), (5, 0, 0), (6, 6 + 64 * 4, 1), (3, 0, 1), (9, 5 + 64 * 2, 1)]:
            e.set_option(L.OPT_TWOSTEP, twostep)
            e.set_option(L.OPT_SHELL2, s2)
            st = e.run(steps)
            took.append((int(st.fused2_pairs), int(st.shell2_pairs)))
        got_f = [e.get_field(c) for c in range(6)]
    assert took == [(1, 1), (2, 0), (0, 0), (3, 3), (0, 0), (4, 4)], took
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c


def test_random_simulations_in_shell2_pairs(emu_lib):
    """scripts/fuzz_shell2.py on the emulator: random grids, walls (CPML / StablePML of random thickness, PEC, PMC on min faces), bodies
    through the layers, random initial fields, dipoles deep inside the bulk, monitors inside the bulk, split runs, the three forms
    of the launches and random tile shapes of the boxes — shell2 pairs == single steps, bit for bit (the GPU suite runs 40 more)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_shell2", os.path.join(os.path.dirname(__file__), "..", "scripts", "fuzz_shell2.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, taken = mod.run_cases(6, seed=21, lib=emu_lib, quiet=True, small=True)
    assert bad == 0 and taken >= 5, (bad, taken)


@pytest.mark.parametrize("name,bspec,w,zc,form", [("one_tile", B_ALL, 5, 3, 1), ("two_x_tiles", B_ODD, 8, 4, 2), ("one_tile_wide", B_STABLE, 6, 6, 3)])
def test_dft_monitors_reaching_into_the_shell_do_not_stop_shell2_pairs(name, bspec, w, zc, form, emu_lib):
    """Flux planes and field monitors normally span the whole cross-section, layers included.  In shell2 pairs the shell's boxes
    copy the middle step out over the monitors' boxes as the bulk sweep does over its own (H^{n+1/2} for a record at the first
    step of a pair, E^{n+1} for one at the middle step; each cell by the launch that owns it): every step goes out in a pair and
    the spectra equal those of single steps, bit for bit.  (The round-4 form ends its pairs on such records.)"""
    N = SHAPES[name]
    mons = [td.FluxMonitor(center=(0.1, 0, 0), size=(0, td.inf, td.inf), freqs=[2.5e14, 3e14], name="flux_x"),
            td.FluxMonitor(center=(0, 0, 0.1), size=(td.inf, td.inf, 0), freqs=[3e14], name="flux_z", interval_space=(1, 1, 1)),
            td.FieldMonitor(center=(0, 0.05, 0), size=(td.inf, 0, td.inf), freqs=[2.5e14, 3e14], name="plane_y", colocate=False),
            td.FieldMonitor(center=(0, 0, 0), size=(0.5, td.inf, 0.3), freqs=[3e14], name="volume", colocate=False)]
    disc = discretize(_sim(N, bspec, structures=MEDIA[:2], monitors=mons), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, ref_r, p0, _, q0, _ = _run(disc.spec, emu_lib, 0, 0)
    got_f, got_r, p1, s1, q1, why1 = _run(disc.spec, emu_lib, w + 64 * zc, form)
    assert p0 == 0 and q0 == 0
    assert p1 == 5 + 7 and q1 == p1 and why1 == 0, (p1, s1, q1, why1)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_r:
        assert np.abs(np.asarray(ref_r[k])).max() > 0 and np.array_equal(np.asarray(got_r[k]), np.asarray(ref_r[k])), k


@pytest.mark.parametrize("kind", ["current_sheet", "plane_wave", "two_sheets"])
def test_shell2_pairs_with_z_holes_for_injecting_source_planes(kind, emu_lib):
    """Source lists the sweeps cannot apply themselves — a current sheet of hundreds of nodes that runs through the layers (what a
    mode plane is), the injection plane of a plane wave (TFSF corrections + its 1-D incident grid) — keep single steps in their own
    planes (+- 2) only: those planes are z holes that take two single steps through the third set, with their psi routed through
    temporary sets; the intervals between them go out as clipped two-step sweeps with their own shell2 boxes.  Pairs from the first
    step on, through the pulse and after it; DFT planes clear of the holes recorded from pairs; same bits as single steps."""
    N = (40, 20, 60)
    size = tuple((n - 1e-6) * DL for n in N)
    pulse = td.GaussianPulse(freq0=3e14, fwidth=2.4e14)
    glass = td.Structure(geometry=td.Box(center=(0, 0, -0.9), size=(td.inf, td.inf, 0.6)), medium=td.Medium(permittivity=2.1))
    ball = td.Structure(geometry=td.Sphere(center=(0.1, 0.0, -0.1), radius=0.25), medium=td.Medium(permittivity=3.0, conductivity=0.02))
    if kind == "plane_wave":
        srcs = [td.PlaneWave(center=(0, 0, 0.9), size=(td.inf, td.inf, 0), source_time=pulse, direction="-")]
        bspec = B_ALL
    else:
        srcs = [td.UniformCurrentSource(center=(0, 0, 0.8), size=(td.inf, td.inf, 0), source_time=pulse, polarization="Ey")]
        if kind == "two_sheets":
            srcs.append(td.UniformCurrentSource(center=(0.1, 0, -0.5), size=(1.0, 0.6, 0), source_time=pulse, polarization="Hx"))
        bspec = B_ODD if kind == "two_sheets" else B_ALL
    mons = [td.FieldTimeMonitor(center=(0.1, 0.05, 0.2), size=(0, 0, 0), name="probe", interval=1, colocate=False),
            td.FluxMonitor(center=(0, 0, 0.3), size=(td.inf, td.inf, 0), freqs=[2.5e14, 3e14], name="flux")]
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1.1e-14, structures=[glass, ball], sources=srcs,
                        monitors=mons, boundary_spec=bspec, shutoff=0)
    disc = discretize(sim)
    disc.spec.decay_every = 0
    spec = disc.spec
    n1 = 41
    assert spec.n_steps > n1 + 30
    ref_f, ref_r, p0, _, q0, _ = _run(spec, emu_lib, 0, 0, runs=(n1, spec.n_steps - n1), fields=False)
    got_f, got_r, p1, s1, q1, why = _run(spec, emu_lib, 8 + 64 * 6, 1, runs=(n1, spec.n_steps - n1), fields=False)
    assert p0 == 0 and q0 == 0
    assert q1 == p1 and p1 >= spec.n_steps // 2 - 2, (p1, s1, q1, spec.n_steps, why)
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), (c, float(np.abs(got_f[c] - ref_f[c]).max()))
    for k in ref_r:
        assert np.abs(np.asarray(ref_r[k])).max() > 0 and np.array_equal(np.asarray(got_r[k]), np.asarray(ref_r[k])), k


# ---- periodic x / y faces and dispersive cells in shell2 pairs (round 5, last part) ---------------------------------------------------
PER = td.Boundary.periodic()
B_PXY_Z = td.BoundarySpec(x=PER, y=PER, z=td.Boundary.pml(num_layers=3))                                    # a metasurface cell: BASELINE config 5's walls
B_PX_YZ = td.BoundarySpec(x=PER, y=td.Boundary(minus=pml(2), plus=pml(4)), z=td.Boundary.pml(num_layers=3))     # a grating: periodic x, layers on y and z
B_PY_XZ = td.BoundarySpec(x=td.Boundary(minus=pml(5), plus=pml(3)), y=PER, z=td.Boundary(minus=td.PECBoundary(), plus=pml(3)))
PER_CASES = [("one_tile", B_PXY_Z, 5, 3, shape_word()), ("one_tile", B_PXY_Z, 8, 4, shape_word(qw=9, ww=3, zcw=2)),
             ("two_x_tiles", B_PXY_Z, 6, 4, shape_word(qw=62, ww=5)), ("three_x_tiles", B_PXY_Z, 6, 32, shape_word(qw=33, ww=4)),
             ("one_tile_100", B_PX_YZ, 6, 3, shape_word(qw=13, ww=4)), ("two_x_tiles", B_PX_YZ, 5, 5, shape_word()),
             ("one_tile", B_PY_XZ, 5, 3, shape_word()), ("two_x_tiles", B_PY_XZ, 7, 4, shape_word(qw=20, ww=6, ws=2, zcs=4))]


@pytest.mark.parametrize("name,bspec,w,zc,shape", PER_CASES)
def test_shell2_pairs_with_periodic_faces(name, bspec, w, zc, shape, emu_lib):
    """A periodic x wraps through the halo lanes of the boxes (and as one more seam of the bulk sweep); a periodic y keeps boxes and
    bulk two rows clear of the wrap — those rows take two single steps through the third set beside them, psi routed through the
    temporary sets like a z hole.  Materials through the wrap, dipoles deep inside, random fields: the same bits as single steps."""
    N = SHAPES[name]
    disc = discretize(_sim(N, bspec, structures=MEDIA), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, _, p0, s0, q0, why0 = _run(disc.spec, emu_lib, 0, 0)
    got_f, _, p1, s1, q1, why1 = _run(disc.spec, emu_lib, w + 64 * zc, 1, shape)
    assert p0 == 0 and q0 == 0
    assert p1 == 5 + 7 and q1 == p1 and why1 == 0, (p1, s1, q1, why1)
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), (c, float(np.abs(got_f[c] - ref_f[c]).max()), np.argwhere(got_f[c] != ref_f[c])[:5])


@pytest.mark.parametrize("walls", ["periodic_xy", "periodic_x", "pml"])
def test_shell2_pairs_with_dispersive_planes_and_a_plane_wave(walls, emu_lib):
    """BASELINE config 5 in small: a unit cell periodic in x and y with layers on z, a dispersive disc (Drude: its planes are a z hole
    of the bulk — the ADE state advances every step — with the memory term applied in the middle step), a lossy substrate, a plane
    wave (its injection plane: one more hole while it injects) and flux planes that span the wrap (their record steps take single
    steps; between them: pairs).  Also with layers on y, and on every face.  Same bits as single steps: fields and records."""
    N = (40, 24, 64)
    size = tuple((n - 1e-6) * DL for n in N)
    pulse = td.GaussianPulse(freq0=3e14, fwidth=2.4e14)
    glass = td.Structure(geometry=td.Box(center=(0, 0, -1.0), size=(td.inf, td.inf, 0.8)), medium=td.Medium(permittivity=2.1, conductivity=0.01))
    disc_m = td.Structure(geometry=td.Cylinder(center=(0.1, 0.05, -0.3), radius=0.3, length=0.2, axis=2), medium=td.Drude(eps_inf=2.0, coeffs=[(1.2e15, 9e13)]))
    srcs = [td.PlaneWave(center=(0, 0, 0.8), size=(td.inf, td.inf, 0), source_time=pulse, direction="-")]
    bspec = {"periodic_xy": td.BoundarySpec(x=PER, y=PER, z=td.Boundary.pml(num_layers=4)),
             "periodic_x": td.BoundarySpec(x=PER, y=td.Boundary.pml(num_layers=3), z=td.Boundary.pml(num_layers=4)),
             "pml": B_ALL}[walls]
    if walls != "periodic_xy":      # (a plane wave needs periodic or far walls on its transverse axes: a finite sheet instead)
        srcs = [td.UniformCurrentSource(center=(0, 0, 0.8), size=(td.inf, td.inf, 0), source_time=pulse, polarization="Ex")]
    mons = [td.FieldTimeMonitor(center=(0.1, 0.05, 0.3), size=(0, 0, 0), name="probe", interval=3, colocate=False),
            td.FluxMonitor(center=(0, 0, 0.5), size=(td.inf, td.inf, 0), freqs=[2.5e14, 3e14], name="R"),
            td.FluxMonitor(center=(0, 0, -1.2), size=(td.inf, td.inf, 0), freqs=[3e14], name="T")]
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1.3e-14, structures=[glass, disc_m], sources=srcs,
                        monitors=mons, boundary_spec=bspec, shutoff=0)
    disc = discretize(sim)
    disc.spec.decay_every = 0
    spec = disc.spec
    n1 = 37
    assert spec.n_steps > n1 + 40
    ref_f, ref_r, p0, _, q0, _ = _run(spec, emu_lib, 0, 0, runs=(n1, spec.n_steps - n1), fields=False)
    got_f, got_r, p1, s1, q1, why = _run(spec, emu_lib, 8 + 64 * 6, 1, runs=(n1, spec.n_steps - n1), fields=False)
    assert p0 == 0 and q0 == 0
    assert q1 >= spec.n_steps // 4, (p1, s1, q1, spec.n_steps, why)
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), (c, float(np.abs(got_f[c] - ref_f[c]).max()))
    for k in ref_r:
        assert np.abs(np.asarray(ref_r[k])).max() > 0 and np.array_equal(np.asarray(got_r[k]), np.asarray(ref_r[k])), k
