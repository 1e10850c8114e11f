"""Shell pairs (fdtd_capi.hip): step pairs on grids walled by CPML — the two-step sweep over the bulk, the shell (CPML slabs +
collar) as two single steps of the production kernels beside it (z slabs and y slabs through fused_step_kernel with its row
exclusion, x strips through strip_step_kernel) — against single steps of the same library on the CPU emulator: the same
formulas in the same order -> the same bits.  Layer counts that are odd, different per face or absent on a face (a PEC or PMC
wall instead), one / two / three x tiles (a seam inside the bulk, a seam next to the bulk's edge), materials running through
the layers, sources inside the shell and next to the bulk's faces, monitors inside the bulk recorded from pairs, monitors
reaching into the shell giving way to single steps."""
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

PER = td.Boundary.periodic()
B_PXY = td.BoundarySpec(x=PER, y=PER, z=td.Boundary.pml(num_layers=3))            # a metasurface's unit cell
B_PALL = td.BoundarySpec(x=PER, y=PER, z=PER)
B_PZ = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary(minus=pml(3), plus=td.PECBoundary()), z=PER)
B_PY = td.BoundarySpec(x=td.Boundary(minus=td.PECBoundary(), plus=pml(4)), y=PER, z=td.Boundary(minus=td.PMCBoundary(), plus=pml(3)))

SHAPES = {
    "one_tile": (48, 22, 20),
    "one_tile_wide": (97, 23, 19),       # + 9 + 6 StablePML layers = 112 columns
    "two_x_tiles": (300, 20, 19),        # seam at column 256 inside the bulk
    "seam_at_bulk_edge": (256, 18, 18),  # + 5 + 3 x layers = 264 columns: the bulk ends at column 256, the seam's right column belongs to the shell
    "three_x_tiles": (536, 18, 17),
}

MEDIA = [td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.3, 0.25)), medium=td.Medium(permittivity=3.0, conductivity=0.02)),
         td.Structure(geometry=td.Sphere(center=(0.15, 0.1, 0), radius=0.22), medium=td.Medium(permittivity=2.0)),
         td.Structure(geometry=td.Box(center=(-0.3, -0.1, 0.1), size=(0.1, 0.1, 0.1)), medium=td.PEC)]


def _sim(N, bspec, structures=(), monitors=(), extra=()):
    size = tuple((n - 1e-6) * DL for n in N)          # (a hair under n cells: ceil(size / dl) must not round up)
    hx, hy, hz = (0.5 * s for s in size)
    srcs = [td.PointDipole(center=(0.02, 0.01, 0.03), source_time=PULSE, polarization="Ez"),
            td.PointDipole(center=(-0.11, 0.12, -0.1), source_time=PULSE, polarization="Ex"),
            # inside the shell: in the x-min layers, in the y-max collar, in a z-max layer; and a magnetic one in the x-max collar
            td.PointDipole(center=(-hx + 1.3 * DL, 0.03, 0.02), source_time=PULSE, polarization="Ey"),
            td.PointDipole(center=(0.1, hy - 3.6 * DL, -0.07), source_time=PULSE, polarization="Ez"),
            td.PointDipole(center=(-0.05, 0.06, hz - 1.2 * DL), source_time=PULSE, polarization="Ex"),
            td.PointDipole(center=(hx - 4.4 * DL, -0.02, 0.04), source_time=PULSE, polarization="Hz")]
    if N[0] > 256:      # on both sides of the seam at column 256
        srcs += [td.PointDipole(center=(-hx + 255.0 * DL, -0.05, 0.04), source_time=PULSE, polarization="Ey"),
                 td.PointDipole(center=(-hx + 256.5 * DL, 0.0, 0.0), source_time=PULSE, polarization="Ex")]
    return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, sources=srcs + list(extra),
                         structures=list(structures), monitors=list(monitors), boundary_spec=bspec, shutoff=0)


def _run(spec, lib, twostep, shell=1, runs=(11, 15), split=1, disp=-1):       # (shell = 1: pairs whatever the cost model says of these small grids)
    with HipEngine(spec, lib=lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
        e.set_option(L.OPT_ROWS, 3)
        if disp >= 0:
            e.set_option(L.OPT_DISP, disp)
        e.set_option(L.OPT_PML_SPLIT, split)
        e.set_option(L.OPT_TWOSTEP, twostep)
        e.set_option(L.OPT_SHELL_PAIRS, shell)
        pairs = shell_pairs = 0
        why = 0
        for r in runs:
            st = e.run(r)
            pairs += int(st.fused2_pairs)
            shell_pairs += int(st.shell_pairs)
            why = int(st.fused2_off_reason)
        return [e.get_field(c) for c in range(6)], e.results(), pairs, shell_pairs, why


CASES = [("one_tile", B_ALL, 5, 3), ("one_tile", B_ODD, 16, 32), ("one_tile", B_XZ, 4, 2), ("one_tile", B_YZ, 8, 5),
         ("one_tile_wide", B_STABLE, 6, 4),
         ("two_x_tiles", B_ALL, 5, 3), ("two_x_tiles", B_ODD, 8, 4),
         ("seam_at_bulk_edge", B_ODD, 6, 5),
         ("three_x_tiles", B_ALL, 6, 32)]


@pytest.mark.parametrize("name,bspec,w,zc", CASES)
def test_shell_pairs_equal_single_steps(name, bspec, w, zc, emu_lib):
    N = SHAPES[name]
    disc = discretize(_sim(N, bspec), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, _, p0, s0, why0 = _run(disc.spec, emu_lib, 0)
    got_f, _, p1, s1, why1 = _run(disc.spec, emu_lib, w + 64 * zc)
    assert p0 == 0 and s0 == 0 and why0 == 1            # switched off
    assert p1 == 5 + 7 and s1 == p1 and why1 == 0, (p1, s1, why1)
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), (c, float(np.abs(got_f[c] - ref_f[c]).max()))


PERIODIC_CASES = [("one_tile", B_PXY, 5, 3), ("two_x_tiles", B_PXY, 6, 4), ("three_x_tiles", B_PALL, 16, 32), ("one_tile", B_PALL, 4, 2),
                  ("one_tile", B_PZ, 5, 4), ("two_x_tiles", B_PY, 8, 5)]


@pytest.mark.parametrize("name,bspec,w,zc", PERIODIC_CASES)
def test_shell_pairs_with_periodic_faces(name, bspec, w, zc, emu_lib):
    """Periodic faces in step pairs: periodic x wraps inside the two-step sweep (the row's last column | its first is one more
    seam, repaired by the seam kernel; with one, two and three x tiles), a periodic y / z face has the two rows / planes next to
    it in the shell (the single steps wrap as always; the middle step's wrapped z planes are refreshed).  Materials through the
    wrap, dipoles of both kinds in the first and last rows / columns / planes.  Same bits as single steps."""
    N = SHAPES[name]
    size = tuple(n * DL for n in N)
    hx, hy, hz = (0.5 * v for v in size)
    extra = [td.PointDipole(center=(-hx + 0.3 * DL, -hy + 0.6 * DL, 0.02), source_time=PULSE, polarization="Ey"),
             td.PointDipole(center=(hx - 0.4 * DL, 0.1, -hz + 0.5 * DL), source_time=PULSE, polarization="Ez"),
             td.PointDipole(center=(hx - 0.6 * DL, hy - 0.3 * DL, 0.1), source_time=PULSE, polarization="Hz"),
             td.PointDipole(center=(0.2, -hy + 1.4 * DL, hz - 0.4 * DL), source_time=PULSE, polarization="Hx")]
    bar = [td.Structure(geometry=td.Box(center=(0, 0.1, 0), size=(td.inf, 0.3, td.inf)), medium=td.Medium(permittivity=2.5, conductivity=0.02)),
           td.Structure(geometry=td.Sphere(center=(-hx + 0.1, -hy + 0.15, 0.0), radius=0.2), medium=td.Medium(permittivity=3.0))]
    disc = discretize(_sim(N, bspec, structures=bar, extra=extra), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, _, p0, _, _ = _run(disc.spec, emu_lib, 0)
    got_f, _, p1, s1, why = _run(disc.spec, emu_lib, w + 64 * zc)
    assert p0 == 0 and p1 == 5 + 7 and s1 == p1, (p1, s1, why)
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), (c, float(np.abs(got_f[c] - ref_f[c]).max()))


@pytest.mark.parametrize("name,bspec,w,zc", [("one_tile", B_ALL, 5, 3), ("two_x_tiles", B_ODD, 8, 4), ("one_tile_wide", B_STABLE, 6, 4)])
def test_shell_pairs_with_materials(name, bspec, w, zc, emu_lib):
    """Non-dispersive media running through the layers (a lossy bar along x through both x slabs, a sphere, a PEC box): the bulk's
    materials instantiation, the strips' per-cell look-ups and the slabs' row-segment words."""
    N = SHAPES[name]
    disc = discretize(_sim(N, bspec, structures=MEDIA), n_steps=26)
    disc.spec.decay_every = 0
    assert len(disc.spec.media) > 2
    ref_f, _, _, _, _ = _run(disc.spec, emu_lib, 0)
    got_f, _, p1, s1, _ = _run(disc.spec, emu_lib, w + 64 * zc)
    assert p1 == 5 + 7 and s1 == p1
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), (c, float(np.abs(got_f[c] - ref_f[c]).max()))


def test_shell_pairs_against_the_one_launch_step_and_the_slab_kernels(emu_lib):
    """The same bits as the single launch of the all-axes instantiation and as the slab-form CPML behind the plain sweep."""
    disc = discretize(_sim(SHAPES["one_tile"], B_ALL, structures=MEDIA), n_steps=26)
    disc.spec.decay_every = 0
    got_f, _, p1, s1, _ = _run(disc.spec, emu_lib, 5 + 64 * 3)
    one_f, _, _, _, _ = _run(disc.spec, emu_lib, 0, split=0)
    assert s1 == 12
    for c in range(6):
        assert np.array_equal(got_f[c], one_f[c]), c
    with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
        e.set_option(L.OPT_ROWS, 3)
        e.set_option(L.OPT_PML_FUSED, 0)
        e.set_option(L.OPT_TWOSTEP, 5 + 64 * 3)
        st = e.run(11)
        assert int(st.shell_pairs) == 0 and int(st.fused2_off_reason) == 4      # slab-kernel CPML asked for: no shell pairs
        e.run(15)
        slab_f = [e.get_field(c) for c in range(6)]
    for c in range(6):
        assert np.array_equal(got_f[c], slab_f[c]), c


def test_shell_pairs_with_monitors(emu_lib):
    """A probe and a DFT plane inside the bulk are recorded from pairs (the sweep's copies of the middle step); a DFT plane that
    reaches into the shell makes the pairs it records in give way to single steps.  Same records, bit for bit."""
    N = SHAPES["one_tile"]
    inner = [td.FieldTimeMonitor(center=(0.1, 0.05, 0.0), size=(0, 0, 0), name="probe", interval=1, colocate=False),
             td.FieldMonitor(center=(0, 0, 0.1), size=(1.0, 0.4, 0), freqs=[3e14, 3.5e14], name="dft_in", colocate=False)]
    outer = [td.FieldMonitor(center=(0, 0, 0), size=(td.inf, td.inf, 0), freqs=[3e14], name="dft_out", interval_space=(1, 1, 1), colocate=False)]
    for mons, want_all in ((inner, True), (inner + outer, False)):
        disc = discretize(_sim(N, B_ALL, structures=MEDIA, monitors=mons), n_steps=40)
        disc.spec.decay_every = 0
        ref_f, ref_m, _, _, _ = _run(disc.spec, emu_lib, 0, runs=(17, 23))
        got_f, got_m, p1, s1, _ = _run(disc.spec, emu_lib, 5 + 64 * 4, runs=(17, 23))
        assert s1 == p1 and (p1 == 8 + 11 if want_all else 0 < p1 < 8 + 11), p1
        for c in range(6):
            assert np.array_equal(got_f[c], ref_f[c]), c
        for k in ref_m:
            assert np.abs(ref_m[k]).max() > 0 and np.array_equal(got_m[k], ref_m[k]), k


def test_pairs_resume_when_a_source_plane_is_spent(emu_lib):
    """A current sheet of several hundred nodes (more than the sweep's node table takes) keeps single steps only for the length of
    its pulse: the waveform ends where the reference says the source ends (SourceTime.end_time), and from there on the run goes
    out in pairs — CPML shell included — with the same bits as single steps.  (A TFSF box likewise: GPU suite.)"""
    N = (32, 16, 14)
    size = tuple((n - 1e-6) * DL for n in N)
    pulse = td.GaussianPulse(freq0=3e14, fwidth=2.4e14)
    srcs = [td.UniformCurrentSource(center=(0, 0, -0.1), size=(1.4, 0.7, 0), source_time=pulse, polarization="Ex")]
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1.3e-14, sources=srcs, structures=MEDIA,
                        monitors=[td.FieldTimeMonitor(center=(0.1, 0.05, 0.0), size=(0, 0, 0), name="probe", interval=1, colocate=False)],
                        boundary_spec=td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.pml(num_layers=2), z=td.Boundary.pml(num_layers=2)),
                        shutoff=0)
    disc = discretize(sim)
    disc.spec.decay_every = 0
    spec = disc.spec
    assert sum(len(s.comp) for s in spec.sources) > 256
    n_src = max(len(s.wave_e) for s in spec.sources)
    assert 50 < n_src < spec.n_steps - 30, (n_src, spec.n_steps)              # the list ends before the run does
    ref_f, ref_m, p0, _, _ = _run(spec, emu_lib, 0, runs=(spec.n_steps,))
    got_f, got_m, p1, s1, why = _run(spec, emu_lib, 5 + 64 * 4, runs=(spec.n_steps,))
    assert p0 == 0 and s1 == p1 and p1 >= (spec.n_steps - n_src) // 2 - 1, (p1, spec.n_steps, n_src)
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    assert np.array_equal(got_m["probe"], ref_m["probe"])


LORENTZ = td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])


@pytest.mark.parametrize("kind", ["ade_layer_cpml", "ade_sphere_pec_box", "sheet_and_ade_periodic", "plane_wave_unit_cell"])
def test_shell_pairs_with_z_holes(kind, emu_lib):
    """z holes: plane ranges inside the bulk's box that take single steps with the shell — the planes that hold dispersive cells
    (their ADE state advances every step) and, while the lists inject, the planes of sources the sweep cannot apply itself (a
    current sheet of hundreds of nodes, the injection plane of a plane wave: TFSF corrections + its 1-D incident grid).  The
    two-step sweep runs once per interval of the remaining planes.  Same bits as single steps, through the pulse and after it."""
    N = (28, 18, 44)
    size = tuple((n - 1e-6) * DL for n in N)
    pulse = td.GaussianPulse(freq0=3e14, fwidth=2.4e14)
    dip = [td.PointDipole(center=(0.1, 0.05, 0.7), source_time=PULSE, polarization="Ex"),
           td.PointDipole(center=(-0.2, 0.1, -0.6), source_time=PULSE, polarization="Hy")]
    film = td.Structure(geometry=td.Box(center=(0, 0, 0.1), size=(td.inf, td.inf, 0.2)), medium=LORENTZ)
    ball = td.Structure(geometry=td.Sphere(center=(0.1, 0.0, -0.1), radius=0.22), medium=td.Drude(eps_inf=1.5, coeffs=[(6e14, 5e13)]))
    glass = td.Structure(geometry=td.Box(center=(0, 0, -0.7), size=(td.inf, td.inf, 0.5)), medium=td.Medium(permittivity=2.1))
    per = td.Boundary.periodic()
    if kind == "ade_layer_cpml":
        sim = dict(structures=[glass, film], sources=dip, boundary_spec=B_ALL)
    elif kind == "ade_sphere_pec_box":
        sim = dict(structures=[glass, ball], sources=dip, boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    elif kind == "sheet_and_ade_periodic":
        sheet = td.UniformCurrentSource(center=(0, 0, 0.75), size=(1.9, 0.8, 0), source_time=pulse, polarization="Ey")
        sim = dict(structures=[glass, film], sources=[sheet], boundary_spec=td.BoundarySpec(x=per, y=per, z=td.Boundary.pml(num_layers=3)))
    else:
        pw = td.PlaneWave(center=(0, 0, 0.8), size=(td.inf, td.inf, 0), source_time=pulse, direction="-")
        sim = dict(structures=[glass, ball], sources=[pw], boundary_spec=td.BoundarySpec(x=per, y=per, z=td.Boundary.pml(num_layers=3)))
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1.1e-14,
                        monitors=[td.FieldTimeMonitor(center=(0.1, 0.05, -0.45), size=(0, 0, 0), name="probe", interval=1, colocate=False)],
                        shutoff=0, **sim)
    disc = discretize(sim)
    disc.spec.decay_every = 0
    spec = disc.spec
    n1 = 60
    assert spec.n_steps > n1 + 40
    # (FDTD_OPT_DISP = 0: the planes of dispersive cells as z holes, the subject here; by default the pairs advance those cells
    #  themselves since round 6, tests/test_emu_disp.py — in a PEC box without any shell at all)
    ref_f, ref_m, p0, _, _ = _run(spec, emu_lib, 0, runs=(n1 - 1, spec.n_steps - n1 + 1), disp=0)
    got_f, got_m, p1, s1, why = _run(spec, emu_lib, 8 + 64 * 6, runs=(n1 - 1, spec.n_steps - n1 + 1), disp=0)
    assert p0 == 0 and s1 == p1 and p1 >= spec.n_steps // 2 - 3, (p1, spec.n_steps, why)
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), (c, float(np.abs(got_f[c] - ref_f[c]).max()))
    assert np.abs(ref_m["probe"]).max() > 0 and np.array_equal(got_m["probe"], ref_m["probe"])
