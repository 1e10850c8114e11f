"""Step pairs WHILE big source lists inject (round 6; FDTD_OPT_SRC_PAGED, fdtd_fused2.hpp SrcP): a TFSF box, a current sheet of
hundreds of nodes, a crowd of dipoles — lists the two-step sweep's node table cannot hold.  In front of each pair list kernels leave
what the lists add at steps n (E side), n + 1 (H side) and n + 1 (E side) in paged storage; the sweep, its seam kernel and the shell's
boxes add them.  Held to single steps of the same library on the CPU emulator, bit for bit, through the pulse and after it; with
FDTD_OPT_SRC_PAGED = 0 the run behaves as in round 5 (single steps or z holes while the lists inject)."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)
SHORT = td.GaussianPulse(freq0=3e14, fwidth=3e14)
PEC = td.BoundarySpec.all_sides(td.PECBoundary())
ABS = td.BoundarySpec.all_sides(td.Absorber(num_layers=4))
PML = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.pml(num_layers=3), z=td.Boundary.pml(num_layers=3))
LOR = td.Lorentz(eps_inf=2.0, coeffs=[(2.0, 4e14, 2e13)])


def run(spec, lib, twostep, runs, paged=-1, shell2=None, seed=None):
    with HipEngine(spec, lib=lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
        e.set_option(L.OPT_ROWS, 3)
        e.set_option(L.OPT_TWOSTEP, twostep)
        if paged >= 0:
            e.set_option(L.OPT_SRC_PAGED, paged)
        if shell2 is not None:
            e.set_option(L.OPT_PML_SPLIT, 1)
            e.set_option(L.OPT_SHELL_PAIRS, 1)
            e.set_option(L.OPT_SHELL2, shell2)
        if seed is not None:
            rng = np.random.default_rng(seed)
            for c in range(6):
                f = e.get_field(c)
                e.set_field(c, ((1e-3 if c < 3 else 1e-3 / 376.73) * rng.uniform(-1, 1, size=f.shape)).astype(np.float32))
        pairs = sp = dp = 0
        for r in runs:
            st = e.run(r)
            pairs += int(st.fused2_pairs)
            sp += int(st.src_paged_pairs)
            dp += int(st.disp_pairs)
        return [e.get_field(c) for c in range(6)], e.results(), pairs, sp, dp


def same(ref, got):
    for c in range(6):
        assert np.array_equal(got[0][c], ref[0][c]), (c, float(np.abs(got[0][c] - ref[0][c]).max()))
    for k in ref[1]:
        assert np.array_equal(np.asarray(got[1][k]), np.asarray(ref[1][k])), k


def sim(N, bspec, sources, structures=(), monitors=()):
    size = tuple((n - 1e-6) * DL for n in N)
    return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, sources=list(sources), structures=list(structures),
                         monitors=list(monitors), boundary_spec=bspec, shutoff=0)


BALL = td.Structure(geometry=td.Sphere(center=(0.05, 0, 0), radius=0.2), medium=td.Medium(permittivity=2.5))
MON = [td.FieldMonitor(center=(0, 0, 0), size=(td.inf, td.inf, 0), freqs=[3e14], name="f", colocate=False),
       td.FieldTimeMonitor(center=(0.1, 0.05, -0.05), size=(0, 0, 0), name="p", interval=1, colocate=False)]


def check(spec, lib, twostep, runs, shell2=None, seed=None, want_disp=False):
    ref = run(spec, lib, 0, runs, shell2=shell2, seed=seed)
    got = run(spec, lib, twostep, runs, shell2=shell2, seed=seed)
    off = run(spec, lib, twostep, runs, paged=0, shell2=shell2, seed=seed)
    assert ref[2] == 0 and got[3] > 0 and got[2] >= sum(r // 2 for r in runs), got[2:]
    assert off[3] == 0 and off[2] < got[2], (off[2:], got[2:])
    if want_disp:
        assert got[4] == got[2], got[2:]
    assert max(float(np.abs(f).max()) for f in ref[0]) > 0
    same(ref, got)
    same(ref, off)
    return got


@pytest.mark.parametrize("N,w,zc", [((40, 30, 28), 5, 3), ((300, 26, 24), 6, 4), ((300, 26, 24), 16, 32)])
def test_tfsf_box_in_pec_walls(N, w, zc, emu_lib):
    """a TFSF box (six faces of E- and H-side corrections read from the 1-D incident grid, which the pair advances through both steps
    up front) around a dielectric sphere, PEC walls: plain pairs; the wide grid puts box faces on both sides of the seam at column 256"""
    sx = (N[0] - 1e-6) * DL
    src = td.TFSF(center=(-0.5 * sx + 256 * DL if N[0] > 256 else 0, 0, 0), size=(1.0, 0.7, 0.6), source_time=PULSE, injection_axis=2, direction="+")
    ball = td.Structure(geometry=td.Sphere(center=(src.center[0] + 0.05, 0, 0), radius=0.2), medium=td.Medium(permittivity=2.5))
    disc = discretize(sim(N, PEC, [src], [ball], MON), n_steps=40)
    disc.spec.decay_every = 0
    check(disc.spec, emu_lib, w + 64 * zc, (11, 16, 13))


@pytest.mark.parametrize("N,w,zc,axis", [((44, 30, 28), 5, 3, 1), ((44, 30, 28), 8, 4, 0), ((300, 26, 24), 6, 4, 2)])
def test_current_sheets_through_the_cpml_layers(N, w, zc, axis, emu_lib):
    """a current sheet normal to y / x / z that spans the whole cross-section, layers included (what a ModeSource plane is to the
    engine: hundreds of nodes): the bulk sweep, the seam kernel and the shell's boxes add its paged terms; a second, shorter pulse
    on a few dipoles: lists of different lengths (alive and spent lists in one pair)"""
    size = [td.inf, td.inf, td.inf]
    size[axis] = 0
    centre = [0.0, 0.0, 0.0]
    centre[axis] = 0.23
    pol = "Ez" if axis != 2 else "Ex"
    srcs = [td.UniformCurrentSource(center=tuple(centre), size=tuple(size), source_time=PULSE, polarization=pol),
            td.PointDipole(center=(0.02, 0.01, 0.03), source_time=SHORT, polarization="Ey"),
            td.PointDipole(center=(-0.1, 0.05, -0.04), source_time=SHORT, polarization="Hx")]
    disc = discretize(sim(N, PML, srcs, [BALL], MON), n_steps=40)
    disc.spec.decay_every = 0
    check(disc.spec, emu_lib, w + 64 * zc, (11, 16, 13), shell2=1, seed=4)


def test_tfsf_box_and_dispersive_sphere_in_cpml(emu_lib):
    """BASELINE config 4 in miniature with a Lorentz sphere: TFSF box + CPML + dispersive cells — the clipped sweep carries the paged
    source terms AND the memory terms (OPT bits 6 and 5), the flux box records its DFT inside the pairs"""
    N = (48, 40, 36)
    src = td.TFSF(center=(0, 0, 0), size=(1.2, 1.0, 0.9), source_time=PULSE, injection_axis=2, direction="+")
    ball = td.Structure(geometry=td.Sphere(center=(0.05, 0, 0), radius=0.25), medium=LOR)
    mons = [td.FluxMonitor(center=(0, 0, 0), size=(1.5, 1.3, 1.2), freqs=[2.5e14, 3e14], name="sca")]
    disc = discretize(sim(N, PML, [src], [ball], mons), n_steps=40)
    disc.spec.decay_every = 0
    check(disc.spec, emu_lib, 6 + 64 * 4, (11, 16, 13), shell2=1, seed=6, want_disp=True)


def test_electric_sheet_inside_absorber_layers(emu_lib):
    """absorber layers (damped inside the sweep) and an ELECTRIC current sheet through them: its terms are added in front of the
    damping, as launch_sources precedes launch_damp.  (Magnetic nodes together with absorber layers keep single steps.)"""
    N = (36, 30, 28)
    srcs = [td.UniformCurrentSource(center=(0, 0.2, 0), size=(td.inf, 0, td.inf), source_time=PULSE, polarization="Ex")]
    disc = discretize(sim(N, ABS, srcs, [BALL], MON[:1]), n_steps=40)
    disc.spec.decay_every = 0
    check(disc.spec, emu_lib, 5 + 64 * 3, (11, 16, 13))


def test_lists_that_meet_on_nodes(emu_lib):
    """two sheets that cross, and a TFSF box with a polarisation angle (two incident grids correcting the same nodes): the second list
    goes to layer 1, added behind layer 0 as the list kernels add one after the other; a third sheet through the crossing would need
    a third layer — such a problem keeps single steps while its lists inject (and the run is what it was)"""
    N = (40, 30, 28)
    two = [td.UniformCurrentSource(center=(0, 0.2, 0), size=(td.inf, 0, td.inf), source_time=PULSE, polarization="Ez"),
           td.UniformCurrentSource(center=(0.1, 0, 0), size=(0, td.inf, td.inf), source_time=PULSE, polarization="Ez")]
    disc = discretize(sim(N, PEC, two, [BALL], MON), n_steps=40)
    disc.spec.decay_every = 0
    check(disc.spec, emu_lib, 5 + 64 * 3, (11, 16, 13))
    box = [td.TFSF(center=(0, 0, 0), size=(1.0, 0.7, 0.6), source_time=PULSE, injection_axis=2, direction="+", pol_angle=0.4)]
    disc = discretize(sim((44, 36, 32), PML, box, [BALL], MON), n_steps=40)
    disc.spec.decay_every = 0
    assert len(disc.spec.tfsf) == 2
    check(disc.spec, emu_lib, 6 + 64 * 4, (11, 16, 13), shell2=1, seed=2)
    three = two + [td.UniformCurrentSource(center=(0, 0, 0.1), size=(td.inf, td.inf, 0), source_time=PULSE, polarization="Ez")]
    disc = discretize(sim(N, PEC, three, [BALL]), n_steps=30)
    disc.spec.decay_every = 0
    ref = run(disc.spec, emu_lib, 0, (11, 16))
    got = run(disc.spec, emu_lib, 5 + 64 * 3, (11, 16))
    assert got[3] == 0, got[2:]
    same(ref, got)


@pytest.mark.parametrize("N,w,zc", [((40, 30, 28), 5, 3), ((300, 26, 24), 16, 32)])
def test_time_monitor_on_the_faces_of_a_tfsf_box(N, w, zc, emu_lib):
    """point probes (stored as the 4 x 4 x 4 cells around them) on a corner and on a face of the TFSF box — they hold H nodes of the box's H-side corrections — recording
    every step while the box injects inside pairs: the H-side terms of step n change H^{n-1/2} in front of the sweep, so the record of
    step n takes its first H half-sample in front of THEM (as for magnetic dipoles).  Found by scripts/fuzz_round6.py on the device
    (seed 11, case 47): the record came back half a term too large, fields identical."""
    sx = (N[0] - 1e-6) * DL
    cx = -0.5 * sx + 256 * DL if N[0] > 256 else 0.0
    src = td.TFSF(center=(cx, 0, 0), size=(1.0, 0.7, 0.6), source_time=SHORT, injection_axis=2, direction="+")
    mons = [td.FieldTimeMonitor(center=(cx + 0.5, 0.35, 0.3), size=(0, 0, 0), name="corner", interval=1, colocate=False),
            td.FieldTimeMonitor(center=(cx - 0.5, 0.0, -0.3), size=(0, 0, 0), name="edge", interval=2, colocate=False,
                                fields=["Hx", "Hy", "Ez"])]
    disc = discretize(sim(N, PEC, [src], [], mons), n_steps=40)
    disc.spec.decay_every = 0
    got = check(disc.spec, emu_lib, w + 64 * zc, (11, 16, 13))
    assert float(np.abs(np.asarray(got[1]["corner"])).max()) > 0
