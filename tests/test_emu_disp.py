"""Dispersive (pole-residue ADE) cells in step pairs (round 6; FDTD_OPT_DISP, fdtd_kernels2.hpp OPT bit 5 + ade2_kernel) against single
steps of the same library on the CPU emulator: the same formulas in the same order -> the same bits.  The two-step sweep subtracts
the memory term cc S(Q^n) — kept in paged storage by every ADE kernel — from E^{n+1} and leaves E^{n+1} for ade2_kernel, which
advances the pole states two steps behind the sweep.  Covered: Lorentz / Drude / multi-pole bodies through the seams between
256-cell x tiles, on tile / chunk edges and on walls; sources on dispersive cells and on seam columns; time monitors on dispersive
cells, DFT planes; absorber layers; PMC min faces; CPML (shell2 pairs, dispersive cells deep inside the bulk; reaching into the
layers: the round-5 z holes); runs cut in two with single steps in between (the paged terms follow single steps too);
FDTD_OPT_DISP = 0 restores the round-5 behaviour."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)
PEC = td.BoundarySpec.all_sides(td.PECBoundary())
ABS = td.BoundarySpec.all_sides(td.Absorber(num_layers=4))
PMC_MIN = td.BoundarySpec(x=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                          y=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                          z=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()))
PML = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.pml(num_layers=3), z=td.Boundary.pml(num_layers=3))

LOR = td.Lorentz(eps_inf=2.0, coeffs=[(2.0, 4e14, 2e13)])
DRU = td.Drude(eps_inf=1.5, coeffs=[(3e14, 1e13), (5e14, 3e13)])
LOR3 = td.Lorentz(eps_inf=1.8, coeffs=[(1.0, 3e14, 2e13), (0.7, 5e14, 4e13), (0.4, 7e14, 3e13)])

SHAPES = {
    "one_tile": (32, 14, 10),
    "ragged_rows": (36, 9, 7),
    "two_x_tiles": (260, 9, 8),
    "three_x_tiles_tall": (516, 6, 13),
}


def bodies(N):
    """a Lorentz bar through the seam at column 256 (wide grids) or left of centre, a two-pole Drude sphere that overlaps it, a
    three-pole block touching the y-min wall, a plain dielectric"""
    sx, sy = N[0] * DL, N[1] * DL
    wide = N[0] > 256
    return [td.Structure(geometry=td.Box(center=(-0.5 * sx + 256 * DL if wide else -0.3, 0, 0), size=(1.0 if wide else 0.3, 0.3, 0.2)), medium=LOR),
            td.Structure(geometry=td.Sphere(center=(0.05, 0, 0), radius=0.2), medium=DRU),
            td.Structure(geometry=td.Box(center=(0.45, -0.5 * sy + 0.05, 0.05), size=(0.2, 0.2, 0.15)), medium=LOR3),
            td.Structure(geometry=td.Box(center=(0.3, 0.1, 0), size=(0.1, 0.1, 0.1)), medium=td.Medium(permittivity=3.0))]


def sim_for(N, bspec=PEC, structures=None, monitors=(), inner=1.0, magnetic=True):
    size = tuple((n - 1e-6) * DL for n in N)
    srcs = [td.PointDipole(center=(0.02, 0.01, 0.03), source_time=PULSE, polarization="Ez"),          # on a Drude cell
            td.PointDipole(center=(-0.11 * inner, 0.06 * inner, -0.05 * inner), source_time=PULSE, polarization="Ex")]
    if magnetic:      # (magnetic nodes together with absorber layers keep single steps, FDTD_F2_OFF_H_SOURCE_ABSORBER)
        srcs.append(td.PointDipole(center=(0.07, -0.04, 0.02), source_time=PULSE, polarization="Hy"))
    if N[0] > 256:      # on both sides of the seam at column 256, inside the Lorentz bar
        srcs += [td.PointDipole(center=(-0.5 * size[0] + 255.0 * DL, -0.05, 0.04), source_time=PULSE, polarization="Ey"),
                 td.PointDipole(center=(-0.5 * size[0] + 256.5 * DL, 0.0, 0.0), source_time=PULSE, polarization="Ex"),
                 td.PointDipole(center=(-0.5 * size[0] + 256.0 * DL, 0.05, 0.0), source_time=PULSE, polarization="Ez")]
    return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, sources=srcs,
                         structures=bodies(N) if structures is None else list(structures), monitors=list(monitors),
                         boundary_spec=bspec, shutoff=0)


def run(spec, lib, twostep, runs=(11, 15), disp=-1, shell2=None, seed=None):
    with HipEngine(spec, lib=lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
        e.set_option(L.OPT_ROWS, 3)
        e.set_option(L.OPT_TWOSTEP, twostep)
        if disp >= 0:
            e.set_option(L.OPT_DISP, disp)
        if shell2 is not None:
            e.set_option(L.OPT_PML_SPLIT, 1)
            e.set_option(L.OPT_SHELL_PAIRS, 1)
            e.set_option(L.OPT_SHELL2, shell2)
        if seed is not None:
            rng = np.random.default_rng(seed)
            for c in range(6):
                f = e.get_field(c)
                e.set_field(c, ((1e-3 if c < 3 else 1e-3 / 376.73) * rng.uniform(-1, 1, size=f.shape)).astype(np.float32))
        pairs = dpairs = s2 = 0
        for r in runs:
            st = e.run(r)
            pairs += int(st.fused2_pairs)
            dpairs += int(st.disp_pairs)
            s2 += int(st.shell2_pairs)
        return [e.get_field(c) for c in range(6)], e.results(), pairs, dpairs, s2


def same(ref, got):
    for c in range(6):
        assert np.array_equal(got[0][c], ref[0][c]), c
    for k in ref[1]:
        assert np.array_equal(np.asarray(got[1][k]), np.asarray(ref[1][k])), k


@pytest.mark.parametrize("name,w,zc", [("one_tile", 16, 32), ("one_tile", 4, 2), ("ragged_rows", 5, 3), ("two_x_tiles", 6, 4),
                                       ("two_x_tiles", 16, 32), ("three_x_tiles_tall", 8, 5)])
def test_dispersive_cells_in_step_pairs(name, w, zc, emu_lib):
    N = SHAPES[name]
    disc = discretize(sim_for(N), n_steps=26)
    disc.spec.decay_every = 0
    from tidy3d_amd.coeffs import material_table
    mt = material_table(disc.spec.media, disc.spec.dt)
    assert sum(mt.is_dispersive(m) for m in range(mt.n_media)) >= 3
    ref = run(disc.spec, emu_lib, 0)
    got = run(disc.spec, emu_lib, w + 64 * zc)
    assert ref[2] == 0 and got[2] == 12 and got[3] == 12, got[2:]
    assert max(float(np.abs(f).max()) for f in ref[0]) > 0
    same(ref, got)


@pytest.mark.parametrize("name,w,zc,bspec", [("one_tile", 5, 3, ABS), ("two_x_tiles", 8, 4, ABS), ("ragged_rows", 5, 3, PMC_MIN),
                                             ("three_x_tiles_tall", 6, 4, PMC_MIN)])
def test_dispersive_pairs_with_absorber_layers_and_pmc_walls(name, w, zc, bspec, emu_lib):
    """absorber layers: E^{n+1} is damped before its memory term is subtracted (launch_damp precedes launch_ade); the Lorentz bar of the
    wide grids runs into the x layers.  PMC min faces: the three-pole block sits on the y-min wall."""
    N = SHAPES[name]
    disc = discretize(sim_for(N, bspec=bspec, magnetic=bspec is not ABS), n_steps=26)
    disc.spec.decay_every = 0
    ref = run(disc.spec, emu_lib, 0)
    got = run(disc.spec, emu_lib, w + 64 * zc)
    assert ref[2] == 0 and got[2] == 12 and got[3] == 12, got[2:]
    same(ref, got)


@pytest.mark.parametrize("name,w,zc", [("one_tile", 5, 3), ("two_x_tiles", 8, 4)])
def test_dispersive_pairs_with_monitors(name, w, zc, emu_lib):
    """a probe ON a dispersive cell recording every step (its E sample of the middle step is taken behind the ADE update), a
    volume time monitor every third step, a DFT plane through the bodies (records on first and middle steps)"""
    N = SHAPES[name]
    mons = [td.FieldTimeMonitor(center=(0.02, 0.01, 0.03), size=(0, 0, 0), name="p", interval=1, colocate=False),
            td.FieldTimeMonitor(center=(0, 0, 0), size=(0.2, 0.2, 0.1), name="v", interval=3, colocate=False, fields=["Ex", "Ez", "Hy"]),
            td.FieldMonitor(center=(0, 0, 0), size=(td.inf, td.inf, 0), freqs=[3e14, 4e14], name="f", colocate=False)]
    disc = discretize(sim_for(N, monitors=mons), n_steps=26)
    disc.spec.decay_every = 0
    ref = run(disc.spec, emu_lib, 0)
    got = run(disc.spec, emu_lib, w + 64 * zc)
    assert ref[2] == 0 and got[2] > 0 and got[3] == got[2], got[2:]
    for k in ("p", "v", "f"):
        assert np.abs(np.asarray(ref[1][k])).max() > 0, k
    same(ref, got)


def test_paged_terms_follow_single_steps_and_the_option(emu_lib):
    """runs of odd lengths and of one step: single steps (ade_kernel) keep the paged memory terms that the next pair subtracts;
    FDTD_OPT_DISP = 0: no pairs without CPML (round 5), the same fields"""
    N = SHAPES["two_x_tiles"]
    disc = discretize(sim_for(N), n_steps=30)
    disc.spec.decay_every = 0
    ref = run(disc.spec, emu_lib, 0, runs=(1, 3, 1, 6, 5, 9))
    got = run(disc.spec, emu_lib, 6 + 64 * 4, runs=(1, 3, 1, 6, 5, 9))
    off = run(disc.spec, emu_lib, 6 + 64 * 4, runs=(1, 3, 1, 6, 5, 9), disp=0)
    assert got[2] == 0 + 1 + 0 + 3 + 2 + 4 and got[3] == got[2], got[2:]
    assert off[2] == 0 and off[3] == 0, off[2:]
    same(ref, got)
    same(ref, off)


@pytest.mark.parametrize("N,w,zc,shell2", [((48, 26, 24), 5, 3, 1), ((300, 24, 22), 6, 4, 1), ((536, 20, 19), 8, 5, 1)])
def test_dispersive_cells_deep_inside_the_bulk_of_shell2_pairs(N, w, zc, shell2, emu_lib):
    """CPML walls: the dispersive bodies lie three or more cells inside the bulk of the shell2 pair — its clipped sweep subtracts
    their memory terms, ade2_kernel follows it beside the shell's boxes; random initial fields fill the layers"""
    sx = (N[0] - 1e-6) * DL
    structures = [td.Structure(geometry=td.Box(center=(-0.5 * sx + 256 * DL if N[0] > 256 else -0.2, 0, 0), size=(0.6 if N[0] > 256 else 0.3, 0.2, 0.2)), medium=LOR),
                  td.Structure(geometry=td.Sphere(center=(0.1, 0.05, 0), radius=0.15), medium=DRU),
                  td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.1, 0.1)), medium=td.Medium(permittivity=2.5))]
    mons = [td.FieldMonitor(center=(0, 0, 0), size=(td.inf, td.inf, 0), freqs=[3e14], name="f", colocate=False)]
    disc = discretize(sim_for(N, bspec=PML, structures=structures, monitors=mons, inner=0.5), n_steps=26)
    disc.spec.decay_every = 0
    ref = run(disc.spec, emu_lib, 0, shell2=shell2, seed=3)
    got = run(disc.spec, emu_lib, w + 64 * zc, shell2=shell2, seed=3)
    assert ref[2] == 0 and got[2] == 12 and got[3] == 12 and got[4] == 12, got[2:]
    same(ref, got)


@pytest.mark.parametrize("N,w,zc", [((48, 26, 24), 5, 3), ((300, 24, 22), 6, 4)])
def test_dispersive_cells_on_the_first_planes_and_rows_of_the_bulk(N, w, zc, emu_lib):
    """A Lorentz block whose lowest plane is the bulk's FIRST plane (and whose first row is the bulk's first row): the shell's z-min / y-min
    boxes read the memory terms of that plane / row for their halo, so ade2_kernel — which rewrites the paged terms — must wait for the
    boxes instead of running beside them.  Found on the device by scripts/fuzz_round6.py (seed 31, case 124; one run in a few there — the
    emulator runs the streams in issue order and showed it every time: 14,000 cells off by 5e-6 after 26 steps)."""
    sy, sz = (N[1] - 1e-6) * DL, (N[2] - 1e-6) * DL
    z0, y0 = -0.5 * sz + 1.75 * DL, -0.5 * sy + 1.75 * DL             # (the collar is two cells: nodes of plane / row 2 of the domain inside, 1 outside)
    structures = [td.Structure(geometry=td.Box(center=(-0.2, 0.1, 0.5 * (z0 + 0.1)), size=(0.5, 0.3, 0.1 - z0)), medium=LOR),
                  td.Structure(geometry=td.Box(center=(0.3, 0.5 * (y0 + 0.05), 0.1), size=(0.3, 0.05 - y0, 0.2)), medium=DRU)]
    disc = discretize(sim_for(N, bspec=PML, structures=structures, inner=0.5), n_steps=26)
    disc.spec.decay_every = 0
    pol = np.array([len(m.poles) for m in disc.spec.media])
    where = np.argwhere((pol[np.asarray(disc.spec.mat_idx)] > 0).any(axis=0))
    assert where[:, 0].min() == 3 + 2 and where[:, 1].min() == 3 + 2, (where[:, 0].min(), where[:, 1].min())      # (3 layers + the collar)
    ref = run(disc.spec, emu_lib, 0, shell2=1, seed=5)
    got = run(disc.spec, emu_lib, w + 64 * zc, shell2=1, seed=5)
    assert ref[2] == 0 and got[2] == 12 and got[3] == 12 and got[4] == 12, got[2:]
    same(ref, got)


@pytest.mark.parametrize("N,w,zc,bspec", [((48, 26, 40), 5, 3, PML), ((300, 24, 22), 6, 4, PML), ((100, 23, 19), 8, 5, "periodic_x"),
                                          ((48, 26, 24), 16, 32, "odd")])
def test_dispersive_cells_inside_the_layers(N, w, zc, bspec, emu_lib):
    """a Lorentz bar running through the x layers, a Drude slab through the y layers, a three-pole block in a corner of the shell: the
    shell's boxes (shell2_step_kernel) subtract the paged memory terms as the bulk sweep does, ade2_kernel follows both; a periodic x
    (the boxes' halo lanes hold the wrapped columns); odd layer counts, a PEC z-min wall"""
    if bspec == "periodic_x":
        bspec = td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.pml(num_layers=3), z=td.Boundary.pml(num_layers=4))
    elif bspec == "odd":
        bspec = td.BoundarySpec(x=td.Boundary(minus=td.PML(num_layers=5), plus=td.PML(num_layers=3)), y=td.Boundary(minus=td.PML(num_layers=2), plus=td.PML(num_layers=4)),
                                z=td.Boundary(minus=td.PECBoundary(), plus=td.PML(num_layers=3)))
    sx, sy, sz = ((n - 1e-6) * DL for n in N)
    structures = [td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.2, 0.2)), medium=LOR),
                  td.Structure(geometry=td.Box(center=(0.1, 0, 0.3), size=(0.3, td.inf, 0.15)), medium=DRU),
                  td.Structure(geometry=td.Box(center=(0.5 * sx, 0.5 * sy, 0.5 * sz), size=(0.5, 0.5, 0.5)), medium=LOR3),
                  td.Structure(geometry=td.Sphere(center=(-0.2, 0.1, -0.2), radius=0.15), medium=td.Medium(permittivity=2.5))]
    mons = [td.FieldMonitor(center=(0, 0, 0), size=(td.inf, td.inf, 0), freqs=[3e14], name="f", colocate=False)]
    disc = discretize(sim_for(N, bspec=bspec, structures=structures, monitors=mons, inner=0.5), n_steps=26)
    disc.spec.decay_every = 0
    ref = run(disc.spec, emu_lib, 0, shell2=1, seed=5)
    got = run(disc.spec, emu_lib, w + 64 * zc, shell2=1, seed=5)
    assert ref[2] == 0 and got[2] == 12 and got[3] == 12 and got[4] == 12, got[2:]
    same(ref, got)
    off = run(disc.spec, emu_lib, w + 64 * zc, shell2=1, seed=5, disp=0)        # the round-5 form: the bodies' planes as z holes — here: every plane
    assert off[3] == 0
    same(ref, off)
