"""Two time steps per sweep (fused2_step_kernel + the seam kernels, fdtd_kernels2.hpp) against single steps of the same
library on the CPU emulator: same formulas in the same order -> the same bits.  Tile edges in x (seams between 256-cell
tiles, ragged last tile), y (two halo rows below, one above, ragged last tile row) and z (chunk prologue, chunks of every
length, the walls), sources between the two steps, pairs giving way to single steps around monitor records."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)
PEC = td.BoundarySpec.all_sides(td.PECBoundary())


MEDIA = [td.Structure(geometry=td.Box(center=(-0.3, 0, 0), size=(0.25, 0.3, 0.2)), medium=td.Medium(permittivity=3.0, conductivity=0.02)),
         td.Structure(geometry=td.Sphere(center=(0.05, 0, 0), radius=0.2), medium=td.Medium(permittivity=2.0)),
         td.Structure(geometry=td.Box(center=(0.3, 0.1, 0), size=(0.1, 0.1, 0.1)), medium=td.PEC)]
# a lossy bar through the seam at column 256, a sphere right of it, a PEC box (wide grids)
MEDIA_WIDE = [td.Structure(geometry=td.Box(center=(-0.5, 0, 0), size=(3.0, 0.3, 0.2)), medium=td.Medium(permittivity=3.0, conductivity=0.02)),
              td.Structure(geometry=td.Sphere(center=(6.3, 0, 0), radius=0.25), medium=td.Medium(permittivity=2.5)),
              td.Structure(geometry=td.Box(center=(0.1, 0.1, 0), size=(0.1, 0.1, 0.1)), medium=td.PEC)]


PMC_MIN = td.BoundarySpec(x=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                          y=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                          z=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()))
PMC_MIX = td.BoundarySpec(x=td.Boundary(minus=td.PECBoundary(), plus=td.PECBoundary()),
                          y=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                          z=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()))


def _sim(N, monitors=True, extra=(), structures=(), bspec=PEC):
    size = tuple(n * DL for n in N)
    srcs = [td.PointDipole(center=(0.3 * size[0] - 0.5 * size[0] + 0.02, 0.01, 0.03), source_time=PULSE, polarization="Ez"),
            td.PointDipole(center=(0.01, 0.02, -0.1), source_time=PULSE, polarization="Ex")]
    if N[0] >= 128:       # next to both x faces and on both sides of the seam at column 256
        srcs += [td.PointDipole(center=(-0.5 * size[0] + 2.3 * DL, 0.03, 0.02), source_time=PULSE, polarization="Ez"),
                 td.PointDipole(center=(0.5 * size[0] - 3.2 * DL, 0.1, -0.07), source_time=PULSE, polarization="Ey"),
                 td.PointDipole(center=(-0.5 * size[0] + 255.0 * DL, -0.05, 0.04), source_time=PULSE, polarization="Ey"),
                 td.PointDipole(center=(-0.5 * size[0] + 256.5 * DL, 0.0, 0.0), source_time=PULSE, polarization="Ex")]
    mons = []
    if monitors:
        mons = [td.FieldMonitor(center=(0, 0, 0), size=(td.inf, td.inf, 0), freqs=[3e14], name="f", interval_space=(1, 1, 1)),
                td.FieldTimeMonitor(center=(0, 0, 0), size=(0.2, 0.2, 0.2), name="t", interval=5, colocate=False)]
    return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, sources=srcs + list(extra),
                         structures=list(structures), monitors=mons, boundary_spec=bspec, shutoff=0)


def _run(spec, lib, twostep, runs=(11, 15), disp=-1):
    with HipEngine(spec, lib=lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
        e.set_option(L.OPT_ROWS, 3)
        e.set_option(L.OPT_TWOSTEP, twostep)
        if disp >= 0:
            e.set_option(L.OPT_DISP, disp)
        pairs = 0
        for r in runs:
            st = e.run(r)
            pairs += int(st.fused2_pairs)
        return [e.get_field(c) for c in range(6)], e.results(), pairs


SHAPES = {
    "one_tile": (32, 14, 10),
    "ragged_rows": (36, 9, 7),
    "two_x_tiles": (260, 9, 8),
    "three_x_tiles_tall": (516, 6, 13),
}


CASES = [("one_tile", 16, 32), ("one_tile", 4, 2),
         ("ragged_rows", 16, 32), ("ragged_rows", 5, 3), ("ragged_rows", 8, 5),
         ("two_x_tiles", 16, 32), ("two_x_tiles", 4, 2), ("two_x_tiles", 8, 3),
         ("three_x_tiles_tall", 5, 3), ("three_x_tiles_tall", 6, 32)]


@pytest.mark.parametrize("name,w,zc", CASES)
def test_two_steps_per_sweep_equal_single_steps(name, w, zc, emu_lib):
    N = SHAPES[name]
    disc = discretize(_sim(N, monitors=False), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, _, p0 = _run(disc.spec, emu_lib, 0)
    got_f, _, p1 = _run(disc.spec, emu_lib, w + 64 * zc)
    assert p0 == 0 and p1 == 5 + 7, p1          # runs of 11 and 15 steps: 5 + 7 pairs and a single step each
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c


@pytest.mark.parametrize("name,w,zc", [("one_tile", 16, 32), ("ragged_rows", 8, 5), ("two_x_tiles", 6, 4), ("three_x_tiles_tall", 8, 5)])
def test_two_steps_per_sweep_with_materials(name, w, zc, emu_lib):
    """Non-dispersive media (a lossy dielectric bar through the seam of the wide grids, a sphere with sub-pixel-averaged
    surface cells, a PEC box): uniform row segments take their coefficients as scalars, mixed ones per cell, the seam kernel
    looks them up per cell — the same bits as single sweeps."""
    N = SHAPES[name]
    disc = discretize(_sim(N, monitors=False, structures=MEDIA_WIDE if N[0] >= 128 else MEDIA), n_steps=26)
    disc.spec.decay_every = 0
    assert len(disc.spec.media) > 2
    ref_f, _, p0 = _run(disc.spec, emu_lib, 0)
    got_f, _, p1 = _run(disc.spec, emu_lib, w + 64 * zc)
    assert p0 == 0 and p1 == 5 + 7, p1
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c


@pytest.mark.parametrize("name,w,zc,bspec", [("ragged_rows", 5, 3, PMC_MIN), ("two_x_tiles", 16, 32, PMC_MIX), ("three_x_tiles_tall", 6, 4, PMC_MIN)])
def test_two_steps_per_sweep_with_pmc_min_faces(name, w, zc, bspec, emu_lib):
    """PMC walls on the min faces (the symmetry planes of a half / quarter / eighth domain): H mirrored with the opposite sign
    behind them in both steps, in the sweep and in the seam kernel; with materials and a probe next to the walls."""
    N = SHAPES[name]
    size = tuple(n * DL for n in N)
    sim = _sim(N, monitors=False, structures=MEDIA_WIDE if N[0] >= 128 else MEDIA, bspec=bspec)
    srcs = list(sim.sources) + [td.PointDipole(center=(-0.5 * size[0] + 0.6 * DL, -0.5 * size[1] + 0.4 * DL, -0.5 * size[2] + 1.2 * DL),
                                               source_time=PULSE, polarization="Ey")]
    mons = [td.FieldTimeMonitor(center=(-0.5 * size[0] + 1.1 * DL, -0.5 * size[1] + 1.2 * DL, -0.5 * size[2] + 0.9 * DL), size=(0, 0, 0),
                                name="corner", interval=1, colocate=False)]
    disc = discretize(sim.updated_copy(sources=srcs, monitors=mons), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, ref_m, p0 = _run(disc.spec, emu_lib, 0)
    got_f, got_m, p1 = _run(disc.spec, emu_lib, w + 64 * zc)
    assert p0 == 0 and p1 == 5 + 7, p1
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    assert np.abs(ref_m["corner"]).max() > 0 and np.array_equal(got_m["corner"], ref_m["corner"])


@pytest.mark.parametrize("name,w,zc", [("one_tile", 16, 32), ("two_x_tiles", 8, 5), ("three_x_tiles_tall", 6, 4)])
def test_two_steps_per_sweep_with_magnetic_dipoles(name, w, zc, emu_lib):
    """H-side point sources: those of step n act on H^{n-1/2} in front of the sweep, those of step n+1 on H^{n+1/2} inside it —
    behind E^{n+1}, which is formed from the value without them; a probe on a source node records H^{n+1/2} without the term
    of step n+1, as between two single steps.  (An H_x node may sit in the column left of a seam; H_y / H_z there: next test.)"""
    N = SHAPES[name]
    size = tuple(n * DL for n in N)
    hs = [td.PointDipole(center=(0.12, 0.03, -0.02), source_time=PULSE, polarization="Hy"),
          td.PointDipole(center=(-0.2 * size[0], -0.1, 0.0), source_time=PULSE, polarization="Hz"),
          td.PointDipole(center=(-0.5 * size[0] + 0.7 * DL, 0.1, 0.11), source_time=PULSE, polarization="Hx")]
    if N[0] > 256:
        hs.append(td.PointDipole(center=(-0.5 * size[0] + 257.5 * DL, 0.02, 0.0), source_time=PULSE, polarization="Hy"))
    mons = [td.FieldTimeMonitor(center=(0.12, 0.03, -0.02), size=(0, 0, 0), name="on_h", interval=1, fields=["Hy", "Ex"], colocate=False)]
    sim = _sim(N, monitors=False, extra=hs).updated_copy(monitors=mons)
    disc = discretize(sim, n_steps=26)
    disc.spec.decay_every = 0
    ref_f, ref_m, p0 = _run(disc.spec, emu_lib, 0)
    got_f, got_m, p1 = _run(disc.spec, emu_lib, w + 64 * zc)
    assert p0 == 0 and p1 == 5 + 7, p1
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    assert np.abs(ref_m["on_h"]).max() > 0 and np.array_equal(got_m["on_h"], ref_m["on_h"])


def test_magnetic_dipole_left_of_a_seam(emu_lib):
    """an H_z node in the column left of a seam: the node table cannot carry it (the seam kernel rebuilds H^{n+3/2} there from
    H^{n+1/2} WITHOUT the term of step n+1) — single steps with FDTD_OPT_SRC_PAGED = 0 (round 5); by default the lists go out as
    paged source terms, which the seam kernel adds too (round 6): pairs, the same bits"""
    N = SHAPES["two_x_tiles"]
    size = tuple(n * DL for n in N)
    sim = _sim(N, monitors=False, extra=[td.PointDipole(center=(-0.5 * size[0] + 255.5 * DL, 0.0, 0.0), source_time=PULSE, polarization="Hz")])
    disc = discretize(sim, n_steps=12)
    disc.spec.decay_every = 0
    ref_f, _, p0 = _run(disc.spec, emu_lib, 0, runs=(12,))
    got_f, _, p1 = _run(disc.spec, emu_lib, 8 + 64 * 4, runs=(12,))
    with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
        e.set_option(L.OPT_ROWS, 3)
        e.set_option(L.OPT_TWOSTEP, 8 + 64 * 4)
        e.set_option(L.OPT_SRC_PAGED, 0)
        p2 = int(e.run(12).fused2_pairs)
        old_f = [e.get_field(c) for c in range(6)]
    assert p0 == 0 and p1 == 6 and p2 == 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
        assert np.array_equal(old_f[c], ref_f[c]), c


ABS = td.BoundarySpec(x=td.Boundary.absorber(num_layers=5, parameters=td.AbsorberParams(sigma_max=1.5)),
                      y=td.Boundary(minus=td.PECBoundary(), plus=td.Absorber(num_layers=3)),
                      z=td.Boundary.absorber(num_layers=3))
ABS_PMC = td.BoundarySpec(x=td.Boundary(minus=td.PMCBoundary(), plus=td.Absorber(num_layers=6)),
                          y=td.Boundary.absorber(num_layers=2), z=td.Boundary(minus=td.PMCBoundary(), plus=td.Absorber(num_layers=4)))


@pytest.mark.parametrize("name,w,zc,bspec", [("ragged_rows", 16, 32, ABS), ("two_x_tiles", 8, 5, ABS), ("two_x_tiles", 6, 4, ABS_PMC),
                                             ("three_x_tiles_tall", 7, 6, ABS)])
def test_two_steps_per_sweep_with_absorber_layers(name, w, zc, bspec, emu_lib):
    """Absorber boundaries (open problems): damp_kernel's factors applied in registers to H^{n-1/2}, E^{n+1} (behind its sources),
    H^{n+1/2} and E^{n+2} — the sweep, its x-halo column, its chunk prologue and the seam kernel — with media, probes inside the
    layers, and (wide grids) source nodes next to a seam, whose terms of step n+1 and the damping behind them are left to the
    caller's launches."""
    N = SHAPES[name]
    size = tuple(n * DL for n in N)
    sim = _sim(N, monitors=False, structures=MEDIA_WIDE if N[0] >= 128 else MEDIA, bspec=bspec)
    mons = [td.FieldTimeMonitor(center=(0.5 * size[0] - 2.2 * DL, 0.5 * size[1] - 1.6 * DL, 0.5 * size[2] - 1.4 * DL), size=(0, 0, 0),
                                name="in_layers", interval=1, colocate=False),
            td.FieldTimeMonitor(center=(0.0, 0.0, 0.0), size=(0, 0, 0), name="mid", interval=2, fields=["Ez", "Hx"], colocate=False)]
    disc = discretize(sim.updated_copy(monitors=mons), n_steps=26)
    disc.spec.decay_every = 0
    assert any(a is not None for a in disc.spec.absorber) if hasattr(disc.spec, "absorber") else True
    ref_f, ref_m, p0 = _run(disc.spec, emu_lib, 0)
    got_f, got_m, p1 = _run(disc.spec, emu_lib, w + 64 * zc)
    assert p0 == 0 and p1 == 5 + 7, p1
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.abs(ref_m[k]).max() > 0 and np.array_equal(got_m[k], ref_m[k]), k


def test_seam_sources_inside_and_outside_the_absorber_layers(emu_lib):
    """An E-side source node next to a seam has its term of step n+1 added behind the launch.  Outside the layers the damping factor
    is exactly 1, so the sweep damps E^{n+2} itself (round 5: the bench's dipole sits on column 256 — six damping launches over
    40 % of the grid went out behind every pair); a seam node INSIDE a layer leaves that damping to the caller's launches, behind
    the source.  Same bits as single steps in both cases."""
    N = SHAPES["two_x_tiles"]
    size = tuple(n * DL for n in N)
    for inside in (False, True):
        y = 0.5 * size[1] - 1.4 * DL if inside else 0.02
        extra = [td.PointDipole(center=(-0.5 * size[0] + 256.0 * DL, y, 0.03), source_time=PULSE, polarization="Ey"),
                 td.PointDipole(center=(-0.5 * size[0] + 255.5 * DL, y, -0.04), source_time=PULSE, polarization="Ex")]
        disc = discretize(_sim(N, monitors=False, structures=MEDIA_WIDE, bspec=ABS, extra=extra), n_steps=26)
        disc.spec.decay_every = 0
        ref_f, _, p0 = _run(disc.spec, emu_lib, 0)
        got_f, _, p1 = _run(disc.spec, emu_lib, 8 + 64 * 5)
        assert p0 == 0 and p1 == 5 + 7, (inside, p1)
        for c in range(6):
            assert np.array_equal(got_f[c], ref_f[c]), (inside, c)


def test_absorber_layers_with_a_magnetic_dipole_take_single_steps(emu_lib):
    N = SHAPES["one_tile"]
    sim = _sim(N, monitors=False, bspec=ABS, extra=[td.PointDipole(center=(0.1, 0, 0), source_time=PULSE, polarization="Hy")])
    disc = discretize(sim, n_steps=12)
    disc.spec.decay_every = 0
    ref_f, _, p0 = _run(disc.spec, emu_lib, 0, runs=(12,))
    got_f, _, p1 = _run(disc.spec, emu_lib, 8 + 64 * 4, runs=(12,))
    assert p0 == 0 and p1 == 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c


@pytest.mark.parametrize("name,w,zc,bspec", [("one_tile", 16, 32, PEC), ("two_x_tiles", 8, 5, PEC), ("two_x_tiles", 6, 4, ABS),
                                             ("three_x_tiles_tall", 5, 3, PMC_MIN)])
def test_dft_monitors_do_not_stop_pairs(name, w, zc, bspec, emu_lib):
    """DFT monitors (field planes normal to x, y and z — the x-normal one through a seam column on wide grids —, a volume, a flux
    plane; each at its own sampling interval): a record on the FIRST step of a pair takes E^n in front of the sweep and its H
    terms from the sweep's copy of H^{n+1/2} over the box; a record on the MIDDLE step takes its E terms from the copy of E^{n+1}
    and its H terms from the write set.  Same spectra, bit for bit; every step goes out in a pair."""
    N = SHAPES[name]
    size = tuple(n * DL for n in N)
    xs = -0.5 * size[0] + 255.4 * DL if N[0] > 256 else 0.1
    mons = [td.FieldMonitor(center=(0, 0, 0.02), size=(td.inf, td.inf, 0), freqs=[3e14, 3.3e14], name="fz", interval_space=(1, 1, 1)),
            td.FieldMonitor(center=(0, 0.03, 0), size=(td.inf, 0, td.inf), freqs=[3e14], name="fy", fields=["Ex", "Hy", "Hz"]),
            td.FieldMonitor(center=(xs, 0, 0), size=(0, td.inf, td.inf), freqs=[2.8e14], name="fx", fields=["Hx", "Hz", "Ey"]),
            td.FieldMonitor(center=(0.05, 0.02, 0.0), size=(0.3, 0.2, 0.15), freqs=[3e14], name="vol"),
            td.FluxMonitor(center=(0, 0, -0.1), size=(td.inf, td.inf, 0), freqs=[3e14, 3.1e14], name="flux"),
            td.FieldMonitor(center=(0, 0, 0), size=(0.2, 0.2, 0), freqs=[3e14], name="e_only", fields=["Ex", "Ey"])]
    sim = _sim(N, monitors=False, structures=MEDIA_WIDE if N[0] >= 128 else MEDIA, bspec=bspec).updated_copy(monitors=mons)
    disc = discretize(sim, n_steps=40)
    disc.spec.decay_every = 0
    ref_f, ref_m, p0 = _run(disc.spec, emu_lib, 0, runs=(17, 23))
    got_f, got_m, p1 = _run(disc.spec, emu_lib, w + 64 * zc, runs=(17, 23))
    assert p0 == 0 and p1 == 8 + 11, p1          # every step of both runs but their odd last one: records never stop a pair
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    assert len(ref_m) >= 6 and set(ref_m) == set(got_m), sorted(ref_m)
    for k in ref_m:
        assert np.abs(ref_m[k]).max() > 0, k
        assert np.array_equal(got_m[k], ref_m[k]), k


@pytest.mark.parametrize("name", ["ragged_rows", "two_x_tiles"])
def test_pairs_give_way_to_monitor_records_and_decay_checks(name, emu_lib):
    N = SHAPES[name]
    disc = discretize(_sim(N), n_steps=26)
    disc.spec.decay_every = 8
    ref_f, ref_m, p0 = _run(disc.spec, emu_lib, 0)
    got_f, got_m, p1 = _run(disc.spec, emu_lib, 6 + 64 * 4)
    assert p0 == 0                        # (the DFT monitor samples at its Nyquist interval: pairs fit in between)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.array_equal(got_m[k], ref_m[k]), k
    # without the DFT monitor: pairs between the time monitor's records (every 5th step) and the decay checks
    sim = _sim(N)
    sim = sim.updated_copy(monitors=[m for m in sim.monitors if m.name == "t"]) if hasattr(sim, "updated_copy") else sim
    disc = discretize(sim, n_steps=26)
    disc.spec.decay_every = 8
    ref_f, ref_m, p0 = _run(disc.spec, emu_lib, 0)
    got_f, got_m, p1 = _run(disc.spec, emu_lib, 6 + 64 * 4)
    assert p0 == 0 and p1 >= 4, p1
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.array_equal(got_m[k], ref_m[k]), k


@pytest.mark.parametrize("name,interval", [("one_tile", 1), ("one_tile", 3), ("two_x_tiles", 1), ("two_x_tiles", 2)])
def test_small_time_monitors_sample_the_middle_step(name, interval, emu_lib):
    """Point-like FieldTimeMonitors (E and H components; H is the mean of two half-steps) do not stop pairs: the sweep copies
    E^{n+1} (behind the sources of step n) and H^{n+1/2} of their cells out, and the records equal those of single steps."""
    N = SHAPES[name]
    size = tuple(n * DL for n in N)
    sim = _sim(N, monitors=False)
    mons = [td.FieldTimeMonitor(center=(0.11, 0.02, -0.03), size=(0, 0, 0), name="p", interval=interval, colocate=False),
            td.FieldTimeMonitor(center=(-0.5 * size[0] + 0.32, 0.1, 0.1), size=(0.1, 0, 0), name="q", interval=1,
                                fields=["Ez", "Hx"], colocate=False, start=1e-15),
            # on the source node of the first dipole: E^{n+1} must be taken behind the source term
            td.FieldTimeMonitor(center=(0.3 * size[0] - 0.5 * size[0] + 0.02, 0.01, 0.03), size=(0, 0, 0), name="s", interval=2,
                                fields=["Ez"], colocate=False)]
    if N[0] > 256:        # H_y / H_z samples in the column left of the seam: repaired only behind the sweep -> recorded around it
        mons.insert(0, td.FieldTimeMonitor(center=(-0.5 * size[0] + 255.4 * DL, 0.0, 0.05), size=(0, 0, 0), name="seam",
                                           interval=1, fields=["Ex", "Hy", "Hz"], colocate=False))
    sim = sim.updated_copy(monitors=mons)
    disc = discretize(sim, n_steps=26)
    disc.spec.decay_every = 0
    ref_f, ref_m, p0 = _run(disc.spec, emu_lib, 0)
    got_f, got_m, p1 = _run(disc.spec, emu_lib, 8 + 64 * 4)
    assert p0 == 0 and p1 == 5 + 7, p1
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    assert set(ref_m) >= {"p", "q", "s"}
    for k in ref_m:
        assert np.abs(ref_m[k]).max() > 0, k
        assert np.array_equal(got_m[k], ref_m[k]), k


def test_not_eligible_runs_take_single_steps(emu_lib):
    """a dispersive medium with FDTD_OPT_DISP = 0 (the round-5 behaviour; by default the pairs advance dispersive cells since round 6,
    tests/test_emu_disp.py): the option changes nothing, no pair is taken.  (A periodic x face was such a case until round 4: now the
    wrap is a seam of the clipped sweep — pairs, same bits; tests/test_emu_shell.py has the periodic cases proper.)"""
    N = (32, 10, 9)
    cases = [dict(bspec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary(minus=td.PECBoundary(), plus=td.PECBoundary()),
                                        z=td.Boundary(minus=td.PECBoundary(), plus=td.PECBoundary()))),
             dict(structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.3, 0.3, 0.2)),
                                           medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)]))])]
    for kw in cases:
        sim = _sim(N, monitors=False, extra=kw.get("extra", ()))
        if "bspec" in kw:
            sim = td.Simulation(size=sim.size, grid_spec=sim.grid_spec, run_time=sim.run_time, sources=list(sim.sources),
                                monitors=[], boundary_spec=kw["bspec"], shutoff=0)
        if "structures" in kw:
            sim = td.Simulation(size=sim.size, grid_spec=sim.grid_spec, run_time=sim.run_time, sources=list(sim.sources),
                                monitors=[], boundary_spec=PEC, structures=kw["structures"], shutoff=0)
        disc = discretize(sim, n_steps=12)
        disc.spec.decay_every = 0
        ref_f, _, p0 = _run(disc.spec, emu_lib, 0, runs=(12,), disp=0)
        got_f, _, p1 = _run(disc.spec, emu_lib, 8 + 64 * 4, runs=(12,), disp=0)
        assert p0 == 0 and p1 == (6 if "bspec" in kw else 0), p1
        for c in range(6):
            assert np.array_equal(got_f[c], ref_f[c]), c


@pytest.mark.parametrize("bspec_name,w,zc", [("pec", 5, 3), ("abs", 6, 4), ("pec", 4, 2)])
def test_background_only_tiles_on_the_plain_instantiation(bspec_name, w, zc, emu_lib):
    """Tile classes (round 5, FDTD_OPT_TILE_SPLIT): in the two-step sweep of a grid with bodies, the workgroup of a tile that — halo
    rows / planes included — holds only the background medium runs the plain sweep inside the materials launch (the uniform
    coefficients are the table's entry 1).  A tall three-tile grid with a small lossy block, a sphere right of the seams and a PEC
    box: most tiles are background-only.  Forced split == never split == single steps, bit for bit; with absorber layers too
    (the plain sweep keeps the damping)."""
    N = (516, 30, 40)
    size = tuple(n * DL for n in N)
    bspec = {"pec": PEC, "abs": ABS}[bspec_name]
    structures = [td.Structure(geometry=td.Box(center=(-0.5, 0.2, 0.3), size=(3.0, 0.2, 0.2)), medium=td.Medium(permittivity=3.0, conductivity=0.02)),
                  td.Structure(geometry=td.Sphere(center=(6.3, -0.3, -0.5), radius=0.25), medium=td.Medium(permittivity=2.5)),
                  td.Structure(geometry=td.Box(center=(0.1, 0.1, 0), size=(0.1, 0.1, 0.1)), medium=td.PEC)]
    mons = [td.FieldTimeMonitor(center=(0.0, 0.0, 0.0), size=(0, 0, 0), name="mid", interval=2, fields=["Ez", "Hx"], colocate=False),
            td.FieldMonitor(center=(0, 0, 0.1), size=(td.inf, td.inf, 0), freqs=[3e14], name="f", colocate=False)]
    disc = discretize(_sim(N, monitors=False, structures=structures, bspec=bspec).updated_copy(monitors=mons), n_steps=26)
    disc.spec.decay_every = 0
    outs = []
    for twostep, split in ((0, -1), (w + 64 * zc, 1), (w + 64 * zc, 0)):
        with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
            e.set_option(L.OPT_ROWS, 3)
            e.set_option(L.OPT_TWOSTEP, twostep)
            e.set_option(L.OPT_TILE_SPLIT, split)
            pairs = sum(int(e.run(r).fused2_pairs) for r in (11, 15))
            outs.append(([e.get_field(c) for c in range(6)], e.results(), pairs))
    assert outs[0][2] == 0 and outs[1][2] == 12 and outs[2][2] == 12, [o[2] for o in outs]
    for got in outs[1:]:
        for c in range(6):
            assert np.array_equal(got[0][c], outs[0][0][c]), c
        for k in outs[0][1]:
            assert np.array_equal(np.asarray(got[1][k]), np.asarray(outs[0][1][k])), k


@pytest.mark.parametrize("name,zc,pf", [("one_tile", 32, 10), ("ragged_rows", 3, 10), ("two_x_tiles", 5, 10), ("three_x_tiles_tall", 4, 10),
                                        ("ragged_rows", 3, 11), ("two_x_tiles", 32, 11), ("one_tile", 2, 12), ("three_x_tiles_tall", 5, 12),
                                        ("one_tile", 32, 13), ("ragged_rows", 3, 13), ("two_x_tiles", 5, 13), ("three_x_tiles_tall", 4, 13),
                                        ("one_tile", 32, 15), ("ragged_rows", 3, 15), ("two_x_tiles", 5, 15), ("three_x_tiles_tall", 4, 15),
                                        ("ragged_rows", 3, 14)])
def test_prefetch_instantiations_equal_single_steps(name, zc, pf, emu_lib):
    """The PREFETCH instantiations of the vacuum sweep (FDTD_OPT_WHATIF = 10 ... 12: part of the next plane travels global memory ->
    LDS by LDS-DMA while this one is computed; E1 exchanged through ONE buffer, published behind the second barrier): the same bits
    as single steps — tile edges in x / y / z, chunk prologues, sources between the steps, runs of odd lengths.  13: E_x of the row
    above handed down through a ninth exchange array instead of a second global load."""
    N = SHAPES[name]
    disc = discretize(_sim(N, monitors=False), n_steps=26)
    disc.spec.decay_every = 0
    ref_f, _, p0 = _run(disc.spec, emu_lib, 0)

    def run_pf():
        with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_FUSED, z_chunk=2) as e:
            e.set_option(L.OPT_ROWS, 3)
            e.set_option(L.OPT_TWOSTEP, 16 + 64 * zc)
            e.set_option(L.OPT_WHATIF, pf)
            pairs = sum(int(e.run(r).fused2_pairs) for r in (11, 15))
            return [e.get_field(c) for c in range(6)], pairs
    got_f, p1 = run_pf()
    assert p0 == 0 and p1 == 5 + 7, p1
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
