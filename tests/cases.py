"""Shared parity cases: small simulations exercising every kernel of the hot path.  Used by the
CPU emulator tests (kernel logic) and by the GPU tests (the real thing), both against the oracle."""
import numpy as np

import tidy3d_amd.schema as td
from tidy3d_amd.discretize import discretize

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)


def _sim(N, bspec, structures=(), sources=None, monitors=None, dl=DL, grid_spec=None):
    size = tuple(n * dl for n in N)
    if sources is None:
        sources = [td.PointDipole(center=(0.03, -0.07, 0.01), source_time=PULSE, polarization="Ez"),
                   td.PointDipole(center=(-0.1, 0.07, 0.06), source_time=PULSE, polarization="Hx")]
    if monitors is None:
        monitors = [td.FieldTimeMonitor(center=(-0.1, 0.1, -0.05), size=(0.3, 0.2, 0.1), name="t",
                                        colocate=False, interval=7),
                    td.FieldMonitor(center=(0, 0, 0), size=(0.4, 0.3, 0), freqs=[2.5e14, 3e14],
                                    name="f")]
    return td.Simulation(size=size, grid_spec=grid_spec or td.GridSpec.uniform(dl=dl), run_time=1e-12,
                         structures=list(structures), sources=sources, monitors=monitors,
                         boundary_spec=bspec, shutoff=0)


def pipelined_slab_case(N=(28, 24, 32)):
    """Everything the pipelined z-slab schedule splits between its two streams: x/y CPML slabs, ADE
    (Lorentz sphere), a lossy box, electric and magnetic dipoles, a plane wave (incident-grid replica
    on the comm stream), time / DFT / flux monitors.  Periodic z: runs as a 1-rank self exchange."""
    structures = [
        td.Structure(geometry=td.Sphere(center=(0.05, 0, 0.1), radius=0.2),
                     medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])),
        td.Structure(geometry=td.Box(center=(-0.3, 0, -0.3), size=(0.25, 0.3, 0.8)),
                     medium=td.Medium(permittivity=3.0, conductivity=0.02))]
    bspec = td.BoundarySpec(x=td.Boundary.pml(num_layers=5), y=td.Boundary.pml(num_layers=4), z=td.Boundary.periodic())
    top = N[2] * DL / 2          # sources and monitors inside the top boundary chunk as well
    sources = [td.PointDipole(center=(0.03, -0.07, 0.01), source_time=PULSE, polarization="Ez"),
               td.PointDipole(center=(-0.1, 0.07, top - 0.04), source_time=PULSE, polarization="Hx"),
               td.PlaneWave(center=(0.3, 0, 0), size=(0, td.inf, td.inf), source_time=PULSE, direction="-",
                            pol_angle=0.4)]
    monitors = [td.FieldTimeMonitor(center=(-0.1, 0.1, -0.05), size=(0.3, 0.2, td.inf), name="t", colocate=False,
                                    interval=7),
                td.FieldMonitor(center=(0, 0, top - 0.05), size=(0.4, 0.3, 0), freqs=[2.5e14, 3e14], name="f"),
                td.FluxMonitor(center=(0, 0, 0), size=(0.5, 0.4, 1.5 * top), freqs=[3e14], name="box")]
    return _sim(N, bspec, structures, sources=sources, monitors=monitors)


def slab_pairs_box(N=(24, 20, 44), periodic_z=False):
    """PEC / PMC walls (or a periodic z), a lossy block and a PEC box, electric and magnetic dipoles next to where 2- and 3-rank
    runs cut the grid, a probe and an x-z plane recorded every 7th / 10th step: what z-slab ranks advance in step pairs
    (tests/test_dist_gloo.py)."""
    sx, sy, sz = (n * DL for n in N)
    zb = td.Boundary.periodic() if periodic_z else td.Boundary(minus=td.PECBoundary(), plus=td.PECBoundary())
    bspec = td.BoundarySpec(x=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                            y=td.Boundary(minus=td.PECBoundary(), plus=td.PECBoundary()), z=zb)
    structures = [td.Structure(geometry=td.Box(center=(0.1, 0, 0), size=(0.4, 0.3, td.inf)), medium=td.Medium(permittivity=2.5, conductivity=0.03)),
                  td.Structure(geometry=td.Box(center=(-0.3, 0.2, 0.05), size=(0.15, 0.15, 0.3)), medium=td.PEC)]
    sources = [td.PointDipole(center=(0.02, 0.01, 0.03), source_time=PULSE, polarization="Ez"),
               td.PointDipole(center=(-0.2, 0.12, -0.5 * sz + 14.6 * DL), source_time=PULSE, polarization="Ex"),      # next to the 3-rank cut at 15
               td.PointDipole(center=(0.25, -0.2, -0.5 * sz + 21.4 * DL), source_time=PULSE, polarization="Hy"),      # below the 2-rank cut at 22
               td.PointDipole(center=(-0.1, -0.3, -0.5 * sz + 22.6 * DL), source_time=PULSE, polarization="Ey"),      # above it
               td.PointDipole(center=(0.3, 0.3, -0.5 * sz + 29.5 * DL), source_time=PULSE, polarization="Hx"),
               td.PointDipole(center=(0.0, 0.2, 0.5 * sz - 1.4 * DL), source_time=PULSE, polarization="Ez")]
    monitors = [td.FieldTimeMonitor(center=(0.1, 0.05, 0.1), size=(0, 0, 0), name="probe", interval=7, colocate=False),
                td.FieldTimeMonitor(center=(0, 0, 0), size=(td.inf, 0, td.inf), name="plane", interval=10, colocate=False)]
    return td.Simulation(size=(sx, sy, sz), grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, structures=structures,
                         sources=sources, monitors=monitors, boundary_spec=bspec, shutoff=0)


def slab_pairs_pml_box(N=(24, 20, 44), layers=(4, 5, 3), near_cut=False, periodic_z=False):
    """CPML on every face (inside nothing: the layers are added to N), a lossy slab through the whole grid along z and a PEC box,
    dipoles of both kinds deep inside the CPML-free box and away from where 2- and 3-rank runs cut the grid, a probe and an x-z plane
    recorded every 7th / 10th step: what CPML-carrying z-slab ranks advance in shell2 pairs (tests/test_dist_gloo.py)."""
    sx, sy, sz = (n * DL for n in N)
    bspec = td.BoundarySpec(x=td.Boundary.pml(num_layers=layers[0]), y=td.Boundary.pml(num_layers=layers[1]),
                            z=td.Boundary.periodic() if periodic_z else td.Boundary.pml(num_layers=layers[2]))
    structures = [td.Structure(geometry=td.Box(center=(0.1, 0, 0), size=(0.4, 0.3, td.inf)), medium=td.Medium(permittivity=2.5, conductivity=0.03)),
                  td.Structure(geometry=td.Box(center=(-0.3, 0.2, 0.05), size=(0.15, 0.15, 0.3)), medium=td.PEC)]
    sources = [td.PointDipole(center=(0.02, 0.01, -0.5 * sz + 9.6 * DL), source_time=PULSE, polarization="Ez"),
               td.PointDipole(center=(-0.1, 0.05, -0.5 * sz + 30.4 * DL), source_time=PULSE, polarization="Hy"),
               td.PointDipole(center=(0.1, -0.1, -0.5 * sz + 35.5 * DL), source_time=PULSE, polarization="Ex")]
    if near_cut:      # one plane above the lower cut of the tall box's 3-rank run (plane 21): that rank keeps single steps while the dipole injects
        sources.append(td.PointDipole(center=(-0.15, 0.1, -0.5 * sz + 19.4 * DL), source_time=PULSE, polarization="Ey"))
    monitors = [td.FieldTimeMonitor(center=(0.1, 0.05, 0.1), size=(0, 0, 0), name="probe", interval=7, colocate=False),
                td.FieldTimeMonitor(center=(0, 0, 0), size=(td.inf, 0, td.inf), name="plane", interval=10, colocate=False)]
    return td.Simulation(size=(sx, sy, sz), grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, structures=structures,
                         sources=sources, monitors=monitors, boundary_spec=bspec, shutoff=0)


def slab_pairs_pml_box_tall():
    """66 planes for three ranks: the dipoles keep five or more planes from every cut a balanced split may choose."""
    return slab_pairs_pml_box(N=(24, 20, 60))


def slab_pairs_pml_box_near_cut():
    return slab_pairs_pml_box(N=(24, 20, 60), near_cut=True)


def random_slab_pml_box(seed, index):
    """A random CPML-walled box for the step pairs of CPML-carrying z-slab ranks (scripts/fuzz_slab_cpml.py, tests/test_dist_gloo.py):
    layers per axis 0 / 2 ... 6 (absent on at most one axis; PMC on a min face without layers now and then), lossy / PEC bodies anywhere
    (through cuts and layers), two to four dipoles of either kind — most runs keep them deep inside, some put one next to a cut or
    into the layers (that rank then keeps single steps while it injects) —, a probe and a plane recorded at random intervals.
    -> (simulation, world, FDTD_OPT_TWOSTEP word, steps)"""
    rng = np.random.default_rng([int(seed), int(index)])
    world = int(rng.integers(2, 5))
    N = (int(rng.integers(6, 10)) * 4, int(rng.integers(18, 30)), int(rng.integers(13, 20)) * world + int(rng.integers(0, 4)))
    layers = [int(rng.integers(2, 7)) for _ in range(3)]
    none = int(rng.integers(0, 5))
    if none < 3:
        layers[none] = 0
    def bnd(a):
        if layers[a]:
            return td.Boundary.pml(num_layers=layers[a]) if rng.random() < 0.7 else td.Boundary.stable_pml(num_layers=layers[a])
        return td.Boundary(minus=td.PMCBoundary() if rng.random() < 0.4 else td.PECBoundary(), plus=td.PECBoundary())
    bspec = td.BoundarySpec(x=bnd(0), y=bnd(1), z=bnd(2))
    size = tuple(n * DL for n in N)
    structures = []
    for _ in range(int(rng.integers(1, 4))):
        c = tuple(float(rng.uniform(-0.4, 0.4) * s) for s in size)
        sz = tuple(float(rng.uniform(0.1, 0.5) * s) if rng.random() < 0.8 else td.inf for s in size)
        med = td.PEC if rng.random() < 0.25 else td.Medium(permittivity=float(rng.uniform(1.5, 4.0)), conductivity=float(rng.choice([0.0, 0.02])))
        structures.append(td.Structure(geometry=td.Box(center=c, size=sz), medium=med))
    sources = []
    wild = rng.random() < 0.3
    for _ in range(int(rng.integers(2, 5))):
        f = 0.45 if wild else 0.18
        c = tuple(float(rng.uniform(-f, f) * s) for s in size)
        sources.append(td.PointDipole(center=c, source_time=PULSE, polarization=str(rng.choice(["Ex", "Ey", "Ez", "Hx", "Hy", "Hz"]))))
    monitors = [td.FieldTimeMonitor(center=tuple(float(rng.uniform(-0.3, 0.3) * s) for s in size), size=(0, 0, 0), name="probe",
                                    interval=int(rng.integers(3, 12)), colocate=False),
                td.FieldTimeMonitor(center=(0, 0, 0), size=(td.inf, 0, td.inf), name="plane", interval=int(rng.integers(6, 16)), colocate=False)]
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, structures=structures, sources=sources,
                        monitors=monitors, boundary_spec=bspec, shutoff=0)
    twostep = int(rng.choice([5, 6, 8, 16])) + 64 * int(rng.integers(3, 9))
    return sim, world, twostep, int(rng.integers(30, 52))


def slab_pairs_box_periodic():
    return slab_pairs_box(periodic_z=True)


def pec_box(N=(20, 16, 12)):
    return _sim(N, td.BoundarySpec.all_sides(td.PECBoundary()))


def pec_box_vec(N=(24, 16, 12)):
    """nx % 4 == 0 after discretisation -> float4 path of the main kernels."""
    return _sim(N, td.BoundarySpec.all_sides(td.PECBoundary()))


def periodic_box(N=(16, 12, 10)):
    return _sim(N, td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(),
                                   z=td.Boundary.periodic()))


def periodic_box_tall(N=(16, 12, 16)):
    """Periodic in z with enough planes for two fused z-slabs (lo == hi neighbour pairing)."""
    return periodic_box(N)


def pml_box(N=(16, 12, 10)):
    return _sim(N, td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.pml(num_layers=5),
                                   z=td.Boundary.pml(num_layers=3)))


def stable_pml_box(N=(12, 10, 8)):
    return _sim(N, td.BoundarySpec(x=td.Boundary.stable_pml(num_layers=6),
                                   y=td.Boundary.stable_pml(num_layers=5),
                                   z=td.Boundary.stable_pml(num_layers=4)))


def media_mix(N=(13, 11, 9)):
    """PML + periodic + PMC/PEC walls, Lorentz sphere (ADE), lossy box, PEC box."""
    structures = [
        td.Structure(geometry=td.Sphere(center=(0.05, 0, 0), radius=0.2),
                     medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])),
        td.Structure(geometry=td.Box(center=(-0.2, 0, 0), size=(0.15, 0.3, 0.2)),
                     medium=td.Medium(permittivity=3.0, conductivity=0.02)),
        td.Structure(geometry=td.Box(center=(0.2, 0.1, 0), size=(0.1, 0.1, 0.1)), medium=td.PEC)]
    bspec = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.periodic(),
                            z=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()))
    return _sim(N, bspec, structures)


def drude_in_pml(N=(16, 12, 12)):
    """Drude (two real poles) slab running through the x/y PML, plus an over-damped Lorentz box
    and a Debye box: exercises multi-pole ADE groups inside CPML slabs."""
    structures = [
        td.Structure(geometry=td.Box(center=(0, 0, -0.15), size=(td.inf, td.inf, 0.2)),
                     medium=td.Drude(eps_inf=1.5, coeffs=[(1.2e15, 8e13)])),
        td.Structure(geometry=td.Box(center=(0.1, 0, 0.15), size=(0.2, 0.2, 0.15)),
                     medium=td.Lorentz(eps_inf=1.2, coeffs=[(0.8, 2e14, 5e14)])),
        td.Structure(geometry=td.Cylinder(center=(-0.2, 0, 0.1), radius=0.1, length=0.2, axis=1),
                     medium=td.Debye(eps_inf=2.0, coeffs=[(1.0, 2e-15)]))]
    return _sim(N, td.BoundarySpec.all_sides(td.PML(num_layers=4)), structures)


def lorentz_sphere(N=(20, 18, 16), pml=True):
    """A volumetric dispersive body away from every wall — SURVEY 8(d)'s V3 in miniature: a Lorentz sphere (one pole pair) with a
    two-pole Drude block inside it and a dielectric collar, CPML on all faces (or PEC walls), the dipole ON a dispersive cell,
    a DFT plane and a probe through the sphere.  At three times the size the step pairs advance its cells themselves (round 6:
    the sweep subtracts the paged memory terms, ade2_kernel follows), tests/test_gpu_disp.py."""
    structures = [
        td.Structure(geometry=td.Sphere(center=(0.02, -0.03, 0.0), radius=0.3), medium=td.Lorentz(eps_inf=2.0, coeffs=[(2.0, 4e14, 2e13)])),
        td.Structure(geometry=td.Box(center=(0.05, 0.0, 0.05), size=(0.2, 0.15, 0.1)), medium=td.Drude(eps_inf=1.5, coeffs=[(3e14, 1e13), (5e14, 3e13)])),
        td.Structure(geometry=td.Box(center=(-0.32, 0.0, 0.0), size=(0.1, 0.3, 0.3)), medium=td.Medium(permittivity=2.5))]
    bspec = td.BoundarySpec.all_sides(td.PML(num_layers=4)) if pml else td.BoundarySpec.all_sides(td.PECBoundary())
    monitors = [td.FieldTimeMonitor(center=(0.1, 0.05, -0.05), size=(0, 0, 0), name="t", colocate=False, interval=1),
                td.FieldMonitor(center=(0, 0, 0), size=(0.5, 0.4, 0), freqs=[2.5e14, 3e14], name="f")]
    return _sim(N, bspec, structures, monitors=monitors)


def nonuniform_grid(N=(16, 12, 10)):
    """CustomGrid with graded steps along x and z."""
    dlx = tuple(0.03 + 0.02 * np.abs(np.linspace(-1, 1, 20)))
    dlz = tuple(0.04 + 0.015 * np.linspace(0, 1, 12))
    gs = td.GridSpec(grid_x=td.CustomGrid(dl=dlx), grid_y=td.UniformGrid(dl=DL),
                     grid_z=td.CustomGrid(dl=dlz))
    size = (float(np.sum(dlx)), 12 * DL, float(np.sum(dlz)))
    sim = _sim((1, 1, 1), td.BoundarySpec(x=td.Boundary.pml(num_layers=3), y=td.Boundary.pec(),
                                          z=td.Boundary.pml(num_layers=3)), grid_spec=gs)
    return sim.copy(size=size)


def tfsf_box(N=(20, 16, 16)):
    """TFSF box around a dielectric sphere, PML everywhere (Mie set-up in miniature)."""
    src = td.TFSF(center=(0, 0, 0), size=(0.5, 0.4, 0.4), source_time=PULSE, injection_axis=2,
                  direction="+", pol_angle=0.4)
    structures = [td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=0.12),
                               medium=td.Medium(permittivity=4.0))]
    mons = [td.FieldTimeMonitor(center=(0, 0, 0), size=(0.7, 0.6, 0.6), name="t", colocate=False, interval=7),
            td.FluxMonitor(center=(0, 0, 0), size=(0.7, 0.6, 0.6), freqs=[2.5e14, 3e14], name="sca")]
    return _sim(N, td.BoundarySpec.all_sides(td.PML(num_layers=4)), structures, sources=[src],
                monitors=mons)


def tfsf_angled_box(N=(20, 16, 16)):
    """TFSF box at oblique incidence (incident grid along k_hat, cubic interpolation: four aux entries per leg) around a
    dielectric sphere."""
    src = td.TFSF(center=(0, 0, 0), size=(0.5, 0.4, 0.4), source_time=PULSE, injection_axis=2, direction="+", pol_angle=0.4,
                  angle_theta=0.45, angle_phi=0.8)
    structures = [td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=0.12), medium=td.Medium(permittivity=4.0))]
    mons = [td.FieldTimeMonitor(center=(0, 0, 0), size=(0.7, 0.6, 0.6), name="t", colocate=False, interval=7),
            td.FluxMonitor(center=(0, 0, 0), size=(0.7, 0.6, 0.6), freqs=[2.5e14, 3e14], name="sca")]
    return _sim(N, td.BoundarySpec.all_sides(td.PML(num_layers=4)), structures, sources=[src], monitors=mons)


def planewave_periodic(N=(8, 12, 20)):
    """PlaneWave launched along -x through a periodic cross-section, PML along x."""
    src = td.PlaneWave(center=(0.2, 0, 0), size=(0, td.inf, td.inf), source_time=PULSE, direction="-",
                       pol_angle=np.pi / 2)
    structures = [td.Structure(geometry=td.Box(center=(-0.1, 0, 0), size=(0.1, td.inf, td.inf)),
                               medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)]))]
    mons = [td.FieldMonitor(center=(-0.3, 0, 0), size=(0, td.inf, td.inf), freqs=[2.5e14, 3e14], name="f"),
            td.FluxMonitor(center=(0.3, 0, 0), size=(0, td.inf, td.inf), freqs=[3e14], name="back")]
    bspec = td.BoundarySpec(x=td.Boundary.pml(num_layers=6), y=td.Boundary.periodic(), z=td.Boundary.periodic())
    return _sim(N, bspec, structures, sources=[src], monitors=mons)


def gold_johnson_christy():
    """Au (Johnson & Christy 1972 fit, 5 pole pairs) exactly as the reference's material library
    serialises it (ref material_library.py:506), read from the committed golden fixture."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schema_golden.json")) as f:
        rec = json.load(f)["media"][-1]
    return td.parse(rec["json"])


def au_array(N=(12, 12, 24), dl=0.01):
    """BASELINE config[4] in miniature: periodic array of Au nano-discs (5-pole dispersive ADE) on a
    dielectric slab, plane wave from above, PML along z."""
    pulse = td.GaussianPulse(freq0=5e14, fwidth=1e14)
    size = tuple(n * dl for n in N)
    structures = [td.Structure(geometry=td.Box(center=(0, 0, -size[2] / 4), size=(td.inf, td.inf, size[2] / 2)),
                               medium=td.Medium(permittivity=2.1)),
                  td.Structure(geometry=td.Cylinder(center=(0, 0, 0.02), radius=0.035, length=0.03, axis=2),
                               medium=gold_johnson_christy())]
    src = td.PlaneWave(center=(0, 0, size[2] / 2 - 0.03), size=(td.inf, td.inf, 0), source_time=pulse, direction="-")
    mons = [td.FluxMonitor(center=(0, 0, size[2] / 2 - 0.015), size=(td.inf, td.inf, 0), freqs=[4.5e14, 5.5e14], name="R"),
            td.FluxMonitor(center=(0, 0, -size[2] / 2 + 0.03), size=(td.inf, td.inf, 0), freqs=[4.5e14, 5.5e14], name="T"),
            td.FieldTimeMonitor(center=(0, 0, 0.02), size=(0.05, 0.05, 0), name="t", interval=9, colocate=False)]
    bspec = td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml(num_layers=6))
    return td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12, structures=structures,
                         sources=[src], monitors=mons, boundary_spec=bspec, shutoff=0)


def two_d(N=(36, 30)):
    """2-D simulation: zero size along y (one cell, periodic; ref simulation.py:2272), PML in x and z,
    a dielectric bar, out-of-plane (TE) and in-plane (TM) dipoles."""
    size = (N[0] * DL, 0.0, N[1] * DL)
    return td.Simulation(
        size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, shutoff=0,
        structures=[td.Structure(geometry=td.Box(center=(0.1, 0, 0.1), size=(0.3, td.inf, 0.2)),
                                 medium=td.Medium(permittivity=4.0))],
        sources=[td.PointDipole(center=(0.03, 0, -0.2), source_time=PULSE, polarization="Ey"),
                 td.PointDipole(center=(-0.2, 0, 0.07), source_time=PULSE, polarization="Ex")],
        monitors=[td.FieldTimeMonitor(center=(0.1, 0, 0.2), size=(0.3, 0, 0.2), name="t", interval=5, colocate=False),
                  td.FieldMonitor(center=(0, 0, 0), size=(0.8, 0, 0.6), freqs=[2.5e14, 3e14], name="f")],
        boundary_spec=td.BoundarySpec(x=td.Boundary.pml(num_layers=6), y=td.Boundary.periodic(),
                                      z=td.Boundary.pml(num_layers=6)))


def one_d(N=60):
    """1-D simulation along z (zero size in x and y): a current sheet, a lossy slab, PML at both ends."""
    return td.Simulation(
        size=(0.0, 0.0, N * DL), grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, shutoff=0,
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0.5), size=(td.inf, td.inf, 0.4)),
                                 medium=td.Medium(permittivity=2.25, conductivity=0.02))],
        sources=[td.UniformCurrentSource(center=(0, 0, -0.8), size=(td.inf, td.inf, 0), source_time=PULSE,
                                         polarization="Ex")],
        monitors=[td.FieldTimeMonitor(center=(0, 0, 1.0), size=(0, 0, 0), name="t", interval=3),
                  td.FluxMonitor(center=(0, 0, 1.1), size=(td.inf, td.inf, 0), freqs=[2.5e14, 3e14], name="f")],
        boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(),
                                      z=td.Boundary.pml(num_layers=8)))


def pmc_plus_mix(N=(20, 16, 14)):
    """PMC on PLUS faces (x and z; SolverSpec.mirror_plus: two ghost cells beyond each wall, mirror images refreshed every
    step) with CPML on the opposite faces and on y, a Lorentz sphere cut by the x wall, a lossy box at the z wall, dipoles
    near both walls."""
    structures = [
        td.Structure(geometry=td.Sphere(center=(0.45, 0, 0.1), radius=0.2), medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])),
        td.Structure(geometry=td.Box(center=(-0.1, 0.1, 0.3), size=(0.3, 0.2, 0.2)), medium=td.Medium(permittivity=3.0, conductivity=0.02))]
    bspec = td.BoundarySpec(x=td.Boundary(minus=td.PML(num_layers=4), plus=td.PMCBoundary()), y=td.Boundary.pml(num_layers=3),
                            z=td.Boundary(minus=td.PML(num_layers=3), plus=td.PMCBoundary()))
    size = tuple(n * DL for n in N)
    sources = [td.PointDipole(center=(size[0] / 2 - 0.16, -0.07, 0.01), source_time=PULSE, polarization="Ez"),
               td.PointDipole(center=(-0.1, 0.07, size[2] / 2 - 0.13), source_time=PULSE, polarization="Hx"),
               td.PointDipole(center=(0.2, 0.02, 0.1), source_time=PULSE, polarization="Ey")]
    monitors = [td.FieldTimeMonitor(center=(0.3, 0.05, 0.2), size=(0.5, 0.2, 0.4), name="t", colocate=False, interval=7),
                td.FieldMonitor(center=(0, 0, 0.1), size=(td.inf, td.inf, 0), freqs=[2.5e14, 3e14], name="f"),
                td.FluxMonitor(center=(size[0] / 2 - 0.2, 0, 0), size=(0, td.inf, td.inf), freqs=[3e14], name="fx")]
    return _sim(N, bspec, structures, sources=sources, monitors=monitors)


def absorber_mix(N=(16, 12, 16)):
    """Absorber layers (ref boundary.py:427) on x (both faces), y+ (PML on y-) and z (both faces), with a
    Drude slab and a lossy box running through them: the damping kernel next to CPML and ADE."""
    structures = [
        td.Structure(geometry=td.Box(center=(0, 0, -0.15), size=(td.inf, td.inf, 0.2)),
                     medium=td.Drude(eps_inf=1.5, coeffs=[(1.2e15, 8e13)])),
        td.Structure(geometry=td.Box(center=(0.2, 0, 0.2), size=(td.inf, 0.2, 0.15)),
                     medium=td.Medium(permittivity=2.5, conductivity=0.01))]
    ab = lambda n, s: td.Absorber(num_layers=n, parameters=td.AbsorberParams(sigma_max=s))
    bspec = td.BoundarySpec(x=td.Boundary(minus=ab(4, 1.0), plus=ab(4, 2.0)),
                            y=td.Boundary(minus=td.PML(num_layers=3), plus=ab(5, 0.8)),
                            z=td.Boundary(minus=ab(3, 1.5), plus=ab(4, 1.5)))
    return _sim(N, bspec, structures)


def absorber_odd_rows(N=(13, 10, 9)):
    """Absorber on x+ / z- with a row length that needs PEC padding (nx % 4 != 0), PMC at x-."""
    ab = td.Absorber(num_layers=4, parameters=td.AbsorberParams(sigma_max=1.2, sigma_min=0.05, sigma_order=2))
    bspec = td.BoundarySpec(x=td.Boundary(minus=td.PMCBoundary(), plus=ab),
                            y=td.Boundary.periodic(),
                            z=td.Boundary(minus=ab, plus=td.PECBoundary()))
    return _sim(N, bspec)


def bloch_box(N=(12, 10, 9)):
    """Bloch boundaries on all three axes (complex fields as a (Re, Im) pair of solvers): Lorentz sphere
    (ADE on both parts), lossy box, electric and magnetic dipoles; odd row length -> scalar kernels."""
    structures = [
        td.Structure(geometry=td.Sphere(center=(0.05, 0, 0.05), radius=0.15),
                     medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])),
        td.Structure(geometry=td.Box(center=(-0.25, 0, -0.2), size=(0.2, td.inf, 0.15)),
                     medium=td.Medium(permittivity=3.0, conductivity=0.02))]
    bspec = td.BoundarySpec(x=td.Boundary.bloch(0.31), y=td.Boundary.bloch(-0.17), z=td.Boundary.bloch(0.45))
    return _sim(N, bspec, structures)


def bloch_xy_pml_z(N=(16, 12, 12)):
    """The usual periodic-array set-up at oblique incidence: Bloch in x and y, CPML in z, a Drude film
    through the cell, float4 rows."""
    structures = [td.Structure(geometry=td.Box(center=(0, 0, -0.15), size=(td.inf, td.inf, 0.1)),
                               medium=td.Drude(eps_inf=1.5, coeffs=[(1.2e15, 8e13)])),
                  td.Structure(geometry=td.Box(center=(0.1, 0.1, 0.1), size=(0.3, 0.2, 0.2)), medium=td.PEC)]
    bspec = td.BoundarySpec(x=td.Boundary.bloch(0.2), y=td.Boundary.bloch(0.35), z=td.Boundary.pml(num_layers=4))
    return _sim(N, bspec, structures)


def bloch_x_only(N=(16, 10, 8)):
    """Bloch along x only; y periodic, z PMC / PEC walls."""
    bspec = td.BoundarySpec(x=td.Boundary.bloch(-0.4), y=td.Boundary.periodic(),
                            z=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()))
    return _sim(N, bspec)


def bloch_planewave(N=(12, 8, 20)):
    """Oblique PlaneWave (current sheets with the Bloch phase gradient) onto a dielectric block, Bloch in
    x and y from the source, CPML in z, flux and field monitors either side."""
    pw = td.PlaneWave(center=(0, 0, -0.35), size=(td.inf, td.inf, 0), source_time=PULSE, direction="+",
                      angle_theta=0.5, angle_phi=0.6, pol_angle=0.4)
    size = tuple(n * DL for n in N)
    bspec = td.BoundarySpec(x=td.Boundary.bloch_from_source(pw, size[0], 0), y=td.Boundary.bloch_from_source(pw, size[1], 1),
                            z=td.Boundary.pml(num_layers=5))
    structures = [td.Structure(geometry=td.Box(center=(0.1, 0, 0.25), size=(0.3, td.inf, 0.2)), medium=td.Medium(permittivity=3.0))]
    monitors = [td.FluxMonitor(center=(0, 0, 0.45), size=(td.inf, td.inf, 0), freqs=[2.8e14, 3e14], name="T"),
                td.FieldMonitor(center=(0, 0, -0.5), size=(td.inf, td.inf, 0), freqs=[3e14], name="r", colocate=False),
                td.FieldTimeMonitor(center=(0, 0, 0.1), size=(0.2, 0.2, 0), name="t", interval=6, colocate=False)]
    return _sim(N, bspec, structures, sources=[pw], monitors=monitors)


def wide_flat(N=(128, 128, 16)):
    """2^18 cells, 16 planes: a z-slab run renames the axes so that a 128-cell axis becomes the slab axis
    (dist.best_slab_shift).  Not in CASES (too slow for the per-case parity lists under the emulator)."""
    structures = [td.Structure(geometry=td.Box(center=(0.5, -0.3, 0), size=(1.0, 0.8, 0.3)), medium=td.Medium(permittivity=3.0))]
    monitors = [td.FieldMonitor(center=(0, 0, 0), size=(td.inf, td.inf, 0), freqs=[3e14], name="f"),
                td.FieldTimeMonitor(center=(0.4, 0.2, 0.1), size=(0.5, 0, 0.3), name="t", interval=2, colocate=False)]
    return _sim(N, td.BoundarySpec.all_sides(td.PECBoundary()), structures, monitors=monitors)


def sheets_box(N=(18, 14, 16)):
    """Medium2D sheets (ref medium.py:6090) and a LumpedResistor: an ohmic / Drude sheet across a dielectric step and through the
    x layers, a resistor patch on another plane; CPML on x, PEC / PMC elsewhere."""
    structures = [td.Structure(geometry=td.Box(center=(0, 0, -0.25), size=(td.inf, td.inf, 0.5)), medium=td.Medium(permittivity=2.25)),
                  td.Structure(geometry=td.Box(center=(0, 0, 0.0), size=(td.inf, 0.4, 0)),
                               medium=td.Medium2D(ss=td.Medium(conductivity=2e-3), tt=td.Drude(eps_inf=1.0, coeffs=[(3e14, 2e13)])))]
    bspec = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary(minus=td.PMCBoundary(), plus=td.PECBoundary()),
                            z=td.Boundary(minus=td.PECBoundary(), plus=td.PECBoundary()))
    sim = _sim(N, bspec, structures)
    sim.lumped_elements = (td.LumpedResistor(center=(0.1, 0.1, 0.2), size=(0.3, 0, 0.2), resistance=120.0, voltage_axis=2, name="R"),)
    return sim


def fully_aniso_box(N=(18, 16, 14)):
    """FullyAnisotropicMedium (ref medium.py:5058): a rotated biaxial sphere (the off-diagonal coupling lists of csrc/fdtd_aniso.hpp)
    next to a lossy block and a PEC box; CPML on x, PEC / CPML on y, periodic z."""
    def rot(axis, ang):
        c, s = np.cos(ang), np.sin(ang)
        return np.array({2: [[c, -s, 0], [s, c, 0], [0, 0, 1]], 1: [[c, 0, s], [0, 1, 0], [-s, 0, c]]}[axis])
    med = td.FullyAnisotropicMedium.from_diagonal(2.0, 5.0, 3.2, rot(2, 0.7) @ rot(1, 0.4))
    structures = [td.Structure(geometry=td.Sphere(center=(0.05, 0, 0.02), radius=0.28), medium=med),
                  td.Structure(geometry=td.Box(center=(-0.2, 0.1, 0), size=(0.2, 0.3, td.inf)), medium=td.Medium(permittivity=2.5, conductivity=0.02)),
                  td.Structure(geometry=td.Box(center=(0.2, -0.15, 0.1), size=(0.15, 0.15, 0.15)), medium=td.PEC)]
    bspec = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary(minus=td.PECBoundary(), plus=td.PML(num_layers=3)), z=td.Boundary.periodic())
    return _sim(N, bspec, structures)


CASES = {
    "fully_aniso_box": fully_aniso_box,
    "sheets_box": sheets_box,
    "bloch_box": bloch_box, "bloch_planewave": bloch_planewave, "bloch_xy_pml_z": bloch_xy_pml_z, "bloch_x_only": bloch_x_only,
    "two_d": two_d, "one_d": one_d, "absorber_mix": absorber_mix, "pmc_plus_mix": pmc_plus_mix, "absorber_odd_rows": absorber_odd_rows,
    "tfsf_box": tfsf_box, "tfsf_angled_box": tfsf_angled_box, "planewave_periodic": planewave_periodic, "au_array": au_array,
    "pec_box": pec_box, "pec_box_vec": pec_box_vec, "periodic_box": periodic_box,
    "periodic_box_tall": periodic_box_tall,
    "pml_box": pml_box, "stable_pml_box": stable_pml_box, "media_mix": media_mix,
    "drude_in_pml": drude_in_pml, "nonuniform_grid": nonuniform_grid, "lorentz_sphere": lorentz_sphere,
}


def rel_err(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def run_case(name, lib, n_steps=60, scale=1, **engine_kw):
    """Run case ``name`` through the C ABI library ``lib`` and through the oracle; return the
    worst rel-L2 error over all monitors and over the final (E, H) state (field errors are
    normalised by the norm of the whole E resp. H triple so that symmetry-suppressed components
    do not inflate the figure)."""
    from oracle.fdtd_numpy import OracleFdtd
    from tidy3d_amd.engine import HipEngine
    fn = CASES[name]
    sim = fn() if scale == 1 else fn(tuple(int(n * scale) for n in fn.__defaults__[0]))
    disc = discretize(sim, n_steps=n_steps)
    o = OracleFdtd(disc.spec)
    ref = o.run()
    with HipEngine(disc.spec, lib=lib, **engine_kw) as e:
        e.run()
        got = e.results()
        fields = [e.get_field(c) for c in range(6)]
    worst = 0.0
    # monitors that (still) hold almost nothing — e.g. the flux plane behind a one-way source before
    # any reflection arrives — are judged against the scale of the largest monitor of the run
    scale = max(np.linalg.norm(v) / np.sqrt(v.size) for v in ref.values()) if ref else 1.0
    for k in ref:
        den = max(np.linalg.norm(ref[k]), 0.5 * scale * np.sqrt(ref[k].size), 1e-300)
        worst = max(worst, float(np.linalg.norm(np.asarray(got[k]) - ref[k]) / den))
    # PMC plus faces: the cells beyond a wall are mirror images that are refreshed every step; what an update leaves in
    # them in between (e.g. the ADE history of an image cell) is implementation detail and never read — compare the inside
    sl = [slice(None)] * 3
    for a, w in enumerate(getattr(disc.spec, "mirror_plus", None) or ()):
        if w >= 0:
            sl[2 - a] = slice(0, w)
    sl = tuple(sl)
    en = np.sqrt(sum(np.linalg.norm(x[sl]) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x[sl]) ** 2 for x in o.H))
    for c in range(3):
        worst = max(worst, float(np.linalg.norm((fields[c] - o.E[c])[sl]) / en))
        worst = max(worst, float(np.linalg.norm((fields[3 + c] - o.H[c])[sl]) / hn))
    return worst, disc


def tilted_slab_waveguide(theta=0.2, dl=0.02):
    """2-D (x-z) slab waveguide (n = 2.0 core, 0.4 um, in n = 1.44) whose axis is tilted by ``theta`` about y, an angled
    ModeSource (ModeSpec.angle_theta; ref plugins/mode/solver.py:89-160 tensorial formulation) on a z-normal plane, an
    angled ModeMonitor and a flux plane 3.5 um downstream, a flux plane behind the source.  VERDICT round 2, missing 3:
    the angled launch checked in an FDTD run."""
    from tidy3d_amd.constants import C_0
    f0 = C_0 / 1.55
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 10)
    t, wx, zz = np.tan(theta), 0.4 / np.cos(theta) / 2, 10.0
    verts = [(-t * zz - wx, -zz), (-t * zz + wx, -zz), (t * zz + wx, zz), (t * zz - wx, zz)]           # (x, z)
    core = td.Structure(geometry=td.PolySlab(vertices=verts, axis=1, slab_bounds=(-1, 1)), medium=td.Medium(permittivity=4.0))
    ms = td.ModeSpec(num_modes=1, angle_theta=theta, angle_phi=0.0, target_neff=2.0)
    zsrc, zmon, plane = -2.0, 1.5, (3.0, td.inf, 0)
    return td.Simulation(
        size=(4.0, 0, 6.0), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1.6e-13, medium=td.Medium(permittivity=1.44 ** 2),
        structures=[core],
        sources=[td.ModeSource(center=(t * zsrc, 0, zsrc), size=plane, source_time=pulse, direction="+", mode_spec=ms, mode_index=0)],
        monitors=[td.ModeMonitor(center=(t * zmon, 0, zmon), size=plane, freqs=[f0], mode_spec=ms, name="mm"),
                  td.FluxMonitor(center=(t * zmon, 0, zmon), size=plane, freqs=[f0], name="fwd"),
                  td.FluxMonitor(center=(t * (zsrc - 0.4), 0, zsrc - 0.4), size=plane, freqs=[f0], name="bwd")],
        boundary_spec=td.BoundarySpec(x=td.Boundary.pml(num_layers=12), y=td.Boundary.periodic(), z=td.Boundary.pml(num_layers=12)),
        shutoff=1e-6)


def check_tilted_launch(sd):
    a = sd["mm"].amps.values
    fwd, bwd = float(sd["fwd"].flux.values[0]), float(sd["bwd"].flux.values[0])
    purity = abs(a[0, 0, 0]) ** 2 * float(sd["mm"].mode_power.values[0, 0, 0]) / fwd
    print(f"[tilted launch] fwd={fwd:.5f} bwd/fwd={bwd / fwd:.3e} purity={purity:.7f} |a-|^2={abs(a[1, 0, 0]) ** 2:.3e}")
    assert abs(1 - purity) < 1e-4                     # all but 1e-4 of the power through the far plane is in the tilted eigenmode
    assert abs(bwd) < 1e-4 * fwd                      # one-way launch: -40 dB behind the source (staircased tilted core)
    assert abs(a[1, 0, 0]) ** 2 < 1e-4 * abs(a[0, 0, 0]) ** 2
