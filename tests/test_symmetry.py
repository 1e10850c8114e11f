"""Simulation.symmetry (ref simulation.py:67, grid_spec.py:76-82,
monitor_data.py:238-284): the solver runs on the upper half of each symmetric axis behind a PMC /
PEC wall and the data are expanded to the user's coordinates.  The discrete mirror image is exact
on the symmetric grid, so a symmetric problem must give the same monitor data with and without the
symmetry flag (fp64 oracle: to rounding; emulated HIP library: to fp32 accumulation)."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import discretize as D
from tidy3d_amd.data import assemble
from tidy3d_amd.exceptions import Tidy3dNotImplementedError
from oracle.fdtd_numpy import OracleFdtd

DL = 0.0625      # a power of two: n * DL is exact, so ceil(size / dl) is the intended cell count
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1e14)


def _sim(symmetry, n=(16, 12, 14), **kw):
    # (monitor edges avoid grid lines: the mirrored grid differs from the plain one in the last ulp,
    # which would move a span edge that sits exactly on a boundary)
    # Ez dipole at the centre: E_z is even in x and y (PMC, +1) and even in z, where it is the
    # normal component (PEC, -1)
    base = dict(size=(n[0] * DL, n[1] * DL, n[2] * DL), center=(0.1, -0.05, 0.0), grid_spec=td.GridSpec.uniform(dl=DL),
                run_time=4e-14, symmetry=symmetry,
                structures=[td.Structure(geometry=td.Sphere(center=(0.1, -0.05, 0.0), radius=0.17),
                                         medium=td.Medium(permittivity=4.0))],
                sources=[td.PointDipole(center=(0.1, -0.05, 0.0), source_time=PULSE, polarization="Ez")],
                monitors=[td.FieldMonitor(center=(0.1, -0.05, 0.1), size=(0.53, 0.42, 0), freqs=[2.5e14, 3e14], name="plane"),
                          td.FieldMonitor(center=(-0.05, -0.15, -0.1), size=(0.22, 0.13, 0.12), freqs=[3e14], name="lower",
                                          colocate=False),
                          td.FieldTimeMonitor(center=(0.0, -0.1, -0.1), size=(0, 0, 0), name="probe", interval=3),
                          td.FieldTimeMonitor(center=(0.1, -0.05, 0.0), size=(0.33, 0, 0.31), name="cross", interval=20),
                          td.FluxMonitor(center=(0.1, -0.05, 0.0), size=(0.42, 0.33, 0.31), freqs=[2.5e14, 3e14], name="box"),
                          td.FluxMonitor(center=(0.1, -0.05, -0.2), size=(td.inf, td.inf, 0), freqs=[3e14], name="below",
                                         normal_dir="-"),
                          td.PermittivityMonitor(center=(0.1, -0.05, 0), size=(0.52, 0.43, 0), freqs=[3e14], name="eps")],
                boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=4)))
    base.update(kw)
    return td.Simulation(**base)


def _run_oracle(sim, n_steps):
    disc = D.discretize(sim, n_steps=n_steps)
    raw = OracleFdtd(disc.spec).run()
    return disc, assemble(disc, raw)


def _compare(a, b, rtol):
    for da, db in zip(a.data, b.data):
        assert da.monitor.name == db.monitor.name
        comps = getattr(da, "field_components", None) or {"flux": da.flux}
        compsb = getattr(db, "field_components", None) or {"flux": db.flux}
        for k in comps:
            va, vb = np.asarray(comps[k].values), np.asarray(compsb[k].values)
            assert va.shape == vb.shape, (da.monitor.name, k, va.shape, vb.shape)
            for dim in comps[k].dims:
                np.testing.assert_allclose(comps[k].coords[dim], compsb[k].coords[dim], atol=1e-9)
            # components that vanish by symmetry hold rounding noise: scale by the field kind (E / H)
            kind = [np.abs(np.asarray(v.values)).max() for n, v in comps.items() if n[0] == k[0]]
            scale = max(max(kind), 1e-300)
            assert np.abs(va - vb).max() <= rtol * scale, (da.monitor.name, k, np.abs(va - vb).max() / scale)


def test_symmetric_grid_rule():
    """odd cell count: the boundary nearest to the centre moves onto it and the upper half is mirrored
    (ref grid_spec.py:76-82) -> an even cell count, symmetric about the centre"""
    sim = _sim((1, 0, 0), n=(15, 12, 14), boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    b = D.make_boundaries(sim)[0]
    c = sim.center[0]
    assert len(b) - 1 in (14, 16)       # the centre is equidistant from two boundaries: rounding decides
    np.testing.assert_allclose(b + b[::-1], 2 * c, atol=1e-12)
    assert np.any(np.isclose(b, c))


@pytest.mark.parametrize("symmetry", [(1, 0, 0), (0, 1, -1), (1, 1, -1)])
def test_symmetric_run_equals_full_run(symmetry):
    full_disc, full = _run_oracle(_sim((0, 0, 0)), 90)
    half_disc, half = _run_oracle(_sim(symmetry), 90)
    n_full = np.prod(full_disc.spec.shape)
    n_half = np.prod(half_disc.spec.shape)
    assert n_half * 2 ** sum(1 for s in symmetry if s) == pytest.approx(n_full, rel=0.35)
    assert n_half < n_full
    _compare(full, half, 3e-6)      # containers are fp32 / complex64


def test_symmetry_through_the_library(emu_lib):
    from tidy3d_amd.web import run
    full = run(_sim((0, 0, 0)), task_name="full", verbose=False, lib=emu_lib, n_steps=60)
    half = run(_sim((1, 1, -1)), task_name="sym", verbose=False, lib=emu_lib, n_steps=60)
    _compare(full, half, 2e-4)


def test_unsupported_combinations_are_named():
    with pytest.raises(Tidy3dNotImplementedError, match="TFSF"):
        D.discretize(_sim((1, 0, 0), sources=[td.TFSF(center=(0.1, -0.05, 0), size=(0.3, 0.3, 0.3), source_time=PULSE,
                                                       injection_axis=2, direction="+")]), n_steps=4)


def _waveguide(symmetry, n_steps_unused=None):
    """Si strip in oxide, TE0 launched along +x.  E_y of TE0 is even in y, where it is the normal
    component (PEC plane, -1), and even in z (PMC plane, +1)."""
    dl = 0.0625
    pulse = td.GaussianPulse(freq0=2e14, fwidth=3e13)
    ms = td.ModeSpec(num_modes=1, target_neff=2.85, precision="double")      # TE0 in both runs
    plane = (0, 1.3, 1.1)
    return td.Simulation(
        size=(32 * dl, 28 * dl, 24 * dl), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-13, symmetry=symmetry,
        medium=td.Medium(permittivity=1.44 ** 2),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, 0.5, 0.25)),
                                 medium=td.Medium(permittivity=3.48 ** 2))],
        sources=[td.ModeSource(center=(-0.5, 0, 0), size=plane, source_time=pulse, mode_spec=ms, mode_index=0,
                               direction="+")],
        monitors=[td.ModeMonitor(center=(0.45, 0, 0), size=plane, freqs=[1.9e14, 2e14], mode_spec=ms, name="modes"),
                  td.FluxMonitor(center=(0.45, 0, 0), size=plane, freqs=[2e14], name="flux"),
                  td.FieldMonitor(center=(0.2, 0, 0), size=(0, 0.83, 0.57), freqs=[2e14], name="xs")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=4)))


def test_mode_source_and_monitor_with_symmetry():
    """ModeSource + ModeMonitor on a quarter domain: the mode plane is solved behind the PEC / PMC
    walls (== reference compute_modes(symmetry=...), tests/test_mode_solver.py), launched with 1 W
    over the WHOLE plane, and decomposed on the computed quarter — same amplitudes, effective
    indices, flux and fields as the full run."""
    _, full = _run_oracle(_waveguide((0, 0, 0)), 620)
    disc, quarter = _run_oracle(_waveguide((0, -1, 1)), 620)
    assert np.prod(disc.spec.shape) < 0.4 * 40 * 36 * 32
    np.testing.assert_allclose(quarter["modes"].n_complex.values, full["modes"].n_complex.values, rtol=1e-6)
    a_f, a_q = full["modes"].amps.values, quarter["modes"].amps.values
    # mode 0 (TE0, the launched one): equal amplitudes; the sign of an eigenvector is arbitrary
    assert np.abs(a_f[0, :, 0]).max() > 0.1
    np.testing.assert_allclose(np.abs(a_q[:, :, 0]), np.abs(a_f[:, :, 0]), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(quarter["flux"].flux.values, full["flux"].flux.values, rtol=2e-4)
    ez_f, ez_q = full["xs"].Ey.values, quarter["xs"].Ey.values
    assert np.abs(ez_f - ez_q).max() <= 2e-4 * np.abs(ez_f).max()


def test_symmetry_with_absorber_and_bloch_boundaries():
    """Absorber layers sit behind the plus faces of the computed half; a Bloch axis (no symmetry along it,
    ref simulation.py bloch_with_symmetry) keeps its phase."""
    ab = td.BoundarySpec(x=td.Boundary.absorber(num_layers=5, parameters=td.AbsorberParams(sigma_max=1.5)),
                         y=td.Boundary.pml(num_layers=4), z=td.Boundary.absorber(num_layers=4))
    _, full = _run_oracle(_sim((0, 0, 0), boundary_spec=ab), 80)
    _, half = _run_oracle(_sim((1, 1, -1), boundary_spec=ab), 80)
    _compare(full, half, 3e-6)
    bl = td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.bloch(0.23), z=td.Boundary.pml(num_layers=4))
    mons = [m for m in _sim((0, 0, 0)).monitors if m.name in ("plane", "box", "below")]
    dfull, full = _run_oracle(_sim((0, 0, 0), boundary_spec=bl, monitors=mons), 80)
    dhalf, half = _run_oracle(_sim((1, 0, -1), boundary_spec=bl, monitors=mons), 80)
    assert dhalf.spec.bloch == pytest.approx(dfull.spec.bloch) and dhalf.spec.bloch[1] != 0
    _compare(full, half, 3e-6)
    from tidy3d_amd.exceptions import SetupError
    with pytest.raises(SetupError, match="symmetry along the same axis"):
        D.discretize(_sim((0, 1, 0), boundary_spec=bl, monitors=mons), n_steps=2)


def test_symmetric_run_keeps_the_users_run_time_and_dft_stride():
    """RunTimeSpec measures the LONGEST side of the user's simulation and the DFT stride follows the monitors'
    frequencies too (ref simulation.py:3677-3711, :4414-4443) — also when the solver only computes one half."""
    from tidy3d_amd.discretize import discretize
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
    kw = dict(size=(4.0, 1.0, 1.0), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=td.RunTimeSpec(quality_factor=3.0),
              sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ez")],
              monitors=[td.FieldMonitor(center=(0.5, 0, 0), size=(1, 1, 0), freqs=[6e14], name="f")],
              boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=4)))
    full = discretize(td.Simulation(**kw))
    half = discretize(td.Simulation(symmetry=(-1, 0, 0), **kw))          # x is the longest AND the symmetric axis
    assert half.spec.n_steps == full.spec.n_steps
    assert half.nyquist_step == full.nyquist_step
    assert np.array_equal(half.spec.monitors[0].steps, full.spec.monitors[0].steps)


@pytest.mark.parametrize("symmetry", [(-1, 0, 0), (0, 1, 0), (-1, 1, 0)])
def test_symmetric_unit_cell_of_a_periodic_array(symmetry):
    """Symmetry on PERIODIC axes (the unit cell of a metasurface, symmetry = (1, -1, 0) style): mirror + translation put a
    second symmetry plane on the cell's ends, so the half cell is closed by the same kind of wall on its plus face — PEC is
    the plain truncation, PMC the mirror-image layout of discretize._discretize_pmc_plus.  x-polarised plane wave onto a
    dielectric disc: E_x is normal to the x plane (-1) and tangential to the y plane (+1).  Fields and flux equal the full
    periodic run to rounding."""
    pulse = td.GaussianPulse(freq0=3e14, fwidth=1e14)

    def sim(sym):
        return td.Simulation(
            size=(16 * DL, 12 * DL, 32 * DL), grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-13, symmetry=sym, shutoff=0,
            structures=[td.Structure(geometry=td.Cylinder(center=(0, 0, 0), radius=0.3, length=0.25, axis=2),
                                     medium=td.Medium(permittivity=6.0))],
            sources=[td.PlaneWave(center=(0, 0, 0.6), size=(td.inf, td.inf, 0), source_time=pulse, direction="-", pol_angle=0.0)],
            monitors=[td.FieldMonitor(center=(0, 0, -0.3), size=(td.inf, td.inf, 0), freqs=[2.5e14, 3e14], name="f"),
                      td.FluxMonitor(center=(0, 0, -0.6), size=(td.inf, td.inf, 0), freqs=[3e14], name="T"),
                      td.FieldTimeMonitor(center=(0.2, 0.1, -0.3), size=(0, 0, 0), name="p", interval=3)],
            boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml(num_layers=6)))
    _, full = _run_oracle(sim((0, 0, 0)), 240)
    disc, half = _run_oracle(sim(symmetry), 240)
    assert disc.spec.shape[0] == (8 if symmetry[0] else 16)
    assert (disc.spec.mirror_plus is not None and disc.spec.mirror_plus[1] == 6) == (symmetry[1] == 1)
    for c in ("Ex", "Ey", "Ez", "Hx", "Hy", "Hz"):
        a, b = np.asarray(full["f"][c].values), np.asarray(half["f"][c].values)
        scale = max(np.abs(np.asarray(full["f"][k].values)).max() for k in (("Ex", "Ey", "Ez") if c[0] == "E" else ("Hx", "Hy", "Hz")))
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-9 * scale, c
    assert float(half["T"].flux.values[0]) == pytest.approx(float(full["T"].flux.values[0]), rel=1e-9)
    np.testing.assert_allclose(np.asarray(half["p"].Ex.values), np.asarray(full["p"].Ex.values), rtol=0,
                               atol=1e-9 * np.abs(np.asarray(full["p"].Ex.values)).max())
