"""Physics validation of the oracle (and of the host pipeline around it): since the reference has
no FDTD solver, field values are pinned by analytic results (SURVEY.md section 8(c) "parity
unpinned" row).  The HIP library is then held to the oracle (tests/test_emu_parity.py on CPU,
tests/test_gpu_parity.py on the GPU)."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd.constants import C_0, EPSILON_0, ETA_0, MU_0
from tidy3d_amd.data import assemble
from tidy3d_amd.discretize import discretize

from oracle.fdtd_numpy import OracleFdtd


def solve(sim, n_steps=None):
    disc = discretize(sim, n_steps=n_steps)
    o = OracleFdtd(disc.spec)
    raw = o.run()
    return assemble(disc, raw, log=""), disc, o


def test_pec_cavity_eigenfrequencies():
    """Peaks of a probe spectrum sit on the exact eigenfrequencies of the discrete Yee cavity:
    sin^2(w dt/2)/(c dt)^2 = sum_i sin^2(k_i d_i/2)/d_i^2 with k_i = m_i pi / L_i."""
    dl = 0.05
    N = (20, 16, 12)
    size = tuple(n * dl for n in N)
    pulse = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                        sources=[td.PointDipole(center=(0.13, -0.07, 0.11), source_time=pulse, polarization="Ez")],
                        monitors=[td.FieldTimeMonitor(center=(-0.21, 0.12, -0.06), size=(0, 0, 0), name="t",
                                                      fields=["Ez"], colocate=False)],
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()), shutoff=0)
    sd, disc, _ = solve(sim, n_steps=12000)
    sig = sd["t"].Ez.values.reshape(-1)
    dt = disc.spec.dt
    n = len(sig)
    spec_ = np.abs(np.fft.rfft(sig * np.hanning(n), 8 * n))
    f = np.fft.rfftfreq(8 * n, dt)
    d = [np.diff(b)[0] for b in disc.spec.boundaries]
    L = [b[-1] - b[0] for b in disc.spec.boundaries]
    modes = []
    for m in range(4):
        for q in range(4):
            for p in range(4):
                if (m > 0) + (q > 0) + (p > 0) < 2:
                    continue
                s = sum(np.sin(np.pi * mm / Li * di / 2) ** 2 / di ** 2 for mm, Li, di in zip((m, q, p), L, d))
                modes.append(2 / dt * np.arcsin(C_0 * dt * np.sqrt(s)) / (2 * np.pi))
    modes = np.array(modes)
    from scipy.signal import find_peaks
    pk, _ = find_peaks(spec_, height=0.05 * spec_.max())
    assert len(pk) >= 4
    res = 1.0 / (n * dt)
    for x in f[pk]:
        assert np.min(np.abs(modes - x)) < 1.5 * res


def test_dipole_radiated_power_matches_hertzian_formula():
    """Closed flux box around a point dipole in PML-terminated vacuum: normalised power
    = eta0 k^2 / (12 pi) for a unit current moment.  Checks source normalisation, CPML, running
    DFT, spectrum normalisation, colocation and the flux integral in absolute terms; also
    FluxMonitor == FieldMonitor(...).flux (ref monitor.py:569 docstring)."""
    lam = 1.0
    f0 = C_0 / lam
    dl = lam / 16
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 6)
    sim = td.Simulation(size=(1.2, 1.2, 1.2), grid_spec=td.GridSpec.uniform(dl=dl), run_time=60 / f0,
                        sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ez")],
                        monitors=[td.FluxMonitor(center=(0, 0, 0), size=(0.7, 0.7, 0.7),
                                                 freqs=[0.9 * f0, f0, 1.1 * f0], name="box"),
                                  td.FluxMonitor(center=(0, 0, 0.35), size=(0.7, 0.7, 0), freqs=[f0], name="top"),
                                  td.FieldMonitor(center=(0, 0, 0.35), size=(0.7, 0.7, 0), freqs=[f0], name="ftop")],
                        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=8)), shutoff=1e-5)
    sd, disc, o = solve(sim)
    assert o.stopped_at is not None            # shutoff reached well before run_time
    fl = sd["box"].flux
    k = 2 * np.pi * fl.coords["f"] / C_0
    ana = ETA_0 * k ** 2 / (12 * np.pi)
    np.testing.assert_allclose(fl.values, ana, rtol=0.02)
    np.testing.assert_allclose(sd["top"].flux.values, sd["ftop"].flux.values, rtol=1e-5)
    assert sd["box"].flux.dtype == np.float32 and sd["ftop"].Ex.dtype == np.complex64


def _sheet_sim(L, npml, structures=(), monitors=(), run_time=4e-13, dl=0.02, shutoff=0.0):
    f0 = 2e14
    pulse = td.GaussianPulse(freq0=f0, fwidth=4e13)
    return td.Simulation(size=(0, 0, L), grid_spec=td.GridSpec.uniform(dl=dl), run_time=run_time,
                         structures=list(structures),
                         sources=[td.UniformCurrentSource(center=(0, 0, -L / 2 + 0.3), size=(td.inf, td.inf, 0),
                                                          source_time=pulse, polarization="Ex")],
                         monitors=list(monitors), shutoff=shutoff,
                         boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(),
                                                       z=td.Boundary.pml(num_layers=npml)))


def test_current_sheet_plane_wave_power_and_pml_reflection():
    """A current sheet J_s radiates eta0 |J_s|^2 / 8 to each side; the default 12-layer CPML
    reflects less than -60 dB."""
    freqs = [1.8e14, 2e14, 2.2e14]
    mons = [td.FluxMonitor(center=(0, 0, 0.5), size=(td.inf, td.inf, 0), freqs=freqs, name="fwd"),
            td.FieldTimeMonitor(center=(0, 0, 0.5), size=(0, 0, 0), name="probe", fields=["Ex"], colocate=False)]
    sd, disc, _ = solve(_sheet_sim(3.0, 12, monitors=mons))
    np.testing.assert_allclose(sd["fwd"].flux.values, ETA_0 / 8, rtol=5e-3)
    # same probe, 3x longer domain: the PML reflection cannot have come back within the window
    sd2, disc2, _ = solve(_sheet_sim(9.0, 12, monitors=[
        td.FieldTimeMonitor(center=(0, 0, -3.0 + 0.5), size=(0, 0, 0), name="probe", fields=["Ex"], colocate=False)]),
        n_steps=disc.spec.n_steps)
    a = sd["probe"].Ex.values.reshape(-1)
    b = sd2["probe"].Ex.values.reshape(-1)
    refl = np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert refl < 1e-3, refl          # -60 dB


def _slab_T(eps, d, freqs):
    """Power transmission of a slab (normal incidence, vacuum both sides)."""
    n = np.sqrt(eps)
    k = 2 * np.pi * freqs / C_0 * n
    r = (1 - n) / (1 + n)
    t = (1 - r ** 2) * np.exp(1j * k * d) / (1 - r ** 2 * np.exp(2j * k * d))
    return np.abs(t) ** 2


@pytest.mark.parametrize("medium", [
    td.Medium(permittivity=4.0),
    td.Medium(permittivity=2.25, conductivity=0.05),
    td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 3.2e14, 2e13)]),
    td.Drude(eps_inf=1.0, coeffs=[(4e14, 3e13)]),
    td.Debye(eps_inf=2.0, coeffs=[(1.0, 3e-15)]),
], ids=["dielectric", "lossy", "lorentz", "drude", "debye"])
def test_slab_transmission_matches_fresnel(medium):
    """Transmission through a (dispersive) slab vs the analytic Airy formula with
    ``medium.eps_model(f)`` (ref medium.py:2900-2913): validates Ca/Cb and the ADE recursion
    (SURVEY.md section 7 step 6)."""
    freqs = np.linspace(1.6e14, 2.4e14, 9)
    d = 0.4
    mon = [td.FluxMonitor(center=(0, 0, 1.0), size=(td.inf, td.inf, 0), freqs=list(freqs), name="T")]
    slab = td.Structure(geometry=td.Box(center=(0, 0, 0.2), size=(td.inf, td.inf, d)), medium=medium)
    sd0, _, _ = solve(_sheet_sim(3.0, 12, monitors=mon, run_time=6e-13, dl=0.01))
    sd1, _, _ = solve(_sheet_sim(3.0, 12, structures=[slab], monitors=mon, run_time=6e-13, dl=0.01))
    T = sd1["T"].flux.values / sd0["T"].flux.values
    ana = _slab_T(medium.eps_model(freqs), d, freqs)
    np.testing.assert_allclose(T, ana, atol=0.02)


def test_energy_conservation_lossless_cavity():
    """Yee leapfrog conserves  eps0 E^n.E^n + mu0 H^{n+1/2}.H^{n-1/2}  exactly (to round-off) in a
    lossless PEC cavity once the source is off."""
    dl = 0.05
    pulse = td.GaussianPulse(freq0=3e14, fwidth=1.5e14)
    sim = td.Simulation(size=(12 * dl, 10 * dl, 8 * dl), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                        structures=[td.Structure(geometry=td.Box(center=(0.1, 0, 0), size=(0.2, 0.2, 0.2)),
                                                 medium=td.Medium(permittivity=3.0))],
                        sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ex")],
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()), shutoff=0)
    disc = discretize(sim, n_steps=900)
    o = OracleFdtd(disc.spec)
    end = int(np.ceil(pulse.end_time() / disc.spec.dt)) + 2
    tab = np.array([m.eps_inf for m in disc.spec.media])      # incl. the sub-pixel averaged interface media
    eps = [np.real(tab[disc.spec.mat_idx[c]]) for c in range(3)]
    vals = []
    for n in range(900):
        h_prev = [h.copy() for h in o.H]
        o._record(n, "pre")
        o.update_h(n)
        if n >= end:
            w = sum(EPSILON_0 * np.sum(e_ * E * E) for e_, E in zip(eps, o.E))
            w += sum(MU_0 * np.sum(h0 * h1) for h0, h1 in zip(h_prev, o.H))
            vals.append(w)
        o._record(n, "post")
        o.update_e(n)
        o.step_index += 1
    vals = np.array(vals)
    assert vals.min() > 0
    assert (vals.max() - vals.min()) / vals.mean() < 1e-10


def test_au_film_on_glass_matches_airy_with_johnson_christy_poles():
    """BASELINE config[4]'s stack with a continuous 40 nm Au film in place of the discs (5 pole pairs of the reference's
    material library -> ADE): R and T vs the Airy formula with ``eps_model(f)``, 0.5 % of the incident power
    (tests/test_gpu_parity.py runs the same on the GPU)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cases import gold_johnson_christy
    from tidy3d_amd.analytic import thin_film_RT
    au = gold_johnson_christy()
    dl, nxy, nz, d = 0.005, 4, 256 - 24, 0.04
    L, Lz = nxy * dl, nz * dl
    freqs = np.array([4.2e14, 4.6e14, 5.0e14, 5.4e14, 5.8e14])
    plane = (td.inf, td.inf, 0)
    sim = td.Simulation(
        size=(L, L, Lz), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1.2e-13,
        structures=[td.Structure(geometry=td.Box(center=(0, 0, -Lz / 4 - d / 2 - Lz), size=(td.inf, td.inf, Lz / 2 + 2 * Lz)),
                                 medium=td.Medium(permittivity=2.1)),
                    td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, td.inf, d)), medium=au)],
        sources=[td.PlaneWave(center=(0, 0, Lz / 2 - 0.1), size=plane, source_time=td.GaussianPulse(freq0=5e14, fwidth=1e14),
                              direction="-")],
        monitors=[td.FluxMonitor(center=(0, 0, Lz / 2 - 0.05), size=plane, freqs=list(freqs), name="R"),
                  td.FluxMonitor(center=(0, 0, -Lz / 2 + 0.1), size=plane, freqs=list(freqs), name="T")],
        boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml()), shutoff=1e-5)
    sd, _, _ = solve(sim)
    R = sd["R"].flux.values / (L * L)
    T = -sd["T"].flux.values / (L * L)
    Ra, Ta = thin_film_RT(au.eps_model(freqs), d, 2.1, freqs)
    assert np.max(np.abs(R - Ra)) < 0.005 and np.max(np.abs(T - Ta)) < 0.005, (R, Ra, T, Ta)
