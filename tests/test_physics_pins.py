"""Physics pins that need neither reference data nor the oracle's agreement with the kernels: what
`discretize.py` hands BOTH the HIP engine and the fp64 oracle — dipole weights, TFSF tables, the source-spectrum normalisation,
flux areas, colocation — is held to closed forms here.

  * a `PointDipole` in vacuum: the power through a closed flux box, normalised as every result is, equals the Hertzian dipole's
    eta0 k^2 / (12 pi) per unit current moment (ref source.py:600-607: "an infinitesimal antenna with a fixed current density") —
    to the grid's O((k dl)^2): 0.1 - 0.25 % at 12 points per wavelength, falling with dl;
  * a `TFSF` box (normal incidence) injects |E0|^2 = 2 / (c eps0) = 2 eta0, "1 W / um^2 for any source size" (ref source.py:1210-
    1214): at the Yee nodes inside an empty box |E| = sqrt(2 eta0) and |H| = sqrt(2 / eta0) to 1e-4 (the discrete plane wave along
    an axis has impedance eta0 exactly); the COLOCATED flux through a grid plane inside is cos(k~ dl / 2) W / um^2 — H is averaged
    over the two half-cells next to the plane — with k~ from the 1-D dispersion relation, to 1e-4; nothing leaves the box (flux through
    a closed surface outside it: 0) and nothing stays in it (closed surface inside: < 2e-4 of the incident power — its faces lie
    between grid planes, where the two interpolations of a travelling wave do not cancel to the last digit);
  * a lossless sphere in the box: the power scattered out (closed box outside the TFSF surface) equals the power taken from the
    total field (closed box inside, around the sphere: net inflow 0 for a lossless body) — closure to 1 % of the scattered power
    at 16 points per wavelength, second order in dl.

CPU: the fp64 oracle at small sizes.  GPU (`-m gpu`): the HIP engine at two resolutions each — the convergence is asserted."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd.constants import C_0, ETA_0
from tidy3d_amd.data import assemble
from tidy3d_amd.discretize import discretize

F0 = 2e14
FREQS = [0.9 * F0, F0, 1.1 * F0]


def _solve(sim, hip_lib=None):
    disc = discretize(sim)
    if hip_lib is None:
        from oracle.fdtd_numpy import OracleFdtd
        raw = OracleFdtd(disc.spec).run()
    else:
        from tidy3d_amd.engine import HipEngine
        with HipEngine(disc.spec, lib=hip_lib) as e:
            st = e.run()
            assert not st.diverged
            raw = e.results()
    return disc, assemble(disc, raw, log="")


def dipole_power_error(ppw, n, npml=8, hip_lib=None):
    dl = C_0 / F0 / ppw
    L = n * dl
    sim = td.Simulation(size=(L, L, L), grid_spec=td.GridSpec.uniform(dl=dl), run_time=40 / F0,
                        sources=[td.PointDipole(center=(0, 0, 0), source_time=td.GaussianPulse(freq0=F0, fwidth=F0 / 6), polarization="Ez")],
                        monitors=[td.FluxMonitor(center=(0, 0, 0), size=(0.6 * L,) * 3, freqs=FREQS, name="box")],
                        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=npml)), shutoff=1e-7)
    _, sd = _solve(sim, hip_lib)
    k = 2 * np.pi * np.array(FREQS) / C_0
    return np.asarray(sd["box"].flux.values) / (ETA_0 * k ** 2 / (12 * np.pi)) - 1


def tfsf_box(ppw, n, npml=6, sphere=False, hip_lib=None):
    dl = C_0 / F0 / ppw
    L = n * dl
    b = 0.5 * L
    structures = [td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=0.12 * L), medium=td.Medium(permittivity=4.0))] if sphere else []
    sim = td.Simulation(size=(L, L, L), grid_spec=td.GridSpec.uniform(dl=dl), run_time=60 / F0, structures=structures,
                        sources=[td.TFSF(center=(0, 0, 0), size=(b, b, b), source_time=td.GaussianPulse(freq0=F0, fwidth=F0 / 6), injection_axis=2,
                                         direction="+", pol_angle=0.3)],
                        monitors=[td.FieldMonitor(center=(0.03 * L, 0.02 * L, 0.05 * L), size=(0, 0, 3 * dl), freqs=FREQS, name="nodes", colocate=False),
                                  td.FluxMonitor(center=(0, 0, 4 * dl), size=(0.3 * L, 0.3 * L, 0), freqs=FREQS, name="plane"),      # (on a grid plane: n is even)
                                  td.FluxMonitor(center=(0, 0, 0), size=(0.36 * L,) * 3, freqs=FREQS, name="inner"),
                                  td.FluxMonitor(center=(0, 0, 0), size=(0.7 * L,) * 3, freqs=FREQS, name="outer")],
                        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=npml)), shutoff=1e-8)
    disc, sd = _solve(sim, hip_lib)
    out = {"dl": dl, "dt": disc.spec.dt, "plane": np.asarray(sd["plane"].flux.values) / (0.3 * L) ** 2,
           "inner": np.asarray(sd["inner"].flux.values) / (0.36 * L) ** 2, "outer": np.asarray(sd["outer"].flux.values) / (0.36 * L) ** 2}
    if not sphere:
        out["E2"] = sum(np.abs(np.asarray(getattr(sd["nodes"], c).values)) ** 2 for c in ("Ex", "Ey", "Ez")).reshape(-1, len(FREQS))
        out["H2"] = sum(np.abs(np.asarray(getattr(sd["nodes"], c).values)) ** 2 for c in ("Hx", "Hy", "Hz")).reshape(-1, len(FREQS))
    return out


def colocation_factor(dl, dt):
    """cos(k~ dl / 2), k~ from the dispersion relation of the Yee scheme along an axis: sin(k~ dl / 2) / dl = sin(w dt / 2) / (c dt)"""
    w = 2 * np.pi * np.array(FREQS)
    return np.cos(np.arcsin(np.clip(dl / (C_0 * dt) * np.sin(w * dt / 2), -1, 1)))


def check_empty_box(r, tol=1e-4):
    assert np.max(np.abs(r["E2"] / (2 * ETA_0) - 1)) < tol, r["E2"] / (2 * ETA_0) - 1           # |E0|^2 = 2 eta0 at every node, every frequency
    assert np.max(np.abs(r["H2"] * ETA_0 / 2 - 1)) < tol, r["H2"] * ETA_0 / 2 - 1                 # impedance eta0
    assert np.max(np.abs(r["plane"] - colocation_factor(r["dl"], r["dt"]))) < tol, (r["plane"], colocation_factor(r["dl"], r["dt"]))
    assert np.max(np.abs(r["outer"])) < 1e-6 and np.max(np.abs(r["inner"])) < 2e-4, (r["outer"], r["inner"])


def test_dipole_radiates_the_hertzian_power_oracle():
    err = dipole_power_error(12, 40)
    assert np.max(np.abs(err)) < 3e-3, err


def test_tfsf_box_injects_one_watt_per_square_micron_oracle():
    check_empty_box(tfsf_box(12, 36))


def test_flux_closure_on_a_lossless_sphere_oracle():
    r = tfsf_box(12, 40, sphere=True)
    assert np.min(r["outer"]) > 0.2                                     # it scatters
    assert np.max(np.abs(r["inner"]) / r["outer"]) < 2e-2, r["inner"] / r["outer"]


@pytest.mark.gpu
def test_dipole_radiates_the_hertzian_power_gpu(hip_lib):
    e1, e2 = dipole_power_error(16, 96, 12, hip_lib), dipole_power_error(32, 192, 12, hip_lib)
    print(f"\n[dipole power / Hertzian - 1] 16 points per wavelength {e1}, 32: {e2}")
    # measured -0.40 ... -0.58 % and -0.057 ... -0.083 %: second order in dl (x 7 per halving)
    assert np.max(np.abs(e1)) < 8e-3 and np.max(np.abs(e2)) < 1.5e-3 and np.max(np.abs(e1)) > 3 * np.max(np.abs(e2)), (e1, e2)


@pytest.mark.gpu
def test_tfsf_box_injects_one_watt_per_square_micron_gpu(hip_lib):
    for ppw, n in ((16, 96), (32, 160)):
        check_empty_box(tfsf_box(ppw, n, 10, hip_lib=hip_lib), tol=5e-4)        # (fp32 fields and DFT sums over ~10^4 steps: measured 2e-4)


@pytest.mark.gpu
def test_flux_closure_on_a_lossless_sphere_gpu(hip_lib):
    # (the oracle test's problem — a sphere of 0.4 wavelengths radius in a box of 3.3 — at 1.5 x and 3 x its resolution)
    r1, r2 = tfsf_box(18, 60, 10, True, hip_lib), tfsf_box(36, 120, 10, True, hip_lib)
    c1, c2 = np.max(np.abs(r1["inner"]) / r1["outer"]), np.max(np.abs(r2["inner"]) / r2["outer"])
    print(f"\n[lossless sphere: |net flux through the inner box| / scattered power] 18 points per wavelength {c1:.2e}, 36: {c2:.2e}")
    assert c1 < 1.5e-2 and c2 < 5e-3, (c1, c2)
