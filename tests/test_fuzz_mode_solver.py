"""The cross-section eigen-solver against the LIVE reference ``compute_modes`` on random problems: strip / rib waveguides of random
size and index on random non-uniform grids, with random combinations of in-plane PML, a bend, an angle, symmetry walls (PEC / PMC
on the min edges) and lossy cores — complex effective index and the six field components of the fundamental mode.  Skipped
where the reference checkout is absent."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from tidy3d_amd.constants import C_0, ETA_0
from tidy3d_amd.mode_solver import solve_modes, solve_modes_angled

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/tidy3d"), reason="reference checkout not present")


@pytest.mark.parametrize("seed", range(24))
def test_random_cross_sections_match_the_reference_solver(seed):
    from oracle.tidy3d_ref_loader import load_mode_solver
    _, solver = load_mode_solver()
    rng = np.random.default_rng(500 + seed)
    nx, ny = int(rng.integers(26, 42)), int(rng.integers(20, 32))
    sym = (int(rng.choice([0, 0, 1, -1])), int(rng.choice([0, 0, 1, -1])))
    xb = np.concatenate(([0.0], np.cumsum(rng.uniform(0.03, 0.06, nx))))
    yb = np.concatenate(([0.0], np.cumsum(rng.uniform(0.03, 0.055, ny))))
    xb -= 0.0 if sym[0] else 0.5 * xb[-1]                 # a symmetry wall is the window's min edge: the core sits on it
    yb -= 0.0 if sym[1] else 0.5 * yb[-1]
    xc, yc = (xb[1:] + xb[:-1]) / 2, (yb[1:] + yb[:-1]) / 2
    w, h = float(rng.uniform(0.35, 0.6)), float(rng.uniform(0.18, 0.3))
    n_core = float(rng.uniform(2.0, 3.5)) + (1j * float(rng.uniform(0, 0.01)) if rng.integers(0, 4) == 0 else 0.0)
    n_clad = float(rng.uniform(1.0, 1.5))
    rib = float(rng.uniform(0, 0.1)) if rng.integers(0, 2) else 0.0

    def eps_at(x, y):
        X, Y = np.meshgrid(x, y, indexing="ij")
        e = np.full(X.shape, n_clad ** 2, complex)
        e[(np.abs(X) <= w / 2) & (np.abs(Y) <= h / 2)] = n_core ** 2
        if rib:
            e[(Y >= -h / 2) & (Y <= -h / 2 + rib)] = n_core ** 2
        return e
    exx, eyy, ezz = eps_at(xc, yb[:-1]), eps_at(xb[:-1], yc), eps_at(xb[:-1], yb[:-1])
    z = np.zeros_like(exx)
    num_pml = (int(rng.choice([0, 0, 6, 8])), int(rng.choice([0, 0, 5, 7])))
    radius = float(rng.choice([-1, 1]) * rng.uniform(4.0, 9.0)) if rng.integers(0, 3) == 0 else None
    bend_axis = int(rng.integers(0, 2))
    theta = float(rng.uniform(-0.3, 0.3)) if rng.integers(0, 3) == 0 else 0.0
    phi = float(rng.uniform(0, 2.0)) if theta else 0.0
    freq = C_0 / float(rng.uniform(1.2, 1.7))
    target = float(np.real(n_core))
    ms = SimpleNamespace(num_modes=1, bend_radius=radius, bend_axis=bend_axis, angle_theta=theta, angle_phi=phi, num_pml=num_pml,
                         target_neff=target, precision="double")
    fields, n_ref, kind = solver.compute_modes(eps_cross=[exx, z, z, z, eyy, z, z, z, ezz], coords=[xb, yb], freq=freq, mode_spec=ms,
                                               symmetry=sym, direction="+")
    pmc_min = (sym[0] == 1, sym[1] == 1)
    pml_min = (sym[0] == 0, sym[1] == 0)
    if theta:
        r = solve_modes_angled(exx, eyy, ezz, xb, yb, freq, theta, phi, num_modes=1, target_neff=target, num_pml=num_pml, pmc_min=pmc_min,
                               pml_min=pml_min, bend_radius=radius, bend_axis=bend_axis)
    else:
        r = solve_modes(exx, eyy, ezz, xb, yb, freq, num_modes=1, target_neff=target, num_pml=num_pml, pmc_min=pmc_min, pml_min=pml_min,
                        bend_radius=radius, bend_axis=bend_axis)
    desc = f"sym={sym} pml={num_pml} bend={radius} axis={bend_axis} theta={theta:.3f} phi={phi:.2f} n={n_core:.3f}/{n_clad:.2f} {kind}"
    assert abs(r.n_complex[0] - n_ref[0]) < 5e-5 * abs(n_ref[0]), (desc, r.n_complex[0], n_ref[0])
    mine = np.concatenate([getattr(r, k)[:, :, 0].ravel() for k in ("Eu", "Ev", "Ew")] + [ETA_0 * getattr(r, k)[:, :, 0].ravel() for k in ("Hu", "Hv", "Hw")])
    ref = np.concatenate([fields[0, c, :, :, 0, 0].ravel() for c in range(3)] + [ETA_0 * fields[1, c, :, :, 0, 0].ravel() for c in range(3)])
    ov = abs(np.vdot(ref, mine)) / (np.linalg.norm(ref) * np.linalg.norm(mine))
    if ov <= 1 - 1e-4:
        # two modes a few 1e-6 apart in index (PML corner modes come in such pairs) mix freely inside their pair: the reference's
        # mode then has to lie in the span of mine of the same index
        kw = dict(num_modes=3, target_neff=target, num_pml=num_pml, pmc_min=pmc_min, pml_min=pml_min, bend_radius=radius, bend_axis=bend_axis)
        r = solve_modes_angled(exx, eyy, ezz, xb, yb, freq, theta, phi, **kw) if theta else solve_modes(exx, eyy, ezz, xb, yb, freq, **kw)
        near = [m for m in range(3) if abs(r.n_complex[m] - n_ref[0]) < 2e-5 * abs(n_ref[0])]
        assert len(near) > 1, (desc, ov, r.n_complex)
        span = np.stack([np.concatenate([getattr(r, k)[:, :, m].ravel() for k in ("Eu", "Ev", "Ew")] +
                                        [ETA_0 * getattr(r, k)[:, :, m].ravel() for k in ("Hu", "Hv", "Hw")]) for m in near], axis=1)
        q, _ = np.linalg.qr(span)
        ov = np.linalg.norm(q.conj().T @ ref) / np.linalg.norm(ref)
    assert ov > 1 - 1e-4, (desc, ov)
