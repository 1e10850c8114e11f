"""Full-vectorial finite-difference eigenmode solver on the 2-D Yee cross-section.

Purpose: the mode profile a ``ModeSource`` injects (ref source.py:993-1085) and the basis of
``ModeMonitor`` decompositions.  When the real tidy3d package is importable its own CPU
``ModeSolver`` stays the reference ("left untouched", north_star); this module is the product-side
implementation for hosts without tidy3d (the GPU box) and is held to the reference's
``EigSolver.compute_modes`` (ref plugins/mode/solver.py:33-269) by golden n_eff / field fixtures
(tests/golden/mode_golden.json, tests/test_mode_solver.py) — it is an independent derivation, not
a copy.

Formulation (e^{-i w t}, fields ~ e^{+i beta w}, w = propagation axis, (u, v, w) right-handed,
Ht = eta0 H):  eliminating E_w and H_w from the six curl equations on the Yee cell gives

    beta E_u =  k0 Ht_v + (1/k0) Duf [ (1/eps_w) (Dub Ht_v - Dvb Ht_u) ]
    beta E_v = -k0 Ht_u + (1/k0) Dvf [ (1/eps_w) (Dub Ht_v - Dvb Ht_u) ]
    beta Ht_u = -k0 eps_v E_v - (1/k0) Dub (Duf E_v - Dvf E_u)
    beta Ht_v =  k0 eps_u E_u - (1/k0) Dvb (Duf E_v - Dvf E_u)

i.e. beta [E] = P [Ht], beta [Ht] = Q [E]  =>  (P Q) [E] = beta^2 [E], with forward differences
(primal steps) acting on E and backward differences (dual steps) acting on Ht, exactly the
staggering of ref plugins/mode/derivatives.py:9-76.  Node layout in the plane (ref
grid/grid.py:465-491): E_u at (uc, vb), E_v at (ub, vc), E_w at (ub, vb), H_u at (ub, vc),
H_v at (uc, vb), H_w at (uc, vc).  The outer edge is PEC (min edge: tangential E on the wall is
zero; max edge: truncation), like the reference's default.  Options held to the reference by direct
comparison (tests/test_mode_solver.py): PMC walls on the min edges (symmetry eigenvalue +1, ref
solver.py:182-197) and the stretched-coordinate PML inside the plane (``ModeSpec.num_pml``, ref
derivatives.py:80-232).

Angled waveguides (``ModeSpec.angle_theta / angle_phi``, ref solver.py:89-160, transforms.py:74-111): the
shear u' = u - tan(theta) cos(phi) w, v' = v - tan(theta) sin(phi) w makes the structure invariant along w;
with J = [[1, 0, a], [0, 1, b], [0, 0, 1]] (a, b = -tan(theta) cos / sin(phi)) the media become the full tensors
eps' = J eps J^T, mu' = J J^T (det J = 1) and the problem no longer reduces to E alone.  With E_w and Ht_w
eliminated from the six curl equations (derivatives in units of k0, n = beta / k0):

    E_w  = (1/eps_ww) [  i (Dub Ht_v - Dvb Ht_u) - eps_wu E_u - eps_wv E_v ]
    Ht_w = (1/mu_ww)  [ -i (Duf E_v - Dvf E_u)   - mu_wu Ht_u - mu_wv Ht_v ]
    i n E_u  = Duf E_w  + i (mu_vu Ht_u + mu_vv Ht_v + mu_vw Ht_w)
    i n E_v  = Dvf E_w  - i (mu_uu Ht_u + mu_uv Ht_v + mu_uw Ht_w)
    i n Ht_u = Dub Ht_w - i (eps_vu E_u + eps_vv E_v + eps_vw E_w)
    i n Ht_v = Dvb Ht_w + i (eps_uu E_u + eps_uv E_v + eps_uw E_w)

a first-order 4N x 4N eigenproblem in [E_u, E_v, Ht_u, Ht_v] (``solve_modes_angled``); tensor entries multiply
pointwise on the flattened Yee arrays (no averaging between staggered nodes — the reference's choice too, which
is what makes the two comparable to 1e-7).  Back in the physical frame F = J^T F' and n_eff = n' cos(theta).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from .constants import C_0, EPSILON_0, ETA_0


@dataclass
class ModeResult:
    n_complex: np.ndarray          # [M]
    # fields[name] complex [Nu, Nv, M] on the component's own Yee nodes, e^{+i beta w} convention,
    # physical H (A/um per V/um); normalised to unit power flux along +w
    Eu: np.ndarray
    Ev: np.ndarray
    Ew: np.ndarray
    Hu: np.ndarray
    Hv: np.ndarray
    Hw: np.ndarray


def _pml_s(direction: str, omega: float, d: np.ndarray, n: int, n_pml: int, at_min: bool, speed) -> np.ndarray:
    """Coordinate-stretching factors s = kappa + i sigma / (omega eps0) of the mode-plane PML
    (ref plugins/mode/derivatives.py:166-232): cubic profiles, kappa 1 -> 3, sigma_max = 2 in units
    of avg_speed / (eta0 dl); 'f' factors sit half a cell off the 'b' factors."""
    s = np.ones(n, dtype=complex)
    if n_pml == 0:
        return s

    def val(dl, step, sp_):
        kappa = 1.0 + 2.0 * step ** 3
        sigma = 2.0 * sp_ / (ETA_0 * dl) * step ** 3
        return kappa + 1j * sigma / (omega * EPSILON_0)
    for i in range(n):
        if direction == "f":
            if i <= n_pml - 1 and at_min:
                s[i] = val(d[0], (n_pml - i - 0.5) / n_pml, speed[0])
            elif i >= n - n_pml:
                s[i] = val(d[-1], (i - (n - n_pml) + 0.5) / n_pml, speed[1])
        else:
            if i < n_pml and at_min:
                s[i] = val(d[0], (n_pml - i) / n_pml, speed[0])
            elif i > n - n_pml:
                s[i] = val(d[-1], (i - (n - n_pml)) / n_pml, speed[1])
    return s


def _diff_ops(nu: int, nv: int, du_p, dv_p, du_d, dv_d, pmc_min=(False, False), s_fac=None, wall_in_fwd: bool = False):
    """Sparse forward (on primal steps) and backward (on dual steps) difference operators for
    fields flattened as index = iu * nv + iv.  Min edge of each axis: PEC wall, or a PMC wall
    (``pmc_min``: the symmetry plane of an even mode, ref derivatives.py:9-62 ``dmin_pmc``)."""
    if s_fac is not None:           # PML: derivatives divided by the stretching factors
        du_p, du_d, dv_p, dv_d = du_p * s_fac[0], du_d * s_fac[1], dv_p * s_fac[2], dv_d * s_fac[3]

    def fwd(n, d, pmc=True):
        main = -np.ones(n)
        # ``wall_in_fwd`` (the tensorial problem, which has no clamped rows): a PEC min wall enters as in ref derivatives.py:9-19 —
        # the wall's own node does not contribute to the first difference
        if wall_in_fwd and not pmc:
            main[0] = 0.0
        D = sp.diags([main, np.ones(n - 1)], [0, 1], shape=(n, n), format="csr")
        return sp.diags(1.0 / d) @ D            # last row: (0 - f[n-1]) -> truncation = PEC
    def bwd(n, d, pmc):
        main = np.ones(n)
        # row 0 lives on the wall.  PEC: it only feeds wall-tangential E (clamped to zero) and
        # wall-normal H (zero on a PEC) -> 0.  PMC: the H node below the wall is minus the one above
        # it -> H[0] - (-H[0]) = 2 H[0]   (ref derivatives.py:24-35)
        main[0] = 2.0 if pmc else 0.0
        D = sp.diags([main, -np.ones(n - 1)], [0, -1], shape=(n, n), format="csr")
        return sp.diags(1.0 / d) @ D
    Iu, Iv = sp.identity(nu, format="csr"), sp.identity(nv, format="csr")
    Duf = sp.kron(fwd(nu, du_p, pmc_min[0]), Iv, format="csr")
    Dvf = sp.kron(Iu, fwd(nv, dv_p, pmc_min[1]), format="csr")
    Dub = sp.kron(bwd(nu, du_d, pmc_min[0]), Iv, format="csr")
    Dvb = sp.kron(Iu, bwd(nv, dv_d, pmc_min[1]), format="csr")
    return Duf, Dvf, Dub, Dvb


def solve_modes(eps_u: np.ndarray, eps_v: np.ndarray, eps_w: np.ndarray, ub: np.ndarray,
                vb: np.ndarray, freq: float, num_modes: int = 1,
                target_neff: Optional[float] = None, precision: str = "double",
                pmc_min: Tuple[bool, bool] = (False, False), num_pml: Tuple[int, int] = (0, 0),
                pml_min: Tuple[bool, bool] = (True, True), bend_radius: Optional[float] = None,
                bend_axis: int = 0) -> ModeResult:
    """eps_* are [Nu, Nv] (complex allowed) sampled at E_u (uc, vb), E_v (ub, vc), E_w (ub, vb);
    ub / vb the Nu+1 / Nv+1 cell boundaries.  ``pmc_min``: PMC instead of PEC on the min edge of
    u / v (a symmetry plane with eigenvalue +1; PEC covers -1 and the default truncation).
    ``num_pml``: PML cells inside the plane along u / v (ref ModeSpec.num_pml, mode.py); ``pml_min``:
    False on an axis whose min edge is a symmetry plane (ref solver.py:196-198).
    ``bend_radius`` / ``bend_axis`` (0 = u, 1 = v: the in-plane axis normal to the bend plane): the
    conformal straightening of ref transforms.py:14-75 — w = R phi, J = diag(1, 1, R / r) — turns the
    bend into diagonal eps' = J eps J^T / det J and mu' = J J^T / det J; n_eff refers to k_w = R k_phi."""
    nu, nv = eps_u.shape
    N = nu * nv
    # lossless cross-sections give a real operator (half the LU cost); "single" follows
    # ModeSpec.precision (ref mode.py:164, solver.py:247: the reference's default) and runs the
    # factorisation and ARPACK in 32-bit
    n_guess = float(np.sqrt(np.max(np.abs([np.max(np.abs(a)) for a in (eps_u, eps_v, eps_w)]))))
    mu_u = mu_v = mu_w = None
    if bend_radius is not None:
        na = 0 if bend_axis == 1 else 1                      # axis along which the radius varies
        c = (ub, vb)[na]
        r = c + (bend_radius - c[(len(c) - 1) // 2])           # plane centre at the bend radius
        s_e = bend_radius / r[:-1]                             # dw/dz at the E nodes ...
        s_h = bend_radius / (r[:-1] + r[1:]) * 2               # ... and at the H nodes (ref :55-56)
        shp = (-1, 1) if na == 0 else (1, -1)
        s_e, s_h = s_e.reshape(shp), s_h.reshape(shp)
        eps_u, eps_v, eps_w = eps_u / s_e, eps_v / s_e, eps_w * s_e
        ones = np.ones((nu, nv))
        mu_u, mu_v, mu_w = ones / s_h, ones / s_h, ones * s_h
    has_pml = any(int(v) > 0 for v in num_pml)
    is_real = all(np.all(np.imag(a) == 0) for a in (eps_u, eps_v, eps_w)) and not has_pml
    if is_real:
        eps_u, eps_v, eps_w = (np.real(a) for a in (eps_u, eps_v, eps_w))
    k0 = 2 * np.pi * freq / C_0
    du_p, dv_p = np.diff(ub), np.diff(vb)
    du_d = np.concatenate(([du_p[0]], 0.5 * (du_p[1:] + du_p[:-1])))
    dv_d = np.concatenate(([dv_p[0]], 0.5 * (dv_p[1:] + dv_p[:-1])))
    s_fac = None
    if has_pml:
        pu, pv = int(num_pml[0]), int(num_pml[1])
        diag = np.stack([np.asarray(eps_u), np.asarray(eps_v), np.asarray(eps_w)])       # (3, Nu, Nv)
        mdiag = np.ones_like(diag) if mu_u is None else np.stack([mu_u, mu_v, mu_w])

        def mean(a):
            return 1.0 if a.size == 0 else np.mean(a)
        # relative wave speed in the four PML regions (ref derivatives.py:131-160)
        regions = [np.s_[:, :pu, :], np.s_[:, nu - pu + 1:, :], np.s_[:, :, :pv], np.s_[:, :, nv - pv + 1:]]
        speed = [1 / np.sqrt(mean(diag[r_]) * mean(mdiag[r_])) for r_ in regions]
        omega = 2 * np.pi * freq
        s_fac = (_pml_s("f", omega, du_p, nu, pu, pml_min[0], speed[:2]),
                 _pml_s("b", omega, du_d, nu, pu, pml_min[0], speed[:2]),
                 _pml_s("f", omega, dv_p, nv, pv, pml_min[1], speed[2:]),
                 _pml_s("b", omega, dv_d, nv, pv, pml_min[1], speed[2:]))
    Duf, Dvf, Dub, Dvb = _diff_ops(nu, nv, du_p, dv_p, du_d, dv_d, pmc_min, s_fac)
    eu, ev, ew = (sp.diags(np.asarray(a).reshape(-1)) for a in (eps_u, eps_v, eps_w))
    ewi = sp.diags(1.0 / np.asarray(eps_w).reshape(-1))
    # PEC on the min edges: the tangential E that sits on a wall is clamped to zero
    mask_u = np.ones((nu, nv))
    mask_v = np.ones((nu, nv))
    mask_w = np.ones((nu, nv))
    if not pmc_min[1]:
        mask_u[:, 0] = 0        # E_u at vb[0]
        mask_w[:, 0] = 0
    if not pmc_min[0]:
        mask_v[0, :] = 0        # E_v at ub[0]
        mask_w[0, :] = 0
    Mu, Mv, Mw = (sp.diags(m.reshape(-1)) for m in (mask_u, mask_v, mask_w))
    I = sp.identity(N, format="csr")
    # E_w = (i/(k0 eps_w)) (Dub Ht_v - Dvb Ht_u)   -> rows of P
    curl_h = sp.hstack([-Dvb, Dub], format="csr")                       # acts on [Ht_u; Ht_v]
    Ew_op = Mw @ ewi @ curl_h                                           # (without the i/k0 factor)
    # diagonal mu (bends):  beta E_u = k0 mu_v Ht_v + ...,  beta E_v = -k0 mu_u Ht_u + ...,
    #                       Ht_w = (-i / (k0 mu_w)) (Duf E_v - Dvf E_u)
    Iu_m = I if mu_u is None else sp.diags(mu_u.reshape(-1))
    Iv_m = I if mu_v is None else sp.diags(mu_v.reshape(-1))
    mwi = I if mu_w is None else sp.diags(1.0 / mu_w.reshape(-1))
    P = sp.vstack([
        Mu @ (sp.hstack([sp.csr_matrix((N, N)), k0 * Iv_m]) + (1 / k0) * Duf @ Ew_op),
        Mv @ (sp.hstack([-k0 * Iu_m, sp.csr_matrix((N, N))]) + (1 / k0) * Dvf @ Ew_op),
    ], format="csr")
    curl_e = mwi @ sp.hstack([-Dvf, Duf], format="csr")                 # (1/mu_w)(Duf E_v - Dvf E_u) on [E_u; E_v]
    Q = sp.vstack([
        sp.hstack([sp.csr_matrix((N, N)), -k0 * ev]) - (1 / k0) * Dub @ curl_e,
        sp.hstack([k0 * eu, sp.csr_matrix((N, N))]) - (1 / k0) * Dvb @ curl_e,
    ], format="csr")
    A = (P @ Q).tocsc()
    if precision == "single":
        A = A.astype(np.float32 if is_real else np.complex64)
    if target_neff is None:
        target_neff = n_guess                       # from the physical eps (ref solver.py:202-207)
    sigma = (target_neff * k0) ** 2
    rng = np.random.default_rng(0)
    v0 = rng.standard_normal(2 * N).astype(A.dtype if not np.iscomplexobj(A) else np.float64)
    vals, vecs = spl.eigs(A, k=num_modes, sigma=A.dtype.type(sigma) if precision == "single" else sigma,
                          v0=v0.astype(A.dtype), tol=(1e-6 if precision == "single" else 1e-10))
    vals = vals.astype(complex)
    vecs = vecs.astype(complex)
    beta = np.sqrt(vals + 0j)
    beta = np.where(beta.real < 0, -beta, beta)
    order = np.argsort(-beta.real)
    beta, vecs = beta[order], vecs[:, order]
    out = {k: np.zeros((nu, nv, num_modes), complex) for k in ("Eu", "Ev", "Ew", "Hu", "Hv", "Hw")}
    for m in range(num_modes):
        E = vecs[:, m]
        Ht = (Q @ E) / beta[m]
        Eu_, Ev_ = E[:N], E[N:]
        Htu, Htv = Ht[:N], Ht[N:]
        Ew_ = (1j / k0) * (Ew_op @ Ht)
        Htw = (-1j / k0) * (curl_e @ E)
        if bend_radius is not None:             # back to the physical frame: F = J^T F' (ref solver.py:254-259)
            Ew_ = Ew_ * np.broadcast_to(s_e, (nu, nv)).reshape(-1)
            Htw = Htw * np.broadcast_to(s_h, (nu, nv)).reshape(-1)
        f = dict(Eu=Eu_, Ev=Ev_, Ew=Ew_, Hu=Htu / ETA_0, Hv=Htv / ETA_0, Hw=Htw / ETA_0)
        f = {k: v.reshape(nu, nv) for k, v in f.items()}
        # gauge: the largest tangential E sample is real and positive (ref mode_solver.py:803-806)
        big = f["Eu"] if np.abs(f["Eu"]).max() >= np.abs(f["Ev"]).max() else f["Ev"]
        ph = np.exp(-1j * np.angle(big.reshape(-1)[np.argmax(np.abs(big))]))
        # unit power along +w: 0.5 Re int (E_u H_v* - E_v H_u*) dA, fields colocated to cell centres
        f = {k: v * ph for k, v in f.items()}
        p = mode_flux(f, ub, vb)
        s = 1.0 / np.sqrt(abs(p)) if p != 0 else 1.0
        for k in out:
            out[k][:, :, m] = f[k] * s
    return ModeResult(n_complex=beta / k0, **out)


def solve_modes_angled(eps_u: np.ndarray, eps_v: np.ndarray, eps_w: np.ndarray, ub: np.ndarray, vb: np.ndarray,
                       freq: float, angle_theta: float, angle_phi: float = 0.0, num_modes: int = 1,
                       target_neff: Optional[float] = None, precision: str = "double",
                       pmc_min: Tuple[bool, bool] = (False, False), num_pml: Tuple[int, int] = (0, 0),
                       pml_min: Tuple[bool, bool] = (True, True), bend_radius: Optional[float] = None,
                       bend_axis: int = 0) -> ModeResult:
    """Modes of a waveguide that crosses the plane at polar angle ``angle_theta`` from its normal, azimuth
    ``angle_phi`` from the plane's u axis (module docstring).  Same arguments and result layout as
    ``solve_modes``; the tensorial problem is complex even for lossless media (eigenvalue i n).
    With ``bend_radius`` the two straightening maps are composed as the reference composes them (ref solver.py:141-147:
    the shear first, then the conformal map): J = diag(1, 1, s) [[1, 0, a], [0, 1, b], [0, 0, 1]], s = R / r, det J = s;
    ``num_pml`` stretches the derivative operators as in ``solve_modes`` (wave speeds from the diagonals of the
    transformed tensors, ref derivatives.py:129-155)."""
    nu, nv = eps_u.shape
    N = nu * nv
    k0 = 2 * np.pi * freq / C_0
    a = -np.tan(angle_theta) * np.cos(angle_phi)
    b = -np.tan(angle_theta) * np.sin(angle_phi)
    eu, ev, ew = (np.asarray(x, complex).reshape(-1) for x in (eps_u, eps_v, eps_w))
    one = np.ones(N, complex)
    s_e = s_h = one
    if bend_radius is not None:
        na = 0 if bend_axis == 1 else 1                      # axis along which the radius varies
        c = (ub, vb)[na]
        r = c + (bend_radius - c[(len(c) - 1) // 2])
        shp = (-1, 1) if na == 0 else (1, -1)
        s_e = np.broadcast_to((bend_radius / r[:-1]).reshape(shp), (nu, nv)).reshape(-1).astype(complex)
        s_h = np.broadcast_to((bend_radius / (r[:-1] + r[1:]) * 2).reshape(shp), (nu, nv)).reshape(-1).astype(complex)
    # eps' = J diag(eu, ev, ew) J^T / det J,  mu' = J J^T / det J
    eps = {"uu": (eu + a * a * ew) / s_e, "uv": a * b * ew / s_e, "uw": a * ew, "vu": a * b * ew / s_e,
           "vv": (ev + b * b * ew) / s_e, "vw": b * ew, "wu": a * ew, "wv": b * ew, "ww": ew * s_e}
    mu = {"uu": (1 + a * a) * one / s_h, "uv": a * b * one / s_h, "uw": a * one, "vu": a * b * one / s_h,
          "vv": (1 + b * b) * one / s_h, "vw": b * one, "wu": a * one, "wv": b * one, "ww": one * s_h}
    du_p, dv_p = np.diff(ub), np.diff(vb)
    du_d = np.concatenate(([du_p[0]], 0.5 * (du_p[1:] + du_p[:-1])))
    dv_d = np.concatenate(([dv_p[0]], 0.5 * (dv_p[1:] + dv_p[:-1])))
    s_fac = None
    if any(int(n_) > 0 for n_ in num_pml):
        pu, pv = int(num_pml[0]), int(num_pml[1])
        diag = np.stack([eps[k].reshape(nu, nv) for k in ("uu", "vv", "ww")])
        mdiag = np.stack([mu[k].reshape(nu, nv) for k in ("uu", "vv", "ww")])

        def mean(x):
            return 1.0 if x.size == 0 else np.mean(x)
        regions = [np.s_[:, :pu, :], np.s_[:, nu - pu + 1:, :], np.s_[:, :, :pv], np.s_[:, :, nv - pv + 1:]]
        speed = [1 / np.sqrt(mean(diag[r_]) * mean(mdiag[r_])) for r_ in regions]
        omega = 2 * np.pi * freq
        s_fac = (_pml_s("f", omega, du_p, nu, pu, pml_min[0], speed[:2]), _pml_s("b", omega, du_d, nu, pu, pml_min[0], speed[:2]),
                 _pml_s("f", omega, dv_p, nv, pv, pml_min[1], speed[2:]), _pml_s("b", omega, dv_d, nv, pv, pml_min[1], speed[2:]))
    # the walls live in the derivative matrices alone, as in the reference's tensorial problem (ref solver.py:595-668,
    # derivatives.py:9-62) — no rows are clamped.  (Clamped like the diagonal solver's, a weakly guided mode that reaches the
    # walls came out 2e-4 off in n_eff: tests/test_fuzz_mode_solver.py)
    Duf, Dvf, Dub, Dvb = (D / k0 for D in _diff_ops(nu, nv, du_p, dv_p, du_d, dv_d, pmc_min, s_fac, wall_in_fwd=True))
    dg = lambda v: sp.diags(np.asarray(v).reshape(-1))           # noqa: E731
    mask_u, mask_v, mask_w = np.ones((nu, nv)), np.ones((nu, nv)), np.ones((nu, nv))
    Mu, Mv, Mw = dg(mask_u), dg(mask_v), dg(mask_w)
    Z = sp.csr_matrix((N, N), dtype=complex)
    Ie, Im = dg(1.0 / eps["ww"]), dg(1.0 / mu["ww"])
    # E_w and Ht_w as operators on x = [E_u, E_v, Ht_u, Ht_v]
    Ew = [Mw @ (-Ie @ dg(eps["wu"])), Mw @ (-Ie @ dg(eps["wv"])), Mw @ (-1j * Ie @ Dvb), Mw @ (1j * Ie @ Dub)]
    Hw = [1j * Im @ Dvf, -1j * Im @ Duf, -Im @ dg(mu["wu"]), -Im @ dg(mu["wv"])]

    def row(Dl, Ow, sgn, t_u, t_v, t_w, on_h: bool, mask=None):
        """D . O_w  + sgn i (t_u X_u + t_v X_v + t_w O2_w): the X are Ht (on_h) or E components."""
        O2 = Hw if on_h else Ew
        blocks = [Dl @ Ow[q] + sgn * 1j * dg(t_w) @ O2[q] for q in range(4)]
        off = 2 if on_h else 0
        blocks[off] = blocks[off] + sgn * 1j * dg(t_u)
        blocks[off + 1] = blocks[off + 1] + sgn * 1j * dg(t_v)
        return [b_ if mask is None else mask @ b_ for b_ in blocks]
    rows = [row(Duf, Ew, +1, mu["vu"], mu["vv"], mu["vw"], True, Mu),
            row(Dvf, Ew, -1, mu["uu"], mu["uv"], mu["uw"], True, Mv),
            row(Dub, Hw, -1, eps["vu"], eps["vv"], eps["vw"], False),
            row(Dvb, Hw, +1, eps["uu"], eps["uv"], eps["uw"], False)]
    A = (-1j * sp.bmat(rows, format="csc")).astype(np.complex64 if precision == "single" else np.complex128)
    cos_t = float(np.cos(angle_theta))
    if target_neff is None:
        target_neff = float(np.sqrt(np.max(np.abs([np.max(np.abs(x)) for x in (eps_u, eps_v, eps_w)]))))
    sigma = target_neff / cos_t
    rng = np.random.default_rng(0)
    v0 = rng.standard_normal(4 * N).astype(A.dtype)
    vals, vecs = spl.eigs(A, k=num_modes, sigma=A.dtype.type(sigma), v0=v0, tol=(1e-6 if precision == "single" else 1e-10))
    vals, vecs = vals.astype(complex), vecs.astype(complex)
    order = np.argsort(-vals.real)
    vals, vecs = vals[order], vecs[:, order]
    out = {k: np.zeros((nu, nv, num_modes), complex) for k in ("Eu", "Ev", "Ew", "Hu", "Hv", "Hw")}
    for m in range(num_modes):
        x = vecs[:, m]
        Eu_, Ev_, Htu, Htv = x[:N], x[N:2 * N], x[2 * N:3 * N], x[3 * N:]
        Ew_p = sum(Ew[q] @ x[q * N:(q + 1) * N] for q in range(4))
        Hw_p = sum(Hw[q] @ x[q * N:(q + 1) * N] for q in range(4))
        # physical frame: F = J^T F'
        f = dict(Eu=Eu_, Ev=Ev_, Ew=a * Eu_ + b * Ev_ + s_e * Ew_p, Hu=Htu / ETA_0, Hv=Htv / ETA_0,
                 Hw=(a * Htu + b * Htv + s_h * Hw_p) / ETA_0)
        f = {k: v.reshape(nu, nv) for k, v in f.items()}
        big = f["Eu"] if np.abs(f["Eu"]).max() >= np.abs(f["Ev"]).max() else f["Ev"]
        ph = np.exp(-1j * np.angle(big.reshape(-1)[np.argmax(np.abs(big))]))
        f = {k: v * ph for k, v in f.items()}
        pw = mode_flux(f, ub, vb)
        sc = 1.0 / np.sqrt(abs(pw)) if pw != 0 else 1.0
        for k in out:
            out[k][:, :, m] = f[k] * sc
    return ModeResult(n_complex=vals * cos_t, **out)


def _to_centres(f: np.ndarray, on_b_u: bool, on_b_v: bool) -> np.ndarray:
    """Average a node field to cell centres (values beyond the max edge are the PEC zero)."""
    g = f
    if on_b_u:
        g = 0.5 * (g + np.concatenate([g[1:], np.zeros_like(g[:1])], axis=0))
    if on_b_v:
        g = 0.5 * (g + np.concatenate([g[:, 1:], np.zeros_like(g[:, :1])], axis=1))
    return g


def mode_flux(f: dict, ub: np.ndarray, vb: np.ndarray) -> float:
    """0.5 Re int (E_u H_v* - E_v H_u*) du dv with all four fields brought to the cell centres."""
    eu = _to_centres(f["Eu"], False, True)
    ev = _to_centres(f["Ev"], True, False)
    hu = _to_centres(f["Hu"], True, False)
    hv = _to_centres(f["Hv"], False, True)
    dA = np.outer(np.diff(ub), np.diff(vb))
    return float(0.5 * np.real(np.sum((eu * np.conj(hv) - ev * np.conj(hu)) * dA)))
