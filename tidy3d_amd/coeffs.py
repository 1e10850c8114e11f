"""fp64 update-coefficient tables derived from a SolverSpec.

Shared by the HIP engine (which casts them to fp32 and uploads them) and by
the NumPy oracle (which uses them in fp64), so both step *the same discrete
system*; the only difference left between the two is fp32-vs-fp64 arithmetic.

Discrete system (textbook Yee leapfrog; no counterpart in the reference,
SURVEY.md section 8(a) K1-K4; conventions pinned by the reference as cited):

  H^{n+1/2} = H^{n-1/2} - (dt/mu0) curl_p E^n                    (primal steps, ref grid.py:393)
  E^{n+1}   = Ca E^n + Cb (curl_d H^{n+1/2} - J^{n+1/2}) - Cc S^n (dual steps,  ref grid.py:404)

with, for a medium eps(w) = eps_inf + i sigma/(w eps0) - sum_k [c_k/(jw+a_k) + cc]
(ref medium.py:2900-2913, e^{-iwt}):  each pole carries a complex auxiliary polarisation
Q_k (= P_k/eps0) obeying dQ/dt = a Q + c E, integrated with the trapezoidal rule

  Q^{n+1} = kap Q^n + bet (E^{n+1} + E^n),  kap = (1 + a dt/2)/(1 - a dt/2),
                                             bet = (c dt/2)/(1 - a dt/2)
  S^n  = sum_k 2 Re[(kap_k - 1) Q_k^n]
  D    = eps_inf + sum_k 2 Re bet_k + sigma dt/(2 eps0)
  Ca   = (eps_inf - sum_k 2 Re bet_k - sigma dt/(2 eps0)) / D
  Cb   = (dt/eps0) / D,   Cc = 1 / D

CPML (stretched coordinate, CFS; Roden & Gedney form, parameters from
ref boundary.py:195-254 with sigma, alpha in units of 2 eps0/dt, ref constants.py:120):
every derivative d/du inside a PML becomes (1/kappa) d/du + psi with
psi <- b psi + c d/du,  b = exp(-2 (sigma_p/kappa + alpha_p)),
c = sigma_p (b - 1) / (kappa (sigma_p + kappa alpha_p)).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .constants import EPSILON_0, MU_0
from .spec import BC_PERIODIC, MediumCoeffs, PmlFace, SolverSpec


@dataclass
class MaterialTable:
    ca: np.ndarray          # [n_media]
    cb: np.ndarray          # [n_media]  (multiplies curl H, already contains dt/eps0)
    cc: np.ndarray          # [n_media]  (multiplies the ADE memory term S)
    kap: List[np.ndarray]   # per medium: complex [n_poles]
    bet: List[np.ndarray]   # per medium: complex [n_poles]

    @property
    def n_media(self) -> int:
        return len(self.ca)

    def is_dispersive(self, m: int) -> bool:
        return len(self.kap[m]) > 0


def material_table(media: List[MediumCoeffs], dt: float) -> MaterialTable:
    n = len(media)
    ca, cb, cc = np.zeros(n), np.zeros(n), np.zeros(n)
    kap, bet = [], []
    for m, med in enumerate(media):
        if med.pec:
            kap.append(np.zeros(0, complex))
            bet.append(np.zeros(0, complex))
            continue
        a = np.array([p[0] for p in med.poles], dtype=complex)
        c = np.array([p[1] for p in med.poles], dtype=complex)
        k = (1 + a * dt / 2) / (1 - a * dt / 2)
        b = (c * dt / 2) / (1 - a * dt / 2)
        sb = float(np.sum(2 * b.real))
        sg = med.sigma * dt / (2 * EPSILON_0)
        D = med.eps_inf + sb + sg
        ca[m] = (med.eps_inf - sb - sg) / D
        cb[m] = (dt / EPSILON_0) / D
        cc[m] = 1.0 / D
        kap.append(k)
        bet.append(b)
    return MaterialTable(ca=ca, cb=cb, cc=cc, kap=kap, bet=bet)


def h_coeff(dt: float) -> float:
    """dt/mu0: multiplies curl E in the H update (vacuum permeability everywhere)."""
    return dt / MU_0


def _poly(vmin: float, vmax: float, order: int, d: np.ndarray) -> np.ndarray:
    return vmin + (vmax - vmin) * d ** order


def _face_profile(face: PmlFace, depth: np.ndarray):
    """(1/kappa, b, c) at normalised depths ``depth`` in [0, 1] (0 = PML entrance, 1 = wall)."""
    d = np.clip(depth, 0.0, 1.0)
    sig = _poly(face.sigma_min, face.sigma_max, face.sigma_order, d)
    kap = _poly(face.kappa_min, face.kappa_max, face.kappa_order, d)
    # CFS alpha is largest at the PML entrance and decays towards the wall
    alp = _poly(face.alpha_min, face.alpha_max, face.alpha_order, 1.0 - d)
    b = np.exp(-2.0 * (sig / kap + alp))
    den = kap * (sig + kap * alp)
    with np.errstate(divide="ignore", invalid="ignore"):
        c = np.where(den > 0, sig * (b - 1.0) / den, 0.0)
    return 1.0 / kap, b, c


@dataclass
class PmlAxis:
    """1-D CPML tables along one axis, full axis length N (identity outside the slabs).

    ``*_e``: sampled at the cell boundaries b[i] (where backward differences of H live),
    ``*_h``: sampled at the cell centres (where forward differences of E live).
    Profile positions follow ref plugins/mode/derivatives.py:174-197 (i/n for E,
    (i+1/2)/n for H)."""

    n_lo: int
    n_hi: int
    kinv_e: np.ndarray
    b_e: np.ndarray
    c_e: np.ndarray
    kinv_h: np.ndarray
    b_h: np.ndarray
    c_h: np.ndarray

    def slab_ranges_e(self, N: int) -> List[Tuple[int, int]]:
        """Index ranges [lo, hi) where the E-side tables differ from identity."""
        r = []
        if self.n_lo > 0:
            r.append((0, self.n_lo))
        if self.n_hi > 1:
            r.append((N - self.n_hi + 1, N))
        return r

    def slab_ranges_h(self, N: int) -> List[Tuple[int, int]]:
        r = []
        if self.n_lo > 0:
            r.append((0, self.n_lo))
        if self.n_hi > 0:
            r.append((N - self.n_hi, N))
        return r


def pml_axis(spec: SolverSpec, axis: int) -> PmlAxis:
    N = spec.shape[axis]
    lo, hi = spec.pml[axis]
    n_lo, n_hi = int(lo.num_layers), int(hi.num_layers)
    if n_lo + n_hi > N:
        raise ValueError("PML layers exceed the grid size along axis %d" % axis)
    kinv_e, b_e, c_e = np.ones(N), np.zeros(N), np.zeros(N)
    kinv_h, b_h, c_h = np.ones(N), np.zeros(N), np.zeros(N)
    i = np.arange(N, dtype=np.float64)
    if n_lo > 0:
        s = slice(0, n_lo)
        kinv_e[s], b_e[s], c_e[s] = _face_profile(lo, (n_lo - i[s]) / n_lo)
        kinv_h[s], b_h[s], c_h[s] = _face_profile(lo, (n_lo - i[s] - 0.5) / n_lo)
    if n_hi > 0:
        s = slice(N - n_hi, N)
        kinv_h[s], b_h[s], c_h[s] = _face_profile(hi, (i[s] + 0.5 - (N - n_hi)) / n_hi)
        if n_hi > 1:
            s = slice(N - n_hi + 1, N)
            kinv_e[s], b_e[s], c_e[s] = _face_profile(hi, (i[s] - (N - n_hi)) / n_hi)
    return PmlAxis(n_lo, n_hi, kinv_e, b_e, c_e, kinv_h, b_h, c_h)


@dataclass
class DampingAxis:
    """Per-step damping factors of the absorber layers along one axis (identity outside):
    ``fb`` at the cell boundaries b[i], ``fc`` at the cell centres; layers [0, n_lo) and [N - n_hi, N)."""

    n_lo: int
    n_hi: int
    fb: np.ndarray
    fc: np.ndarray


def damping_tables(spec: SolverSpec):
    """Adiabatic absorber (ref boundary.py:427-476: "a multilayer system with gradually increasing
    conductivity" in front of a PEC wall; the reference's discretisation of it is server-side) as a
    *matched* conductivity: the same decay rate sigma/eps0 for E and H, whatever the medium, so the
    layers are impedance-matched at normal incidence in the continuum limit and media (also
    dispersive ones) may run through them.  The decay is integrated exactly over one step, operator-
    split from the curl updates:

        H^{n-1/2} <- f_H H^{n-1/2}   (start of step n),
        E^{n+1}   <- f_E (Ca E^n + Cb (curl H - J) + CPML terms) - Cc S^n    (memory term added after)
        f = exp(-2 (s_x + s_y + s_z)),   s_a = profile of axis a at the component's Yee location
                                         (sigma in units of 2 eps0/dt: sigma dt/eps0 = 2 s).

    Returns None or one DampingAxis per axis; a component's factor is the product of its three axis
    factors (``fc`` along the axes where it sits on cell centres, ``fb`` elsewhere)."""
    if spec.absorber is None:
        return None
    out = []
    for sb, sc, n_lo, n_hi in spec.absorber:
        out.append(DampingAxis(n_lo=int(n_lo), n_hi=int(n_hi), fb=np.exp(-2.0 * np.asarray(sb, np.float64)),
                               fc=np.exp(-2.0 * np.asarray(sc, np.float64))))
    return out


def inv_steps(spec: SolverSpec):
    """(1/primal, 1/dual) per axis: what the curls multiply differences with."""
    ip = [1.0 / spec.primal_steps(a) for a in range(3)]
    idl = [1.0 / spec.dual_steps(a) for a in range(3)]
    return ip, idl
