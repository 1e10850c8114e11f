"""TEST INFRASTRUCTURE — fp64 NumPy restatement of the FDTD hot path (the oracle).

Not part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.  The product path
(``tidy3d_amd.web.run`` -> ``libfdtd_hip.so``) never falls back to it.

PARITY STATUS: **the FDTD field values are "parity unpinned" against the
reference** — flexcompute/tidy3d contains no time-stepper at all (SURVEY.md
section 0.1, section 8(c)); there is no reference output, golden vector or test for field
values.  What *is* pinned by the reference and checked in tests/test_golden_*.py:
Yee staggering (ref grid/grid.py:465-491), primal/dual steps (grid.py:393-417),
PEC/PMC edge rules of the difference operators (ref plugins/mode/derivatives.py:9-62),
stretched-coordinate PML sampling positions (derivatives.py:158-232), pole-residue
dispersion (ref medium.py:2900-2913), time step (ref simulation.py:4194-4211) and the
source waveform/spectrum (ref source.py:174-193, time.py:72-105).  The arithmetic of
the leapfrog/CPML/ADE recursions is the textbook scheme documented in
``tidy3d_amd/coeffs.py``; it is validated by physics in tests/test_physics_oracle.py
(PEC-cavity eigenfrequencies, PML reflection, Fresnel transmission of a dispersive
slab, energy conservation) and by a structural property that needs no reference data: discrete Lorentz
reciprocity, held to 1e-11 through CPML, walls of either kind, periodic axes, graded cells and lossy /
dispersive / anisotropic bodies (tests/test_reciprocity.py).

It consumes the same ``SolverSpec`` (the *statement* of the problem: grid boundaries,
material indices, source lists, monitor boxes) as the HIP engine, but derives every
update coefficient ITSELF from that statement — 1/steps, dt/mu0, (Ca, Cb, Cc), the ADE
(kappa, beta) pairs, the CPML (1/kappa, b, c) profiles and their sampling positions, the
absorber factors and the DFT phase tables — with the formulas written out below
(``_own_*``), not by importing ``tidy3d_amd.coeffs``.  tests/test_oracle_coeffs.py asserts
the two derivations agree to 1e-13, so a coefficient error in the product is not
common-mode with the oracle.  GPU-vs-oracle differences are then pure fp32 round-off.
It also *is* the "naive NumPy curl-loop" CPU baseline that BASELINE.md section 4 asks to
be timed next to the GPU number.

Discrete system restated here (all times in s, lengths in um, e^{-i w t}):

  H^{n+1/2} = H^{n-1/2} - (dt/mu0) curl_primal E^n
  E^{n+1}   = Ca E^n + Cb curl_dual H^{n+1/2} - Cc S^n          S^n = sum_k 2 Re[(kap_k - 1) Q_k^n]
  Q_k^{n+1} = kap_k Q_k^n + bet_k (E^{n+1} + E^n)

  medium  eps(w) = eps_inf + i sigma/(w eps0) - sum_k [c_k/(jw + a_k) + c.c.]   (ref medium.py:2900-2913)
      kap_k = (2 + a_k dt)/(2 - a_k dt),   bet_k = c_k dt/(2 - a_k dt)           (trapezoidal rule)
      D  = eps_inf + sum_k 2 Re bet_k + sigma dt/(2 eps0)
      Ca = (2 eps_inf - D)/D,   Cb = dt/(eps0 D),   Cc = 1/D
  CPML   d/du -> (1/kappa) d/du + psi,  psi <- b psi + c d/du   (ref boundary.py:195-254, units 2 eps0/dt)
      b = exp(-2 (sigma/kappa + alpha)),  c = sigma (b - 1)/(kappa (sigma + kappa alpha))
      profiles  p(d) = p_min + (p_max - p_min) d^order  at depth d in [0, 1] from the PML entrance
      (alpha runs the other way: 1 - d); sampled at i/n (E side) and (i + 1/2)/n (H side)
      (ref plugins/mode/derivatives.py:174-197)
  DFT    acc += w(t) (stride dt / sqrt(2 pi)) exp(+2 pi i f t) F,  E at t_n, H at t_n + dt/2
      (same kernel as SourceTime.spectrum, ref time.py:95-105; w = ApodizationSpec window,
      ref apodization.py:87-94)
  decay  W = sum |E|^2 + (mu0/eps0) sum |H|^2 over all cells (H half a step later than E), evaluated every
      decay_every steps; field_decay = W / max W; shutoff when it falls below Simulation.shutoff after
      the sources have ended (ref simulation.py:2089-2096); non-finite W = diverged (ref sim_data.py:909)
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from tidy3d_amd.spec import BC_PEC, BC_PERIODIC, BC_PMC, SolverSpec

# physical constants in tidy3d's unit system (um, s): ref constants.py:16-32
_C0 = 2.99792458e14
_MU0 = 1.25663706212e-12
_EPS0 = 1.0 / (_MU0 * _C0 * _C0)
_ETA0_SQ = _MU0 / _EPS0


# ---------------------------------------------------------------------------------------------
# the oracle's own coefficient derivations (deliberately NOT shared with tidy3d_amd/coeffs.py)
# ---------------------------------------------------------------------------------------------
class _OwnMaterials:
    """(Ca, Cb, Cc) and the ADE pairs of every medium of the spec, one medium at a time in plain Python
    complex arithmetic."""

    def __init__(self, media, dt: float):
        self.ca, self.cb, self.cc, self.kap, self.bet = [], [], [], [], []
        for med in media:
            if med.pec:
                self.ca.append(0.0); self.cb.append(0.0); self.cc.append(0.0)
                self.kap.append(np.zeros(0, complex)); self.bet.append(np.zeros(0, complex))
                continue
            kaps, bets, two_re_beta = [], [], 0.0
            for a, c in med.poles:
                a, c = complex(a), complex(c)
                kaps.append((2.0 + a * dt) / (2.0 - a * dt))
                beta = c * dt / (2.0 - a * dt)
                bets.append(beta)
                two_re_beta += 2.0 * beta.real
            D = med.eps_inf + two_re_beta + med.sigma * dt / (2.0 * _EPS0)
            self.ca.append((2.0 * med.eps_inf - D) / D)
            self.cb.append(dt / (_EPS0 * D))
            self.cc.append(1.0 / D)
            self.kap.append(np.array(kaps, complex))
            self.bet.append(np.array(bets, complex))
        self.ca, self.cb, self.cc = np.array(self.ca), np.array(self.cb), np.array(self.cc)

    @property
    def n_media(self) -> int:
        return len(self.ca)

    def is_dispersive(self, m: int) -> bool:
        return len(self.kap[m]) > 0


class _OwnPml:
    """CPML tables of one axis, identity outside the layers: cell i of the low face lies at depth
    (n_lo - i)/n_lo (E side, on the cell boundary) and (n_lo - i - 1/2)/n_lo (H side, cell centre); cell i
    of the high face at (i - (N - n_hi))/n_hi and (i + 1/2 - (N - n_hi))/n_hi."""

    def __init__(self, spec: SolverSpec, axis: int):
        N = spec.shape[axis]
        lo, hi = spec.pml[axis]
        self.n_lo, self.n_hi = int(lo.num_layers), int(hi.num_layers)
        self.kinv_e, self.b_e, self.c_e = np.ones(N), np.zeros(N), np.zeros(N)
        self.kinv_h, self.b_h, self.c_h = np.ones(N), np.zeros(N), np.zeros(N)
        for i in range(self.n_lo):
            self._set("e", i, lo, (self.n_lo - i) / self.n_lo)
            self._set("h", i, lo, (self.n_lo - i - 0.5) / self.n_lo)
        for i in range(N - self.n_hi, N):
            if i > N - self.n_hi:                       # the first boundary of the high face IS the entrance
                self._set("e", i, hi, (i - (N - self.n_hi)) / self.n_hi)
            self._set("h", i, hi, (i + 0.5 - (N - self.n_hi)) / self.n_hi)

    def _set(self, side: str, i: int, face, depth: float):
        d = min(max(depth, 0.0), 1.0)
        sigma = face.sigma_min + (face.sigma_max - face.sigma_min) * d ** face.sigma_order
        kappa = face.kappa_min + (face.kappa_max - face.kappa_min) * d ** face.kappa_order
        alpha = face.alpha_min + (face.alpha_max - face.alpha_min) * (1.0 - d) ** face.alpha_order
        b = np.exp(-2.0 * (sigma / kappa + alpha))
        den = kappa * (sigma + kappa * alpha)
        c = sigma * (b - 1.0) / den if den > 0 else 0.0
        getattr(self, "kinv_" + side)[i] = 1.0 / kappa
        getattr(self, "b_" + side)[i] = b
        getattr(self, "c_" + side)[i] = c


def _own_inv_steps(spec: SolverSpec):
    """1/primal (forward differences of E) and 1/dual (backward differences of H) steps per axis; the dual
    step of cell 0 pairs it with cell N-1 on a periodic axis and with itself otherwise (ref grid.py:393-417)."""
    ip, idl = [], []
    for a in range(3):
        b = np.asarray(spec.boundaries[a], np.float64)
        d = b[1:] - b[:-1]
        dd = np.empty_like(d)
        dd[1:] = 0.5 * (d[1:] + d[:-1])
        dd[0] = 0.5 * (d[0] + d[-1]) if spec.bc[a][0] == BC_PERIODIC else d[0]
        ip.append(1.0 / d)
        idl.append(1.0 / dd)
    return ip, idl


def _own_dft_phases(m, dt: float):
    """(phase_e, phase_h) [n_rec, nf] of a running-DFT monitor from its recording steps, stride, frequencies
    and apodisation (MonitorSpec.stride / .apod); falls back to the tables in the spec when the spec does not
    carry them (specs built by hand in older tests)."""
    if getattr(m, "stride", None) is None:
        return np.asarray(m.phase_e), np.asarray(m.phase_h)
    f = np.asarray(m.freqs, np.float64)[None, :]
    pref = m.stride * dt / np.sqrt(2.0 * np.pi)

    def window(t):
        w = np.ones_like(t)
        if m.apod is not None:
            start, end, width = m.apod
            if start is not None:
                w = np.where(t < start, w * np.exp(-0.5 * ((t - start) / width) ** 2), w)
            if end is not None:
                w = np.where(t > end, w * np.exp(-0.5 * ((t - end) / width) ** 2), w)
        return w
    te = np.asarray(m.steps, np.float64) * dt
    th = te + 0.5 * dt
    return (window(te)[:, None] * pref * np.exp(2j * np.pi * f * te[:, None]),
            window(th)[:, None] * pref * np.exp(2j * np.pi * f * th[:, None]))


def _ax(axis: int) -> int:
    """numpy axis of physical axis (arrays are [z, y, x])."""
    return 2 - axis


def _bcast(v: np.ndarray, axis: int) -> np.ndarray:
    shp = [1, 1, 1]
    shp[_ax(axis)] = -1
    return v.reshape(shp)


class OracleFdtd:
    def __init__(self, spec: SolverSpec, dtype=np.float64):
        self.spec = spec
        # Bloch boundaries (ref boundary.py:55-79): complex fields, F(r + L_a) = exp(i phi_a) F(r); the
        # HIP engine carries them as a (Re, Im) pair of real solvers (fdtd_run_bloch)
        self.bloch = getattr(spec, "bloch", None)
        self.rdtype = dtype
        if self.bloch is not None:
            dtype = np.complex128
        self.dtype = dtype
        nx, ny, nz = spec.shape
        shp = (nz, ny, nx)
        self.E = [np.zeros(shp, dtype) for _ in range(3)]
        self.H = [np.zeros(shp, dtype) for _ in range(3)]
        ip, idl = _own_inv_steps(spec)
        self.ip = [_bcast(v.astype(dtype), a) for a, v in enumerate(ip)]
        self.id = [_bcast(v.astype(dtype), a) for a, v in enumerate(idl)]
        self.ch = dtype(spec.dt / _MU0)
        # absorber layers: per-step damping factors at each component's Yee location
        # (tidy3d_amd.coeffs.damping_tables; Absorber of ref boundary.py:427-476)
        self.damp_e = self.damp_h = None
        if spec.absorber is not None:
            # per-step factor exp(-2 s) of each axis (s = conductivity in units of 2 eps0/dt at the cell
            # boundaries / centres); a component's factor is the product over the axes at ITS Yee location
            def factor(c, is_h):
                f = np.ones(shp, np.float64)
                for a in range(3):
                    s_b, s_c = spec.absorber[a][0], spec.absorber[a][1]
                    on_center = (a == c) != is_h
                    f = f * _bcast(np.exp(-2.0 * np.asarray(s_c if on_center else s_b, np.float64)), a)
                return f.astype(self.rdtype)
            self.damp_e = [factor(c, False) for c in range(3)]
            self.damp_h = [factor(c, True) for c in range(3)]
        self.mt = _OwnMaterials(spec.media, spec.dt)
        if spec.mat_idx is not None:
            self.ca = [self.mt.ca[spec.mat_idx[c]].astype(self.rdtype) for c in range(3)]
            self.cb = [self.mt.cb[spec.mat_idx[c]].astype(self.rdtype) for c in range(3)]
        else:
            self.ca = [self.rdtype(self.mt.ca[1])] * 3
            self.cb = [self.rdtype(self.mt.cb[1])] * 3
        # CPML
        self.pml = [_OwnPml(spec, a) for a in range(3)]
        self.has_pml = [p.n_lo + p.n_hi > 0 for p in self.pml]
        self.psi_h: Dict = {}
        self.psi_e: Dict = {}
        for a in range(3):
            if self.has_pml[a]:
                for c in range(3):
                    if c != a:
                        self.psi_h[(c, a)] = np.zeros(shp, dtype)
                        self.psi_e[(c, a)] = np.zeros(shp, dtype)
        # ADE state: per component, per dispersive medium: (flat idx, Q[n_poles, n_cells])
        self.ade = []
        if spec.mat_idx is not None:
            for c in range(3):
                flat = spec.mat_idx[c].ravel()
                for m in range(self.mt.n_media):
                    if self.mt.is_dispersive(m):
                        idx = np.nonzero(flat == m)[0]
                        if idx.size:
                            q = np.zeros((2 if self.bloch is not None else 1, len(self.mt.kap[m]), idx.size), complex)
                            self.ade.append([c, m, idx, q, np.zeros(idx.size, dtype)])
        elif self.mt.is_dispersive(1):
            for c in range(3):
                idx = np.arange(nx * ny * nz)
                q = np.zeros((2 if self.bloch is not None else 1, len(self.mt.kap[1]), idx.size), complex)
                self.ade.append([c, 1, idx, q, np.zeros(idx.size, dtype)])
        # TFSF auxiliary 1-D grids
        self.tfsf_state = [self._tfsf_init(t) for t in spec.tfsf]
        # monitor accumulators
        self.mon_data = []
        self.mon_count = []
        self.mon_phase = [(_own_dft_phases(m, spec.dt) if m.kind != "time" else None) for m in spec.monitors]
        for m in spec.monitors:
            bz, by, bx = m.shape
            if m.kind == "time":
                self.mon_data.append(np.zeros((len(m.steps), len(m.comps), bz, by, bx), dtype))
            else:
                self.mon_data.append(np.zeros((len(m.freqs), len(m.comps), bz, by, bx), complex))
            self.mon_count.append(0)
        self.step_index = 0
        self.energy_max = 0.0
        self.decay = 1.0
        self.diverged = False
        self.stopped_at: Optional[int] = None

    # ------------------------------------------------------------------ differences
    def _fwd(self, F: np.ndarray, axis: int) -> np.ndarray:
        """(F[i+1] - F[i]) / primal[i]; F[N] = 0 at a wall, F[0] for periodic
        (ref derivatives.py:9-40: truncation at the max edge = PEC)."""
        ax = _ax(axis)
        nxt = np.roll(F, -1, axis=ax)
        sl = [slice(None)] * 3
        sl[ax] = -1
        if self.spec.bc[axis][1] != BC_PERIODIC:
            nxt[tuple(sl)] = 0
        elif self.bloch is not None and self.bloch[axis] != 0.0:
            nxt[tuple(sl)] *= np.exp(1j * self.bloch[axis])          # F[N] = exp(+i phi) F[0]
        return (nxt - F) * self.ip[axis]

    def _bwd(self, F: np.ndarray, axis: int) -> np.ndarray:
        """(F[i] - F[i-1]) / dual[i]; F[-1] = F[N-1] periodic, -F[0] PMC (ref derivatives.py:
        43-62: element [0,0] = 2), 0 PEC (the wall component is zeroed afterwards anyway)."""
        ax = _ax(axis)
        prv = np.roll(F, 1, axis=ax)
        bc = self.spec.bc[axis][0]
        sl = [slice(None)] * 3
        sl[ax] = 0
        if bc != BC_PERIODIC:
            prv[tuple(sl)] = -F[tuple(sl)] if bc == BC_PMC else 0
        elif self.bloch is not None and self.bloch[axis] != 0.0:
            prv[tuple(sl)] *= np.exp(-1j * self.bloch[axis])         # F[-1] = exp(-i phi) F[N-1]
        return (F - prv) * self.id[axis]

    def _pml_h(self, c: int, a: int, d: np.ndarray) -> np.ndarray:
        if not self.has_pml[a]:
            return d
        p = self.pml[a]
        psi = self.psi_h[(c, a)]
        psi *= _bcast(p.b_h, a)
        psi += _bcast(p.c_h, a) * d
        return _bcast(p.kinv_h, a) * d + psi

    def _pml_e(self, c: int, a: int, d: np.ndarray) -> np.ndarray:
        if not self.has_pml[a]:
            return d
        p = self.pml[a]
        psi = self.psi_e[(c, a)]
        psi *= _bcast(p.b_e, a)
        psi += _bcast(p.c_e, a) * d
        return _bcast(p.kinv_e, a) * d + psi

    # ------------------------------------------------------------------ updates
    def update_h(self, n: int):
        E, H, ch = self.E, self.H, self.ch
        for c in range(3):
            a1, a2 = (c + 1) % 3, (c + 2) % 3      # curl_c = d_{a1} F_{a2} - d_{a2} F_{a1}
            d1 = self._pml_h(c, a1, self._fwd(E[a2], a1))
            d2 = self._pml_h(c, a2, self._fwd(E[a1], a2))
            H[c] -= ch * (d1 - d2)
        self._tfsf_h(n)
        for s in self.spec.sources:
            if n >= len(s.wave_h):                   # the list is spent (its waveform ends where the source ends)
                continue
            w = s.w_re + 1j * s.w_im
            for cc in range(3, 6):
                m = s.comp == cc
                if m.any():
                    ijk = s.ijk[m]
                    np.add.at(H[cc - 3], (ijk[:, 2], ijk[:, 1], ijk[:, 0]), self._amp(w[m] * s.wave_h[n]))

    def _amp(self, z: np.ndarray) -> np.ndarray:
        """Injected amplitude: the real part for real fields, the full complex value under Bloch boundaries."""
        return z.astype(self.dtype) if self.bloch is not None else np.real(z).astype(self.dtype)

    def update_e(self, n: int):
        E, H = self.E, self.H
        spec = self.spec
        e_old = [E[c].ravel()[idx].copy() for (c, m, idx, q, _) in self.ade]
        aniso = getattr(spec, "aniso", None) or []
        e_prev = [E[c].copy() for c in range(3)] if aniso else None
        for c in range(3):
            a1, a2 = (c + 1) % 3, (c + 2) % 3
            d1 = self._pml_e(c, a1, self._bwd(H[a2], a1))
            d2 = self._pml_e(c, a2, self._bwd(H[a1], a2))
            E[c] *= self.ca[c]
            E[c] += self.cb[c] * (d1 - d2)
        self._tfsf_e(n)
        for s in spec.sources:
            if n >= len(s.wave_e):
                continue
            w = s.w_re + 1j * s.w_im
            for cc in range(3):
                m = s.comp == cc
                if m.any():
                    ijk = s.ijk[m]
                    np.add.at(E[cc], (ijk[:, 2], ijk[:, 1], ijk[:, 0]), self._amp(w[m] * s.wave_e[n]))
        # fully anisotropic bodies (spec.AnisoSet): dE_a(i) = (dt / eps0) sum_j g(i, j) curl_b(j), the curl recovered from what the
        # update above did to E_b(j) — (E_b^{n+1} - Ca E_b^n) / Cb, zero on PEC nodes — all components from the unpatched values
        if aniso:
            # (the kernels write the wall-tangential E as zero inside the sweep: the curl recovered at a wall node is zero — the
            #  walls are applied here as well, before the coupling reads those nodes)
            self._pec_walls()
            deltas = []
            for st in aniso:
                d = np.zeros(len(st.ijk), dtype=E[0].dtype)
                for slot in range(8):
                    b = st.nbr_comp[slot]
                    j = st.nbr_ijk[:, slot]
                    ok = j[:, 0] >= 0
                    kk, jj, ii = j[ok, 2], j[ok, 1], j[ok, 0]
                    ca_b = self.ca[b][kk, jj, ii] if np.ndim(self.ca[b]) else self.ca[b]
                    cb_b = self.cb[b][kk, jj, ii] if np.ndim(self.cb[b]) else self.cb[b]
                    cb_safe = np.where(cb_b == 0, 1.0, cb_b)
                    curl = np.where(cb_b == 0, 0.0, (E[b][kk, jj, ii] - ca_b * e_prev[b][kk, jj, ii]) / cb_safe)
                    d[ok] += (spec.dt / _EPS0) * st.g[ok, slot] * curl
                deltas.append(d)
            for st, d in zip(aniso, deltas):
                E[st.comp][st.ijk[:, 2], st.ijk[:, 1], st.ijk[:, 0]] += d
        # absorber layers: damp the updated E before the ADE memory term is added
        if self.damp_e is not None:
            for c in range(3):
                E[c] *= self.damp_e[c]
        # ADE memory term + auxiliary update
        for (c, m, idx, q, _), eo in zip(self.ade, e_old):
            kap, bet = self.mt.kap[m][:, None], self.mt.bet[m][:, None]
            flat = E[c].reshape(-1)
            S = np.sum(2.0 * np.real((kap - 1.0) * q[0]), axis=0)
            if self.bloch is not None:            # Re and Im parts are two independent real problems
                S = S + 1j * np.sum(2.0 * np.real((kap - 1.0) * q[1]), axis=0)
            en = flat[idx] - self.mt.cc[m] * S
            flat[idx] = en
            q *= kap[None]
            q[0] += bet * np.real(en + eo)[None, :]
            if self.bloch is not None:
                q[1] += bet * np.imag(en + eo)[None, :]
        self._pec_walls()

    def _pec_walls(self):
        """PEC walls at the min faces (tangential components living on the wall)"""
        for c in range(3):
            for a in range(3):
                if a != c and self.spec.bc[a][0] == BC_PEC:
                    sl = [slice(None)] * 3
                    sl[_ax(a)] = 0
                    self.E[c][tuple(sl)] = 0

    # ------------------------------------------------------------------ TFSF (1-D auxiliary grid)
    def _tfsf_init(self, t):
        return dict(e=np.zeros(t.n_aux + 1, np.float64), h=np.zeros(t.n_aux, np.float64))

    def _tfsf_h(self, n: int):
        """Advance the incident 1-D H (uses e_inc^n), then correct the 3-D H nodes whose
        stencil straddles the TFSF surface with the incident E they miss / have in excess."""
        for t, st in zip(self.spec.tfsf, self.tfsf_state):
            if n >= len(t.wave):                     # spent: no corrections, the incident grid rests
                continue
            e1, h1 = st["e"], st["h"]
            # 3-D corrections use e_inc at t_n (the same time level as E^n in the H update)
            if len(t.h_corr_w):
                vals = t.h_corr_w * e1[t.h_corr_aux]
                for c in range(3):
                    m = t.h_corr_comp == c + 3
                    if m.any():
                        ijk = t.h_corr_ijk[m]
                        np.add.at(self.H[c], (ijk[:, 2], ijk[:, 1], ijk[:, 0]),
                                  vals[m].astype(self.dtype))
            h1 *= t.ah
            h1 -= t.bh * (e1[1:] - e1[:-1])

    def _tfsf_e(self, n: int):
        for t, st in zip(self.spec.tfsf, self.tfsf_state):
            if n >= len(t.wave):
                continue
            e1, h1 = st["e"], st["h"]
            if len(t.e_corr_w):
                vals = t.e_corr_w * h1[t.e_corr_aux]
                for c in range(3):
                    m = t.e_corr_comp == c
                    if m.any():
                        ijk = t.e_corr_ijk[m]
                        np.add.at(self.E[c], (ijk[:, 2], ijk[:, 1], ijk[:, 0]),
                                  vals[m].astype(self.dtype))
            # advance 1-D E: interior nodes (lossy pads at both ends, PEC end nodes), soft source
            e1[1:-1] = t.ae[1:-1] * e1[1:-1] - t.be[1:-1] * (h1[1:] - h1[:-1])
            e1[t.src_cell] += t.wave[n]

    # ------------------------------------------------------------------ monitors
    def _box(self, F: np.ndarray, m) -> np.ndarray:
        return F[m.lo[2]:m.hi[2], m.lo[1]:m.hi[1], m.lo[0]:m.hi[0]]

    def _record(self, n: int, phase: str):
        """phase 'pre' : before the H update (E^n at t_n, first half of H(t_n));
        phase 'post': after the H update (second half of H(t_n); DFT of H at t_n + dt/2)."""
        for im, m in enumerate(self.spec.monitors):
            k = self.mon_count[im]
            if k >= len(m.steps) or m.steps[k] != n:
                continue
            data = self.mon_data[im]
            for ic, c in enumerate(m.comps):
                if m.kind == "time":
                    if c < 3 and phase == "pre":
                        data[k, ic] = self._box(self.E[c], m)
                    elif c >= 3:
                        data[k, ic] += 0.5 * self._box(self.H[c - 3], m)
                else:
                    pe, ph = self.mon_phase[im]
                    if c < 3 and phase == "pre":
                        data[:, ic] += pe[k][:, None, None, None] * self._box(self.E[c], m)
                    elif c >= 3 and phase == "post":
                        data[:, ic] += ph[k][:, None, None, None] * self._box(self.H[c - 3], m)
            if phase == "post":
                self.mon_count[im] += 1

    # ------------------------------------------------------------------ driver
    def energy(self) -> float:
        """W = sum |E|^2 + (mu0/eps0) sum |H|^2 (E^{n}, H^{n-1/2} as they stand after a step)."""
        we = sum(np.sum(np.square(np.abs(e), dtype=np.float64)) for e in self.E)
        wh = sum(np.sum(np.square(np.abs(h), dtype=np.float64)) for h in self.H)
        return float(we + _ETA0_SQ * wh)

    def _mirror_fill(self):
        """PMC on plus faces (SolverSpec.mirror_plus): beyond the wall index N of axis a the fields are the mirror image of
        the inside — components on cell boundaries along a (E_tan, H_a): F[N + 1] = +F[N - 1]; components on cell centres
        (E_a, H_tan): F[N] = -F[N - 1], F[N + 1] = -F[N - 2]."""
        mp = getattr(self.spec, "mirror_plus", None)
        if not mp:
            return
        for a, N in enumerate(mp):
            if N < 0:
                continue
            ax = 2 - a                      # numpy axis of the (z, y, x) arrays
            def sl(i):
                idx = [slice(None)] * 3
                idx[ax] = i
                return tuple(idx)
            for c in range(3):
                for F, is_h in ((self.E[c], False), (self.H[c], True)):
                    on_center = (c == a) != is_h
                    if on_center:
                        F[sl(N)] = -F[sl(N - 1)]
                        F[sl(N + 1)] = -F[sl(N - 2)]
                    else:
                        F[sl(N + 1)] = F[sl(N - 1)]

    def step(self):
        n = self.step_index
        self._record(n, "pre")
        self._mirror_fill()
        if self.damp_h is not None:
            for c in range(3):
                self.H[c] *= self.damp_h[c]
        self.update_h(n)
        self._record(n, "post")
        self.update_e(n)
        self.step_index = n + 1

    def run(self, n_steps: Optional[int] = None, progress=None):
        spec = self.spec
        n_steps = spec.n_steps if n_steps is None else n_steps
        for _ in range(n_steps):
            self.step()
            n = self.step_index
            if spec.decay_every and n % spec.decay_every == 0:
                en = self.energy()
                if not np.isfinite(en):
                    self.diverged = True
                    self.stopped_at = n
                    break
                self.energy_max = max(self.energy_max, en)
                self.decay = en / self.energy_max if self.energy_max > 0 else 1.0
                if progress is not None:
                    progress(n, n * spec.dt, self.decay)
                if spec.shutoff > 0 and n > spec.decay_ref_step and self.decay < spec.shutoff:
                    self.stopped_at = n
                    break
        return self.results()

    def results(self) -> Dict[str, np.ndarray]:
        return {m.name: d for m, d in zip(self.spec.monitors, self.mon_data)}
