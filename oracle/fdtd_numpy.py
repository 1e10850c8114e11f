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
slab, energy conservation).

It consumes the same ``SolverSpec`` and the same fp64 coefficient tables
(``tidy3d_amd.coeffs``) as the HIP engine, so GPU-vs-oracle differences are pure
fp32 round-off.  It also *is* the "naive NumPy curl-loop" CPU baseline that
BASELINE.md section 4 asks to be timed next to the GPU number.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from tidy3d_amd.coeffs import damping_tables, h_coeff, inv_steps, material_table, pml_axis
from tidy3d_amd.spec import BC_PEC, BC_PERIODIC, BC_PMC, SolverSpec


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
        ip, idl = inv_steps(spec)
        self.ip = [_bcast(v.astype(dtype), a) for a, v in enumerate(ip)]
        self.id = [_bcast(v.astype(dtype), a) for a, v in enumerate(idl)]
        self.ch = dtype(h_coeff(spec.dt))
        # absorber layers: per-step damping factors at each component's Yee location
        # (tidy3d_amd.coeffs.damping_tables; Absorber of ref boundary.py:427-476)
        self.damp_e = self.damp_h = None
        dm = damping_tables(spec)
        if dm is not None:
            def factor(c, is_h):
                f = np.ones(shp, np.float64)
                for a in range(3):
                    on_center = (a == c) != is_h
                    f = f * _bcast(dm[a].fc if on_center else dm[a].fb, a)
                return f.astype(self.rdtype)
            self.damp_e = [factor(c, False) for c in range(3)]
            self.damp_h = [factor(c, True) for c in range(3)]
        self.mt = material_table(spec.media, spec.dt)
        if spec.mat_idx is not None:
            self.ca = [self.mt.ca[spec.mat_idx[c]].astype(self.rdtype) for c in range(3)]
            self.cb = [self.mt.cb[spec.mat_idx[c]].astype(self.rdtype) for c in range(3)]
        else:
            self.ca = [self.rdtype(self.mt.ca[1])] * 3
            self.cb = [self.rdtype(self.mt.cb[1])] * 3
        # CPML
        self.pml = [pml_axis(spec, a) for a in range(3)]
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
        for c in range(3):
            a1, a2 = (c + 1) % 3, (c + 2) % 3
            d1 = self._pml_e(c, a1, self._bwd(H[a2], a1))
            d2 = self._pml_e(c, a2, self._bwd(H[a1], a2))
            E[c] *= self.ca[c]
            E[c] += self.cb[c] * (d1 - d2)
        self._tfsf_e(n)
        for s in spec.sources:
            w = s.w_re + 1j * s.w_im
            for cc in range(3):
                m = s.comp == cc
                if m.any():
                    ijk = s.ijk[m]
                    np.add.at(E[cc], (ijk[:, 2], ijk[:, 1], ijk[:, 0]), self._amp(w[m] * s.wave_e[n]))
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
        # PEC walls at the min faces (tangential components living on the wall)
        for c in range(3):
            for a in range(3):
                if a != c and spec.bc[a][0] == BC_PEC:
                    sl = [slice(None)] * 3
                    sl[_ax(a)] = 0
                    E[c][tuple(sl)] = 0

    # ------------------------------------------------------------------ TFSF (1-D auxiliary grid)
    def _tfsf_init(self, t):
        return dict(e=np.zeros(t.n_aux + 1, np.float64), h=np.zeros(t.n_aux, np.float64))

    def _tfsf_h(self, n: int):
        """Advance the incident 1-D H (uses e_inc^n), then correct the 3-D H nodes whose
        stencil straddles the TFSF surface with the incident E they miss / have in excess."""
        for t, st in zip(self.spec.tfsf, self.tfsf_state):
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
                    if c < 3 and phase == "pre":
                        data[:, ic] += m.phase_e[k][:, None, None, None] * self._box(self.E[c], m)
                    elif c >= 3 and phase == "post":
                        data[:, ic] += (m.phase_h[k][:, None, None, None]
                                        * self._box(self.H[c - 3], m))
            if phase == "post":
                self.mon_count[im] += 1

    # ------------------------------------------------------------------ driver
    def energy(self) -> float:
        return float(sum(np.sum(np.square(np.abs(e), dtype=np.float64)) for e in self.E))

    def step(self):
        n = self.step_index
        self._record(n, "pre")
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
