"""Field sources.  PlaneWave (normal incidence) and TFSF (normal incidence: incident grid along the axis, exact; oblique:
incident grid along k_hat with matched numerical dispersion, ``make_tfsf_angled``) -> TfsfSpec (1-D auxiliary incident grid +
surface correction lists); oblique PlaneWave, GaussianBeam / AstigmaticGaussianBeam and CustomFieldSource -> a sheet
of electric and magnetic currents on the source plane (``_sheet_source``: the same surface legs, fed with
an analytic or tabulated incident field instead of the 1-D grid).

ref components/source.py:1090 (PlaneWave), :1204-1257 (TFSF: the box is the total-field region,
the plane wave carries 1 W/um^2 along the injection axis), :966-990 (polarisation vector:
pol_angle = 0 puts E on the first tangential axis in x,y,z order).

Formulation (total-field / scattered-field, Taflove & Hagness ch. 5) reduced to lists:
a node of component F belongs to the total-field (TF) region iff, along every axis, its index lies
in [lo, hi] where the component sits on cell boundaries and in [lo, hi-1] where it sits on cell
centres.  For every curl term that couples the updated node to a node of the incident components
(E_e, H_q) the update must see *total* field if the updated node is TF and *scattered* field
otherwise; the stored neighbour is total iff it is TF.  Hence each stencil leg gets
``(in_TF(updated) - in_TF(neighbour)) * incident(neighbour)`` times the leg's signed update
coefficient.  Legs with a non-zero factor exist only on the box surface; they are enumerated on
the host, once.

The incident field lives on a 1-D Yee grid that shares the 3-D grid's steps along the propagation
axis (so numerical dispersion matches exactly), extended beyond the domain and terminated by
matched lossy pads.  Its E_e / H_q are the physical components (signs included):
  eps dE_e/dt = s dH_q/dp,  mu dH_q/dt = s dE_e/dp,   s = +1 if (e, p, q) is cyclic else -1.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np

from . import schema as td
from .constants import C_0, EPSILON_0, ETA_0, MU_0
from .exceptions import SetupError, Tidy3dNotImplementedError
from .spec import SolverSpec, TfsfSpec

N_PAD = 40          # cells of each lossy pad of the auxiliary grid
N_EXT = 6           # loss-free extension cells between pad and domain (the soft source sits there)


def _aux_grid(spec: SolverSpec, p: int, s: float, eps_bg: float):
    """Coefficient arrays of the padded 1-D grid along axis p.  Returns (n_aux, off, ae, be, ah,
    bh, d_primal) with ``off`` = aux index of 3-D boundary index 0."""
    d = spec.primal_steps(p)
    off = N_PAD + N_EXT
    dp = np.concatenate([np.full(off, d[0]), d, np.full(off, d[-1])])       # primal steps
    n = len(dp)
    dd = np.empty(n + 1)
    dd[1:-1] = 0.5 * (dp[1:] + dp[:-1])
    dd[0], dd[-1] = dp[0], dp[-1]
    dt = spec.dt
    # matched loss: sigma/eps = sigma*/mu = g(x), cubic grading, g_max chosen for ~1e-8 round trip
    def g_at(pos):          # pos: index coordinate (boundaries = integers, centres = +0.5)
        depth = np.maximum(N_PAD - pos, pos - (n - N_PAD)) / N_PAD
        depth = np.clip(depth, 0.0, 1.0)
        gmax = 4.0 * C_0 / (np.sqrt(eps_bg) * N_PAD * d[0]) * np.log(1e4) / 2.0
        return gmax * depth ** 3
    ge = g_at(np.arange(n + 1, dtype=float))
    gh = g_at(np.arange(n, dtype=float) + 0.5)
    xe, xh = ge * dt / 2, gh * dt / 2
    ae = (1 - xe) / (1 + xe)
    ah = (1 - xh) / (1 + xh)
    # e1 -= ce * (h[i]-h[i-1]) / dd  with  ce = -s dt/(eps0 eps_bg)   (see module docstring)
    be = (-s * dt / (EPSILON_0 * eps_bg)) / dd / (1 + xe)
    bh = (-s * dt / MU_0) / dp / (1 + xh)
    return n, off, ae, be, ah, bh, dp


def replay_aux(t: TfsfSpec, n_steps: int, probe: int) -> np.ndarray:
    """Host replay of the 1-D incident grid: e1[probe] at t_n for n = 0..n_steps-1 (what the
    3-D H-phase corrections read).  Used for source normalisation and by tests."""
    e1 = np.zeros(t.n_aux + 1)
    h1 = np.zeros(t.n_aux)
    out = np.empty(n_steps)
    for n in range(n_steps):
        out[n] = e1[probe]
        h1 *= t.ah
        h1 -= t.bh * (e1[1:] - e1[:-1])
        e1[1:-1] = t.ae[1:-1] * e1[1:-1] - t.be[1:-1] * (h1[1:] - h1[:-1])
        e1[t.src_cell] += t.wave[n]
    return out


def _mask_1d(n: int, lo: int, hi: int, on_center: bool) -> np.ndarray:
    """TF membership along one axis for node indices 0..n-1 (lo/hi may lie outside the grid)."""
    idx = np.arange(n)
    return (idx >= lo) & (idx <= (hi - 1 if on_center else hi))


def _on_center(comp: int, axis: int) -> bool:
    return ((comp % 3) == axis) != (comp >= 3)


def surface_legs(spec: SolverSpec, mt, lo, hi, inc_e, inc_h):
    """Enumerate the stencil legs that straddle the TF/SF surface of the region [lo, hi] for an
    incident field whose non-zero components are ``inc_e`` (E axes) and ``inc_h`` (H axes).
    Returns {"e": (comp, ijk, w, nb_comp, nb_ijk), "h": (...)}: E-phase legs add
    w * H_inc[nb_comp](nb_ijk) to E[comp](ijk); H-phase legs add w * E_inc[nb_comp](nb_ijk)."""
    from .coeffs import h_coeff, inv_steps
    from .discretize import cb_at
    N = spec.shape
    ip, idl = inv_steps(spec)
    ch = h_coeff(spec.dt)
    out = {"e": [], "h": []}

    def in_tf(comp, ijk):
        m = np.ones(len(ijk), bool)
        for a in range(3):
            oc = _on_center(comp, a)
            m &= (ijk[:, a] >= lo[a]) & (ijk[:, a] <= (hi[a] - 1 if oc else hi[a]))
        return m

    _cand_cache = {}

    def candidates(comp):
        """Nodes of `comp` within one cell of the box surface (union of 6 face slabs), clipped.  (The node set does not
        depend on the component — only the box does — and both curl terms of a component ask for it: built once.)"""
        if "nodes" not in _cand_cache:
            _cand_cache["nodes"] = _candidates()
        return _cand_cache["nodes"]

    def _candidates():
        rng = [np.arange(max(lo[a] - 1, 0), min(hi[a] + 2, N[a])) for a in range(3)]
        pts = []
        for a in range(3):
            for face in (lo[a], hi[a]):
                sl = [r for r in rng]
                sl[a] = np.arange(max(face - 1, 0), min(face + 2, N[a]))
                if any(len(x) == 0 for x in sl):
                    continue
                I, J, K = np.meshgrid(*sl, indexing="ij")
                pts.append(np.stack([I.ravel(), J.ravel(), K.ravel()], axis=1))
        if not pts:
            return np.zeros((0, 3), int)
        # union of the slabs through ONE integer key per node (np.unique over rows sorts structured records:
        # 14 s of a 10 s set-up for a plane wave across a 1024 x 1024 period)
        allp = np.concatenate(pts)
        key = np.unique((allp[:, 0].astype(np.int64) * N[1] + allp[:, 1]) * N[2] + allp[:, 2])
        out_ = np.empty((key.size, 3), dtype=allp.dtype)
        out_[:, 2] = key % N[2]
        key //= N[2]
        out_[:, 1] = key % N[1]
        out_[:, 0] = key // N[1]
        return out_

    # Membership in the total-field region factorises per axis: one small boolean table per (component, axis) over the indices
    # -1 .. N (a shifted neighbour may lie one node outside the grid), looked up through contiguous index columns — the candidate
    # set of a plane wave across a 1024 x 1024 period holds nine million nodes, and the six comparisons per axis and component on
    # strided int64 columns, the copies of the shifted node lists and the Cb look-up of EVERY candidate were most of its set-up time.
    tabs = {}

    def table(comp):
        if comp not in tabs:
            t_ = []
            for a_ in range(3):
                i_ = np.arange(-1, N[a_] + 1)
                t_.append((i_ >= lo[a_]) & (i_ <= (hi[a_] - 1 if _on_center(comp, a_) else hi[a_])))
            tabs[comp] = t_
        return tabs[comp]

    def member(comp, cols, axis=None, shift=0):
        t_ = table(comp)
        m = None
        for a_ in range(3):
            v = t_[a_][cols[a_] + (1 + shift if a_ == axis else 1)]
            m = v if m is None else m & v
        return m

    cols_cache = {}

    def columns(nodes):
        if "c" not in cols_cache:
            cols_cache["c"] = [np.ascontiguousarray(nodes[:, a_]) for a_ in range(3)]
        return cols_cache["c"]

    # ---- E-phase: E_c += Cb * sgn * (H_q[n] - H_q[n - e_a]) / dual_a for the term d_a H_q in (curl H)_c
    from .spec import BC_PEC
    for c in range(3):
        for (a, f, sgn) in (((c + 1) % 3, (c + 2) % 3, 1.0), ((c + 2) % 3, (c + 1) % 3, -1.0)):
            if f not in inc_h:
                continue
            nodes = candidates(c)
            if len(nodes) == 0:
                continue
            cols = columns(nodes)
            in_e = member(c, cols)
            # wall nodes of E are forced to zero by the main kernels: never correct them
            wall = np.zeros(len(nodes), bool)
            for b in range(3):
                if b != c and spec.bc[b][0] == BC_PEC:
                    wall |= cols[b] == 0
            for shift, leg_sign in ((0, 1.0), (-1, -1.0)):
                na = cols[a] + shift
                sel = (in_e != member(3 + f, cols, a, shift)) & (na >= 0) & (na < N[a]) & ~wall
                if not sel.any():
                    continue
                rows = np.flatnonzero(sel)
                here = nodes[rows]
                nb = here.copy()
                nb[:, a] += shift
                fac = in_e[rows].astype(float) - in_tf(3 + f, nb).astype(float)
                w = cb_at(spec, mt, c, here) * sgn * leg_sign * idl[a][here[:, a]] * fac
                out["e"].append((np.full(rows.size, c), here, w, np.full(rows.size, 3 + f), nb))
    # ---- H-phase: H_c -= ch * sgn * (E_e[n + e_a] - E_e[n]) / primal_a for the term d_a E_e in (curl E)_c
    for c in range(3):
        for (a, f, sgn) in (((c + 1) % 3, (c + 2) % 3, 1.0), ((c + 2) % 3, (c + 1) % 3, -1.0)):
            if f not in inc_e:
                continue
            nodes = candidates(3 + c)
            if len(nodes) == 0:
                continue
            cols = columns(nodes)
            in_h = member(3 + c, cols)
            for shift, leg_sign in ((1, 1.0), (0, -1.0)):
                na = cols[a] + shift
                sel = (in_h != member(f, cols, a, shift)) & (na >= 0) & (na < N[a])
                if not sel.any():
                    continue
                rows = np.flatnonzero(sel)
                here = nodes[rows]
                nb = here.copy()
                nb[:, a] += shift
                fac = in_h[rows].astype(float) - in_tf(f, nb).astype(float)
                w = -ch * sgn * leg_sign * ip[a][here[:, a]] * fac
                out["h"].append((np.full(rows.size, 3 + c), here, w, np.full(rows.size, f), nb))
    res = {}
    for k in ("e", "h"):
        if out[k]:
            comp = np.concatenate([x[0] for x in out[k]]).astype(np.int32)
            ijk = np.concatenate([x[1] for x in out[k]]).astype(np.int32)
            w = np.concatenate([x[2] for x in out[k]])
            nbc = np.concatenate([x[3] for x in out[k]]).astype(np.int32)
            nbi = np.concatenate([x[4] for x in out[k]]).astype(np.int32)
        else:
            comp, ijk, w, nbc, nbi = (np.zeros(0, np.int32), np.zeros((0, 3), np.int32), np.zeros(0),
                                      np.zeros(0, np.int32), np.zeros((0, 3), np.int32))
        res[k] = (comp, ijk, w, nbc, nbi)
    return res


def _legs(spec: SolverSpec, mt, lo, hi, p: int, e: int, q: int, off: int):
    """Plane-wave specialisation: incident components (E_e, H_q) read from the 1-D grid along p."""
    legs = surface_legs(spec, mt, lo, hi, (e,), (q,))
    return {k: (v[0], v[1], v[2], (v[4][:, p] + off).astype(np.int32)) for k, v in legs.items()}


def make_tfsf(spec: SolverSpec, mt, lo, hi, p: int, direction: int, e: int, e_scale: float,
              wave_fn: Callable[[np.ndarray], np.ndarray], eps_bg: float, name: str = "") -> Tuple[TfsfSpec, int]:
    """One linearly polarised, axis-aligned plane wave.  ``wave_fn(t)`` is the desired incident
    E_e(t) at the injection face (before scaling); returns (TfsfSpec, aux index of that face)."""
    q = 3 - p - e
    s = 1.0 if (e + 1) % 3 == p else -1.0
    n_aux, off, ae, be, ah, bh, dp = _aux_grid(spec, p, s, eps_bg)
    N = spec.shape[p]
    if direction > 0:
        src = off - 3
        face = max(lo[p], 0) + off
    else:
        src = off + N + 3
        face = min(hi[p], N) + off
    # soft source: a sheet adding w per step radiates E = w * d/(2 v dt) to each side
    v = C_0 / np.sqrt(eps_bg)
    d_src = 0.5 * (dp[src] + dp[src - 1])
    # delay so that the pulse arrives at the face with the requested waveform timing
    dist = abs(np.sum(dp[min(src, face):max(src, face)]))
    tmesh = spec.dt * np.arange(spec.n_steps)
    wave = e_scale * wave_fn(tmesh + spec.dt + 0 * dist) * (2 * v * spec.dt / d_src)
    legs = _legs(spec, mt, lo, hi, p, e, q, off)
    # the H-phase legs read e1 (boundary-located along p): aux index = boundary index + off  (done)
    # the E-phase legs read h1 (centre-located along p):   aux index = centre index + off    (done)
    t = TfsfSpec(n_aux=n_aux, ae=ae, be=be, ah=ah, bh=bh, src_cell=int(src), wave=wave,
                 e_corr_comp=legs["e"][0], e_corr_ijk=legs["e"][1], e_corr_w=legs["e"][2],
                 e_corr_aux=legs["e"][3], h_corr_comp=legs["h"][0], h_corr_ijk=legs["h"][1],
                 h_corr_w=legs["h"][2], h_corr_aux=legs["h"][3], name=name)
    return t, int(face)


def numerical_wavenumber(omega: float, dt: float, u: np.ndarray, dl, v: float) -> float:
    """k~ of a plane wave along the unit vector ``u`` ON THE YEE GRID (steps ``dl``, speed ``v`` in the background):
        sin^2(omega dt / 2) / (v dt)^2 = sum_a sin^2(k~ u_a dl_a / 2) / dl_a^2          (Newton from k = omega / v)."""
    rhs = (np.sin(omega * dt / 2) / (v * dt)) ** 2
    u, dl = np.asarray(u, float), np.asarray(dl, float)
    k = omega / v
    for _ in range(50):
        f = np.sum(np.sin(k * u * dl / 2) ** 2 / dl ** 2) - rhs
        df = np.sum(np.sin(k * u * dl) * u / (2 * dl))
        step = f / df
        k -= step
        if abs(step) < 1e-14 * k:
            break
    return float(k)


def matched_aux_step(omega: float, dt: float, k_num: float, v: float, d_max: float) -> float:
    """Step of a 1-D Yee grid (same dt) whose numerical wavenumber at ``omega`` equals ``k_num``:
        sin(k~ d / 2) / d = sin(omega dt / 2) / (v dt)   (bisection; the left side falls monotonically for k~ d < pi)."""
    rhs = np.sin(omega * dt / 2) / (v * dt)
    lo, hi = v * dt, 4.0 * d_max
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if np.sin(k_num * mid / 2) / mid > rhs:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def _lagrange4(fr: np.ndarray):
    """weights of the cubic through the samples at -1, 0, 1, 2 evaluated at fr in [0, 1)"""
    return (-fr * (fr - 1) * (fr - 2) / 6, (fr + 1) * (fr - 1) * (fr - 2) / 2, -(fr + 1) * fr * (fr - 2) / 2, (fr + 1) * fr * (fr - 1) / 6)


def make_tfsf_angled(spec: SolverSpec, mt, lo, hi, k_hat: np.ndarray, e_hat: np.ndarray, e_scale: float,
                     wave_fn: Callable[[np.ndarray], np.ndarray], eps_bg: float, freq0: float, ref_point,
                     name: str = "") -> Tuple[TfsfSpec, int]:
    """A plane wave at OBLIQUE incidence into a TFSF box (ref source.py:1204 ``TFSF(AngledFieldSource, ...)``, angles
    :899-990).  The incident wave still lives on a 1-D auxiliary Yee grid — along the propagation direction now, with a
    uniform step chosen so that its numerical wavenumber at the centre frequency equals that of the 3-D grid along k_hat
    (matched numerical dispersion, Taflove & Hagness 5.9) — and every stencil leg across the box surface reads it at
    zeta = k_hat . (r - r_entry) by cubic interpolation: four (weight, aux index) entries per leg, polarisation folded into
    the weight, for the SAME device machinery as normal incidence (nothing new in the kernels).
        E_inc = e_hat E_s(zeta),   H_inc = (e_hat x k_hat) H_s(zeta)     [(E_s, H_s): the s = +1 pair of the module docstring]
    Needs a uniform 3-D grid around the box.  Returns (TfsfSpec, aux index nearest ``ref_point``)."""
    N = spec.shape
    b = [np.asarray(x, float) for x in spec.boundaries]
    dl = []
    for a in range(3):
        i0, i1 = max(lo[a] - 2, 0), min(hi[a] + 2, N[a])
        d = np.diff(b[a][i0:i1 + 1])
        if d.size and np.ptp(d) > 1e-9 * d.mean():
            raise Tidy3dNotImplementedError("an angled TFSF source needs a uniform grid around its box")
        dl.append(float(d.mean()) if d.size else float(np.diff(b[a]).mean()))
    v = C_0 / np.sqrt(eps_bg)
    dt = spec.dt
    omega = 2 * np.pi * freq0
    k_num = numerical_wavenumber(omega, dt, k_hat, dl, v)
    d_aux = matched_aux_step(omega, dt, k_num, v, max(dl))
    h_hat = np.cross(e_hat, k_hat)
    # entry corner of the (clipped) box: the corner with the smallest projection on k_hat
    cl = [b[a][int(np.clip(lo[a], 0, N[a]))] for a in range(3)]
    ch = [b[a][int(np.clip(hi[a], 0, N[a]))] for a in range(3)]
    r_entry = np.array([cl[a] if k_hat[a] >= 0 else ch[a] for a in range(3)])
    r_exit = np.array([ch[a] if k_hat[a] >= 0 else cl[a] for a in range(3)])
    span = float(k_hat @ (r_exit - r_entry))
    off = N_PAD + N_EXT + 4                                  # aux index of zeta = 0 (4 nodes of room for the cubic and the legs one cell outside)
    n = off + int(np.ceil((span + 3 * max(dl)) / d_aux)) + 4 + N_PAD + N_EXT
    # uniform padded 1-D grid (s = +1), matched lossy pads as in _aux_grid
    def g_at(pos):
        depth = np.clip(np.maximum(N_PAD - pos, pos - (n - N_PAD)) / N_PAD, 0.0, 1.0)
        return 4.0 * v / (N_PAD * d_aux) * np.log(1e4) / 2.0 * depth ** 3
    xe, xh = g_at(np.arange(n + 1, dtype=float)) * dt / 2, g_at(np.arange(n, dtype=float) + 0.5) * dt / 2
    ae, ah = (1 - xe) / (1 + xe), (1 - xh) / (1 + xh)
    be = (-dt / (EPSILON_0 * eps_bg)) / d_aux / (1 + xe)
    bh = (-dt / MU_0) / d_aux / (1 + xh)
    src = off - 6
    tmesh = dt * np.arange(spec.n_steps)
    wave = e_scale * wave_fn(tmesh + dt) * (2 * v * dt / d_aux)
    inc_e = tuple(a for a in range(3) if abs(e_hat[a]) > 1e-12)
    inc_h = tuple(a for a in range(3) if abs(h_hat[a]) > 1e-12)
    legs = surface_legs(spec, mt, lo, hi, inc_e, inc_h)
    out = {}
    for key in ("e", "h"):
        comp, ijk, w, nbc, nbi = legs[key]
        pol = np.zeros(len(w))
        zeta = np.zeros(len(w))
        for c in np.unique(nbc):
            m = nbc == c
            xs = spec.yee_coords(int(c))
            r = np.stack([xs[a][nbi[m, a]] for a in range(3)], axis=1)
            zeta[m] = (r - r_entry[None, :]) @ k_hat
            pol[m] = h_hat[c - 3] if c >= 3 else e_hat[c]
        # E-phase legs read H_s (h1[m] sits at (m + 1/2) d_aux), H-phase legs read E_s (e1[m] at m d_aux)
        pos = zeta / d_aux + off - (0.5 if key == "e" else 0.0)
        m0 = np.floor(pos).astype(np.int64)
        wts = _lagrange4(pos - m0)
        comp4 = np.repeat(comp, 4)
        ijk4 = np.repeat(ijk, 4, axis=0)
        w4 = (np.stack(wts, axis=1) * (w * pol)[:, None]).reshape(-1)
        aux4 = (m0[:, None] + np.arange(-1, 3)[None, :]).reshape(-1)
        keep = w4 != 0
        out[key] = (comp4[keep].astype(np.int32), ijk4[keep].astype(np.int32), w4[keep], aux4[keep].astype(np.int32))
        if keep.any() and (aux4[keep].min() < src + 2 or aux4[keep].max() > n - N_PAD - 1):
            raise SetupError("angled TFSF: the incident grid does not cover the box")           # (internal consistency)
    t = TfsfSpec(n_aux=n, ae=ae, be=be, ah=ah, bh=bh, src_cell=int(src), wave=wave,
                 e_corr_comp=out["e"][0], e_corr_ijk=out["e"][1], e_corr_w=out["e"][2], e_corr_aux=out["e"][3],
                 h_corr_comp=out["h"][0], h_corr_ijk=out["h"][1], h_corr_w=out["h"][2], h_corr_aux=out["h"][3], name=name)
    ref_idx = int(round(float((np.asarray(ref_point, float) - r_entry) @ k_hat) / d_aux)) + off
    return t, int(np.clip(ref_idx, src + 1, n - N_PAD - 1))


def _rot(vec: np.ndarray, axis: int, angle: float) -> np.ndarray:
    """Right-handed rotation about a coordinate axis (ref geometry/base.py rotate_points)."""
    c, s = np.cos(angle), np.sin(angle)
    a, b = [(1, 2), (2, 0), (0, 1)][axis]
    out = np.array(vec, float)
    out[a], out[b] = c * vec[a] - s * vec[b], s * vec[a] + c * vec[b]
    return out


def direction_vectors(src) -> Tuple[np.ndarray, np.ndarray]:
    """(propagation unit vector, E polarisation unit vector) of an angled field source in x, y, z
    (ref source.py:948-990 ``_dir_vector`` / ``_pol_vector``: built for injection along z, then moved
    to the injection axis with unpop_axis; pol_angle = 0 is P polarisation)."""
    p = int(src.injection_axis)
    sgn = 1.0 if src.direction == "+" else -1.0
    th, ph = float(src.angle_theta), float(src.angle_phi)
    d_loc = sgn * np.array([np.cos(ph) * np.sin(th), np.sin(ph) * np.sin(th), np.cos(th)])
    e_loc = _rot(_rot(_rot(np.array([1.0, 0.0, 0.0]), 2, float(src.pol_angle)), 1, th), 2, ph)

    def unpop(v):
        out = [v[0], v[1]]
        out.insert(p, v[2])
        return np.array(out)
    return unpop(d_loc), unpop(e_loc)


def _sheet_source(disc, mt, src, incident, name: str) -> Callable:
    """Launch an incident field from the source plane with a sheet of electric and magnetic currents:
    the total-field/scattered-field legs of the plane (``surface_legs``; total field on the side the
    source points to) weighted with ``incident(comp, x, y, z)`` — the complex amplitude of the incident
    E (comp 0-2) or H (comp 3-5) component at the leg's own Yee location — times the source time
    (E at t_n, H at t_n + dt/2).  Points outside the source rectangle carry nothing."""
    from .spec import PointSourceSet
    spec, tmesh = disc.spec, disc.tmesh
    st = src.source_time
    p = int(src.injection_axis)
    u, v = [a for a in range(3) if a != p]
    b = spec.boundaries
    big = 10 ** 9
    lo, hi = [-big] * 3, [big] * 3
    face = int(np.argmin(np.abs(b[p] - src.center[p])))
    if src.direction == "+":
        lo[p] = face
    else:
        hi[p] = face
    legs = surface_legs(spec, mt, lo, hi, (u, v), (u, v))
    comps, ijks, ws = [], [], []
    for key in ("e", "h"):
        comp, ijk, w, nbc, nbi = legs[key]
        val = np.zeros(len(w), complex)
        for c in np.unique(nbc):
            m = nbc == c
            xs = spec.yee_coords(int(c))
            pos = [xs[a][nbi[m, a]] for a in range(3)]
            inside = np.ones(m.sum(), bool)
            for a in (u, v):
                if np.isfinite(src.size[a]) and spec.shape[a] > 1:
                    inside &= np.abs(pos[a] - src.center[a]) <= src.size[a] / 2
            val[m] = np.where(inside, incident(int(c), *pos), 0.0)
        keep = val != 0
        comps.append(comp[keep])
        ijks.append(ijk[keep])
        ws.append(w[keep] * val[keep])
    comp = np.concatenate(comps).astype(np.int32)
    ijk = np.concatenate(ijks).astype(np.int32)
    w = np.concatenate(ws)
    dt = spec.dt
    spec.sources.append(PointSourceSet(
        comp=comp, ijk=ijk, w_re=w.real.copy(), w_im=w.imag.copy(),
        wave_e=np.asarray(st.amp_time(tmesh + dt / 2), complex),
        wave_h=np.asarray(st.amp_time(tmesh), complex), name=getattr(src, "name", None) or name))

    def fn(freqs):
        return st.spectrum(tmesh, np.asarray(freqs, float), dt, complex_fields=spec.bloch is not None)
    return fn


def _background_index(spec: SolverSpec) -> float:
    if spec.media[1].poles or spec.media[1].sigma:
        raise Tidy3dNotImplementedError("field-source injection needs a lossless, dispersionless background")
    return float(np.sqrt(np.real(spec.media[1].eps_inf)))


def build_angled_planewave(disc, mt, src) -> Callable:
    """PlaneWave at oblique incidence (or any PlaneWave under Bloch boundaries): the sheet source of
    ``_sheet_source`` fed with the incident plane wave of the centre frequency,

        E_inc(r, t) = E0 exp(i k . (r - r0)) amp(t),   H_inc = k_hat x E_inc / eta,   |k| = 2 pi f0 n / c.

    Like the reference's angled sources the in-plane wave vector is fixed by the centre frequency (ref
    boundary.py:84-88: "only the frequency components near the center frequency will exhibit angled
    incidence at the expected angle"), which is also what the Bloch boundaries impose.  Amplitude as for
    normal incidence: 1 W/um^2 along the propagation direction for a unit-amplitude source time (ref
    source.py:1210-1214)."""
    from .spec import BC_PERIODIC
    sim, spec = disc.sim, disc.spec
    st = src.source_time
    p = int(src.injection_axis)
    u, v = [a for a in range(3) if a != p]
    b = spec.boundaries
    for a in (u, v):
        if np.isfinite(src.size[a]) and src.size[a] < (b[a][-1] - b[a][0]) * 0.999 and spec.shape[a] > 1:
            raise Tidy3dNotImplementedError("an angled PlaneWave must span the whole cross-section of the domain")
    n_bg = _background_index(spec)
    k_hat, e_hat = direction_vectors(src)
    kvec = 2 * np.pi * st.freq0 * n_bg / C_0 * k_hat
    # the transverse wave vector must be what the boundaries impose (ref simulation.py:2309-2389)
    for a in (u, v):
        if spec.shape[a] == 1 and sim.size[a] == 0:
            continue
        L = b[a][-1] - b[a][0]
        if spec.bc[a][0] == BC_PERIODIC:
            want = kvec[a] * L
            have = spec.bloch[a] if spec.bloch is not None else 0.0
            if abs(np.angle(np.exp(1j * (want - have)))) > 1e-3:
                raise SetupError(
                    f"The Bloch vector along axis {a} does not match the angled PlaneWave: expected bloch_vec = "
                    f"{want / (2 * np.pi):.6g} (+ integer), see BlochBoundary.from_source (ref boundary.py:81).")
        elif abs(kvec[a]) > 1e-12:
            raise SetupError("An angled PlaneWave needs Bloch boundaries along the axes in which it is tilted "
                             "(ref simulation.py:2309-2389).")
    E0 = np.sqrt(2 * ETA_0 / n_bg) * e_hat                # 1 W/um^2 along k for amplitude 1
    H0 = np.cross(k_hat, E0) * n_bg / ETA_0
    r0 = np.array([float(c) if np.isfinite(c) else 0.0 for c in src.center])
    r0[p] = b[p][int(np.argmin(np.abs(b[p] - src.center[p])))]

    def incident(c, x, y, z):
        amp = E0[c] if c < 3 else H0[c - 3]
        return amp * np.exp(1j * (kvec[0] * (x - r0[0]) + kvec[1] * (y - r0[1]) + kvec[2] * (z - r0[2])))
    return _sheet_source(disc, mt, src, incident, "PlaneWave")


def build_gaussian_beam(disc, mt, src) -> Callable:
    """GaussianBeam / AstigmaticGaussianBeam (ref source.py:1109-1201): the paraxial (simple astigmatic)
    Gaussian beam of the centre frequency on the source plane, launched by ``_sheet_source``.  In beam
    coordinates (zeta along the propagation direction, x' along the P-polarisation vector of
    ``direction_vectors`` at pol_angle = 0, y' = k_hat x x'), per transverse axis q with waist w0_q at
    zeta_q = zeta + waist_distance_q (a positive waist distance puts the waist behind the source):

        E = E0 prod_q sqrt(w0_q / w_q) exp(-q^2 / w_q^2) exp(i (k q^2 / (2 R_q) - psi_q / 2)) * exp(i k zeta)
        w_q = w0_q sqrt(1 + (zeta_q / zR_q)^2),  R_q = zeta_q (1 + (zR_q / zeta_q)^2),  psi_q = atan(zeta_q / zR_q),
        zR_q = k w0_q^2 / 2,   H = k_hat x E / eta.

    Amplitude: the reference's normalisation of beams is server-side (parity unpinned); here the peak
    intensity at a circular waist is 1 W/um^2 for a unit-amplitude source time, i.e. the beam carries
    pi w0x w0y / 2 W."""
    import dataclasses
    spec = disc.spec
    st = src.source_time
    n_bg = _background_index(spec)
    k_hat, e_hat = direction_vectors(src)
    x_hat = direction_vectors(dataclasses.replace(src, pol_angle=0.0))[1]
    y_hat = np.cross(k_hat, x_hat)
    if src.direction == "-":
        y_hat = -y_hat                                  # keep (x', y', propagation) as built for '+'
    k = 2 * np.pi * st.freq0 * n_bg / C_0
    if isinstance(src, td.AstigmaticGaussianBeam):
        w0, wd = [float(v) for v in src.waist_sizes], [float(v) for v in src.waist_distances]
    else:
        w0, wd = [float(src.waist_radius)] * 2, [float(src.waist_distance)] * 2
    E0 = np.sqrt(2 * ETA_0 / n_bg) * e_hat
    H0 = np.cross(k_hat, E0) * n_bg / ETA_0
    r0 = np.array([float(c) for c in src.center])

    def incident(c, x, y, z):
        s = np.stack([x - r0[0], y - r0[1], z - r0[2]])
        zeta = k_hat @ s
        out = np.exp(1j * k * zeta).astype(complex)
        for q_hat, w0q, wdq in ((x_hat, w0[0], wd[0]), (y_hat, w0[1], wd[1])):
            q = q_hat @ s
            zq, zr = zeta + wdq, k * w0q ** 2 / 2
            wq = w0q * np.sqrt(1 + (zq / zr) ** 2)
            inv_r = zq / (zq ** 2 + zr ** 2)                            # 1 / R_q, regular at the waist
            out = out * np.sqrt(w0q / wq) * np.exp(-q ** 2 / wq ** 2 + 1j * (k * q ** 2 * inv_r / 2 - np.arctan(zq / zr) / 2))
        return (E0[c] if c < 3 else H0[c - 3]) * out
    return _sheet_source(disc, mt, src, incident, src.type)


def build_custom_field_source(disc, mt, src) -> Callable:
    """CustomFieldSource (ref source.py:781-900): the tangential E and H of the dataset (coordinates relative
    to the source centre, entry nearest the centre frequency) become the equivalent currents J = n x H,
    M = -n x E of the plane, n = +axis — the jump conditions ``_sheet_source`` imposes with the data as the
    incident field.  Data of a wave travelling along +n are reproduced on the + side and nothing is sent
    back; E-only or H-only data radiate to both sides."""
    ds = td._need_data(src.field_dataset, "CustomFieldSource")
    p = int(src.injection_axis)
    comps = {}
    for name, arr in ds.field_components.items():
        c = "xyz".index(name[1]) + (3 if name[0] == "H" else 0)
        if c % 3 == p:
            continue                                     # normal components play no role (ref :792-795)
        if "f" in arr.dims:
            f = np.asarray(arr.coords["f"], float)
            arr = arr.sel(f=float(f[np.argmin(np.abs(f - src.source_time.freq0))]))
        comps[c] = arr
    if not comps:
        raise SetupError("CustomFieldSource needs at least one field component tangential to its plane "
                         "(ref source.py:792-795).")

    def incident(c, x, y, z):
        if c not in comps:
            return np.zeros(np.shape(x), complex)
        pts = {"x": x - src.center[0], "y": y - src.center[1], "z": z - src.center[2]}
        return td.interp_dataset(comps[c], pts, "linear").astype(complex)
    return _sheet_source(disc, mt, src, incident, "CustomFieldSource")


def build_planewave(disc, mt, src) -> Callable:
    """Discretise a PlaneWave or TFSF source; returns its normalisation spectrum function."""
    sim, spec = disc.sim, disc.spec
    if isinstance(src, td.PlaneWave) and (src.angle_theta != 0.0 or spec.bloch is not None):
        return build_angled_planewave(disc, mt, src)
    is_box = isinstance(src, td.TFSF)
    p = int(src.injection_axis)
    direction = 1 if src.direction == "+" else -1
    N = spec.shape
    b = spec.boundaries
    big = 10 ** 9
    if is_box:
        lo, hi = [], []
        for a in range(3):
            c0, c1 = src.center[a] - src.size[a] / 2, src.center[a] + src.size[a] / 2
            if not np.isfinite(src.size[a]):
                lo.append(-big), hi.append(big)
                continue
            lo.append(int(np.argmin(np.abs(b[a] - c0))))
            hi.append(int(np.argmin(np.abs(b[a] - c1))))
            if hi[-1] - lo[-1] < 1:
                raise SetupError("TFSF box is thinner than one cell")
    else:
        lo, hi = [-big] * 3, [big] * 3
        face = int(np.argmin(np.abs(b[p] - src.center[p])))
        if direction > 0:
            lo[p] = face
        else:
            hi[p] = face
        for a in range(3):
            if a != p and np.isfinite(src.size[a]) and src.size[a] < (b[a][-1] - b[a][0]) * 0.999:
                c0, c1 = src.center[a] - src.size[a] / 2, src.center[a] + src.size[a] / 2
                lo[a] = int(np.argmin(np.abs(b[a] - c0)))
                hi[a] = int(np.argmin(np.abs(b[a] - c1)))
    eps_bg = float(np.real(spec.media[1].eps_inf))
    if spec.media[1].poles or spec.media[1].sigma:
        raise Tidy3dNotImplementedError("plane-wave injection needs a lossless, dispersionless background")
    tang = [a for a in range(3) if a != p]
    st = src.source_time
    pol = float(src.pol_angle)
    # 1 W/um^2 for amplitude 1: |E0|^2 = 2 eta / n  (ref source.py:1210-1214)
    e_unit = np.sqrt(2 * ETA_0 / np.sqrt(eps_bg))
    if src.angle_theta != 0.0:
        # oblique incidence into the box (ref source.py:1204, angles :899-990): one incident grid along k_hat
        k_hat, e_hat = direction_vectors(src)
        for a in range(3):
            if a != p and abs(k_hat[a]) > 1e-9 and not np.isfinite(src.size[a]):
                raise Tidy3dNotImplementedError("an angled TFSF source must be finite along the axes in which it is tilted "
                                                "(an infinite extent needs Bloch boundaries: use an angled PlaneWave)")
        ref = [float(c) if np.isfinite(c) else 0.0 for c in src.center]
        ref[p] = b[p][int(np.clip(lo[p] if direction > 0 else hi[p], 0, N[p]))]
        t, face_aux = make_tfsf_angled(spec, mt, lo, hi, k_hat, e_hat, e_unit, lambda tt: np.real(st.amp_time(tt)), eps_bg,
                                       float(st.freq0), ref, name=getattr(src, "name", None) or src.type)
        spec.tfsf.append(t)
        inc_a = replay_aux(t, spec.n_steps, face_aux) / e_unit
        tmesh_a = disc.tmesh

        def fn_angled(freqs):
            freqs = np.atleast_1d(np.asarray(freqs, float))
            ph = np.exp(2j * np.pi * freqs[:, None] * tmesh_a[None, :])
            return spec.dt / np.sqrt(2 * np.pi) * (ph @ inc_a)
        return fn_angled
    comps = [(tang[0], np.cos(pol)), (tang[1], np.sin(pol))]
    specs = []
    for e_axis, wgt in comps:
        if abs(wgt) < 1e-12:
            continue
        t, face_aux = make_tfsf(spec, mt, lo, hi, p, direction, e_axis, e_unit * wgt,
                                lambda tt: np.real(st.amp_time(tt)), eps_bg,
                                name=getattr(src, "name", None) or src.type)
        spec.tfsf.append(t)
        specs.append((t, face_aux, wgt))

    t0, face0, w0 = specs[0]
    inc = replay_aux(t0, spec.n_steps, face0) / (e_unit * w0)     # incident E(t_n)/E_unit at the face
    tmesh = disc.tmesh

    def fn(freqs):
        freqs = np.atleast_1d(np.asarray(freqs, float))
        ph = np.exp(2j * np.pi * freqs[:, None] * tmesh[None, :])
        return spec.dt / np.sqrt(2 * np.pi) * (ph @ inc)
    return fn
