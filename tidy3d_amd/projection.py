"""Near-field to far-field projection (ref components/
field_projection.py, monitor.py:640-1040, data/monitor_data.py:2100-2260).

The near fields are recorded on the surfaces of the projection monitor like a flux box (running
DFT of the tangential components, colocated); after the run they are turned into equivalent surface
currents J = n x H, M = -n x E (ref field_projection.py:231-278), resampled to a regular lattice of
10 points per wavelength (ref :280-349), and integrated against the far-field phase

    N(theta, phi) = int J exp(-i k r_hat . r') dS,     L(theta, phi) = int M exp(-i k r_hat . r') dS,
    E_theta = -(L_phi + eta N_theta),  E_phi = L_theta - eta N_phi,  H = r_hat x E / eta

(Balanis 8.33-8.34, ref :370-521), times the propagation factor -i k exp(i k r) / (4 pi r)
(ref monitor_data.py:2170-2178).  Far-field approximation only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .constants import C_0, ETA_0
from .discretize import flux_surfaces

PTS_PER_WVL = 10          # ref field_projection.py:40


@dataclass
class FieldProjectionAngleData:
    """Mirror of tidy3d FieldProjectionAngleData (ref monitor_data.py:2300): Er .. Hphi, dims
    (r, theta, phi, f), relative to the monitor's local origin."""
    monitor: object
    Er: object = None
    Etheta: object = None
    Ephi: object = None
    Hr: object = None
    Htheta: object = None
    Hphi: object = None

    @property
    def field_components(self):
        return {k: getattr(self, k) for k in ("Er", "Etheta", "Ephi", "Hr", "Htheta", "Hphi")}

    @property
    def power(self):
        """Radiated power density 0.5 Re(E x H*) . r_hat  (ref monitor_data.py:2230-2240)."""
        from .data import DataArray
        e_t, e_p = np.asarray(self.Etheta.values), np.asarray(self.Ephi.values)
        h_t, h_p = np.asarray(self.Htheta.values), np.asarray(self.Hphi.values)
        return DataArray(0.5 * np.real(e_t * np.conj(h_p) - e_p * np.conj(h_t)), self.Etheta.coords)


_trap = getattr(np, "trapezoid", None) or np.trapz        # numpy >= 2 / < 2


def _trapz2(f: np.ndarray, u: np.ndarray, v: np.ndarray) -> complex:
    g = _trap(f, u, axis=0) if len(u) > 1 else f[0]
    return _trap(g, v, axis=0) if len(v) > 1 else g[0]


def _surface_currents(disc, plan, raw, norm, medium):
    """Per surface of the monitor: (axis, u, v, sample points [3 arrays], J {axis: (u, v, f)}, M {...}):
    equivalent currents J = n x H, M = -n x E (ref field_projection.py:231-278) on a regular lattice of
    10 points per wavelength clipped to the simulation domain (ref :280-349)."""
    from .data import FieldData, _field_container, interp_axis
    mon, sim, spec = plan.monitor, disc.sim, disc.spec
    freqs = np.asarray(mon.freqs, float)
    names = "xyz"
    # Simulation.symmetry: the surfaces were recorded on their images inside the computed sub-domain (discretize.py
    # symmetry_box_map) and are expanded to the user's surfaces with the reference's parity rules (ref monitor_data.py:
    # 238-284), exactly like the surfaces of a FluxMonitor
    sym = tuple(getattr(disc, "symmetry", (0, 0, 0)))
    pfull = disc.plans_full[[id(p_) for p_ in disc.plans].index(id(plan))] if any(sym) else None
    for isurf, (fp, (sname, box, axis, sign)) in enumerate(zip(plan.fields, flux_surfaces(mon))):
        class _M:
            pass
        m = _M()
        m.size, m.center, m.geometry, m.name = box.size, box.center, box, fp.spec_name
        full = None if pfull is None else (pfull.fields[isurf], disc.spec_full, sym)
        fd = _field_container(FieldData, m, spec, fp, raw[fp.spec_name], "f", freqs, sim.center,
                              np.complex128, full).normalize(norm)
        u, v = [a for a in range(3) if a != axis]
        n_idx = float(np.real(np.sqrt(complex(np.asarray(medium.eps_model(float(freqs.max()))).ravel()[0]))))
        wavelength = C_0 / float(freqs.max()) / n_idx
        pts = [None, None, None]
        pts[axis] = np.array([box.center[axis]])
        for a in (u, v):
            start = max(box.center[a] - box.size[a] / 2.0, sim.center[a] - sim.size[a] / 2.0)
            stop = min(box.center[a] + box.size[a] / 2.0, sim.center[a] + sim.size[a] / 2.0)
            n_pts = int(np.ceil(PTS_PER_WVL * (stop - start) / wavelength))
            pts[a] = np.linspace(start, stop, max(n_pts, 2)) if stop > start else np.array([start])
        # J = n x H, M = -n x E with the reference's sign table (ref :247-265)
        signs = np.array([-1.0, 1.0])
        if axis % 2 != 0:
            signs = -signs
        if sign < 0:
            signs = -signs
        cu, cv = names[u], names[v]

        def sampled(comp):
            arr = np.asarray(fd[comp].values)                         # (x, y, z, f) on the colocated nodes
            for a in (u, v):
                arr = interp_axis(arr, np.asarray(fd[comp].coords[names[a]]), pts[a], axis=a)
            return np.take(arr, 0, axis=axis)                         # (u, v, f) in x, y, z order
        J = {u: signs[0] * sampled("H" + cv), v: signs[1] * sampled("H" + cu)}
        M = {v: signs[0] * sampled("E" + cu), u: signs[1] * sampled("E" + cv)}
        win = _window(mon, (u, v), pts)
        if win is not None:
            J = {a: arr * win for a, arr in J.items()}
            M = {a: arr * win for a, arr in M.items()}
        yield axis, u, v, pts, J, M


WINDOW_FACTOR = 15          # ref monitor.py:44


def _window(mon, plane_axes, pts):
    """``window_size`` of a SURFACE projection monitor (ref monitor.py:819-848 field, :898-951 ``window_parameters`` /
    ``window_function``, applied to the equivalent currents in ref field_projection.py:522-558): along each tangential
    direction the currents are scaled by exp(-WINDOW_FACTOR/2 ((x - x_edge)/w)^2) beyond the points x_edge that lie
    w = window_size[i] * size / 2 inside the (sampled, i.e. clipped to the simulation) ends of the surface.  Returns
    the (u, v, 1) factor, or None without windowing."""
    ws = tuple(getattr(mon, "window_size", (0, 0)) or (0, 0))
    if not any(ws) or list(mon.size).count(0.0) != 1:
        return None
    fac = []
    for i, a in enumerate(plane_axes):
        p = np.asarray(pts[a], float)
        f = np.ones_like(p)
        lo_m, hi_m = mon.center[a] - mon.size[a] / 2.0, mon.center[a] + mon.size[a] / 2.0
        size = min(mon.size[a], p[-1] - p[0])
        w = ws[i] * size / 2.0
        if w > 0:
            minus, plus = max(lo_m, p[0]) + w, min(hi_m, p[-1]) - w
            f[p < minus] = np.exp(-0.5 * WINDOW_FACTOR * ((p[p < minus] - minus) / w) ** 2)
            f[p > plus] = np.exp(-0.5 * WINDOW_FACTOR * ((p[p > plus] - plus) / w) ** 2)
        fac.append(f)
    return fac[0][:, None, None] * fac[1][None, :, None]


def _medium_params(mon, sim, medium, freqs):
    if medium is None:
        medium = mon.medium if mon.medium is not None else sim.medium
    eps_f = np.array([complex(np.asarray(medium.eps_model(float(f))).ravel()[0]) for f in freqs])
    return medium, eps_f, 2 * np.pi * freqs * np.sqrt(eps_f) / C_0, ETA_0 / np.sqrt(eps_f)


def _trap_weights(x: np.ndarray) -> np.ndarray:
    """weights w with  sum(w * f) == np.trapz(f, x)  (1 for a single point: ``_trapz2`` takes the value itself)"""
    x = np.asarray(x, float)
    if x.size == 1:
        return np.ones(1)
    w = np.zeros(x.size)
    d = np.diff(x)
    w[:-1] += 0.5 * d
    w[1:] += 0.5 * d
    return w


def _far_fields(disc, plan, raw, norm, theta: np.ndarray, phi: np.ndarray, medium=None, f_sel=None, lib=None, device=0):
    """E_theta, E_phi (without the propagation factor) at the direction PAIRS (theta[n], phi[n]), summed
    over the monitor's surfaces: arrays [n, n_freq]; also k and eta per frequency.  ``f_sel`` restricts
    the evaluation to some frequency indices (the other columns stay 0).

    ``lib`` = the loaded HIP library: the surface integrals (directions x lattice points x frequencies — all of the
    cost) run on the device (``fdtd_far_field``, kernel K9); ``web.run`` passes it.  Without it the same sums are
    taken in NumPy — that is how stored data are projected after the fact, as the reference's ``FieldProjector`` does
    (ref field_projection.py:370), and what the device path is tested against."""
    mon, sim = plan.monitor, disc.sim
    freqs = np.asarray(mon.freqs, float)
    medium, eps_f, k_f, eta_f = _medium_params(mon, sim, medium, freqs)
    origin = mon.local_origin
    st, ct, sp_, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
    r_hat = np.stack([st * cp, st * sp_, ct])                       # (3, n)
    e_t = np.zeros((len(theta), len(freqs)), complex)
    e_p = np.zeros_like(e_t)
    for axis, u, v, pts, J, M in _surface_currents(disc, plan, raw, norm, medium):
        rel = [pts[a] - origin[a] for a in range(3)]
        for i_f in (range(len(freqs)) if f_sel is None else f_sel):
            k, eta = k_f[i_f], eta_f[i_f]
            Jv = np.zeros((3, len(theta)), complex)
            Mv = np.zeros_like(Jv)
            if lib is not None:
                cur = np.stack([J[u][:, :, i_f], J[v][:, :, i_f], M[u][:, :, i_f], M[v][:, :, i_f]])
                got = lib.far_field(rel[u], rel[v], _trap_weights(pts[u]), _trap_weights(pts[v]), cur, float(rel[axis][0]), k,
                                    r_hat[u], r_hat[v], r_hat[axis], device=device)
                Jv[u], Jv[v], Mv[u], Mv[v] = got[:, 0], got[:, 1], got[:, 2], got[:, 3]
            for n in (range(len(theta)) if lib is None else ()):
                ph = (np.exp(-1j * k * rel[u] * r_hat[u, n])[:, None] * np.exp(-1j * k * rel[v] * r_hat[v, n])[None, :] *
                      np.exp(-1j * k * rel[axis][0] * r_hat[axis, n]))
                for a in (u, v):
                    Jv[a, n] = _trapz2(J[a][:, :, i_f] * ph, pts[u], pts[v])
                    Mv[a, n] = _trapz2(M[a][:, :, i_f] * ph, pts[u], pts[v])
            n_t = Jv[0] * ct * cp + Jv[1] * ct * sp_ - Jv[2] * st
            n_p = -Jv[0] * sp_ + Jv[1] * cp
            l_t = Mv[0] * ct * cp + Mv[1] * ct * sp_ - Mv[2] * st
            l_p = -Mv[0] * sp_ + Mv[1] * cp
            e_t[:, i_f] += -(l_p + eta * n_t)
            e_p[:, i_f] += l_t - eta * n_p
    return e_t, e_p, k_f, eta_f


def _exact_fields(disc, plan, raw, norm, points: np.ndarray, medium=None):
    """E and H (Cartesian, [3, n_points, n_freq]) at the observation ``points`` [n, 3] (relative to the
    monitor's local origin) from the homogeneous-medium Green's function without the far-field
    approximation (ref field_projection.py:831-1010), e^{-i w t} convention:

        E = i w mu int [G J + (1/k^2) grad grad G . J] dS - int grad G x M dS
        H = i w eps int [G M + (1/k^2) grad grad G . M] dS + int grad G x J dS
        G = exp(i k R) / (4 pi R),  grad G = R_hat G',  grad grad G = R_hat R_hat G'' + (1 - R_hat R_hat) G' / R."""
    from .constants import EPSILON_0, MU_0
    mon, sim = plan.monitor, disc.sim
    freqs = np.asarray(mon.freqs, float)
    medium, eps_f, k_f, eta_f = _medium_params(mon, sim, medium, freqs)
    origin = np.asarray(mon.local_origin, float)
    E = np.zeros((3, len(points), len(freqs)), complex)
    H = np.zeros_like(E)
    for axis, u, v, pts, J, M in _surface_currents(disc, plan, raw, norm, medium):
        src = np.zeros((len(pts[u]), len(pts[v]), 3))
        src[..., u], src[..., v], src[..., axis] = pts[u][:, None], pts[v][None, :], pts[axis][0]
        zero = np.zeros_like(J[u][:, :, 0])
        for i_f, f in enumerate(freqs):
            k, w = k_f[i_f], 2 * np.pi * f
            Jc = [zero, zero, zero]
            Mc = [zero, zero, zero]
            for a in (u, v):
                Jc[a], Mc[a] = J[a][:, :, i_f], M[a][:, :, i_f]
            for n, pt in enumerate(points):
                Rv = (pt + origin)[None, None, :] - src
                R = np.sqrt(np.sum(Rv ** 2, axis=-1))
                Rh = [Rv[..., a] / R for a in range(3)]
                G = np.exp(1j * k * R) / (4 * np.pi * R)
                G1 = G * (1j * k - 1.0 / R)
                G2 = G1 * (1j * k - 1.0 / R) + G / R ** 2
                for cur, oth, out, oth_out, const in ((Jc, Mc, E, H, 1j * w * MU_0), (Mc, Jc, H, E, 1j * w * EPSILON_0 * eps_f[i_f])):
                    rd = Rh[0] * cur[0] + Rh[1] * cur[1] + Rh[2] * cur[2]
                    for a in range(3):
                        pot = G * cur[a] + (Rh[a] * rd * G2 + (cur[a] - Rh[a] * rd) * G1 / R) / k ** 2
                        out[a, n, i_f] += const * _trapz2(pot, pts[u], pts[v])
                # curl terms: E -= int grad G x M,  H += int grad G x J
                for cur, out, sgn in ((Mc, E, -1.0), (Jc, H, 1.0)):
                    cx = [Rh[1] * cur[2] - Rh[2] * cur[1], Rh[2] * cur[0] - Rh[0] * cur[2], Rh[0] * cur[1] - Rh[1] * cur[0]]
                    for a in range(3):
                        out[a, n, i_f] += sgn * _trapz2(G1 * cx[a], pts[u], pts[v])
    return E, H


def _package_exact(cls, mon, E, H, theta, phi, shape, coords):
    """Spherical components (about the local origin) of Cartesian E, H [3, n, f]."""
    from .data import DataArray
    st, ct, sp_, cp = np.sin(theta)[:, None], np.cos(theta)[:, None], np.sin(phi)[:, None], np.cos(phi)[:, None]

    def sph(F):
        return (F[0] * st * cp + F[1] * st * sp_ + F[2] * ct, F[0] * ct * cp + F[1] * ct * sp_ - F[2] * st,
                -F[0] * sp_ + F[1] * cp)
    er, et, ep = sph(E)
    hr, ht, hp = sph(H)
    comps = {"Er": er, "Etheta": et, "Ephi": ep, "Hr": hr, "Htheta": ht, "Hphi": hp}
    return cls(monitor=mon, **{k: DataArray(np.asarray(a).reshape(shape), coords) for k, a in comps.items()})


def _package(cls, mon, e_t, e_p, k_f, eta_f, r, shape, coords):
    """Apply the propagation factor -i k exp(i k r) / (4 pi r) (ref monitor_data.py:2170-2178) per
    point and frequency and box the six spherical components."""
    from .data import DataArray
    prop = -1j * k_f[None, :] * np.exp(1j * k_f[None, :] * r[:, None]) / (4 * np.pi * r[:, None])
    et, ep = e_t * prop, e_p * prop
    comps = {"Er": np.zeros_like(et), "Etheta": et, "Ephi": ep, "Hr": np.zeros_like(et),
             "Htheta": -ep / eta_f[None, :], "Hphi": et / eta_f[None, :]}
    return cls(monitor=mon, **{k: DataArray(v.reshape(shape), coords) for k, v in comps.items()})


def project_angle(disc, plan, raw, norm, lib=None, device: int = 0) -> FieldProjectionAngleData:
    mon = plan.monitor
    freqs = np.asarray(mon.freqs, float)
    theta, phi = np.asarray(mon.theta, float), np.asarray(mon.phi, float)
    T, P = np.meshgrid(theta, phi, indexing="ij")
    coords = {"r": np.atleast_1d(float(mon.proj_distance)), "theta": theta, "phi": phi, "f": freqs}
    if not mon.far_field_approx:
        t, p_, r0 = T.ravel(), P.ravel(), float(mon.proj_distance)
        pts = r0 * np.stack([np.sin(t) * np.cos(p_), np.sin(t) * np.sin(p_), np.cos(t)], axis=1)
        E, H = _exact_fields(disc, plan, raw, norm, pts)
        return _package_exact(FieldProjectionAngleData, mon, E, H, t, p_, (1, len(theta), len(phi), len(freqs)), coords)
    e_t, e_p, k_f, eta_f = _far_fields(disc, plan, raw, norm, T.ravel(), P.ravel(), lib=lib, device=device)
    r = np.full(T.size, float(mon.proj_distance))
    return _package(FieldProjectionAngleData, mon, e_t, e_p, k_f, eta_f, r, (1, len(theta), len(phi), len(freqs)), coords)


@dataclass
class FieldProjectionCartesianData(FieldProjectionAngleData):
    """ref monitor_data.py FieldProjectionCartesianData: dims (x, y, z, f), local Cartesian points."""


@dataclass
class FieldProjectionKSpaceData(FieldProjectionAngleData):
    """ref monitor_data.py FieldProjectionKSpaceData: dims (ux, uy, r, f)."""


def project_cartesian(disc, plan, raw, norm, lib=None, device: int = 0) -> FieldProjectionCartesianData:
    """Observation points on a plane at ``proj_distance`` along ``proj_axis`` (ref field_projection.py:665-746)."""
    mon = plan.monitor
    freqs = np.asarray(mon.freqs, float)
    loc = [np.atleast_1d(np.asarray(mon.x, float)), np.atleast_1d(np.asarray(mon.y, float))]
    loc.insert(int(mon.proj_axis), np.atleast_1d(float(mon.proj_distance)))      # unpop_axis
    X, Y, Z = np.meshgrid(*loc, indexing="ij")
    r = np.sqrt(X ** 2 + Y ** 2 + Z ** 2).ravel()
    theta = np.arccos(Z.ravel() / r)
    phi = np.arctan2(Y.ravel(), X.ravel())
    coords = {"x": loc[0], "y": loc[1], "z": loc[2], "f": freqs}
    if not mon.far_field_approx:
        E, H = _exact_fields(disc, plan, raw, norm, np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1))
        return _package_exact(FieldProjectionCartesianData, mon, E, H, theta, phi, X.shape + (len(freqs),), coords)
    e_t, e_p, k_f, eta_f = _far_fields(disc, plan, raw, norm, theta, phi, lib=lib, device=device)
    return _package(FieldProjectionCartesianData, mon, e_t, e_p, k_f, eta_f, r, X.shape + (len(freqs),), coords)


def project_kspace(disc, plan, raw, norm, lib=None, device: int = 0) -> FieldProjectionKSpaceData:
    """Observation directions given by the in-plane unit-vector components (ux, uy) around
    ``proj_axis`` (ref field_projection.py:748-829, geometry/base.py:963-985)."""
    mon = plan.monitor
    freqs = np.asarray(mon.freqs, float)
    ux, uy = np.atleast_1d(np.asarray(mon.ux, float)), np.atleast_1d(np.asarray(mon.uy, float))
    UX, UY = np.meshgrid(ux, uy, indexing="ij")
    phi_l = np.arctan2(UY, UX)
    with np.errstate(invalid="ignore"):
        theta_l = np.arcsin(np.sqrt(UX ** 2 + UY ** 2))
    if int(mon.proj_axis) == 2:
        theta, phi = theta_l, phi_l
    else:
        x, y, z = np.cos(theta_l), np.sin(theta_l) * np.cos(phi_l), np.sin(theta_l) * np.sin(phi_l)
        if int(mon.proj_axis) == 1:
            x, y = y, x
        theta, phi = np.arccos(z), np.arctan2(y, x)
    valid = np.isfinite(theta).ravel()
    th, ph = np.where(valid, theta.ravel(), 0.0), np.where(valid, phi.ravel(), 0.0)
    e_t, e_p, k_f, eta_f = _far_fields(disc, plan, raw, norm, th, ph, lib=lib, device=device)
    e_t[~valid], e_p[~valid] = np.nan, np.nan              # evanescent directions (ux^2 + uy^2 > 1)
    r = np.full(th.size, float(mon.proj_distance))
    coords = {"ux": ux, "uy": uy, "r": np.atleast_1d(float(mon.proj_distance)), "f": freqs}
    return _package(FieldProjectionKSpaceData, mon, e_t, e_p, k_f, eta_f, r, (len(ux), len(uy), 1, len(freqs)), coords)


# ----------------------------------------------------------------------------------------
# diffraction orders of a periodic structure  (ref monitor.py:1353, monitor_data.py:2672-2900)
# ----------------------------------------------------------------------------------------

@dataclass
class DiffractionData(FieldProjectionAngleData):
    """Mirror of tidy3d DiffractionData (ref monitor_data.py:2672): Er .. Hphi with dims
    (orders_x, orders_y, f) in the monitor's local frame (z' = monitor normal, x', y' = the transverse
    axes in x, y, z order), normalised like the reference's: ``abs(amps)**2`` = power carried by an
    order and polarisation through one period (ref :2840-2860)."""
    sim_size: tuple = (0.0, 0.0)
    bloch_vecs: tuple = (0.0, 0.0)
    medium: object = None
    structure_index: int = -1            # structure whose medium fills the monitor plane (-1: background)

    @property
    def orders_x(self):
        return np.atleast_1d(np.asarray(self.Etheta.coords["orders_x"]))

    @property
    def orders_y(self):
        return np.atleast_1d(np.asarray(self.Etheta.coords["orders_y"]))

    @property
    def f(self):
        return np.asarray(self.Etheta.coords["f"], float)

    @property
    def eta(self):
        """Wave impedance of the projection medium per frequency."""
        eps = np.array([complex(np.asarray(self.medium.eps_model(float(f))).ravel()[0]) for f in self.f])
        return np.real(ETA_0 / np.sqrt(eps))

    def _u(self, orders, size, bloch):
        """ref monitor_data.py:2758-2767 reciprocal_coords."""
        if size == 0:
            return np.zeros((1, len(self.f)))
        eps = np.array([complex(np.asarray(self.medium.eps_model(float(f))).ravel()[0]) for f in self.f])
        return (bloch + np.atleast_1d(orders))[:, None] / size * C_0 / self.f[None, :] / np.real(np.sqrt(eps))[None, :]

    @property
    def ux(self):
        return self._u(self.orders_x, self.sim_size[0], self.bloch_vecs[0])

    @property
    def uy(self):
        return self._u(self.orders_y, self.sim_size[1], self.bloch_vecs[1])

    @property
    def angles(self):
        """(theta, phi) per (orders_x, orders_y, f) in the local frame, NaN outside the light cone
        (ref :2769-2781, :2830-2838)."""
        from .data import DataArray
        ux, uy = self.ux[:, None, :], self.uy[None, :, :]
        with np.errstate(invalid="ignore"):
            theta = np.arcsin(np.sqrt(ux ** 2 + uy ** 2))
        phi = np.where(np.isfinite(theta), np.arctan2(uy, ux) + 0.0 * theta, np.nan)
        return DataArray(theta, self.Etheta.coords), DataArray(phi, self.Etheta.coords)

    @property
    def amps(self):
        """ref :2840-2860: dims (orders_x, orders_y, f, polarization = [s, p])."""
        from .data import DataArray
        cos_theta = np.cos(np.nan_to_num(self.angles[0].values))
        cos_theta[cos_theta <= 0] = np.inf
        nrm = 1.0 / np.sqrt(2.0 * self.eta)[None, None, :] / np.sqrt(cos_theta)
        coords = dict(self.Etheta.coords)
        coords["polarization"] = np.array(["s", "p"])
        return DataArray(np.stack([self.Ephi.values * nrm, self.Etheta.values * nrm], axis=3), coords)

    @property
    def power(self):
        """Total power per order, both polarisations (ref :2862-2868)."""
        from .data import DataArray
        return DataArray(np.sum(np.abs(self.amps.values) ** 2, axis=3), self.Etheta.coords)


def _medium_at(sim, point):
    """(medium, structure index) that fills ``point``: the last structure containing it, else the
    background with index -1 (ref simulation.py:1228-1230 overwrite order)."""
    x, y, z = (np.array([float(v)]) for v in point)
    for i in reversed(range(len(sim.structures))):
        st = sim.structures[i]
        if bool(np.asarray(st.geometry.inside(x, y, z)).ravel()[0]):
            return st.medium, i
    return sim.medium, -1


def diffraction(disc, plan, raw, norm, lib=None, device: int = 0) -> DiffractionData:
    """Order amplitudes from the surface-equivalence integrals over ONE period: a periodic sheet of
    currents radiates the plane waves  E_mn = [far-field integrand at the order's direction] /
    (2 A cos(theta_mn))  (A = area of the period; the uniform sheet J_s radiating -eta J_s / 2 is the
    m = n = 0 case), and an order carries |E_mn|^2 A cos(theta) / (2 eta).  Stored like the reference
    (ref monitor_data.py:2840-2850: power = |E|^2 / (2 eta cos(theta))):  E = E_mn cos(theta) sqrt(A).
    Orders outside the light cone hold 0.  The reference's own order bookkeeping is server-side."""
    from .data import DataArray
    mon, sim = plan.monitor, disc.sim
    freqs = np.asarray(mon.freqs, float)
    axis = [a for a in range(3) if mon.size[a] == 0][0]
    u, v = [a for a in range(3) if a != axis]
    sgn = 1.0 if (mon.normal_dir or "+") == "+" else -1.0
    Lu, Lv = float(sim.size[u]), float(sim.size[v])
    medium, s_index = _medium_at(sim, mon.center)
    eps_f = np.array([complex(np.asarray(medium.eps_model(float(f))).ravel()[0]) for f in freqs])
    n_f = np.real(np.sqrt(eps_f))
    n_max = float(np.max(n_f * freqs)) / C_0                       # 1 / shortest wavelength in the medium
    # Bloch vectors in units of 2 pi / size (ref monitor_data.py:2746-2751): orders are m + bloch_vec
    bl = disc.spec.bloch if disc.spec.bloch is not None else (0.0, 0.0, 0.0)
    bu, bv = bl[u] / (2 * np.pi), bl[v] / (2 * np.pi)

    def order_range(L, bvec):
        if L <= 0:
            return np.arange(0, 1)
        return np.arange(int(np.ceil(-L * n_max - bvec - 1e-9)), int(np.floor(L * n_max - bvec + 1e-9)) + 1)
    ox, oy = order_range(Lu, bu), order_range(Lv, bv)
    area = (Lu if Lu > 0 else 1.0) * (Lv if Lv > 0 else 1.0)
    shape = (len(ox), len(oy), len(freqs))
    e_th, e_ph = np.zeros(shape, complex), np.zeros(shape, complex)
    for i_f, f in enumerate(freqs):
        lam = C_0 / (f * n_f[i_f])
        UX = ((ox[:, None] + bu) * lam / Lu if Lu > 0 else np.zeros((1, 1))) + np.zeros((len(ox), len(oy)))
        UY = ((oy[None, :] + bv) * lam / Lv if Lv > 0 else np.zeros((1, 1))) + np.zeros((len(ox), len(oy)))
        ok = (UX ** 2 + UY ** 2) < 1.0 - 1e-9
        if not ok.any():
            continue
        uxv, uyv = UX[ok], UY[ok]
        uz = np.sqrt(1.0 - uxv ** 2 - uyv ** 2)
        d = np.zeros((3, uxv.size))                                 # global propagation direction
        d[u], d[v], d[axis] = uxv, uyv, sgn * uz
        th_g, ph_g = np.arccos(np.clip(d[2], -1, 1)), np.arctan2(d[1], d[0])
        et, ep, _, _ = _far_fields(disc, plan, raw, norm, th_g, ph_g, medium=medium, f_sel=[i_f], lib=lib, device=device)
        # global Cartesian field vector, then the local spherical components
        st, ct, sp_, cp = np.sin(th_g), np.cos(th_g), np.sin(ph_g), np.cos(ph_g)
        t_hat = np.stack([ct * cp, ct * sp_, -st])
        p_hat = np.stack([-sp_, cp, np.zeros_like(cp)])
        E = et[:, i_f][None, :] * t_hat + ep[:, i_f][None, :] * p_hat
        th_l, ph_l = np.arcsin(np.sqrt(uxv ** 2 + uyv ** 2)), np.arctan2(uyv, uxv)
        stl, ctl, spl, cpl = np.sin(th_l), np.cos(th_l), np.sin(ph_l), np.cos(ph_l)
        E_l = np.stack([E[u], E[v], sgn * E[axis]])                # local (x', y', z') components
        e_t_l = E_l[0] * ctl * cpl + E_l[1] * ctl * spl - E_l[2] * stl
        e_p_l = -E_l[0] * spl + E_l[1] * cpl
        scale = 1.0 / (2.0 * np.sqrt(area))
        tmp_t, tmp_p = np.zeros(ok.shape, complex), np.zeros(ok.shape, complex)
        tmp_t[ok], tmp_p[ok] = e_t_l * scale, e_p_l * scale
        e_th[:, :, i_f], e_ph[:, :, i_f] = tmp_t, tmp_p
    eta = np.real(ETA_0 / np.sqrt(eps_f))[None, None, :]
    coords = {"orders_x": ox, "orders_y": oy, "f": freqs}
    comps = {"Er": np.zeros(shape, complex), "Etheta": e_th, "Ephi": e_ph, "Hr": np.zeros(shape, complex),
             "Htheta": -e_ph / eta, "Hphi": e_th / eta}
    return DiffractionData(monitor=mon, sim_size=(Lu, Lv), bloch_vecs=(float(bu), float(bv)), medium=medium,
                           structure_index=s_index,
                           **{k: DataArray(a, coords) for k, a in comps.items()})
