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


def _far_fields(disc, plan, raw, norm, theta: np.ndarray, phi: np.ndarray):
    """E_theta, E_phi (without the propagation factor) at the direction PAIRS (theta[n], phi[n]), summed
    over the monitor's surfaces: arrays [n, n_freq]; also k and eta per frequency."""
    from .data import FieldData, _field_container, interp_axis
    mon, sim, spec = plan.monitor, disc.sim, disc.spec
    freqs = np.asarray(mon.freqs, float)
    medium = mon.medium if mon.medium is not None else sim.medium
    origin = mon.local_origin
    names = "xyz"
    eps_f = np.array([complex(np.asarray(medium.eps_model(float(f))).ravel()[0]) for f in freqs])
    k_f = 2 * np.pi * freqs * np.sqrt(eps_f) / C_0
    eta_f = ETA_0 / np.sqrt(eps_f)
    st, ct, sp_, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
    r_hat = np.stack([st * cp, st * sp_, ct])                       # (3, n)
    e_t = np.zeros((len(theta), len(freqs)), complex)
    e_p = np.zeros_like(e_t)
    for fp, (sname, box, axis, sign) in zip(plan.fields, flux_surfaces(mon)):
        class _M:
            pass
        m = _M()
        m.size, m.center, m.geometry, m.name = box.size, box.center, box, fp.spec_name
        fd = _field_container(FieldData, m, spec, fp, raw[fp.spec_name], "f", freqs, sim.center,
                              np.complex128).normalize(norm)
        u, v = [a for a in range(3) if a != axis]
        # regular sample lattice on the surface: 10 points per wavelength in the projection medium at the
        # highest frequency, clipped to the simulation domain (ref field_projection.py:292-343)
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
        rel = [pts[a] - origin[a] for a in range(3)]
        for i_f in range(len(freqs)):
            k, eta = k_f[i_f], eta_f[i_f]
            Jv = np.zeros((3, len(theta)), complex)
            Mv = np.zeros_like(Jv)
            for n in range(len(theta)):
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


def _package(cls, mon, e_t, e_p, k_f, eta_f, r, shape, coords):
    """Apply the propagation factor -i k exp(i k r) / (4 pi r) (ref monitor_data.py:2170-2178) per
    point and frequency and box the six spherical components."""
    from .data import DataArray
    prop = -1j * k_f[None, :] * np.exp(1j * k_f[None, :] * r[:, None]) / (4 * np.pi * r[:, None])
    et, ep = e_t * prop, e_p * prop
    comps = {"Er": np.zeros_like(et), "Etheta": et, "Ephi": ep, "Hr": np.zeros_like(et),
             "Htheta": -ep / eta_f[None, :], "Hphi": et / eta_f[None, :]}
    return cls(monitor=mon, **{k: DataArray(v.reshape(shape), coords) for k, v in comps.items()})


def project_angle(disc, plan, raw, norm) -> FieldProjectionAngleData:
    mon = plan.monitor
    freqs = np.asarray(mon.freqs, float)
    theta, phi = np.asarray(mon.theta, float), np.asarray(mon.phi, float)
    T, P = np.meshgrid(theta, phi, indexing="ij")
    e_t, e_p, k_f, eta_f = _far_fields(disc, plan, raw, norm, T.ravel(), P.ravel())
    r = np.full(T.size, float(mon.proj_distance))
    coords = {"r": np.atleast_1d(float(mon.proj_distance)), "theta": theta, "phi": phi, "f": freqs}
    return _package(FieldProjectionAngleData, mon, e_t, e_p, k_f, eta_f, r, (1, len(theta), len(phi), len(freqs)), coords)


@dataclass
class FieldProjectionCartesianData(FieldProjectionAngleData):
    """ref monitor_data.py FieldProjectionCartesianData: dims (x, y, z, f), local Cartesian points."""


@dataclass
class FieldProjectionKSpaceData(FieldProjectionAngleData):
    """ref monitor_data.py FieldProjectionKSpaceData: dims (ux, uy, r, f)."""


def project_cartesian(disc, plan, raw, norm) -> FieldProjectionCartesianData:
    """Observation points on a plane at ``proj_distance`` along ``proj_axis`` (ref field_projection.py:665-746)."""
    mon = plan.monitor
    freqs = np.asarray(mon.freqs, float)
    loc = [np.atleast_1d(np.asarray(mon.x, float)), np.atleast_1d(np.asarray(mon.y, float))]
    loc.insert(int(mon.proj_axis), np.atleast_1d(float(mon.proj_distance)))      # unpop_axis
    X, Y, Z = np.meshgrid(*loc, indexing="ij")
    r = np.sqrt(X ** 2 + Y ** 2 + Z ** 2).ravel()
    theta = np.arccos(Z.ravel() / r)
    phi = np.arctan2(Y.ravel(), X.ravel())
    e_t, e_p, k_f, eta_f = _far_fields(disc, plan, raw, norm, theta, phi)
    coords = {"x": loc[0], "y": loc[1], "z": loc[2], "f": freqs}
    return _package(FieldProjectionCartesianData, mon, e_t, e_p, k_f, eta_f, r, X.shape + (len(freqs),), coords)


def project_kspace(disc, plan, raw, norm) -> FieldProjectionKSpaceData:
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
    e_t, e_p, k_f, eta_f = _far_fields(disc, plan, raw, norm, th, ph)
    e_t[~valid], e_p[~valid] = np.nan, np.nan              # evanescent directions (ux^2 + uy^2 > 1)
    r = np.full(th.size, float(mon.proj_distance))
    coords = {"ux": ux, "uy": uy, "r": np.atleast_1d(float(mon.proj_distance)), "f": freqs}
    return _package(FieldProjectionKSpaceData, mon, e_t, e_p, k_f, eta_f, r, (len(ux), len(uy), 1, len(freqs)), coords)
