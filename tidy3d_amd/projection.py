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


def project_angle(disc, plan, raw, norm) -> FieldProjectionAngleData:
    from .data import DataArray, FieldData, _field_container, interp_axis
    mon, sim, spec = plan.monitor, disc.sim, disc.spec
    freqs = np.asarray(mon.freqs, float)
    theta, phi = np.asarray(mon.theta, float), np.asarray(mon.phi, float)
    medium = mon.medium if mon.medium is not None else sim.medium
    origin = mon.local_origin
    names = "xyz"
    out = {k: np.zeros((1, len(theta), len(phi), len(freqs)), complex)
           for k in ("Er", "Etheta", "Ephi", "Hr", "Htheta", "Hphi")}
    st, ct, sp_, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
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
            arr = np.take(arr, 0, axis=axis)                          # (u, v, f) in x, y, z order
            return arr
        J = {u: signs[0] * sampled("H" + cv), v: signs[1] * sampled("H" + cu)}
        M = {v: signs[0] * sampled("E" + cu), u: signs[1] * sampled("E" + cv)}
        rel = [pts[a] - origin[a] for a in range(3)]
        for i_f, f in enumerate(freqs):
            eps = complex(np.asarray(medium.eps_model(float(f))).ravel()[0])
            k = 2 * np.pi * f * np.sqrt(eps) / C_0
            eta = ETA_0 / np.sqrt(eps)
            Jv = np.zeros((3, len(theta), len(phi)), complex)
            Mv = np.zeros_like(Jv)
            for i_t in range(len(theta)):
                for i_p in range(len(phi)):
                    r_hat = (st[i_t] * cp[i_p], st[i_t] * sp_[i_p], ct[i_t])
                    ph = (np.exp(-1j * k * rel[u] * r_hat[u])[:, None] * np.exp(-1j * k * rel[v] * r_hat[v])[None, :] *
                          np.exp(-1j * k * rel[axis][0] * r_hat[axis]))
                    for a in (u, v):
                        Jv[a, i_t, i_p] = _trapz2(J[a][:, :, i_f] * ph, pts[u], pts[v])
                        Mv[a, i_t, i_p] = _trapz2(M[a][:, :, i_f] * ph, pts[u], pts[v])
            ctcp, ctsp = ct[:, None] * cp[None, :], ct[:, None] * sp_[None, :]
            n_t = Jv[0] * ctcp + Jv[1] * ctsp - Jv[2] * st[:, None]
            n_p = -Jv[0] * sp_[None, :] + Jv[1] * cp[None, :]
            l_t = Mv[0] * ctcp + Mv[1] * ctsp - Mv[2] * st[:, None]
            l_p = -Mv[0] * sp_[None, :] + Mv[1] * cp[None, :]
            e_t = -(l_p + eta * n_t)
            e_p = l_t - eta * n_p
            prop = -1j * k * np.exp(1j * k * mon.proj_distance) / (4 * np.pi * mon.proj_distance)
            out["Etheta"][0, :, :, i_f] += e_t * prop
            out["Ephi"][0, :, :, i_f] += e_p * prop
            out["Htheta"][0, :, :, i_f] += -e_p / eta * prop
            out["Hphi"][0, :, :, i_f] += e_t / eta * prop
    coords = {"r": np.atleast_1d(float(mon.proj_distance)), "theta": theta, "phi": phi, "f": freqs}
    return FieldProjectionAngleData(monitor=mon, **{k: DataArray(v, coords) for k, v in out.items()})
