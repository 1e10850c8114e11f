"""Closed-form references used to pin field values (the reference has no solver to compare with;
SURVEY.md section 8(c)): Mie series (Bohren & Huffman, ch. 4), PEC-cavity eigenfrequencies of the
discrete Yee grid, Airy transmission of a slab."""
from __future__ import annotations

import numpy as np
from scipy import special

from .constants import C_0


def mie_efficiencies(m: complex, x: float, n_max: int = None):
    """(Q_ext, Q_sca) of a homogeneous sphere: relative index m = n_sphere/n_medium, size
    parameter x = k_medium * radius.  Riccati-Bessel form of the Mie coefficients."""
    if n_max is None:
        n_max = int(np.ceil(x + 4 * x ** (1 / 3) + 2)) + 2
    n = np.arange(1, n_max + 1)

    def psi(z):      # z j_n(z) and derivative
        j = special.spherical_jn(n, z)
        jd = special.spherical_jn(n, z, derivative=True)
        return z * j, j + z * jd

    def xi(z):       # z h_n^(1)(z) and derivative
        j = special.spherical_jn(n, z)
        y = special.spherical_yn(n, z)
        jd = special.spherical_jn(n, z, derivative=True)
        yd = special.spherical_yn(n, z, derivative=True)
        h = j + 1j * y
        hd = jd + 1j * yd
        return z * h, h + z * hd
    px, pxd = psi(x)
    pm, pmd = psi(m * x)
    xx, xxd = xi(x)
    a = (m * pm * pxd - px * pmd) / (m * pm * xxd - xx * pmd)
    b = (pm * pxd - m * px * pmd) / (pm * xxd - m * xx * pmd)
    q_ext = 2 / x ** 2 * np.sum((2 * n + 1) * np.real(a + b))
    q_sca = 2 / x ** 2 * np.sum((2 * n + 1) * (np.abs(a) ** 2 + np.abs(b) ** 2))
    return float(q_ext), float(q_sca)


def mie_cross_sections(radius: float, eps_sphere: complex, freqs, eps_medium: float = 1.0):
    """(sigma_ext, sigma_sca) in um^2 at each frequency (tidy3d units)."""
    out_e, out_s = [], []
    nm = np.sqrt(eps_medium)
    for f, es in zip(np.atleast_1d(freqs), np.broadcast_to(eps_sphere, np.shape(np.atleast_1d(freqs)))):
        k = 2 * np.pi * f / C_0 * nm
        qe, qs = mie_efficiencies(np.sqrt(complex(es)) / nm, k * radius)
        out_e.append(qe * np.pi * radius ** 2)
        out_s.append(qs * np.pi * radius ** 2)
    return np.array(out_e), np.array(out_s)


def yee_cavity_modes(lengths, steps, dt, max_index: int = 4):
    """Eigenfrequencies (Hz) of a PEC cavity on a uniform Yee grid:
    sin^2(w dt/2)/(c dt)^2 = sum_i sin^2(k_i d_i/2)/d_i^2,  k_i = m_i pi / L_i."""
    modes = []
    for m in range(max_index):
        for q in range(max_index):
            for p in range(max_index):
                if (m > 0) + (q > 0) + (p > 0) < 2:
                    continue
                s = sum(np.sin(np.pi * mm / L * d / 2) ** 2 / d ** 2
                        for mm, L, d in zip((m, q, p), lengths, steps))
                modes.append(2 / dt * np.arcsin(C_0 * dt * np.sqrt(s)) / (2 * np.pi))
    return np.array(sorted(modes))


def slab_transmission(eps, thickness: float, freqs):
    """Power transmission of a slab at normal incidence, vacuum on both sides."""
    n = np.sqrt(np.asarray(eps, complex))
    k = 2 * np.pi * np.asarray(freqs) / C_0 * n
    r = (1 - n) / (1 + n)
    t = (1 - r ** 2) * np.exp(1j * k * thickness) / (1 - r ** 2 * np.exp(2j * k * thickness))
    return np.abs(t) ** 2


def thin_film_RT(eps_film, d: float, eps_substrate: float, freqs):
    """Power reflectance and transmittance (into a lossless substrate) of a film of (complex) permittivity ``eps_film`` and
    thickness ``d`` [um] on a substrate, from vacuum at normal incidence: the Airy sum of the two interfaces.  With
    ``medium.eps_model(freqs)`` (ref medium.py:2900-2913, exp(-i w t) convention) it pins dispersive-medium runs."""
    import numpy as np
    from .constants import C_0
    freqs = np.asarray(freqs, dtype=np.float64)
    n1, n2, n3 = 1.0, np.sqrt(np.asarray(eps_film) + 0j), np.sqrt(eps_substrate)
    k2 = 2 * np.pi * freqs / C_0 * n2
    r12, r23 = (n1 - n2) / (n1 + n2), (n2 - n3) / (n2 + n3)
    t12, t23 = 2 * n1 / (n1 + n2), 2 * n2 / (n2 + n3)
    ph = np.exp(1j * k2 * d)
    den = 1 + r12 * r23 * ph ** 2
    r = (r12 + r23 * ph ** 2) / den
    t = t12 * t23 * ph / den
    return np.abs(r) ** 2, float(np.real(n3)) / n1 * np.abs(t) ** 2
