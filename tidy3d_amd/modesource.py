"""ModeSource (ref source.py:993-1085) and ModeMonitor (ref monitor.py:631) on top of the
cross-section eigenmode solver.

Injection: the source plane splits space into a total-field half-space (downstream of the plane
along ``direction``) and a scattered-field half-space; the incident field is the waveguide mode
E_m(u,v) a(t), H_m(u,v) a(t) (profile computed at freq0, ref source.py:1003: 1 W at the central
frequency).  The stencil legs across the plane (tidy3d_amd.planewave.surface_legs) become a
PointSourceSet with complex weights  leg_coefficient x mode_field(node): E-phase legs carry the
tangential H of the mode half a cell off the plane (phase e^{-i beta dz/2}) sampled at t_{n+1/2},
H-phase legs the tangential E on the plane at t_n — a one-way launch.

When the real tidy3d package is importable, ``mode_profile`` can be swapped for tidy3d's own
``ModeSolver(simulation, plane=source, ...)`` (north_star: the CPU ModeSolver stays the reference);
here the independent solver of tidy3d_amd.mode_solver is used and pinned to the reference's
arithmetic by tests/golden/mode_golden.json.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Tuple

import numpy as np

from . import schema as td
from .constants import C_0
from .discretize import discretize_inds
from .exceptions import SetupError, Tidy3dNotImplementedError
from .mode_solver import ModeResult, solve_modes, solve_modes_angled
from .planewave import surface_legs
from .spec import BC_PMC, PointSourceSet, SolverSpec


@dataclass
class ModePlane:
    p: int                       # propagation axis
    u: int
    v: int
    k0: int                      # boundary index of the plane along p
    lo: Tuple[int, int]          # first cell of the window along (u, v)
    hi: Tuple[int, int]
    result: ModeResult
    # propagation constants [rad/um] of the modes ON THE GRID when the plane was solved with the grid's own
    # dispersion (``grid_dispersion``), else the continuum values 2 pi f n / c
    beta: np.ndarray = None


def grid_frequency(freq: float, dt: float) -> float:
    """The frequency the leapfrog scheme really applies to a field oscillating at ``freq``: its time
    difference is -i w~ with  w~ = (2/dt) sin(w dt/2)."""
    return float(np.sin(np.pi * freq * dt) / (np.pi * dt))


def _eps_plane(spec: SolverSpec, comp_axis: int, p: int, k: int, lo, hi, u: int, v: int, freq: float):
    """eps(freq) at the Yee nodes of E_{comp_axis} on plane index k (p axis), window [lo, hi)."""
    from .data import medium_eps_table
    eps_tab = medium_eps_table(spec, freq)
    nu, nv = hi[0] - lo[0], hi[1] - lo[1]
    if spec.mat_idx is None:
        return np.full((nu, nv), eps_tab[1])
    k = int(np.clip(k, 0, spec.shape[p] - 1))
    sl = [None, None, None]
    sl[p] = k
    sl[u] = slice(lo[0], hi[0])
    sl[v] = slice(lo[1], hi[1])
    m = spec.mat_idx[comp_axis][sl[2], sl[1], sl[0]]      # array axes are (z, y, x) minus the fixed one
    # bring to (u, v) order
    rem = [a for a in (2, 1, 0) if a != p]                 # physical axes of m's dims, in order
    if rem == [v, u]:
        m = m.T
    return eps_tab[m]


def mode_profile(spec: SolverSpec, box: td.Box, mode_spec, freq: float, symmetry=(0, 0, 0),
                 grid_dispersion: bool = False) -> ModePlane:
    """Solve the cross-section eigenproblem on the plane of ``box`` (one zero-size dimension).
    ``symmetry``: the simulation's symmetry — a plane edge lying on a symmetry plane gets no PML.

    ``grid_dispersion``: solve for the mode AS THE YEE GRID PROPAGATES IT.  A field ~ exp(i (beta w - omega t))
    sampled on the staggered grid obeys the same six curl equations with the derivative along w replaced by
    i b~, b~ = (2/dw) sin(beta dw/2), and the time derivative by -i w~, w~ = (2/dt) sin(omega dt/2) (phasors taken
    at each component's own staggered position and time).  So the cross-section problem is solved at k0~ = w~/c;
    its eigenvalue is b~^2, its eigenvector the E/H profile with the impedance the grid supports, and
    beta = (2/dw) asin(b~ dw/2).  A source built from it launches no backward wave to order (beta dw)^2
    (the continuum mode leaves an impedance mismatch of (beta dw)^2/24, -75 dB at 24 cells per wavelength)."""
    zd = [a for a in range(3) if box.size[a] == 0]
    if len(zd) != 1:
        raise SetupError("a mode plane needs exactly one zero-size dimension")
    p = zd[0]
    u, v = (p + 1) % 3, (p + 2) % 3
    angle_theta = float(getattr(mode_spec, "angle_theta", 0.0) or 0.0)
    angle_phi = float(getattr(mode_spec, "angle_phi", 0.0) or 0.0)
    bend_radius = getattr(mode_spec, "bend_radius", None)
    bend_axis = 0
    if bend_radius is not None:
        # ModeSpec.bend_axis counts the plane's axes in x, y, z order (ref mode.py bend_axis); the
        # solver's (u, v) are cyclic
        phys = [a for a in range(3) if a != p][int(getattr(mode_spec, "bend_axis", 0) or 0)]
        bend_axis = (u, v).index(phys)
    b = spec.boundaries
    k0 = int(np.argmin(np.abs(b[p] - box.center[p])))
    k0 = int(np.clip(k0, 1, spec.shape[p] - 1))
    span = discretize_inds(list(b), box)
    lo = (max(span[u][0], 0), max(span[v][0], 0))
    hi = (min(span[u][1], spec.shape[u]), min(span[v][1], spec.shape[v]))
    eps_u = _eps_plane(spec, u, p, k0, lo, hi, u, v, freq)
    eps_v = _eps_plane(spec, v, p, k0, lo, hi, u, v, freq)
    eps_w = _eps_plane(spec, p, p, k0, lo, hi, u, v, freq)
    ub = b[u][lo[0]:hi[0] + 1]
    vb = b[v][lo[1]:hi[1] + 1]
    # a plane that reaches a PMC wall of the grid (a symmetry plane with eigenvalue +1) gets the PMC
    # edge there; PEC walls / the plane's own truncation are the solver's default (ref solver.py:182-197)
    pmc_min = tuple(bool(lo[i] == 0 and spec.bc[a][0] == BC_PMC) for i, a in enumerate((u, v)))
    f_solve = grid_frequency(freq, spec.dt) if grid_dispersion else freq
    if angle_theta:
        # ModeSpec.angle_phi counts from the FIRST in-plane axis in x, y, z order (ref mode.py angle_phi); the
        # solver's (u, v) are cyclic: for a y-normal plane they are (z, x), the azimuth is measured from v there
        phi_uv = angle_phi if (u, v) == tuple(sorted((u, v))) else 0.5 * np.pi - angle_phi
        res = solve_modes_angled(eps_u, eps_v, eps_w, ub, vb, f_solve, angle_theta, phi_uv,
                                 num_modes=int(mode_spec.num_modes), target_neff=mode_spec.target_neff,
                                 precision=getattr(mode_spec, "precision", "single") or "single", pmc_min=pmc_min,
                                 num_pml=tuple(int(n) for n in (getattr(mode_spec, "num_pml", (0, 0)) or (0, 0))),
                                 pml_min=tuple(not (lo[i] == 0 and symmetry[a] != 0) for i, a in enumerate((u, v))),
                                 bend_radius=bend_radius, bend_axis=bend_axis)
    else:
        res = solve_modes(eps_u, eps_v, eps_w, ub, vb, f_solve, num_modes=int(mode_spec.num_modes),
                          target_neff=mode_spec.target_neff,
                          precision=getattr(mode_spec, "precision", "single") or "single", pmc_min=pmc_min,
                          num_pml=tuple(int(n) for n in (getattr(mode_spec, "num_pml", (0, 0)) or (0, 0))),
                          pml_min=tuple(not (lo[i] == 0 and symmetry[a] != 0) for i, a in enumerate((u, v))),
                          bend_radius=bend_radius, bend_axis=bend_axis)
    fp = getattr(mode_spec, "filter_pol", None)
    if fp in ("te", "tm"):
        # ModeSpec.filter_pol (ref mode_solver.py:523-549): modes whose field intensity sits mostly on
        # the FIRST tangential axis (x, y, z order; ref monitor_data.py:1626-1653) are "te"; stable
        # partition, the selected polarisation first
        first = min(u, v)
        w_area = np.outer(np.diff(ub), np.diff(vb))
        e1 = res.Eu if first == u else res.Ev
        e2 = res.Ev if first == u else res.Eu
        te_int = np.array([np.sum(w_area * np.abs(e1[:, :, m]) ** 2) for m in range(e1.shape[2])])
        tm_int = np.array([np.sum(w_area * np.abs(e2[:, :, m]) ** 2) for m in range(e1.shape[2])])
        te_frac = te_int / (te_int + tm_int)
        if fp == "te":
            order = np.concatenate((np.where(te_frac >= 0.5)[0], np.where(te_frac < 0.5)[0]))
        else:
            order = np.concatenate((np.where(te_frac <= 0.5)[0], np.where(te_frac > 0.5)[0]))
        res = type(res)(n_complex=res.n_complex[order], **{k: getattr(res, k)[:, :, order]
                                                           for k in ("Eu", "Ev", "Ew", "Hu", "Hv", "Hw")})
    # wave number ALONG THE NORMAL: an angled mode advances by k n_eff / cos(theta) per unit of w at fixed
    # transverse position of the sheared frame (the reference's grid correction uses the same, ref
    # plugins/mode/mode_solver.py:883-887)
    cos_t = float(np.cos(angle_theta))
    if grid_dispersion:
        dw = float(spec.dual_steps(p)[k0])                       # spacing of the H planes either side of the plane
        b_tilde = res.n_complex * (2 * np.pi * f_solve / C_0) / cos_t
        beta = (2.0 / dw) * np.arcsin(b_tilde * dw / 2.0 + 0j)
    else:
        beta = res.n_complex * (2 * np.pi * freq / C_0) / cos_t
    return ModePlane(p=p, u=u, v=v, k0=k0, lo=lo, hi=hi, result=res, beta=beta)


CHEB_GRID_WIDTH = 1.5            # ref source.py:55: the Chebyshev grid of a broadband source spans freq0 +- 1.5 fwidth


def _leg_weights(disc, mt, src, freq: float, align_to=None):
    """Complex weights of the stencil legs across the source plane for the mode solved at ``freq``:
    (plane, comp, ijk, w, tangential E fields).  ``align_to``: tangential E of the reference frequency — the eigenvector
    of every other frequency is rotated to the same phase (an eigen-solver returns an arbitrary one per call)."""
    sim, spec = disc.sim, disc.spec
    plane = mode_profile(spec, src.geometry, src.mode_spec, freq, getattr(sim, "_symmetry", (0, 0, 0)),
                         grid_dispersion=bool(getattr(disc, "mode_grid_dispersion", True)))
    p, u, v, k0 = plane.p, plane.u, plane.v, plane.k0
    r = plane.result
    mi = int(src.mode_index)
    if mi >= len(r.n_complex):
        raise SetupError("mode_index exceeds mode_spec.num_modes")
    direction = 1 if src.direction == "+" else -1
    beta = plane.beta[mi]
    big = 10 ** 9
    lo, hi = [-big] * 3, [big] * 3
    if direction > 0:
        lo[p] = k0
    else:
        hi[p] = k0
    legs = surface_legs(spec, mt, lo, hi, (u, v), (u, v))
    d = spec.primal_steps(p)
    # backward-travelling mode: tangential H flips sign
    # symmetry: the solved plane is one half / quarter of the user's plane and its mode carries
    # unit power on that part — the launched mode carries 1 W over the whole plane (ref source.py:1003)
    sym = getattr(sim, "_symmetry", (0, 0, 0))
    scale = 1.0 + 0j
    for i, a in enumerate((u, v)):
        if sym[a] != 0 and plane.lo[i] == 0:
            scale /= np.sqrt(2.0)
    e_tan = (r.Eu[:, :, mi], r.Ev[:, :, mi])
    if align_to is not None:
        ov = np.vdot(align_to[0], e_tan[0]) + np.vdot(align_to[1], e_tan[1])
        if abs(ov) > 0:
            scale = scale * np.exp(-1j * np.angle(ov))
    fields = {u: scale * r.Eu[:, :, mi], v: scale * r.Ev[:, :, mi], 3 + u: scale * direction * r.Hu[:, :, mi],
              3 + v: scale * direction * r.Hv[:, :, mi]}

    def sample(nb_comp, nb_ijk):
        iu = nb_ijk[:, u] - plane.lo[0]
        iv = nb_ijk[:, v] - plane.lo[1]
        ok = (iu >= 0) & (iu < plane.hi[0] - plane.lo[0]) & (iv >= 0) & (iv < plane.hi[1] - plane.lo[1])
        out = np.zeros(len(nb_ijk), complex)
        for c, arr in fields.items():
            m = ok & (nb_comp == c)
            out[m] = arr[iu[m], iv[m]]
        return out

    comps, ijks, ws = [], [], []
    for key in ("e", "h"):
        comp, ijk, w, nbc, nbi = legs[key]
        val = sample(nbc, nbi)
        if key == "e":
            # incident H sits on the cell centre next to the plane: half a cell of propagation phase
            kc = nbi[:, p]
            dist = 0.5 * d[np.clip(kc, 0, len(d) - 1)]
            val = val * np.exp(-1j * beta * dist)
        comps.append(comp)
        ijks.append(ijk)
        ws.append(w * val)
    return (plane, np.concatenate(comps).astype(np.int32), np.concatenate(ijks).astype(np.int32), np.concatenate(ws),
            (scale * e_tan[0], scale * e_tan[1]))


def chebyshev_frequency_grid(source_time, num_freqs: int) -> np.ndarray:
    """ref source.py:750-758 ``BroadbandSource.frequency_grid``: Chebyshev nodes on freq0 +- CHEB_GRID_WIDTH fwidth."""
    freq_min, freq_max = source_time.frequency_range(num_fwidth=CHEB_GRID_WIDTH)
    freq_avg, freq_diff = 0.5 * (freq_min + freq_max), 0.5 * (freq_max - freq_min)
    uni_points = (2 * np.arange(num_freqs) + 1) / (2 * num_freqs)
    return freq_avg + freq_diff * np.cos(np.pi * np.flip(uni_points))


def _filtered_waveforms(source_time, times: np.ndarray, dt: float, n_terms: int, f_avg: float, f_diff: float) -> np.ndarray:
    """g_m(t) = the source's complex waveform filtered by T_m(x(f)), x = (f - f_avg) / f_diff held at +-1 outside the
    Chebyshev interval (a polynomial must not be extrapolated under the Gaussian's tails: T_6(2) = 1351), m = 0 .. n_terms-1;
    [n_terms, len(times)].  The waveform ~ exp(-i 2 pi f0 t) sits at the NEGATIVE frequencies of numpy's FFT."""
    a = np.asarray(source_time.amp_time(times), complex)
    n = len(a)
    nfft = 1 << int(np.ceil(np.log2(4 * n)))
    spec_a = np.fft.fft(a, nfft)
    f_phys = -np.fft.fftfreq(nfft, dt)
    x = np.clip((f_phys - f_avg) / f_diff, -1.0, 1.0)
    th = np.arccos(x)
    return np.stack([np.fft.ifft(spec_a * np.cos(m * th))[:n] for m in range(n_terms)])


def build_mode_source(disc, mt, src) -> Callable:
    """``num_freqs == 1``: the mode profile of freq0 times the source's waveform.  ``num_freqs > 1`` (ref source.py:737-772
    ``BroadbandSource``: "a Chebyshev interpolation is used" for the frequency dependence of the injected field): the
    leg weights W(r, f_j) of the modes solved at the Chebyshev nodes f_j are expanded as sum_m c_m(r) T_m(x(f)) and the
    source becomes ``num_freqs`` point-source sets on the same nodes — weights c_m(r), waveform g_m(t) = the pulse filtered
    by T_m — so that every frequency of the pulse is launched with (the interpolant of) its own profile, propagation
    constant and impedance.  The reference's realisation is server-side (parity unpinned); pinned here physically: the
    power launched BACKWARDS at the band edges drops by orders of magnitude against num_freqs = 1
    (tests/test_mode_solver.py)."""
    sim, spec, tmesh = disc.sim, disc.spec, disc.tmesh
    st = src.source_time
    dt = spec.dt
    name = getattr(src, "name", None) or "ModeSource"
    nf = int(getattr(src, "num_freqs", 1) or 1)
    plane, comp, ijk, w, e_ref = _leg_weights(disc, mt, src, st.freq0)
    if nf == 1:
        keep = w != 0
        spec.sources.append(PointSourceSet(
            comp=comp[keep], ijk=ijk[keep], w_re=w[keep].real.copy(), w_im=w[keep].imag.copy(),
            wave_e=np.asarray(st.amp_time(tmesh + dt / 2), complex),
            wave_h=np.asarray(st.amp_time(tmesh), complex), name=name))
    else:
        if not hasattr(st, "fwidth"):
            raise SetupError("a broadband ModeSource (num_freqs > 1) needs a source time with a bandwidth (GaussianPulse)")
        fgrid = chebyshev_frequency_grid(st, nf)
        f_min, f_max = st.frequency_range(num_fwidth=CHEB_GRID_WIDTH)
        f_avg, f_diff = 0.5 * (f_min + f_max), 0.5 * (f_max - f_min)
        W = np.stack([_leg_weights(disc, mt, src, float(f), align_to=e_ref)[3] for f in fgrid])       # [nf, points]
        xj = (fgrid - f_avg) / f_diff
        Tm = np.cos(np.arange(nf)[:, None] * np.arccos(np.clip(xj, -1, 1))[None, :])                  # [m, j]
        c = (2.0 / nf) * Tm @ W                                                                        # discrete Chebyshev transform
        c[0] *= 0.5
        keep = np.any(W != 0, axis=0)
        g_e = _filtered_waveforms(st, tmesh + dt / 2, dt, nf, f_avg, f_diff)
        g_h = _filtered_waveforms(st, tmesh, dt, nf, f_avg, f_diff)
        for m in range(nf):
            spec.sources.append(PointSourceSet(
                comp=comp[keep], ijk=ijk[keep], w_re=c[m][keep].real.copy(), w_im=c[m][keep].imag.copy(),
                wave_e=g_e[m], wave_h=g_h[m], name=f"{name}[T{m}]"))
    disc.mode_planes = getattr(disc, "mode_planes", {})
    disc.mode_planes[id(src)] = plane

    def fn(freqs):
        return st.spectrum(tmesh, np.asarray(freqs, float), dt, complex_fields=spec.bloch is not None)
    return fn


def _interp_nodes(arr: np.ndarray, src_u, src_v, dst_u, dst_v) -> np.ndarray:
    from .data import interp_axis
    return interp_axis(interp_axis(arr, src_u, dst_u, 0), src_v, dst_v, 1)


def colocated_mode(plane: ModePlane, spec: SolverSpec, mi: int, direction: int, dst_u, dst_v) -> Dict[str, np.ndarray]:
    """Tangential mode fields interpolated from their Yee nodes to the nodes (dst_u, dst_v) — the
    same linear colocation the monitor data undergo (ref dataset.py:83-147)."""
    r = plane.result
    b = spec.boundaries
    ub = b[plane.u][plane.lo[0]:plane.hi[0] + 1]
    vb = b[plane.v][plane.lo[1]:plane.hi[1] + 1]
    uc, vc = 0.5 * (ub[1:] + ub[:-1]), 0.5 * (vb[1:] + vb[:-1])
    return {
        "Eu": _interp_nodes(r.Eu[:, :, mi], uc, vb[:-1], dst_u, dst_v),
        "Ev": _interp_nodes(r.Ev[:, :, mi], ub[:-1], vc, dst_u, dst_v),
        "Hu": direction * _interp_nodes(r.Hu[:, :, mi], ub[:-1], vc, dst_u, dst_v),
        "Hv": direction * _interp_nodes(r.Hv[:, :, mi], uc, vb[:-1], dst_u, dst_v),
    }


def overlap(a: Dict[str, np.ndarray], b: Dict[str, np.ndarray], w: np.ndarray) -> complex:
    """(a, b) = 1/4 int (E_a* x H_b + H_a* x E_b) . n dA   (ref monitor_data.py:640-697), n = +w."""
    e1h2 = np.conj(a["Eu"]) * b["Hv"] - np.conj(a["Ev"]) * b["Hu"]
    h1e2 = np.conj(a["Hu"]) * b["Ev"] - np.conj(a["Hv"]) * b["Eu"]
    return complex(0.25 * np.sum((e1h2 - h1e2) * w))


@dataclass
class ModeData:
    """Mirror of tidy3d ModeData (ref monitor_data.py:1223): complex amplitudes of the forward (+)
    and backward (-) modes, ``amps`` dims (direction, f, mode_index); ``n_complex`` (f, mode_index).
    Modes are normalised to unit directed flux, amps = (mode, field)/(mode, mode) (CHANGELOG:534-538).
    ``mode_power`` (not part of the reference's container): the flux (mode, mode) a unit-amplitude mode carries
    AS THIS MONITOR MEASURES IT — on fields colocated to the plane like the monitor data — so that
    |amps|^2 mode_power is directly comparable with a FluxMonitor on the same plane."""
    monitor: object
    amps: object = None
    n_complex: object = None
    mode_power: object = None


def grid_correction(spec: SolverSpec, p: int, pos: float, beta) -> Tuple[complex, complex]:
    """Factors that bring a mode given on the plane ``pos`` to what a monitor records there (the reference's
    ``ModeSolver._grid_correction``, ref plugins/mode/mode_solver.py:847-904): monitor fields are linear
    interpolations between the grid planes that carry tangential E (cell boundaries along the normal axis p)
    resp. tangential H (cell centres); a field ~ exp(i beta w) interpolated that way is multiplied by the
    interpolated phase — cos(beta dw/2) for H on a uniform grid when the plane sits on a cell boundary."""
    b = np.asarray(spec.boundaries[p], float)
    c = 0.5 * (b[1:] + b[:-1])
    out = []
    for nodes in (b, c):
        if len(nodes) < 2:
            out.append(1.0 + 0j)
            continue
        ph = np.exp(1j * beta * (nodes - pos))
        out.append(complex(np.interp(pos, nodes, ph.real) + 1j * np.interp(pos, nodes, ph.imag)))
    return out[0], out[1]


def mode_monitor_data(disc, plan, raw, norm):
    from .data import DataArray, FieldData, _diff_area, _field_container
    mon = plan.monitor
    spec = disc.spec
    fp = plan.fields[0]
    box = fp.box if fp.box is not None else mon.geometry       # symmetry: the image inside the computed half
    freqs = np.asarray(mon.freqs, float)
    fd = _field_container(FieldData, mon, spec, fp, raw[fp.spec_name], "f", freqs, disc.sim.center,
                          np.complex128).normalize(norm)
    zd = [a for a in range(3) if mon.size[a] == 0]
    p = zd[0]
    u, v = (p + 1) % 3, (p + 2) % 3
    names = "xyz"
    gu = np.asarray(fd["E" + names[u]].coords[names[u]])
    gv = np.asarray(fd["E" + names[u]].coords[names[v]])

    def plane_vals(comp):
        arr = fd[comp].values                     # (x, y, z, f)
        arr = np.take(arr, 0, axis=p)             # drop the normal axis -> remaining spatial dims in x,y,z order
        rem = [a for a in range(3) if a != p]
        if rem != [u, v]:
            arr = np.swapaxes(arr, 0, 1)
        return arr                                # (u, v, f)
    F = {"Eu": plane_vals("E" + names[u]), "Ev": plane_vals("E" + names[v]),
         "Hu": plane_vals("H" + names[u]), "Hv": plane_vals("H" + names[v])}

    # symmetry planes inside the monitor plane: the decomposition runs on the computed part; a
    # component that is odd across the plane and straddles it is zero ON it (the one-sided
    # colocation returned the upper node), and amplitudes refer to modes of unit power over the
    # whole plane: sqrt(2) per halving
    from .data import _SYM_EIG
    sym = tuple(getattr(disc, "symmetry", (0, 0, 0)))
    comp_of = {"Eu": "E" + names[u], "Ev": "E" + names[v], "Hu": "H" + names[u], "Hv": "H" + names[v]}
    wall = []                                    # (array axis, in-plane axis) of symmetry walls touched
    for i, (a, g) in enumerate(((u, gu), (v, gv))):
        c = disc.sim.center[a]
        if sym[a] != 0 and len(g) and abs(g[0] - c) <= 1e-9 * max(1.0, abs(c)):
            wall.append((i, a))
    amp_scale = np.sqrt(2.0) ** len(wall)

    def zero_on_walls(fields):
        for key, cname in comp_of.items():
            for i, a in wall:
                straddles = (cname[0] == "E") == ("xyz".index(cname[1]) == a)
                if straddles and sym[a] * _SYM_EIG[cname][a] < 0:
                    idx = [slice(None)] * fields[key].ndim
                    idx[i] = 0
                    fields[key][tuple(idx)] = 0.0
        return fields
    F = zero_on_walls({k: np.array(a) for k, a in F.items()})
    w = _diff_area(box, None, None, p, gu, gv)
    nm = int(mon.mode_spec.num_modes)
    amps = np.zeros((2, len(freqs), nm), complex)
    power = np.zeros((2, len(freqs), nm))
    neff = np.zeros((len(freqs), nm), complex)
    pos = float(box.center[p])
    for i, f in enumerate(freqs):
        plane = mode_profile(spec, box, mon.mode_spec, float(f), sym)
        neff[i] = plane.result.n_complex
        Ff = {k: a[:, :, i] for k, a in F.items()}
        for m in range(nm):
            for d_i, direction in enumerate((1, -1)):
                M = zero_on_walls(colocated_mode(plane, spec, m, direction, gu, gv))
                ce, chh = grid_correction(spec, p, pos, direction * plane.beta[m])
                M = {k: a * (ce if k[0] == "E" else chh) for k, a in M.items()}
                mm = overlap(M, M, w)
                amps[d_i, i, m] = amp_scale * overlap(M, Ff, w) / mm
                power[d_i, i, m] = mm.real / amp_scale ** 2
    coords = {"direction": np.array(["+", "-"]), "f": freqs, "mode_index": np.arange(nm)}
    return ModeData(monitor=mon, amps=DataArray(amps.astype(np.complex64), coords),
                    n_complex=DataArray(neff, {"f": freqs, "mode_index": np.arange(nm)}),
                    mode_power=DataArray(power, coords))
