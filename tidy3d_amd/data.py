"""Output side of the drop-in boundary: raw monitor buffers -> monitor data containers.

Mirror of the part of ``tidy3d.components.data`` a solver must produce
(SURVEY.md section 8(a) A12-A13): ``SimulationData`` holding one ``FieldData`` /
``FieldTimeData`` / ``FluxData`` / ``FluxTimeData`` per monitor, on exactly the
coordinates the reference's fake backend ``run_emulated`` uses
(ref tests/utils.py:862-1035), normalised by the source spectrum
(ref sim_data.py:931-953, monitor_data.py:972-979, :1955-1960), complex64/float32
(ref monitor.py:35-36).  ``DataArray`` is a small numpy-backed stand-in for the
xarray subclass (ref data_array.py:65) with the same ``dims``/``coords``/``values``/
``sel``/``isel`` surface; ``tidy3d_amd.adapter.to_tidy3d`` converts to the real
classes when tidy3d is importable.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import schema as td
from .discretize import Discretization, FieldPlan, MonitorPlan
from .exceptions import DataError
from .spec import BC_PERIODIC, COMP_ID, SolverSpec


class DataArray:
    """Minimal labelled array: ``values`` + ordered ``coords`` (ref data_array.py:65-140)."""

    def __init__(self, values, coords: Dict[str, Sequence], dims: Optional[Tuple[str, ...]] = None):
        self.dims = tuple(dims) if dims is not None else tuple(coords.keys())
        self.coords = {d: np.asarray(coords[d]) for d in self.dims}
        self.values = np.asarray(values)
        if self.values.shape != tuple(len(self.coords[d]) for d in self.dims):
            raise DataError(f"DataArray shape {self.values.shape} does not match coords "
                            f"{[len(self.coords[d]) for d in self.dims]}")

    # numpy interop
    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    @property
    def shape(self):
        return self.values.shape

    @property
    def dtype(self):
        return self.values.dtype

    @property
    def real(self):
        return DataArray(self.values.real, self.coords, self.dims)

    @property
    def imag(self):
        return DataArray(self.values.imag, self.coords, self.dims)

    @property
    def abs(self):
        return DataArray(np.abs(self.values), self.coords, self.dims)

    def conj(self):
        return DataArray(np.conj(self.values), self.coords, self.dims)

    def _binary(self, other, op):
        o = other.values if isinstance(other, DataArray) else other
        return DataArray(op(self.values, o), self.coords, self.dims)

    def __mul__(self, o): return self._binary(o, np.multiply)
    __rmul__ = __mul__
    def __add__(self, o): return self._binary(o, np.add)
    def __sub__(self, o): return self._binary(o, np.subtract)
    def __truediv__(self, o): return self._binary(o, np.divide)

    def isel(self, **indexers):
        vals, coords, dims = self.values, dict(self.coords), list(self.dims)
        for d, idx in indexers.items():
            ax = dims.index(d)
            vals = np.take(vals, idx, axis=ax)
            if np.ndim(idx) == 0:
                dims.pop(ax)
                coords.pop(d)
            else:
                coords[d] = coords[d][idx]
        return DataArray(vals, coords, tuple(dims))

    def sel(self, method: Optional[str] = None, **indexers):
        out = self
        for d, v in indexers.items():
            c = out.coords[d]
            if np.ndim(v) == 0:
                i = int(np.argmin(np.abs(c - v)))
                if method != "nearest" and not np.isclose(c[i], v, rtol=1e-9, atol=0):
                    raise KeyError(f"{v} not found along '{d}' (use method='nearest')")
                out = out.isel(**{d: i})
            else:
                ii = [int(np.argmin(np.abs(c - x))) for x in v]
                out = out.isel(**{d: ii})
        return out

    def interp(self, **points):
        """Linear interpolation along the given dims (edge values held outside)."""
        out = self
        for d, v in points.items():
            ax = out.dims.index(d)
            tgt = np.atleast_1d(np.asarray(v, float))
            vals = interp_axis(out.values, out.coords[d], tgt, ax)
            coords = dict(out.coords)
            coords[d] = tgt
            out = DataArray(vals, coords, out.dims)
            if np.ndim(v) == 0:
                out = out.isel(**{d: 0})
        return out

    def squeeze(self):
        keep = [d for d in self.dims if len(self.coords[d]) != 1]
        return DataArray(self.values.reshape([len(self.coords[d]) for d in keep]),
                         {d: self.coords[d] for d in keep}, tuple(keep))

    def __repr__(self):
        return f"DataArray(dims={self.dims}, shape={self.shape}, dtype={self.dtype})"


def interp_axis(arr: np.ndarray, src: np.ndarray, dst: np.ndarray, axis: int) -> np.ndarray:
    """Linear interpolation of ``arr`` from coordinates ``src`` to ``dst`` along ``axis``; values
    are held constant beyond the ends (the raw boxes are clipped at the domain walls)."""
    src = np.asarray(src, float)
    dst = np.asarray(dst, float)
    if len(src) == len(dst) and np.array_equal(src, dst):
        return arr
    if len(src) == 1:
        return np.repeat(arr, len(dst), axis=axis)
    j = np.clip(np.searchsorted(src, dst, side="right") - 1, 0, len(src) - 2)
    w = np.clip((dst - src[j]) / (src[j + 1] - src[j]), 0.0, 1.0)
    a0 = np.take(arr, j, axis=axis)
    a1 = np.take(arr, j + 1, axis=axis)
    shp = [1] * arr.ndim
    shp[axis] = -1
    w = w.reshape(shp)
    return a0 * (1 - w) + a1 * w


# ----------------------------------------------------------------------------------------------
# monitor data containers
# ----------------------------------------------------------------------------------------------

@dataclass
class _FieldLike:
    monitor: object
    Ex: Optional[DataArray] = None
    Ey: Optional[DataArray] = None
    Ez: Optional[DataArray] = None
    Hx: Optional[DataArray] = None
    Hy: Optional[DataArray] = None
    Hz: Optional[DataArray] = None
    symmetry: Tuple[int, int, int] = (0, 0, 0)
    symmetry_center: Optional[Tuple[float, float, float]] = None
    grid_expanded: Optional[Dict[str, np.ndarray]] = None      # boundaries x, y, z of the sub-grid

    @property
    def field_components(self) -> Dict[str, DataArray]:
        return {k: getattr(self, k) for k in ("Ex", "Ey", "Ez", "Hx", "Hy", "Hz")
                if getattr(self, k) is not None}

    def __getitem__(self, name):
        return self.field_components[name]


@dataclass
class FieldData(_FieldLike):
    """ref monitor_data.py:927 — dims (x, y, z, f), complex64."""

    def normalize(self, spectrum_fn: Callable) -> "FieldData":
        """Divide by the source spectrum (ref monitor_data.py:972-979)."""
        out = FieldData(monitor=self.monitor, symmetry=self.symmetry,
                        symmetry_center=self.symmetry_center, grid_expanded=self.grid_expanded)
        for k, v in self.field_components.items():
            s = np.asarray(spectrum_fn(v.coords["f"]))
            setattr(out, k, DataArray((v.values / s[None, None, None, :]).astype(np.complex64),
                                      v.coords, v.dims))
        return out

    @property
    def flux(self) -> DataArray:
        """Power through a planar monitor (ref monitor_data.py:582-618), FluxDataArray(f)."""
        axis = [a for a in range(3) if self.monitor.size[a] == 0]
        if len(axis) != 1:
            raise DataError("flux needs a planar monitor")
        return plane_flux(self, axis[0], self.monitor)


@dataclass
class FieldTimeData(_FieldLike):
    """ref monitor_data.py:1119 — dims (x, y, z, t), float32."""


@dataclass
class FluxData:
    """ref monitor_data.py:1898 — flux(f) float32."""
    monitor: object
    flux: DataArray = None

    def normalize(self, spectrum_fn: Callable) -> "FluxData":
        """Divide by |spectrum|^2 (ref monitor_data.py:1955-1960)."""
        s = np.abs(np.asarray(spectrum_fn(self.flux.coords["f"]))) ** 2
        return FluxData(monitor=self.monitor,
                        flux=DataArray((self.flux.values / s).astype(np.float32), self.flux.coords))


@dataclass
class PermittivityData:
    """ref monitor_data.py:1190-1222: eps_xx / eps_yy / eps_zz (x, y, z, f) at the Ex / Ey / Ez
    Yee locations inside the monitor."""
    monitor: object
    eps_xx: DataArray = None
    eps_yy: DataArray = None
    eps_zz: DataArray = None
    grid_expanded: Optional[Dict[str, np.ndarray]] = None

    @property
    def field_components(self) -> Dict[str, DataArray]:
        return {"eps_xx": self.eps_xx, "eps_yy": self.eps_yy, "eps_zz": self.eps_zz}


@dataclass
class FluxTimeData:
    """ref monitor_data.py:1992 — flux(t) float32."""
    monitor: object
    flux: DataArray = None


@dataclass
class SimulationData:
    """ref sim_data.py:826: simulation + tuple of monitor data + solver log + diverged flag."""

    simulation: object
    data: Tuple[object, ...] = ()
    log: Optional[str] = None
    diverged: bool = False

    def __post_init__(self):
        names = [d.monitor.name for d in self.data]
        sim_names = [m.name for m in self.simulation.monitors]
        for n in names:          # ref base_sim/data/sim_data.py:54-84
            if n not in sim_names:
                raise DataError(f"Data with monitor name '{n}' supplied but not found in the "
                                "original Simulation.")
        if len(set(names)) != len(names):
            raise DataError("monitor data names must be unique")

    @property
    def monitor_data(self) -> Dict[str, object]:
        return {d.monitor.name: d for d in self.data}

    def __getitem__(self, monitor_name: str):
        try:
            return self.monitor_data[monitor_name]
        except KeyError:
            raise DataError(f"monitor '{monitor_name}' not found in the simulation data") from None

    @property
    def final_decay_value(self) -> float:
        """ref sim_data.py:916-929."""
        if self.log is None:
            raise DataError("No log string in the SimulationData object, can't find final decay value.")
        lines = [ln for ln in self.log.split("\n") if "field decay" in ln]
        return float(lines[-1].split("field decay: ")[-1]) if lines else 1.0


# ----------------------------------------------------------------------------------------------
# post-processing
# ----------------------------------------------------------------------------------------------

def _colocate_box(raw: np.ndarray, spec: SolverSpec, fp: FieldPlan, ic: int, fname: str) -> np.ndarray:
    """raw [n_lead, n_comps, bz, by, bx] -> [n_lead, nz_t, ny_t, nx_t] on the target coordinates of
    field ``fname`` (separable linear interpolation from the component's Yee coordinates; this is
    both the colocation to the primal nodes, ref dataset.py:83-147 / CHANGELOG:467-470, and the
    snapping of zero-size dimensions to the exact plane position, ref simulation.py:1019-1026)."""
    comp = COMP_ID[fname]
    yee = spec.yee_coords(comp)
    arr = raw[:, ic]
    for a in range(3):
        lo = fp.lo[a]
        n = arr.shape[3 - a]
        src = yee[a][lo:lo + n]
        on_boundary = (a == comp % 3) == (comp >= 3)      # nodes on the grid lines along a
        if on_boundary and lo + n == spec.shape[a] and spec.bc[a][1] != BC_PERIODIC and n > 1:
            # the top wall is not stored: wall-tangential E / wall-normal H are zero on it
            src = np.append(src, spec.boundaries[a][-1])
            pad = [(0, 0)] * arr.ndim
            pad[3 - a] = (0, 1)
            arr = np.pad(arr, pad)
        elif lo == 0 and n == spec.shape[a] and n > 1 and spec.bc[a][0] == BC_PERIODIC and spec.bc[a][1] == BC_PERIODIC:
            # a box that spans a whole PERIODIC axis: the samples one period away close the interpolation at both ends (the
            # node on the plus face IS node 0; without this the last target took the value of its neighbour — 1e-3 of a flux
            # through a unit cell, found against the symmetric half-cell run, tests/test_symmetry.py)
            period = float(spec.boundaries[a][-1] - spec.boundaries[a][0])
            first = np.take(arr, [0], axis=3 - a)
            last = np.take(arr, [n - 1], axis=3 - a)
            phi = 0.0 if getattr(spec, "bloch", None) is None else float(spec.bloch[a])
            if phi != 0.0:                      # Bloch axis: F(r + L) = exp(i phi) F(r)   (ref boundary.py:55-79)
                first, last = first * np.exp(1j * phi), last * np.exp(-1j * phi)
            src = np.concatenate([[src[-1] - period], src, [src[0] + period]])
            arr = np.concatenate([last, arr, first], axis=3 - a)
        arr = interp_axis(arr, src, fp.target[fname][a], axis=3 - a)
    return arr


def _extended_subspace(coords: np.ndarray, ind_beg: int, ind_end: int, periodic: bool) -> np.ndarray:
    """Boundaries [ind_beg, ind_end) with out-of-range indices padded by the periodic image or the
    mirror image about the end planes (ref grid.py:546-603)."""
    coords = np.asarray(coords, float)
    padded = coords
    n = coords.size - 1
    reverse = True
    while ind_beg < 0:
        if periodic or not reverse:
            padded = np.concatenate([coords[:-1] + (padded[0] - coords[-1]), padded])
            reverse = True
        else:
            padded = np.concatenate([(padded[0] + coords[0]) - coords[:0:-1], padded])
            reverse = False
        ind_beg += n
        ind_end += n
    reverse = True
    while ind_end >= padded.size:
        if periodic or not reverse:
            padded = np.concatenate([padded, coords[1:] + (padded[-1] - coords[0])])
            reverse = True
        else:
            padded = np.concatenate([padded, (padded[-1] + coords[-1]) - coords[-2::-1]])
            reverse = False
    return padded[ind_beg:ind_end]


def _grid_expanded(spec: SolverSpec, fp: FieldPlan) -> Dict[str, np.ndarray]:
    """``Simulation.discretize_monitor`` (ref simulation.py:1068-1073): sub-grid of the span (padded
    beyond the walls, ref :973-987), zero-size monitor dimensions snapped to the monitor position
    (ref grid.py:605-627), single-pixel simulation axes to the grid centre (ref :1019-1026)."""
    out = {}
    for a, d in enumerate("xyz"):
        b = np.asarray(spec.boundaries[a], float)
        periodic = spec.bc[a][0] == BC_PERIODIC and spec.bc[a][1] == BC_PERIODIC
        sub = _extended_subspace(b, int(fp.span[a, 0]), int(fp.span[a, 1]) + 1, periodic)
        if fp.box is not None and fp.box.size[a] == 0:
            c = float(fp.box.center[a])
            sub = np.array([c, c])
        if len(b) == 2:
            c = 0.5 * (b[0] + b[1])
            sub = np.array([c, c])
        out[d] = sub
    return out


def _diff_area(plan_box: td.Box, coords1: np.ndarray, coords2: np.ndarray, axis: int,
               gb1: np.ndarray, gb2: np.ndarray) -> np.ndarray:
    """Integration weights for values colocated to the grid boundaries gb1 x gb2, truncated to
    the monitor bounds (ref monitor_data.py:426-463)."""
    (lo, hi) = plan_box.bounds
    t = [a for a in range(3) if a != axis]

    def sizes(gb, mlo, mhi):
        if gb.size <= 1:
            return np.array([1.0])
        c = 0.5 * (gb[1:] + gb[:-1])
        c = np.concatenate(([gb[0]], c, [gb[-1]]))
        c = np.clip(c, mlo, mhi)
        return c[1:] - c[:-1]
    s1 = sizes(gb1, lo[t[0]], hi[t[0]])
    s2 = sizes(gb2, lo[t[1]], hi[t[1]])
    return np.outer(s1, s2)


def plane_flux(fd: _FieldLike, axis: int, mon, sign: float = 1.0, box: Optional[td.Box] = None,
               lead: str = "f") -> DataArray:
    """0.5 Re(E x H*) . n integrated over the plane (ref monitor_data.py:582-618); for real
    time-domain fields the factor 0.5 and the conjugate drop out (ref monitor_data.py:1158)."""
    t = [a for a in range(3) if a != axis]
    d1, d2 = "xyz"[t[0]], "xyz"[t[1]]
    e1, e2 = fd["E" + d1], fd["E" + d2]
    h1, h2 = fd["H" + d1], fd["H" + d2]
    # (x, y, z, lead) -> drop the normal axis
    def sq(v):
        return np.take(v.values, 0, axis=axis)
    if lead == "f":
        s = 0.5 * np.real(sq(e1) * np.conj(sq(h2)) - sq(e2) * np.conj(sq(h1)))
    else:
        s = sq(e1) * sq(h2) - sq(e2) * sq(h1)
    if axis == 1:       # (x, z) ordering is left-handed w.r.t. +y  (ref monitor_data.py:491-493)
        s = -s
    # colocated data sit on the sub-grid boundaries with the last one dropped
    # (= colocation_boundaries, ref monitor_data.py:372-395), which is what _diff_area expects
    gb1, gb2 = np.asarray(e1.coords[d1]), np.asarray(e1.coords[d2])
    w = _diff_area(box or mon.geometry, None, None, axis, gb1, gb2)
    flux = sign * np.tensordot(w, s, axes=([0, 1], [0, 1]))
    return DataArray(flux.astype(np.float32), {lead: e1.coords[lead]})


# parity of a field component under the mirror x_a -> -x_a (ref dataset.py:210-220); the value on the
# lower side is  symmetry[a] * eigenvalue * value(mirror point)  (ref monitor_data.py:238-284)
_SYM_EIG = {"Ex": (-1, 1, 1), "Ey": (1, -1, 1), "Ez": (1, 1, -1),
            "Hx": (1, -1, -1), "Hy": (-1, 1, -1), "Hz": (-1, -1, 1)}


def expand_symmetry(arr: np.ndarray, half: Sequence[np.ndarray], full: Sequence[np.ndarray], fname: str,
                    symmetry, center) -> np.ndarray:
    """arr (x, y, z, lead) on the computed upper-half coordinates -> the monitor's own coordinates:
    points below a symmetry plane take the nearest mirror sample times the parity."""
    for a in range(3):
        if symmetry[a] == 0:
            continue
        cf, ch, c = np.asarray(full[a], float), np.asarray(half[a], float), center[a]
        flip = cf < c
        want = np.where(flip, 2 * c - cf, cf)
        idx = np.abs(ch[None, :] - want[:, None]).argmin(axis=1)
        arr = np.take(arr, idx, axis=a)
        parity = float(symmetry[a] * _SYM_EIG[fname][a])
        sign = np.where(flip, parity, 1.0)
        # the mirror image of the lowest grid line is the top wall, which a recorded span never
        # includes: wall-tangential E and wall-normal H vanish there (every outer face is PEC-backed)
        if len(ch) > 1:
            sign = np.where(want > ch[-1] + 0.5 * (ch[-1] - ch[-2]), 0.0, sign)
        # a colocated sample ON the plane of a component whose nodes straddle it (E_a along a, H_b
        # along a != b) is the mean of the two mirror nodes: zero for odd parity (the half-domain
        # interpolation, which has no node below the plane, returned the upper node instead)
        straddles = (fname[0] == "E") == ("xyz".index(fname[1]) == a)
        if parity < 0 and straddles:
            sign = np.where(np.isclose(cf, c, rtol=0, atol=1e-9 * max(1.0, abs(c))), 0.0, sign)
        shape = [1] * arr.ndim
        shape[a] = len(cf)
        arr = arr * sign.reshape(shape).astype(arr.real.dtype)
    return arr


def _field_container(cls, mon, spec: SolverSpec, fp: FieldPlan, raw: np.ndarray, lead: str,
                     lead_coords: np.ndarray, sim_center, dtype, full=None):
    """``full`` = (FieldPlan on the full grid, full-grid spec, symmetry) when the solver ran on the
    symmetry-reduced domain."""
    kw = {}
    for ic, fname in enumerate(fp.fields):
        arr = _colocate_box(raw, spec, fp, ic, fname)           # [lead, z, y, x]
        arr = np.transpose(arr, (3, 2, 1, 0)).astype(dtype)      # (x, y, z, lead)
        tx, ty, tz = fp.target[fname]
        if full is not None:
            fp_full, _, symmetry = full
            arr = expand_symmetry(arr, (tx, ty, tz), fp_full.target[fname], fname, symmetry, sim_center)
            tx, ty, tz = fp_full.target[fname]
        kw[fname] = DataArray(arr, {"x": tx, "y": ty, "z": tz, lead: lead_coords})
    gexp = _grid_expanded(spec, fp) if full is None else _grid_expanded(full[1], full[0])
    return cls(monitor=mon, symmetry=(0, 0, 0), symmetry_center=tuple(sim_center),
               grid_expanded=gexp, **kw)


def medium_eps_table(spec: SolverSpec, freq: float) -> np.ndarray:
    """Complex eps_r(freq) per entry of ``spec.media`` (ref medium.py:1016-1038 conductivity term,
    :2900-2913 pole-residue sum; PEC -> pec_val, ref constants.py:64)."""
    from .constants import EPSILON_0
    w = 2 * np.pi * freq
    tab = []
    for med in spec.media:
        if med.pec:
            tab.append(-1e8 + 0j)
            continue
        e = med.eps_inf + 0j
        if med.sigma:
            e += 1j * med.sigma / (w * EPSILON_0)
        for a, c in med.poles:
            e -= c / (1j * w + a) + np.conj(c) / (1j * w + np.conj(a))
        tab.append(e)
    return np.array(tab)


def permittivity_data(sim, spec: SolverSpec, plan) -> "PermittivityData":
    """PermittivityMonitor: background, then structures in order, sampled per component at its own
    Yee nodes of the monitor sub-grid — zero-size dimensions snapped to the monitor plane — i.e. the
    reference's epsilon_on_grid recipe (ref simulation.py:1135-1241) with the staircase rule the
    kernels use (identical to ``spec.mat_idx`` wherever the nodes coincide)."""
    mon, fp = plan.monitor, plan.fields[0]
    freqs = np.asarray(mon.freqs, float)
    kw = {}
    for c, (name, fname) in enumerate((("eps_xx", "Ex"), ("eps_yy", "Ey"), ("eps_zz", "Ez"))):
        tx, ty, tz = fp.target[fname]
        X, Y, Z = np.meshgrid(tx, ty, tz, indexing="ij")
        vals = np.empty(X.shape + (len(freqs),), dtype=complex)
        vals[...] = np.asarray(sim.medium.eps_model(freqs), complex)
        for st in sim.structures:
            if isinstance(st.medium, td.Medium2D):
                continue                   # (a sheet has no volume; its volumetric equivalent on the plane's nodes is not reported here)
            inside = st.geometry.inside(X, Y, Z)
            if isinstance(st.medium, td.FullyAnisotropicMedium):      # the tensor's own diagonal (ref medium.py eps_comp), not what the sweep uses
                vals[inside] = st.medium.eps_tensor[c, c]
                continue
            med = st.medium.component(c) if hasattr(st.medium, "component") else st.medium
            if getattr(med, "is_pec", False):
                vals[inside] = -1e8 + 0j                      # pec_val, ref constants.py:64
            else:
                vals[inside] = np.asarray(med.eps_model(freqs), complex)
        kw[name] = DataArray(vals, {"x": tx, "y": ty, "z": tz, "f": freqs})
    return PermittivityData(monitor=mon, grid_expanded=_grid_expanded(spec, fp), **kw)


def source_spectrum_fn(disc: Discretization, index: Optional[int]) -> Callable:
    """ref sim_data.py:931-953: spectrum / amplitude / exp(i phase) of source ``index``."""
    sim = disc.sim
    if index is None or len(sim.sources) == 0:
        return lambda f: np.ones(len(np.atleast_1d(f)), complex)
    st = sim.sources[index].source_time
    base = disc.source_norm[index]

    def fn(freqs):
        return base(freqs) / st.amplitude / np.exp(1j * st.phase)
    return fn


def assemble(disc: Discretization, raw: Dict[str, np.ndarray], log: str = "", diverged: bool = False,
             n_steps_run: Optional[int] = None, device_lib=None, device: int = 0) -> SimulationData:
    """Raw monitor buffers (name -> array as returned by the engine / oracle) -> SimulationData.
    ``n_steps_run``: time steps actually taken (a run that stopped on the shutoff criterion or diverged took
    fewer than ``spec.n_steps``): time-domain monitors then keep only the samples that were recorded — steps
    after the stop never happened and must not come back as zeros on the full ``tmesh`` axis.
    ``device_lib``: the loaded HIP library — projection and diffraction monitors then integrate on the device
    (``device``: the GPU that ran the solve)."""
    sim, spec = disc.sim, disc.spec

    def recorded(steps):
        """number of leading entries of ``steps`` that were reached"""
        steps = np.asarray(steps)
        return len(steps) if n_steps_run is None else int(np.searchsorted(steps, int(n_steps_run), side="left"))

    norm = source_spectrum_fn(disc, sim.normalize_index)
    out = []
    sym = tuple(getattr(disc, "symmetry", (0, 0, 0)))
    plans_full = disc.plans_full if any(sym) else [None] * len(disc.plans)

    def full_of(pf, i):
        return None if pf is None else (pf.fields[i], disc.spec_full, sym)

    for plan, pfull in zip(disc.plans, plans_full):
        mon = plan.monitor
        if plan.kind == "field":
            fp = plan.fields[0]
            fd = _field_container(FieldData, mon, spec, fp, raw[fp.spec_name], "f",
                                  np.asarray(mon.freqs, float), sim.center, np.complex64, full_of(pfull, 0))
            out.append(fd.normalize(norm))
        elif plan.kind == "field_time":
            fp = plan.fields[0]
            nk = recorded(plan.steps)
            t = disc.tmesh[plan.steps[:nk]]
            # real fields; complex ones under Bloch boundaries (ref simulation.py:4396-4411 complex_fields)
            out.append(_field_container(FieldTimeData, mon, spec, fp, raw[fp.spec_name][:nk], "t", t,
                                        sim.center, np.complex64 if spec.bloch is not None else np.float32,
                                        full_of(pfull, 0)))
        elif plan.kind in ("flux", "flux_time"):
            is_time = plan.kind == "flux_time"
            lead = "t" if is_time else "f"
            nk = recorded(plan.steps) if is_time else None
            lead_coords = disc.tmesh[plan.steps[:nk]] if is_time else np.asarray(mon.freqs, float)
            total = None
            from .discretize import flux_surfaces
            for isurf, (fp, (sname, box, axis, sign)) in enumerate(zip(plan.fields, flux_surfaces(mon))):
                class _M:
                    pass
                m = _M()
                m.size, m.center, m.geometry = box.size, box.center, box
                # the time-domain flux of a complex-field (Bloch) run is that of the physical field Re(E), Re(H)
                fd = _field_container(FieldTimeData if is_time else FieldData, m, spec, fp,
                                      np.real(raw[fp.spec_name][:nk]) if is_time else raw[fp.spec_name], lead, lead_coords, sim.center,
                                      np.float64 if is_time else np.complex128, full_of(pfull, isurf))
                fl = plane_flux(fd, axis, m, sign=sign, box=box, lead=lead)
                total = fl if total is None else DataArray(total.values + fl.values, fl.coords)
            if is_time:
                out.append(FluxTimeData(monitor=mon, flux=DataArray(total.values.astype(np.float32),
                                                                   total.coords)))
            else:
                out.append(FluxData(monitor=mon, flux=total).normalize(norm))
        elif plan.kind == "mode":
            from .modesource import mode_monitor_data
            out.append(mode_monitor_data(disc, plan, raw, norm))
        elif plan.kind.startswith("projection_"):
            from . import projection
            fn = {"projection_angle": projection.project_angle, "projection_cartesian": projection.project_cartesian,
                  "projection_kspace": projection.project_kspace}[plan.kind]
            out.append(fn(disc, plan, raw, norm, lib=device_lib, device=device))
        elif plan.kind == "mode_solver":
            from .plugins.mode import ModeSolver
            ms = ModeSolver(simulation=sim, plane=mon.geometry, mode_spec=mon.mode_spec, freqs=mon.freqs,
                            direction=mon.direction, colocate=bool(mon.colocate))
            md = ms.solve(spec=spec, disc=disc if any(sym) else None)
            md.monitor = mon
            md.grid_expanded = _grid_expanded(spec, plan.fields[0]) if pfull is None else _grid_expanded(disc.spec_full, pfull.fields[0])
            out.append(md)
        elif plan.kind == "diffraction":
            from . import projection
            out.append(projection.diffraction(disc, plan, raw, norm, lib=device_lib, device=device))
        elif plan.kind == "permittivity":
            out.append(permittivity_data(sim, spec if pfull is None else disc.spec_full,
                                         plan if pfull is None else pfull))
        else:
            raise DataError(f"unknown monitor plan kind '{plan.kind}'")
    return SimulationData(simulation=sim, data=tuple(out), log=log, diverged=diverged)
