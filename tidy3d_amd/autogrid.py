"""AutoGrid: the graded non-uniform mesher behind ``GridSpec.auto`` — tidy3d's DEFAULT grid.

Restatement of the reference's ``GradedMesher`` (ref components/grid/mesher.py:72-1270) and of
``AutoGrid._make_coords_initial`` (ref components/grid/grid_spec.py:386-520) so that a Simulation
built with the reference's defaults lands on the SAME cell boundaries here.  Two stages per axis:

1. ``parse_structures`` (ref mesher.py:133-308): project the bounding boxes of all structures on the
   axis, walking from the topmost structure down so that covered structures do not contribute,
   and obtain intervals with a largest admissible step each — ``wavelength / (n * min_steps_per_wvl)``
   of the densest structure present (ref :473-521), mesh-override structures replacing that rule.
2. ``grid_multiple_intervals`` (ref mesher.py:637-731): fill every interval with steps that respect its
   maximum, agree with the neighbours' edge steps within ``max_scale`` and sum to the interval length:
   geometric growth / plateau / decay sequences with the left-over length absorbed by one extra step
   or a uniform rescale (ref :816-1220).

Pinned by running the reference's own mesher (tests/golden/make_golden.py; its three third-party
calls are shimmed in oracle/tidy3d_ref_loader.py) on a set of simulations: tests/golden/
autogrid_golden.json, tests/test_autogrid.py.
"""
from __future__ import annotations

from math import isclose
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import schema as td
from .constants import C_0, fp_eps
from .exceptions import SetupError

MIN_STEP_SCALE = 0.9999          # ref mesher.py:29
ROOTS_TOL = 1e-10                # ref mesher.py:24


# ----------------------------------------------------------------------------------------------
# stage 1: intervals and their maximum steps
# ----------------------------------------------------------------------------------------------

class _Item:
    """One entry of the meshing list: a structure (medium) or a mesh-override box (dl)."""

    def __init__(self, bounds, medium=None, dl=None, enforce=False):
        self.bmin, self.bmax = (tuple(float(v) for v in bounds[0]), tuple(float(v) for v in bounds[1]))
        self.medium, self.dl, self.enforce = medium, dl, bool(enforce)

    @property
    def is_override(self) -> bool:
        return self.dl is not None


def _index_for_step(medium, freq: float, axis: int) -> float:
    """Refractive index that sets the step inside a medium (ref mesher.py:495-517): PEC (and a PEC
    component along the axis) count as vacuum, otherwise the largest |n| or |k| over the diagonal."""
    if getattr(medium, "is_pec", False) or isinstance(medium, td.Medium2D):      # (2-D sheets never set the step, ref mesher.py:499-506)
        return 1.0
    if hasattr(medium, "eps_diagonal"):       # spatially varying media: the value of largest modulus over the data (ref medium.py:1324-1336)
        eps = np.asarray(medium.eps_diagonal(freq), complex)
        nk = np.sqrt(eps)
        return float(max(np.max(np.abs(nk.real)), np.max(np.abs(nk.imag))))
    if isinstance(medium, td.AnisotropicMedium):
        if getattr(medium.component(axis), "is_pec", False):
            return 1.0
        # (a spatially varying component — CustomAnisotropicMedium — sets the step with its value of largest modulus, as a
        #  CustomMedium does, ref medium.py:1324-1336; eps_comp would return the component's spatial mean)
        def diag(c):
            m_c = medium.component(c)
            if hasattr(m_c, "eps_diagonal"):
                e3 = np.asarray(m_c.eps_diagonal(freq), complex).ravel()
                return complex(e3[int(np.argmax(np.abs(e3)))])
            return complex(np.asarray(medium.eps_comp(c, freq)).ravel()[0])
        eps = np.array([diag(c) for c in range(3)])
    else:
        if isinstance(medium, td.Unsupported):
            medium.fail()
        eps = np.array([complex(np.asarray(medium.eps_model(freq)).ravel()[0])] * 3)
    nk = np.sqrt(eps)
    return float(max(np.max(np.abs(nk.real)), np.max(np.abs(nk.imag))))


def _steps(items: Sequence[_Item], wavelength: float, min_steps_per_wvl: float, dl_min: float, axis: int) -> np.ndarray:
    out = []
    for it in items:
        if it.is_override:
            out.append(max(dl_min, it.dl[axis]))
        else:
            out.append(max(dl_min, wavelength / _index_for_step(it.medium, C_0 / wavelength, axis) / min_steps_per_wvl))
    return np.array(out)


def _drop_short(coords: Sequence[float], steps: Sequence[float]) -> Tuple[List[float], List[float]]:
    """ref mesher.py:619-635: intervals shorter than the smallest maximum step are merged away."""
    smallest = np.amin(steps)
    kept, kept_steps = [coords[0]], []
    for i, c in enumerate(coords[1:]):
        if c - kept[-1] >= smallest:
            kept.append(c)
            kept_steps.append(steps[i])
    return kept, kept_steps


def parse_structures(axis: int, items: Sequence[_Item], wavelength: float, min_steps_per_wvl: float,
                     dl_min: float) -> Tuple[np.ndarray, np.ndarray]:
    """Interval boundaries along ``axis`` and the maximum step of every interval."""
    domain = np.array([items[0].bmin[axis], items[0].bmax[axis]])
    items = [it for it in items if not (it.is_override and it.dl[axis] is None)]        # ref :443-471
    enforced = [it.is_override and it.enforce for it in items]
    n_free = len(items)
    if any(enforced):                                                                   # ref :408-441
        free = [it for it, e in zip(items, enforced) if not e]
        items = free + [it for it, e in zip(items, enforced) if e]
        n_free = len(free)
    steps = _steps(items, wavelength, min_steps_per_wvl, dl_min, axis)
    min_step = MIN_STEP_SCALE * np.amin(steps)
    if len(items) == 1:
        c, s = _drop_short(list(domain), list(steps))
        return np.array(c), np.array(s)

    # boxes with the meshing axis last: rows (min, max), columns (plane axis 0, plane axis 1, axis)
    plane = [a for a in range(3) if a != axis]
    boxes: List[Optional[np.ndarray]] = [
        np.array([[it.bmin[plane[0]], it.bmin[plane[1]], it.bmin[axis]],
                  [it.bmax[plane[0]], it.bmax[plane[1]], it.bmax[axis]]]) for it in items]
    flat = [b.copy() for b in boxes]            # the 2-D overlap query works on the original boxes

    def overlaps_2d(i: int) -> List[int]:
        a = flat[i]
        return [j for j, b in enumerate(flat)
                if not (b[1, 0] < a[0, 0] or b[0, 0] > a[1, 0] or b[1, 1] < a[0, 1] or b[0, 1] > a[1, 1])]

    coords: List[float] = list(domain)
    present: List[List[int]] = [[]]             # structures physically present in each interval

    def near(x: float, k: int, tol: float) -> bool:
        return 0 <= k < len(coords) and isclose(x, coords[k], abs_tol=tol)

    for si in range(len(items) - 1, -1, -1):    # topmost first: later structures override earlier ones
        box = boxes[si]
        if box is None:
            continue
        hits = overlaps_2d(si)
        # structures below that this one swallows entirely (3-D) take no part any more
        for j in hits:
            if j < si and boxes[j] is not None:
                b = boxes[j]
                if all(b[0, d] + fp_eps >= box[0, d] and b[1, d] <= box[1, d] + fp_eps for d in range(3)):
                    boxes[j] = None
        # structures above whose footprint contains this one's footprint
        covers = [boxes[j] for j in hits if j > si and boxes[j] is not None]
        covers = [b for b in covers
                  if box[0, 0] + fp_eps >= b[0, 0] and box[1, 0] <= b[1, 0] + fp_eps and
                  box[0, 1] + fp_eps >= b[0, 1] and box[1, 1] <= b[1, 1] + fp_eps]

        def hidden(z: float) -> bool:
            return any(b[0, 2] <= z <= b[1, 2] for b in covers)

        # where do the two faces of the box go?  (ref mesher.py:310-406)
        tol = MIN_STEP_SCALE * min_step
        lo = box[0, 2]
        i_lo = int(np.nonzero(lo <= np.array(coords))[0][0])
        if near(lo, i_lo - 1, tol):
            i_lo -= 1
        elif not near(lo, i_lo, tol) and not hidden(lo) and si > 0:
            coords.insert(i_lo, lo)
            present.insert(i_lo, list(present[max(0, i_lo - 1)]))
        hi = box[1, 2]
        i_hi = int(np.nonzero(hi >= np.array(coords))[0][-1])
        close_l, close_r = near(hi, i_hi, tol), near(hi, i_hi + 1, tol)
        if close_r:
            i_hi += 1
        elif not close_l and not hidden(hi) and si > 0:
            i_hi += 1
            coords.insert(i_hi, hi)
            present.insert(i_hi, list(present[min(i_hi - 1, len(present) - 1)]))
        for k in range(i_lo, i_hi):
            if not hidden(0.5 * (coords[k] + coords[k + 1])):
                present[k].append(si)
        if i_lo >= i_hi and (box[1, 2] - box[0, 2]) > 0:
            boxes[si] = None                    # thinner than the mesh: invisible to what lies below

    arr = np.array(coords)
    inside = np.nonzero((arr >= domain[0]) * (arr <= domain[1]))[0]
    n_int = len(present)
    coords = [coords[int(i)] for i in inside]
    present = [present[int(i)] for i in inside if i < n_int]
    max_steps = []
    for k in range(len(coords) - 1):
        top = max(present[k])
        # an enforced override wins outright; otherwise the densest structure present decides
        max_steps.append(steps[top] if top >= n_free else np.amin(steps[present[k]]))
    coords, max_steps = _drop_short(coords, max_steps)
    return np.array(coords), np.array(max_steps)


def insert_snapping_points(axis: int, coords: np.ndarray, max_dl: np.ndarray, points) -> Tuple[np.ndarray, np.ndarray]:
    """ref mesher.py:76-131."""
    if coords.size == 1 or len(points) < 1:
        return coords, max_dl
    min_step = np.amin(max_dl) * 0.5
    for pt in points:
        x = pt[axis]
        if x >= coords[-1] or x <= coords[0]:
            continue
        k = int(np.searchsorted(coords, x, side="left"))
        if abs(x - coords[k]) < min_step or abs(x - coords[k - 1]) < min_step:
            continue
        coords = np.insert(coords, k, x)
        max_dl = np.insert(max_dl, k - 1, max_dl[k - 1])
    return coords, max_dl


# ----------------------------------------------------------------------------------------------
# stage 2: steps inside the intervals
# ----------------------------------------------------------------------------------------------

def _geo_len(first: float, scale: float, n: int) -> float:
    return first * (1 - scale ** n) / (1 - scale)


def _geo(first: float, scale: float, n: int) -> np.ndarray:
    return np.array([first * scale ** i for i in range(n)])


def _absorb(rise: np.ndarray, rest: float, seed: float) -> Optional[np.ndarray]:
    """Put a left-over length into an ascending sequence as one more step, if it is at least ``seed``."""
    if rest >= seed:
        return np.insert(rise, np.searchsorted(rise, rest), rest)
    return None


def _grow(small: float, scale: float, length: float) -> np.ndarray:
    """Steps growing from ``small`` by ``scale`` until the interval is used up (ref mesher.py:1136-1220)."""
    n = int(np.floor(np.log(1 - length / small * (1 - scale)) / np.log(scale)))
    seq = _geo(small, scale, n)
    rest = length - _geo_len(small, scale, n)
    if isclose(rest, 0):
        return seq
    done = _absorb(seq, rest, small)
    if done is not None:
        return done
    if n >= 2 and rest >= small - (1 - 1.0 / scale ** 2) * seq[-1]:
        seq = np.append(small, seq)             # repeat the first step, stretch the last
        seq[-1] += rest - small
        return seq
    even_rest = length - n * small
    if isclose(even_rest, small):
        return np.array([small] * (n + 1))
    if even_rest > small:
        def f(s):
            if isclose(s, 1.0):
                return length - small * (1 + n)
            return length - small * (1 - s ** n) / (1 - s) - small
        root = _brentq(f, 1, scale)
        if root is not None and abs(f(root)) <= ROOTS_TOL:
            return np.append(small, _geo(small, root, n))
    seq = np.append(small, seq)
    return seq * (length / np.sum(seq))


def _brentq(f, a, b):
    from scipy.optimize import brentq
    try:
        return brentq(f, a, b, xtol=ROOTS_TOL * 1e-3, rtol=8.9e-16, maxiter=500)
    except Exception:       # noqa: BLE001 - no bracket: the caller falls back to rescaling
        return None


def _grow_plateau(small: float, large: float, scale: float, length: float) -> np.ndarray:
    """ref mesher.py:1081-1134."""
    n = 1 + int(np.floor(np.log(large / small) / np.log(scale)))
    rise = _geo(small, scale, n)
    used = _geo_len(small, scale, n)
    n_flat = int(np.floor((length - used) / large))
    flat = np.array([large] * n_flat)
    rest = length - used - large * n_flat
    if isclose(rest, 0):
        return np.append(rise, flat)
    done = _absorb(rise, rest, small)
    if done is not None:
        return np.append(done, flat)
    seq = np.append(np.append(small, rise), flat)
    return seq * (length / np.sum(seq))


def _grow_plateau_decay(left: float, right: float, top: float, scale: float, length: float) -> np.ndarray:
    """ref mesher.py:923-991."""
    nl = 1 + int(np.floor(np.log(top / left) / np.log(scale)))
    nr = 1 + int(np.floor(np.log(top / right) / np.log(scale)))
    up, down = _geo(left, scale, nl), _geo(right, scale, nr)
    used = _geo_len(left, scale, nl) + _geo_len(right, scale, nr)
    n_flat = int(np.floor((length - used) / top))
    flat = np.array([top] * n_flat)
    rest = length - _geo_len(left, scale, nl) - _geo_len(right, scale, nr) - n_flat * top
    if isclose(rest, 0):
        return np.concatenate((up, flat, np.flip(down)))
    done = _absorb(up, rest, left)
    if done is not None:
        return np.concatenate((done, flat, np.flip(down)))
    done = _absorb(down, rest, right)
    if done is not None:
        return np.concatenate((up, flat, np.flip(done)))
    if left <= right:
        up = np.append(left, up)
    else:
        down = np.append(right, down)
    seq = np.concatenate((up, flat, np.flip(down)))
    return seq * (length / np.sum(seq))


def _grow_decay(left: float, right: float, scale: float, length: float) -> np.ndarray:
    """ref mesher.py:993-1079."""
    if length < left + right:
        even = min(left, right)
        n = int(np.floor(length / even))
        if n * even < length:
            n += 1
        return np.array([length / n] * n)
    tl = ((left + right) - length * (1 - scale)) / 2 / left
    tr = ((left + right) - length * (1 - scale)) / 2 / right
    nl = max(int(np.floor(np.log(tl) / np.log(scale))), 0)
    nr = max(int(np.floor(np.log(tr) / np.log(scale))), 0)
    up, down = _geo(left, scale, nl), _geo(right, scale, nr)
    rest = length - _geo_len(left, scale, nl) - _geo_len(right, scale, nr)
    if isclose(rest, 0):
        return np.append(up, np.flip(down))
    while len(up) > 0 and rest >= up[-1]:
        up = np.append(up, up[-1])
        rest -= up[-1]
    while len(down) > 0 and rest >= down[-1]:
        down = np.append(down, down[-1])
        rest -= down[-1]
    done = _absorb(up, rest, left)
    if done is not None:
        return np.append(done, np.flip(down))
    done = _absorb(down, rest, right)
    if done is not None:
        return np.append(up, np.flip(done))
    if left <= right:
        up = np.append(left, up)
    else:
        down = np.append(right, down)
    seq = np.append(up, np.flip(down))
    return seq * (length / np.sum(seq))


def grid_in_interval(left_nb: float, right_nb: float, top: float, scale: float, length: float) -> np.ndarray:
    """Steps of one interval given the neighbours' edge steps (ref mesher.py:816-921, 1222-1264)."""
    left, right = min(top, left_nb), min(top, right_nb)
    if length <= min(left, right, top):
        return np.array([length])
    if isclose(scale, 1) or (top <= left and top <= right):
        n = int(np.ceil(length / min(left, right)))
        return np.array([length / n] * n)
    small, large = min(left, right), max(left, right)
    if top <= left or top <= right:             # grows from one side only
        n = 1 + int(np.floor(np.log(large / small) / np.log(scale)))
        if length - _geo_len(small, scale, n) < large:
            seq = _grow(small, scale, length)
        else:
            seq = _grow_plateau(small, large, scale, length)
        return seq if left <= right else np.flip(seq)
    nl = 1 + int(np.floor(np.log(top / left) / np.log(scale)))
    nr = 1 + int(np.floor(np.log(top / right) / np.log(scale)))
    if length - _geo_len(left, scale, nl) - _geo_len(right, scale, nr) >= top:
        return _grow_plateau_decay(left, right, top, scale, length)
    return _grow_decay(left, right, scale, length)


def _edge_steps(max_dl: np.ndarray, lens: np.ndarray, scale: float, periodic: bool) -> Tuple[np.ndarray, np.ndarray]:
    """Edge steps of every interval before integer step counts are imposed (ref mesher.py:733-814)."""
    right = np.roll(max_dl, -1)
    left = np.roll(max_dl, 1)
    if not periodic:
        right[-1], left[0] = max_dl[-1], max_dl[0]
    right, left = np.minimum(max_dl, right), np.minimum(max_dl, left)
    again = True
    while again:
        again = False
        n = np.maximum(np.log(1 - lens / left * (1 - scale)) / np.log(scale), 1)
        reach = left * scale ** (n - 1)
        upd = reach < right
        right[upd] = reach[upd]
        if not periodic:
            upd[-1] = False
        if np.any(upd):
            again = True
            left[np.roll(upd, 1)] = reach[upd]
        n = np.maximum(np.log(1 - lens / right * (1 - scale)) / np.log(scale), 1)
        reach = right * scale ** (n - 1)
        upd = reach < left
        left[upd] = reach[upd]
        if not periodic:
            upd[0] = False
        if np.any(upd):
            again = True
            right[np.roll(upd, -1)] = reach[upd]
    if not periodic:
        left[0], right[-1] = max_dl[0], max_dl[-1]
    return left, right


def grid_multiple_intervals(max_dl: np.ndarray, lens: np.ndarray, scale: float, periodic: bool) -> List[np.ndarray]:
    """ref mesher.py:637-731."""
    m = len(lens)
    left, right = _edge_steps(np.array(max_dl, float), np.array(lens, float), scale, periodic)
    out = [grid_in_interval(left[k], right[k], max_dl[k], scale, lens[k]) for k in range(m)]
    changed = 1
    while changed > 0:
        changed = 0
        for k in range(m):
            l_dl, r_dl = out[k][0], out[k][-1]
            l_nb, r_nb = out[k - 1][-1], out[(k + 1) % m][0]
            if not periodic:
                if k == 0:
                    l_nb = l_dl
                if k == m - 1:
                    r_nb = r_dl
            local = 0
            if l_dl / l_nb > scale:
                l_dl = l_nb * (scale - fp_eps)
                changed += 1
                local += 1
            if r_dl / r_nb > scale:
                r_dl = r_nb * (scale - fp_eps)
                changed += 1
                local += 1
            if local:
                out[k] = grid_in_interval(l_dl, r_dl, max_dl[k], scale, lens[k])
    return out


# ----------------------------------------------------------------------------------------------
# AutoGrid along one axis of a Simulation
# ----------------------------------------------------------------------------------------------

def wavelength_of(sim) -> float:
    """ref grid_spec.py:626-646 / :699-703."""
    w = getattr(sim.grid_spec, "wavelength", None)
    if w is not None:
        return float(w)
    if len(sim.sources) == 0:
        raise SetupError("Automatic grid generation requires the input of 'wavelength' or sources.")
    def f0(src):
        # source (or source-time) types the solver does not run still carry freq0 in their JSON form
        st = src.raw.get("source_time", {}) if isinstance(src, td.Unsupported) else src.source_time
        if isinstance(st, td.Unsupported):
            st = st.raw
        return float(st["freq0"] if isinstance(st, dict) else st.freq0)
    freqs = np.array([f0(s) for s in sim.sources])
    if not np.all(np.isclose(freqs, freqs[0])):
        raise SetupError("Sources of different central frequencies are supplied. "
                         "Please supply a 'wavelength' value for 'grid_spec'.")
    return float(C_0 / freqs[0])


def make_coords_initial(sim, axis: int, g1d, wavelength: float, periodic: bool) -> np.ndarray:
    """Cell boundaries of the (symmetry-reduced) domain along ``axis`` before the symmetry mirror and
    the PML cells are added (ref grid_spec.py:430-520)."""
    cen, size = list(sim.center), list(sim.size)
    for d in range(3):
        if sim.symmetry[d] != 0:
            cen[d] += size[d] / 4
            size[d] /= 2
    dom = td.Box(center=tuple(cen), size=tuple(size))
    (d0, d1) = dom.bounds
    items = [_Item(dom.bounds, medium=sim.medium)]
    for st in sim.structures:
        if isinstance(st.geometry, td.Unsupported):
            st.geometry.fail()
        b0, b1 = st.geometry.bounds
        if all(b0[d] <= d1[d] and b1[d] >= d0[d] for d in range(3)):        # ref geometry/base.py:275-311
            items.append(_Item((b0, b1), medium=st.medium))
    for ov in (getattr(sim.grid_spec, "override_structures", None) or ()):
        if isinstance(ov, td.Unsupported) or isinstance(getattr(ov, "geometry", None), td.Unsupported):
            raise SetupError("unsupported mesh override structure")
        b0, b1 = ov.geometry.bounds
        if all(b0[d] <= d1[d] and b1[d] >= d0[d] for d in range(3)):
            if hasattr(ov, "dl"):
                items.append(_Item((b0, b1), dl=tuple(ov.dl), enforce=getattr(ov, "enforce", False)))
            else:       # a plain Structure among the overrides meshes like a structure of the simulation
                items.append(_Item((b0, b1), medium=ov.medium))           # (ref grid_spec.py:471-480 StructureType)
    is_periodic = bool(periodic) and sim.symmetry[axis] == 0
    coords, max_dl = parse_structures(axis, items, wavelength, float(g1d.min_steps_per_wvl), float(g1d.dl_min or 0.0))
    coords, max_dl = insert_snapping_points(axis, coords, max_dl, getattr(sim.grid_spec, "snapping_points", None) or ())
    if coords.size == 1:                        # a 2-D-like simulation: one pixel
        dl = wavelength / g1d.min_steps_per_wvl
        return np.array([cen[axis] - dl / 2, cen[axis] + dl / 2])
    coords = np.array(coords).flatten()
    steps = grid_multiple_intervals(np.array(max_dl).flatten(), coords[1:] - coords[:-1], float(g1d.max_scale),
                                    is_periodic)
    bounds = np.append(0.0, np.cumsum(np.concatenate(steps))) + coords[0]
    ends = [d0[axis], d1[axis]]
    if not np.all(np.isclose(bounds[[0, -1]], ends)):
        raise SetupError(f"AutoGrid coordinates along axis {axis} do not match the simulation domain")
    bounds[[0, -1]] = ends
    return np.array(bounds)
