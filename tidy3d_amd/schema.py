"""tidy3d-free mirror of the part of the ``tidy3d.components`` schema that the
FDTD hot path consumes.

Why this exists: the solver must accept a ``tidy3d.Simulation``; the real
package is not importable on the GPU box (h5py/xarray/shapely/autograd are
absent), so the boundary is defined on tidy3d's *JSON/dict form* — the output
of ``Simulation.dict()`` / ``.json()`` (fixture: reference
tests/sims/simulation_sample.json).  ``parse(obj)`` turns such a dict into the
light classes below; ``tidy3d_amd.adapter`` feeds real tidy3d objects through
the same path.  Class names, field names, defaults and semantics follow the
reference (cited per class) so that scripts and tests read like tidy3d's own::

    import tidy3d_amd.schema as td
    sim = td.Simulation(size=(4, 3, 2), grid_spec=td.GridSpec.uniform(dl=0.05), ...)

Only arithmetic needed by the solver is implemented (grid, dt, pole-residue
conversion, source waveforms, monitor index spans).  Plotting, validation
beyond what protects the solver, file IO and the cloud client are out of
scope.
"""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .constants import (C_0, EPSILON_0, DFT_CUTOFF, END_TIME_FACTOR_GAUSSIAN, fp_eps, inf,
                        LARGE_NUMBER)
from .exceptions import SetupError, Tidy3dNotImplementedError, ValidationError

_REGISTRY: Dict[str, type] = {}


def _register(cls):
    _REGISTRY[cls.__name__] = cls
    return cls


def _tup(x):
    if x is None:
        return None
    return tuple(_tup(v) if isinstance(v, (list, tuple)) else v for v in x)


def _to_float(v):
    """tidy3d serialises +-inf as the strings "Infinity" / "-Infinity"."""
    if isinstance(v, str):
        return float(v.replace("Infinity", "inf"))
    return v


def _clean(v):
    if isinstance(v, dict):
        return parse(v)
    if isinstance(v, (list, tuple)):
        return tuple(_clean(x) for x in v)
    if isinstance(v, str) and v in ("Infinity", "-Infinity", "NaN"):
        return _to_float(v)
    return v


@dataclass
class Unsupported:
    """Placeholder for a schema type the solver does not implement.  Parsing never
    fails on it; the solver raises ``Tidy3dNotImplementedError`` naming the feature
    only if the object is actually used (SURVEY.md section 8(b) "Errors")."""

    type: str = ""
    raw: dict = field(default_factory=dict)

    def fail(self):
        raise Tidy3dNotImplementedError(
            f"'{self.type}' is not supported by the MI355X local FDTD solver.")


def parse(obj: Any):
    """dict (tidy3d ``.dict()`` / JSON form, discriminated by ``"type"``) -> mirror object."""
    if isinstance(obj, (list, tuple)):
        return tuple(parse(o) for o in obj)
    if not isinstance(obj, dict):
        return _to_float(obj) if isinstance(obj, str) and "Infinity" in obj else obj
    tname = obj.get("type")
    cls = _REGISTRY.get(tname)
    if cls is None:
        return Unsupported(type=str(tname), raw=obj)
    # fields that change the physics and that this solver does not model: a medium that carries one becomes a placeholder
    # that raises when the medium is used — never silently the linear / static medium
    for k in ("nonlinear_spec", "modulation_spec"):
        if obj.get(k) is not None:
            return Unsupported(type=f"{tname} with '{k}'", raw=obj)
    names = {f.name for f in dataclasses.fields(cls)}
    kwargs = {}
    for k, v in obj.items():
        if k in ("type", "attrs") or k not in names:
            continue
        kwargs[k] = _clean(v)
    return cls(**kwargs)


class _Model:
    """Common helpers (subset of ref components/base.py Tidy3dBaseModel)."""

    @property
    def type(self) -> str:
        return self.__class__.__name__

    def copy(self, **update):
        return dataclasses.replace(self, **update)

    updated_copy = copy

    def dict(self) -> dict:
        def conv(v):
            if isinstance(v, _Model):
                return v.dict()
            if isinstance(v, Unsupported):
                return v.raw
            if isinstance(v, (list, tuple)):
                return [conv(x) for x in v]
            if isinstance(v, complex):
                return {"real": v.real, "imag": v.imag}
            if isinstance(v, np.ndarray):
                return v.tolist()
            return v
        out = {"type": self.type}
        for f in dataclasses.fields(self):
            out[f.name] = conv(getattr(self, f.name))
        return out


def _complex(v) -> complex:
    """tidy3d ComplexNumber JSON form {"real":..,"imag":..} or (re, im) or python complex."""
    if isinstance(v, dict):
        return complex(v["real"], v["imag"])
    if isinstance(v, Unsupported):
        return complex(v.raw.get("real", 0.0), v.raw.get("imag", 0.0))
    if isinstance(v, (tuple, list)) and len(v) == 2:
        return complex(v[0], v[1])
    return complex(v)


# --------------------------------------------------------------------------------------
# geometry  (ref components/geometry/base.py, primitives.py)
# --------------------------------------------------------------------------------------

@_register
@dataclass
class Box(_Model):
    """Rectangular prism (ref geometry/base.py:1799; ``inside`` :2043-2067 inclusive <=)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)

    def __post_init__(self):
        self.center = tuple(float(_to_float(c)) for c in self.center)
        self.size = tuple(float(_to_float(s)) for s in self.size)

    @property
    def bounds(self):
        c, s = np.array(self.center), np.array(self.size)
        with np.errstate(invalid="ignore"):
            lo = np.where(np.isinf(s), -inf, c - s / 2)
            hi = np.where(np.isinf(s), inf, c + s / 2)
        return tuple(lo), tuple(hi)

    def inside(self, x, y, z):
        x0, y0, z0 = self.center
        lx, ly, lz = self.size
        return ((np.abs(x - x0) <= lx / 2) & (np.abs(y - y0) <= ly / 2)
                & (np.abs(z - z0) <= lz / 2))

    @classmethod
    def from_bounds(cls, rmin, rmax, **kw):
        # (ref geometry/base.py:411-424 _get_center: a dimension unbounded on both sides is centred at 0, on one side only an error)
        center = []
        for lo, hi in zip(rmin, rmax):
            lo, hi = float(lo), float(hi)
            if np.isneginf(lo) and np.isposinf(hi):
                center.append(0.0)
            elif np.isneginf(lo) or np.isposinf(hi):
                raise SetupError(f"Bounds of ({lo}, {hi}) supplied along one dimension: a single inf value in the bounds of a Box is not supported")
            else:
                center.append((lo + hi) / 2.0)
        return cls(center=tuple(center), size=tuple(float(hi) - float(lo) for lo, hi in zip(rmin, rmax)), **kw)

    @property
    def zero_dims(self) -> List[int]:
        return [d for d, s in enumerate(self.size) if s == 0]

    def surfaces(self):
        """Six faces ordered x-,x+,y-,y+,z-,z+ (ref geometry/base.py:1836-1900)."""
        (x0, y0, z0), (lx, ly, lz) = self.center, self.size
        c = [(x0 - lx / 2, y0, z0), (x0 + lx / 2, y0, z0), (x0, y0 - ly / 2, z0),
             (x0, y0 + ly / 2, z0), (x0, y0, z0 - lz / 2), (x0, y0, z0 + lz / 2)]
        s = [(0, ly, lz)] * 2 + [(lx, 0, lz)] * 2 + [(lx, ly, 0)] * 2
        names = ["x-", "x+", "y-", "y+", "z-", "z+"]
        return [(n, Box(center=ci, size=si)) for n, ci, si in zip(names, c, s)]


@_register
@dataclass
class Sphere(_Model):
    """ref geometry/primitives.py:36; ``inside`` :44-68 (dist^2 <= r^2)."""

    radius: float = 1.0
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)

    @property
    def bounds(self):
        c = np.array(self.center, float)
        return tuple(c - self.radius), tuple(c + self.radius)

    def inside(self, x, y, z):
        x0, y0, z0 = self.center
        return ((x - x0) ** 2 + (y - y0) ** 2 + (z - z0) ** 2) <= self.radius ** 2


@_register
@dataclass
class Cylinder(_Model):
    """ref geometry/primitives.py:179; slanted side walls as ref :720-737 ``_radius_z`` (the radius
    shrinks by tan(sidewall_angle) per unit length along +axis from its value on the reference plane)."""

    radius: float = 1.0
    length: float = 1.0
    axis: int = 2
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    sidewall_angle: float = 0.0
    reference_plane: str = "middle"

    def __post_init__(self):
        self.length = float(_to_float(self.length))
        if abs(self.sidewall_angle) >= np.pi / 2:
            raise ValidationError("Cylinder.sidewall_angle must lie in (-pi/2, pi/2) (ref base.py:1613).")
        if self.sidewall_angle != 0.0 and np.isinf(self.length) and self.reference_plane != "middle":
            raise SetupError("A slanted cylinder of infinite length needs reference_plane 'middle' "
                             "(ref primitives.py:212-226).")

    @property
    def _finite_length(self) -> float:
        return min(self.length, 1e10)                     # ref base.py:1642 LARGE_NUMBER

    def _radius_z(self, z):
        """ref primitives.py:720-737."""
        if np.isclose(self.sidewall_angle, 0):
            return self.radius + 0.0 * np.asarray(z, float)
        tanq = np.tan(self.sidewall_angle)
        r_mid = self.radius
        if self.reference_plane == "top":
            r_mid += self._finite_length / 2 * tanq
        elif self.reference_plane == "bottom":
            r_mid -= self._finite_length / 2 * tanq
        return r_mid - (np.asarray(z, float) - self.center[self.axis]) * tanq

    @property
    def bounds(self):
        """ref primitives.py:636-648 (radius_max = the larger of the two end radii)."""
        c = np.array(self.center, float)
        z0 = self.center[self.axis]
        r_max = float(max(self._radius_z(z0 - self._finite_length / 2), self._radius_z(z0 + self._finite_length / 2)))
        h = np.full(3, r_max)
        h[self.axis] = self.length / 2
        return tuple(c - h), tuple(c + h)

    def inside(self, x, y, z):
        """ref primitives.py:600-633."""
        p = [x, y, z]
        c = list(self.center)
        za = p.pop(self.axis)
        z0 = c.pop(self.axis)
        r = self._radius_z(za)
        r2 = (p[0] - c[0]) ** 2 + (p[1] - c[1]) ** 2
        return (r > 0) & (r2 <= r ** 2) & (np.abs(za - z0) <= self._finite_length / 2)


@_register
@dataclass
class PolySlab(_Model):
    """Polygon extruded along ``axis`` (ref geometry/polyslab.py:37), with ``dilation`` and slanted
    side walls.  ``inside`` = slab bounds AND point-in-polygon of the cross-section at that height;
    the reference delegates the latter to ``matplotlib.path.Path.contains_points`` (polyslab.py:511-552,
    third party, absent here; pinned against the real function in tests/test_polyslab.py): restated as
    the even-odd crossing-number test with the half-open edge rule ``(y_i <= y) != (y_j <= y)`` —
    points exactly on an edge are implementation-defined in both.  The cross-section at height z is
    the middle polygon with every edge moved outward by -(z - z_mid) tan(sidewall_angle), mitred at the
    vertices (ref ``_shift_vertices`` :1214-1274).  The reference repairs polygons that self-intersect
    after dilation with shapely (``_heal_polygon`` :1313); that repair is not available here: an offset
    that makes an edge vanish is refused instead."""

    vertices: Any = ()
    slab_bounds: Tuple[float, float] = (0.0, 0.0)
    axis: int = 2
    sidewall_angle: float = 0.0
    dilation: float = 0.0
    reference_plane: str = "middle"

    _RTOL = np.finfo(float).eps                               # ref polyslab.py:38

    def __post_init__(self):
        self.vertices = tuple(tuple(float(c) for c in v) for v in np.asarray(self.vertices, float).reshape(-1, 2))
        self.slab_bounds = tuple(float(_to_float(v)) for v in self.slab_bounds)
        if len(self.vertices) < 3:
            raise ValidationError("PolySlab needs at least 3 vertices.")
        if abs(self.sidewall_angle) >= np.pi / 2:
            raise ValidationError("PolySlab.sidewall_angle must lie in (-pi/2, pi/2) (ref base.py:1613).")
        if self.sidewall_angle != 0.0 and self.reference_plane != "middle" and not np.isfinite(self.slab_bounds).all():
            raise SetupError("A slanted PolySlab of infinite length needs reference_plane 'middle' "
                             "(ref polyslab.py:66-84).")
        self._cache = {}
        if self.sidewall_angle != 0.0 or self.dilation != 0.0:
            # most negative offset any cross-section gets, against the offset at which an edge vanishes
            prop = self._proper_vertices(self.vertices)
            worst = min(self.dilation, 0.0)
            ref = self.reference_polygon
            tq, L = abs(np.tan(self.sidewall_angle)), self._finite_length
            reach = {"middle": tq * L / 2, "bottom": max(0.0, np.tan(self.sidewall_angle)) * L,
                     "top": max(0.0, -np.tan(self.sidewall_angle)) * L}[self.reference_plane]
            if -worst >= self._first_edge_event(prop) or reach >= self._first_edge_event(ref):
                raise Tidy3dNotImplementedError(
                    "PolySlab: the dilation / side-wall slant erodes an edge of the polygon completely; the "
                    "polygon healing the reference applies in that case needs shapely and is not supported.")

    # ---- polygon helpers (static, same arithmetic as the reference's) -------------------------
    @staticmethod
    def _area(v: np.ndarray) -> float:
        """Signed area, positive for CCW (ref polyslab.py:1019-1039)."""
        w = np.roll(v, axis=0, shift=-1)
        return float(np.sum(v[:, 0] * w[:, 1] - v[:, 1] * w[:, 0]) * 0.5)

    @classmethod
    def _proper_vertices(cls, vertices) -> np.ndarray:
        """Duplicate neighbours removed, CCW orientation (ref polyslab.py:1066-1113)."""
        v = np.asarray(vertices, float).reshape(-1, 2)
        d = np.linalg.norm(v - np.roll(v, shift=-1, axis=0), axis=1)
        v = v[~np.isclose(d, 0, rtol=cls._RTOL)]
        return v if cls._area(v) > 0 else v[::-1, :]

    @classmethod
    def _shift_vertices(cls, vertices: np.ndarray, dist: float):
        """Move every edge outward by ``dist`` and re-intersect neighbours (mitre joints); returns
        (vertices, parallel shift per vertex) (ref polyslab.py:1214-1274)."""
        if np.isclose(dist, 0):
            return vertices, np.zeros(vertices.shape[0])
        rot90 = lambda u: np.stack((-u[1], u[0]), axis=0)
        cross = lambda u, w: u[0] * w[1] - u[1] * w[0]
        vs = vertices.T.copy()
        nxt, prv = np.roll(vs, axis=-1, shift=-1), np.roll(vs, axis=-1, shift=+1)
        asp = (nxt - vs) / np.linalg.norm(nxt - vs, axis=0)
        asm = (vs - prv) / np.linalg.norm(vs - prv, axis=0)
        det = cross(asm, asp)
        flat = np.isclose(det, 0, rtol=cls._RTOL)
        tan_half = np.where(flat, 0.0, cross(asm, rot90(asm - asp)) / (det + flat))
        par = dist * tan_half
        return (vs + (-dist) * rot90(asm) + par * asm).T, par

    @classmethod
    def _first_edge_event(cls, vertices: np.ndarray) -> float:
        """Inward offset at which the FIRST edge shrinks to nothing (edge lengths change linearly with
        the offset, ref polyslab.py:1277-1301); beyond it the mitred offset polygon self-intersects."""
        nxt = np.roll(vertices, axis=0, shift=-1)
        length = np.linalg.norm(nxt - vertices, axis=1)
        par = cls._shift_vertices(vertices, 1.0)[1]
        grow = par + np.roll(par, shift=-1)                  # growth of each edge per unit outward offset
        shrink = grow > np.finfo(np.float32).eps
        return float(np.min(length[shrink] / grow[shrink])) if shrink.any() else np.inf

    @classmethod
    def _maximal_erosion(cls, vertices: np.ndarray) -> float:
        """Inward offset that reduces ALL edges to nothing (ref polyslab.py:1303-1310)."""
        nxt = np.roll(vertices, axis=0, shift=-1)
        length = np.linalg.norm(nxt - vertices, axis=1)
        par = cls._shift_vertices(vertices, 1.0)[1]
        red = -(par + np.roll(par, shift=-1))
        nz = np.abs(red) > np.finfo(np.float32).eps
        return float(-np.min(length[nz] / red[nz])) if nz.any() else np.inf

    # ---- derived polygons -----------------------------------------------------------------------
    @property
    def _finite_length(self) -> float:
        z0, z1 = self.slab_bounds
        return min(z1 - z0, 1e10)                           # ref base.py:1642 LARGE_NUMBER

    @property
    def _center_axis(self) -> float:
        """ref polyslab.py:365-373."""
        z0, z1 = self.slab_bounds
        if np.isneginf(z0) and np.isposinf(z1):
            return 0.0
        return (min(z1, 1e10) + max(z0, -1e10)) / 2.0

    @property
    def _tanq(self) -> float:
        return 0.0 if np.isclose(self.sidewall_angle, 0) else float(np.tan(self.sidewall_angle))

    @property
    def reference_polygon(self) -> np.ndarray:
        """ref polyslab.py:381-394 (without the shapely repair)."""
        if "ref" not in self._cache:
            v = self._proper_vertices(self.vertices)
            if not np.isclose(self.dilation, 0):
                v = self._shift_vertices(v, self.dilation)[0]
            self._cache["ref"] = v
        return self._cache["ref"]

    @property
    def middle_polygon(self) -> np.ndarray:
        """ref polyslab.py:396-412."""
        if "mid" not in self._cache:
            dist = -(self._finite_length / 2) * self._tanq
            v = self.reference_polygon
            if self.reference_plane == "bottom":
                v = self._shift_vertices(v, dist)[0]
            elif self.reference_plane == "top":
                v = self._shift_vertices(v, -dist)[0]
            self._cache["mid"] = v
        return self._cache["mid"]

    @property
    def _planar_axes(self):
        return [a for a in range(3) if a != self.axis]       # (x,y,z) minus axis, order kept (ref pop_axis)

    @property
    def bounds(self):
        """ref polyslab.py:970-1010: the largest offset any cross-section can have, applied to the
        given vertices (an over-estimate by construction)."""
        max_offset = self.dilation
        if self._tanq != 0.0:
            L = self._finite_length
            if self.reference_plane == "bottom":
                max_offset += max(0.0, -self._tanq * L)
            elif self.reference_plane == "top":
                max_offset += max(0.0, self._tanq * L)
            else:
                max_offset += max(0.0, abs(self._tanq) * L / 2)
        v = np.array(self.vertices)
        if max_offset > 0:
            v = self._shift_vertices(self._proper_vertices(self.vertices), max_offset)[0]
        lo, hi = [0.0] * 3, [0.0] * 3
        u, w = self._planar_axes
        lo[u], hi[u] = v[:, 0].min(), v[:, 0].max()
        lo[w], hi[w] = v[:, 1].min(), v[:, 1].max()
        lo[self.axis], hi[self.axis] = self.slab_bounds
        return tuple(lo), tuple(hi)

    @staticmethod
    def _in_polygon(v: np.ndarray, xs: np.ndarray, ys: np.ndarray) -> np.ndarray:
        odd = np.zeros(xs.shape, bool)
        xj, yj = v[-1]
        for xi, yi in v:
            cross = (yi <= ys) != (yj <= ys)
            with np.errstate(divide="ignore", invalid="ignore"):
                xint = xi + (ys - yi) * (xj - xi) / (yj - yi)
            odd ^= cross & (xs < xint)
            xj, yj = xi, yi
        return odd

    def inside(self, x, y, z):
        """ref polyslab.py:464-559."""
        p = [x, y, z]
        za = p.pop(self.axis)
        px, py, za = np.broadcast_arrays(np.asarray(p[0], float), np.asarray(p[1], float), np.asarray(za, float))
        zc = self._center_axis
        ok = np.abs(za - zc) <= self._finite_length / 2
        if not np.any(ok):
            return ok
        xs, ys = px[ok], py[ok]
        if self._tanq == 0.0:
            odd = self._in_polygon(self.reference_polygon if self.dilation else np.array(self.vertices), xs, ys)
        else:
            zs = za[ok]
            odd = np.zeros(xs.shape, bool)
            mid = self.middle_polygon
            uz, inv = np.unique(zs, return_inverse=True)      # one offset polygon per height
            order = np.argsort(inv.ravel(), kind="stable")
            cuts = np.searchsorted(inv.ravel()[order], np.arange(len(uz) + 1))
            for q, zv in enumerate(uz):
                sel = order[cuts[q]:cuts[q + 1]]
                vz = self._shift_vertices(mid, -(zv - zc) * self._tanq)[0]
                odd[sel] = self._in_polygon(vz, xs[sel], ys[sel])
        out = np.zeros(ok.shape, bool)
        out[ok] = odd
        return out


@_register
@dataclass
class TriangleMeshDataset(_Model):
    """ref data/dataset.py TriangleMeshDataset: ``surface_mesh`` = TriangleMeshDataArray
    (face_index, vertex_index, axis)."""

    surface_mesh: Any = None


@_register
@dataclass
class TriangleMesh(_Model):
    """Closed triangulated surface (ref geometry/mesh.py:31).  ``inside`` counts the crossings of a ray
    along +z with the faces (the reference asks trimesh for the same test, mesh.py ``inside``; third party,
    absent here): odd = inside.  Points exactly on a face, edge or vertex are implementation-defined in
    both; the ray origin is nudged by 1e-9 of the mesh size off such degeneracies."""

    mesh_dataset: Any = None

    @property
    def triangles(self) -> np.ndarray:
        v = getattr(self.mesh_dataset, "surface_mesh", None)
        if v is None or isinstance(v, str):
            raise SetupError("TriangleMesh: the JSON form carries no data; load the simulation from its .hdf5 "
                             "file (Simulation.from_file) or use TriangleMesh.from_triangles.")
        return np.asarray(v.values, float).reshape(-1, 3, 3)

    @classmethod
    def from_triangles(cls, triangles):
        """ref geometry/mesh.py:100-125."""
        from .data import DataArray
        tri = np.asarray(triangles, float).reshape(-1, 3, 3)
        arr = DataArray(tri, {"face_index": np.arange(len(tri)), "vertex_index": np.arange(3), "axis": np.arange(3)})
        arr.tag = "TriangleMeshDataArray"
        return cls(mesh_dataset=TriangleMeshDataset(surface_mesh=arr))

    @classmethod
    def from_vertices_faces(cls, vertices, faces):
        """ref geometry/mesh.py:160-180."""
        return cls.from_triangles(np.asarray(vertices, float)[np.asarray(faces, int)])

    @property
    def bounds(self):
        t = self.triangles.reshape(-1, 3)
        return tuple(t.min(axis=0)), tuple(t.max(axis=0))

    def inside(self, x, y, z):
        tri = self.triangles
        x, y, z = (np.asarray(v, float) for v in (x, y, z))
        shape = np.broadcast(x, y, z).shape
        span = float(np.max(tri.max(axis=(0, 1)) - tri.min(axis=(0, 1)))) or 1.0
        ex, ey = 1.2345e-9 * span, 2.7183e-9 * span          # off edges / vertices shared by several faces
        # a lattice (x along the last axis, y along the one before, z along the first — what the rasteriser
        # passes) is handled face by face on the footprint of the face only
        grid = (x.ndim == 3 and x.shape[:2] == (1, 1) and y.ndim == 3 and y.shape[0] == 1 and y.shape[2] == 1
                and z.ndim == 3 and z.shape[1:] == (1, 1))
        if not grid:
            xf, yf, zf = (np.broadcast_to(v, shape).ravel() for v in (x, y, z))
            odd = np.zeros(xf.shape, bool)
            for a, b, c in tri:
                odd ^= self._hits(a, b, c, xf + ex, yf + ey, zf)
            return odd.reshape(shape)
        xs, ys, zs = x.ravel() + ex, y.ravel() + ey, z.ravel()
        odd = np.zeros((len(zs), len(ys), len(xs)), bool)
        for a, b, c in tri:
            lo, hi = np.minimum(np.minimum(a, b), c), np.maximum(np.maximum(a, b), c)
            i0, i1 = np.searchsorted(xs, lo[0]), np.searchsorted(xs, hi[0], side="right")
            j0, j1 = np.searchsorted(ys, lo[1]), np.searchsorted(ys, hi[1], side="right")
            if i1 <= i0 or j1 <= j0:
                continue
            X, Y = np.meshgrid(xs[i0:i1], ys[j0:j1], indexing="xy")        # (ny, nx)
            zt, ok = self._cross_z(a, b, c, X, Y)
            if not ok.any():
                continue
            odd[:, j0:j1, i0:i1] ^= ok[None] & (zs[:, None, None] < zt[None])
        return odd

    @staticmethod
    def _cross_z(a, b, c, X, Y):
        """z at which the vertical line through (X, Y) meets the plane of triangle abc, and whether it
        does so inside the triangle (barycentric test in the xy projection)."""
        d = (b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1])
        if d == 0.0:
            return np.zeros_like(X), np.zeros(X.shape, bool)             # vertical face: never crossed
        l1 = ((X - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (Y - a[1])) / d
        l2 = ((b[0] - a[0]) * (Y - a[1]) - (X - a[0]) * (b[1] - a[1])) / d
        ok = (l1 >= 0) & (l2 >= 0) & (l1 + l2 <= 1)
        return a[2] + l1 * (b[2] - a[2]) + l2 * (c[2] - a[2]), ok

    @classmethod
    def _hits(cls, a, b, c, x, y, z):
        zt, ok = cls._cross_z(a, b, c, x, y)
        return ok & (z < zt)


@_register
@dataclass
class GeometryGroup(_Model):
    """Union of geometries sharing one medium (ref geometry/base.py:2304)."""

    geometries: Tuple[Any, ...] = ()

    @property
    def bounds(self):
        b = [g.bounds for g in self.geometries]
        lo = tuple(min(x[0][a] for x in b) for a in range(3))
        hi = tuple(max(x[1][a] for x in b) for a in range(3))
        return lo, hi

    def inside(self, x, y, z):
        out = None
        for g in self.geometries:
            if isinstance(g, Unsupported):
                g.fail()
            i = g.inside(x, y, z)
            out = i if out is None else (out | i)
        return out


@_register
@dataclass
class Transformed(_Model):
    """Affine image of a geometry (ref geometry/base.py:2495): ``inside`` tests the base geometry at
    the inverse-transformed points (:2603-2632), ``bounds`` are the reference's (over)estimate from the
    8 transformed corners of the base bounds (:2566-2578); nested transforms collapse (:2525-2531)."""

    geometry: Any = None
    transform: Any = None

    def __post_init__(self):
        t = np.eye(4) if self.transform is None else np.asarray(self.transform, dtype=np.float64)
        if t.shape != (4, 4):
            raise ValidationError("Transformed.transform must be a 4 x 4 matrix.")
        while isinstance(self.geometry, Transformed):
            t = np.dot(t, np.asarray(self.geometry.transform, dtype=np.float64))
            self.geometry = self.geometry.geometry
        try:
            np.linalg.inv(t)                                   # ref :2509-2513
        except np.linalg.LinAlgError as e:
            raise ValidationError("Transformed.transform is not invertible.") from e
        self.transform = tuple(tuple(float(v) for v in row) for row in t)
        if not isinstance(self.geometry, Unsupported) and not np.isfinite(self.geometry.bounds).all():
            raise ValidationError("Transformations are only supported on geometries with finite dimensions "
                                  "(ref geometry/base.py:2515-2523).")

    @property
    def inverse(self):
        return np.linalg.inv(np.asarray(self.transform))

    @property
    def bounds(self):
        if isinstance(self.geometry, Unsupported):
            self.geometry.fail()
        (x0, y0, z0), (x1, y1, z1) = self.geometry.bounds
        v = np.array(((x0, x0, x0, x0, x1, x1, x1, x1), (y0, y0, y1, y1, y0, y0, y1, y1),
                      (z0, z1, z0, z1, z0, z1, z0, z1), (1.0,) * 8))
        v = np.dot(np.asarray(self.transform), v)[:3]
        return tuple(v.min(axis=1)), tuple(v.max(axis=1))

    def inside(self, x, y, z):
        if isinstance(self.geometry, Unsupported):
            self.geometry.fail()
        x, y, z = np.broadcast_arrays(np.asarray(x, float), np.asarray(y, float), np.asarray(z, float))
        xyz = np.dot(self.inverse, np.vstack((x.ravel(), y.ravel(), z.ravel(), np.ones(x.size))))
        return np.asarray(self.geometry.inside(xyz[0], xyz[1], xyz[2])).reshape(x.shape)

    @staticmethod
    def translation(x: float, y: float, z: float):
        """ref geometry/base.py:2648."""
        t = np.eye(4)
        t[:3, 3] = (x, y, z)
        return t

    @staticmethod
    def scaling(x: float = 1.0, y: float = 1.0, z: float = 1.0):
        """ref geometry/base.py:2676."""
        if np.isclose((x, y, z), 0.0).any():
            raise ValidationError("Scaling factors cannot be zero in any dimensions.")
        return np.diag((float(x), float(y), float(z), 1.0))

    @staticmethod
    def rotation(angle: float, axis):
        """ref geometry/base.py:2706 (rotation matrix of ref components/transformation.py:113-131)."""
        n = np.zeros(3)
        if isinstance(axis, (int, np.integer)):
            n[int(axis)] = 1.0
        else:
            n = np.asarray(axis, float) / np.linalg.norm(axis)
        c, s_ = np.cos(angle), np.sin(angle)
        K = np.array(((0, -n[2], n[1]), (n[2], 0, -n[0]), (-n[1], n[0], 0)))
        t = np.eye(4)
        t[:3, :3] = c * np.eye(3) + s_ * K + (1 - c) * np.outer(n, n)
        return t


@_register
@dataclass
class ClipOperation(_Model):
    """Set operation between two geometries (ref geometry/base.py:2772): ``inside`` combines the two
    membership tests, ``bounds`` are the reference's (over)estimates (:2894-2921)."""

    operation: str = "union"
    geometry_a: Any = None
    geometry_b: Any = None

    _OPS = {"union": np.logical_or, "intersection": np.logical_and,
            "difference": lambda a, b: a & ~b, "symmetric_difference": np.logical_xor}

    def __post_init__(self):
        if self.operation not in self._OPS:
            raise ValidationError("'operation' must be one of 'union', 'intersection', 'difference', or "
                                  "'symmetric_difference'.")

    def _parts(self):
        for g in (self.geometry_a, self.geometry_b):
            if isinstance(g, Unsupported):
                g.fail()
        return self.geometry_a, self.geometry_b

    @property
    def bounds(self):
        a, b = self._parts()
        if self.operation == "difference":
            return a.bounds
        ba, bb = a.bounds, b.bounds
        if self.operation == "intersection":
            lo = tuple(max(ba[0][i], bb[0][i]) for i in range(3))
            hi = tuple(min(ba[1][i], bb[1][i]) for i in range(3))
            if any(lo[i] > hi[i] for i in range(3)):
                return (0, 0, 0), (0, 0, 0)
            return lo, hi
        return (tuple(min(ba[0][i], bb[0][i]) for i in range(3)), tuple(max(ba[1][i], bb[1][i]) for i in range(3)))

    def inside(self, x, y, z):
        a, b = self._parts()
        return self._OPS[self.operation](np.asarray(a.inside(x, y, z), bool), np.asarray(b.inside(x, y, z), bool))


# --------------------------------------------------------------------------------------
# media  (ref components/medium.py)
# --------------------------------------------------------------------------------------

class _AbstractMedium(_Model):
    is_pec = False

    def pole_residue(self) -> Tuple[float, float, Tuple[Tuple[complex, complex], ...]]:
        """(eps_inf, conductivity, ((a_k, c_k), ...)) with
        eps(w) = eps_inf + i sigma/(w eps0) - sum_k [c_k/(jw + a_k) + c_k*/(jw + a_k*)]
        (ref medium.py:2900-2913; e^{-iwt} convention, ref medium.py:1016-1038)."""
        raise NotImplementedError

    def eps_model(self, frequency):
        eps_inf, sigma, poles = self.pole_residue()
        w = 2 * np.pi * np.asarray(frequency, dtype=float)
        eps = eps_inf + 0j * w
        if sigma:
            eps = eps + 1j * sigma / (w * EPSILON_0)
        for a, c in poles:
            eps = eps - c / (1j * w + a) - np.conj(c) / (1j * w + np.conj(a))
        return eps

    @property
    def n_cfl(self) -> float:
        """ref medium.py:1591 (sqrt(eps)), :2744 (sqrt(eps_inf)), :1482 (PEC -> 1)."""
        eps_inf = self.pole_residue()[0]
        return float(np.real(np.sqrt(complex(eps_inf))))


@_register
@dataclass
class Medium(_AbstractMedium):
    """Dispersionless medium (ref medium.py:1499)."""

    permittivity: float = 1.0
    conductivity: float = 0.0
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None

    def __post_init__(self):
        if self.permittivity < 1.0:
            raise ValidationError("Medium.permittivity must be >= 1.")

    def pole_residue(self):
        return float(self.permittivity), float(self.conductivity), ()

    @classmethod
    def from_nk(cls, n: float, k: float, freq: float, **kw):
        """ref medium.py:63-83 test + AbstractMedium.nk_to_eps_sigma."""
        eps_c = (n + 1j * k) ** 2
        sigma = 2 * np.pi * freq * eps_c.imag * EPSILON_0
        return cls(permittivity=eps_c.real, conductivity=sigma, **kw)


def interp_dataset(arr, points: Dict[str, np.ndarray], method: str = "linear") -> np.ndarray:
    """Values of a labelled array (dims = a subset of x, y, z [, f, t]) at POINTS given per dimension
    (equal-shaped arrays); dimensions of length one are constant, positions outside the coordinate range
    take the edge value (what xarray's interp with extrapolation gives for the reference's use:
    ref medium.py:1077-1120 ``_interp`` / ``fill_value="extrapolate"`` on nearest data)."""
    from scipy.interpolate import RegularGridInterpolator
    vals = np.asarray(arr.values)
    dims = list(arr.dims)
    keep, grids, pts = [], [], []
    shape = np.broadcast(*[np.asarray(points[d]) for d in dims if d in points]).shape if points else ()
    index = []
    for i, d in enumerate(dims):
        c = np.asarray(arr.coords[d], float)
        if len(c) == 1 or d not in points:
            index.append(0)
            continue
        order = np.argsort(c)
        vals = np.take(vals, order, axis=i)
        c = c[order]
        index.append(slice(None))
        grids.append(c)
        pts.append(np.clip(np.broadcast_to(np.asarray(points[d], float), shape), c[0], c[-1]).ravel())
    vals = vals[tuple(index)]
    if not grids:
        return np.full(shape, vals)
    rgi = RegularGridInterpolator(tuple(grids), vals, method="nearest" if method == "nearest" else "linear")
    return rgi(np.stack(pts, axis=1)).reshape(shape)


@_register
@dataclass
class PermittivityDataset(_Model):
    """ref data/dataset.py PermittivityDataset: eps_xx / eps_yy / eps_zz on (x, y, z, f)."""

    eps_xx: Any = None
    eps_yy: Any = None
    eps_zz: Any = None


@_register
@dataclass
class CustomMedium(_AbstractMedium):
    """Spatially varying dispersionless medium (ref medium.py:1648-2050): ``permittivity`` (+ optional
    ``conductivity``) as SpatialDataArrays, or ``eps_dataset`` with one complex permittivity per E component
    at one frequency; evaluated at every Yee location with ``interp_method`` (edge values outside the
    data).  Datasets are not part of the JSON form: load the simulation from .hdf5 (Simulation.from_file)."""

    permittivity: Any = None
    conductivity: Any = None
    eps_dataset: Any = None
    interp_method: str = "nearest"
    subpixel: bool = False
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None
    is_custom = True

    def _check(self):
        """Raised on use, not on parsing (unsupported / data-less objects may sit unused in a file)."""
        from .data import DataArray
        if isinstance(self.permittivity, Unsupported) or isinstance(self.conductivity, Unsupported):
            raise Tidy3dNotImplementedError("CustomMedium on unstructured grids (TriangularGridDataset / "
                                            "TetrahedralGridDataset) is not supported")
        if isinstance(self.permittivity, str) or (self.eps_dataset is not None and isinstance(
                getattr(self.eps_dataset, "eps_xx", None), str)):
            raise SetupError("CustomMedium: the JSON form carries no data (only the placeholder "
                             f"'{self.permittivity}'); load the simulation from its .hdf5 file "
                             "(Simulation.from_file) or pass DataArrays.")
        if self.permittivity is None and self.eps_dataset is None:
            raise SetupError("CustomMedium needs 'permittivity' or 'eps_dataset' (ref medium.py:1740-1760).")
        if self.permittivity is not None:
            if not isinstance(self.permittivity, DataArray):
                raise SetupError("CustomMedium.permittivity must be a SpatialDataArray.")
            if np.any(np.asarray(self.permittivity.values).real < 1.0):
                raise SetupError("CustomMedium.permittivity must be >= 1 everywhere (ref medium.py:1800-1815).")

    def _component_arrays(self, c: int):
        """(eps DataArray, sigma DataArray or None, frequency of an eps_dataset or None) of E component c."""
        self._check()
        if self.permittivity is not None:
            return self.permittivity, self.conductivity, None
        arr = getattr(self.eps_dataset, ("eps_xx", "eps_yy", "eps_zz")[c])
        return arr, None, float(np.asarray(arr.coords["f"]).ravel()[0])

    def eps_sigma_at(self, c: int, x, y, z):
        """(permittivity, conductivity) of E component c at the points (ref medium.py:1016-1038
        eps_complex_to_eps_sigma for the dataset form)."""
        arr, sig, freq = self._component_arrays(c)
        pts = {"x": x, "y": y, "z": z}
        if freq is None:
            eps = np.real(interp_dataset(arr, pts, self.interp_method))
            sigma = np.real(interp_dataset(sig, pts, self.interp_method)) if sig is not None else np.zeros_like(eps)
            return eps, sigma
        ec = interp_dataset(arr, pts, self.interp_method)
        return np.real(ec), np.imag(ec) * 2 * np.pi * freq * EPSILON_0

    def pole_residue(self):
        """Representative medium (mean permittivity) — what keys the structure in the material table; the
        per-cell values are written by the rasteriser."""
        arr = self._component_arrays(0)[0]
        return float(np.mean(np.real(arr.values))), 1e-300, ()

    @property
    def n_cfl(self):
        """ref medium.py:1703-1720: sqrt of the smallest permittivity in the data."""
        lo = min(float(np.min(np.real(self._component_arrays(c)[0].values))) for c in range(3))
        return float(np.sqrt(lo))

    def eps_model(self, frequency):
        eps, sig, _ = self.pole_residue()
        return eps + 0j * np.asarray(frequency, float)

    def eps_diagonal(self, frequency):
        """ref medium.py:1324-1336: per component the data value of largest modulus (what AutoGrid sizes its steps with)."""
        w = 2 * np.pi * float(np.atleast_1d(frequency)[0])
        out = []
        for c in range(3):
            arr, sig, freq = self._component_arrays(c)
            e = np.asarray(arr.values, complex).ravel()
            if freq is None and sig is not None:
                e = e + 1j * np.asarray(sig.values, float).ravel() / (w * EPSILON_0)
            out.append(e[np.argmax(np.abs(e))])
        return tuple(out)

    def dict(self):
        out = {"type": "CustomMedium", "interp_method": self.interp_method, "subpixel": self.subpixel,
               "name": self.name, "frequency_range": self.frequency_range, "permittivity": self.permittivity,
               "conductivity": self.conductivity,
               "eps_dataset": None if self.eps_dataset is None else {
                   "type": "PermittivityDataset", **{k: getattr(self.eps_dataset, k) for k in ("eps_xx", "eps_yy", "eps_zz")}}}
        return out


class _CustomDispersive(_AbstractMedium):
    """Spatially varying dispersive media (ref medium.py:3275 CustomPoleResidue, :3804 CustomSellmeier, :4110 CustomLorentz,
    :4455 CustomDrude, :4720 CustomDebye): the model's coefficients as SpatialDataArrays.  ``pole_params_at`` interpolates the
    coefficients at the Yee nodes (``interp_method``, edge values outside the data) and converts them to (eps_inf, poles) with
    the formulas of the uniform models, point by point; the rasteriser groups the nodes into table entries
    (discretize.rasterize).  Datasets are not part of the JSON form: load the simulation from .hdf5."""

    is_custom_dispersive = True

    def _arrays(self):
        raise NotImplementedError

    def _check(self):
        from .data import DataArray
        for a in self._arrays():
            if isinstance(a, Unsupported):
                raise Tidy3dNotImplementedError(f"{self.type} on unstructured grids (TriangularGridDataset / TetrahedralGridDataset) "
                                                "is not supported")
            if isinstance(a, str):
                raise SetupError(f"{self.type}: the JSON form carries no data (only the placeholder '{a}'); load the simulation "
                                 "from its .hdf5 file (Simulation.from_file) or pass DataArrays.")
            if not isinstance(a, DataArray):
                raise SetupError(f"{self.type}: coefficients must be SpatialDataArrays")

    def _at(self, arr, x, y, z):
        return interp_dataset(arr, {"x": x, "y": y, "z": z}, self.interp_method)

    def pole_params_at(self, x, y, z):
        """-> (eps_inf [points], [(a [points], c [points]), ...])"""
        raise NotImplementedError

    def pole_residue(self):
        """Representative medium (the data's mean eps_inf, no poles) — what keys the structure in the material table."""
        e, _ = self.pole_params_at(*[np.asarray([float(np.mean(np.asarray(self._arrays()[0].coords[d], float)))]) for d in "xyz"])
        return float(np.real(e[0])), 1e-300, ()

    @property
    def n_cfl(self):
        """ref medium.py: sqrt of the smallest eps_inf in the data (1 for Sellmeier)."""
        self._check()
        return float(np.sqrt(max(self._eps_inf_min(), 1e-12)))

    def _eps_inf_min(self):
        return float(np.min(np.real(np.asarray(self.eps_inf.values))))

    def _eps_on_data(self, frequency: float) -> np.ndarray:
        """eps(frequency) at every point of the coefficients' own grid"""
        self._check()
        first = self._arrays()[0]
        X, Y, Z = np.meshgrid(*[np.asarray(first.coords[d], float) for d in "xyz"], indexing="ij")
        eps_inf, poles = self.pole_params_at(X.ravel(), Y.ravel(), Z.ravel())
        w = 2 * np.pi * float(frequency)
        eps = np.asarray(eps_inf, complex).copy()
        for a, c in poles:
            ok = c != 0
            eps[ok] -= (c[ok] / (1j * w + a[ok]) + np.conj(c[ok]) / (1j * w + np.conj(a[ok])))
        return eps

    def eps_model(self, frequency):
        """ref medium.py:1315-1322: the spatial mean."""
        f = np.atleast_1d(np.asarray(frequency, float))
        out = np.array([np.mean(self._eps_on_data(v)) for v in f])
        return out if np.ndim(frequency) else complex(out[0])

    def eps_diagonal(self, frequency):
        """ref medium.py:1324-1336: the value of largest modulus over the data (what AutoGrid sizes its steps with)."""
        e = self._eps_on_data(float(np.atleast_1d(frequency)[0]))
        v = e[np.argmax(np.abs(e))]
        return (v, v, v)


@_register
@dataclass
class CustomPoleResidue(_CustomDispersive):
    eps_inf: Any = None
    poles: Tuple[Tuple[Any, Any], ...] = ()
    interp_method: str = "nearest"
    subpixel: bool = False
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None
    allow_gain: Optional[bool] = None

    def _arrays(self):
        return [self.eps_inf] + [v for p in self.poles for v in p]

    def pole_params_at(self, x, y, z):
        self._check()
        return np.real(self._at(self.eps_inf, x, y, z)), [(self._at(a, x, y, z) + 0j, self._at(c, x, y, z) + 0j) for a, c in self.poles]


@_register
@dataclass
class CustomLorentz(_CustomDispersive):
    eps_inf: Any = None
    coeffs: Tuple[Tuple[Any, Any, Any], ...] = ()
    interp_method: str = "nearest"
    subpixel: bool = False
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None
    allow_gain: Optional[bool] = None

    def _arrays(self):
        return [self.eps_inf] + [v for p in self.coeffs for v in p]

    def pole_params_at(self, x, y, z):
        self._check()
        poles = []
        for de, f, delta in self.coeffs:            # the point-wise form of Lorentz.pole_residue (ref medium.py:4021-4047)
            de_, w, d = (np.real(self._at(v, x, y, z)) for v in (de, f, delta))
            w, d = 2 * np.pi * w, 2 * np.pi * d
            over = d * d > w * w
            r = np.sqrt(np.abs(d * d - w * w))
            r_safe = np.where(r == 0, 1.0, r)
            c_over = de_ * w ** 2 / 4 / r_safe
            poles.append((np.where(over, -d + r, -d - 1j * r), np.where(over, c_over + 0j, 1j * de_ * w ** 2 / 2 / r_safe)))
            poles.append((np.where(over, -d - r, 0.0) + 0j, np.where(over, -c_over, 0.0) + 0j))
        return np.real(self._at(self.eps_inf, x, y, z)), poles


@_register
@dataclass
class CustomDrude(_CustomDispersive):
    eps_inf: Any = None
    coeffs: Tuple[Tuple[Any, Any], ...] = ()
    interp_method: str = "nearest"
    subpixel: bool = False
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None
    allow_gain: Optional[bool] = None

    def _arrays(self):
        return [self.eps_inf] + [v for p in self.coeffs for v in p]

    def pole_params_at(self, x, y, z):
        self._check()
        poles = []
        for f, delta in self.coeffs:                # Drude.pole_residue point by point (ref medium.py:4384-4409)
            w, d = (2 * np.pi * np.real(self._at(v, x, y, z)) for v in (f, delta))
            c0 = (w ** 2) / 2 / d + 0j
            poles.append((np.zeros_like(c0), c0))
            poles.append((-d + 0j, -c0))
        return np.real(self._at(self.eps_inf, x, y, z)), poles


@_register
@dataclass
class CustomDebye(_CustomDispersive):
    eps_inf: Any = None
    coeffs: Tuple[Tuple[Any, Any], ...] = ()
    interp_method: str = "nearest"
    subpixel: bool = False
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None
    allow_gain: Optional[bool] = None

    def _arrays(self):
        return [self.eps_inf] + [v for p in self.coeffs for v in p]

    def pole_params_at(self, x, y, z):
        self._check()
        poles = []
        for de, tau in self.coeffs:                 # Debye.pole_residue point by point (ref medium.py:4652-4666)
            de_, tau_ = (np.real(self._at(v, x, y, z)) for v in (de, tau))
            a = -2 * np.pi / tau_ + 0j
            poles.append((a, -0.5 * de_ * a))
        return np.real(self._at(self.eps_inf, x, y, z)), poles


@_register
@dataclass
class CustomSellmeier(_CustomDispersive):
    coeffs: Tuple[Tuple[Any, Any], ...] = ()
    interp_method: str = "nearest"
    subpixel: bool = False
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None
    allow_gain: Optional[bool] = None

    def _arrays(self):
        return [v for p in self.coeffs for v in p]

    def _eps_inf_min(self):
        return 1.0

    def pole_params_at(self, x, y, z):
        self._check()
        poles = []
        for B, C in self.coeffs:                    # Sellmeier.pole_residue point by point (ref medium.py:3677-3686)
            B_, C_ = (np.real(self._at(v, x, y, z)) for v in (B, C))
            beta = 2 * np.pi * C_0 / np.sqrt(C_)
            poles.append((1j * beta, -0.5j * beta * B_))
        return np.ones(np.broadcast(x, y, z).shape), poles


@_register
@dataclass
class PECMedium(_AbstractMedium):
    """Perfect electric conductor (ref medium.py:1454)."""

    name: Optional[str] = None
    is_pec = True

    def pole_residue(self):
        return 1.0, 0.0, ()

    @property
    def n_cfl(self):
        return 1.0

    def eps_model(self, frequency):
        return -1e8 + 0j * np.asarray(frequency, float)   # pec_val, ref constants.py:64


PEC = PECMedium(name="PEC")


@_register
@dataclass
class PoleResidue(_AbstractMedium):
    """ref medium.py:2843; eps(w) formula :2900-2913."""

    eps_inf: float = 1.0
    poles: Tuple[Tuple[complex, complex], ...] = ()
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None

    def __post_init__(self):
        self.poles = tuple((_complex(a), _complex(c)) for a, c in self.poles)

    def pole_residue(self):
        return float(self.eps_inf), 0.0, self.poles


@_register
@dataclass
class Lorentz(_AbstractMedium):
    """eps = eps_inf + sum de f^2/(f^2 - 2j f delta - f_^2) (ref medium.py:3943; poles :4021-4047)."""

    eps_inf: float = 1.0
    coeffs: Tuple[Tuple[float, float, float], ...] = ()
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None

    def pole_residue(self):
        poles = []
        for de, f, delta in self.coeffs:
            w = 2 * np.pi * f
            d = 2 * np.pi * delta
            if d * d > w * w:                       # over-damped: two real poles
                r = np.sqrt(d * d - w * w) + 0j
                c0 = de * w ** 2 / 4 / r
                poles.append((-d + r, c0))
                poles.append((-d - r, -c0))
            else:
                r = np.sqrt(w * w - d * d)
                poles.append((complex(-d, -r), 1j * de * w ** 2 / 2 / r))
        return float(self.eps_inf), 0.0, tuple(poles)


@_register
@dataclass
class Drude(_AbstractMedium):
    """eps = eps_inf - sum f^2/(f_^2 + j f_ delta) (ref medium.py:4327; poles :4384-4409)."""

    eps_inf: float = 1.0
    coeffs: Tuple[Tuple[float, float], ...] = ()
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None

    def pole_residue(self):
        poles = []
        for f, delta in self.coeffs:
            w = 2 * np.pi * f
            d = 2 * np.pi * delta
            c0 = (w ** 2) / 2 / d + 0j
            poles.append((0j, c0))
            poles.append((-d + 0j, -c0))
        return float(self.eps_inf), 0.0, tuple(poles)


@_register
@dataclass
class Sellmeier(_AbstractMedium):
    """n^2 = 1 + sum B l^2/(l^2 - C) (ref medium.py:3584; poles :3677-3686)."""

    coeffs: Tuple[Tuple[float, float], ...] = ()
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None

    def pole_residue(self):
        poles = []
        for B, C in self.coeffs:
            beta = 2 * np.pi * C_0 / np.sqrt(C)
            alpha = -0.5 * beta * B
            poles.append((1j * beta, 1j * alpha))
        return 1.0, 0.0, tuple(poles)


@_register
@dataclass
class Debye(_AbstractMedium):
    """eps = eps_inf + sum de/(1 - j f tau) (ref medium.py:4579; poles :4652-4666)."""

    eps_inf: float = 1.0
    coeffs: Tuple[Tuple[float, float], ...] = ()
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None

    def pole_residue(self):
        poles = []
        for de, tau in self.coeffs:
            a = -2 * np.pi / tau + 0j
            poles.append((a, -0.5 * de * a))
        return float(self.eps_inf), 0.0, tuple(poles)


@_register
@dataclass
class AnisotropicMedium(_AbstractMedium):
    """Diagonally anisotropic medium (ref medium.py:4863): an isotropic medium per E component —
    exactly what the per-component material indices of the kernels hold."""

    xx: Any = None
    yy: Any = None
    zz: Any = None
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None
    allow_gain: Optional[bool] = None

    @property
    def components(self):
        return [self.xx, self.yy, self.zz]

    @property
    def is_pec(self):
        return False          # per component: see ``component(c).is_pec``

    def component(self, c: int):
        m = self.components[c]
        if isinstance(m, Unsupported):
            m.fail()
        return m

    @property
    def n_cfl(self):
        """ref medium.py:4950-4957: the smallest of the components."""
        return min(m.n_cfl for m in self.components)

    def eps_comp(self, c: int, frequency):
        return self.component(c).eps_model(frequency)

    def eps_model(self, frequency):
        """ref medium.py:4960-4963: mean of the diagonal."""
        return np.mean([self.eps_comp(c, frequency) for c in range(3)], axis=0)

    def pole_residue(self):
        raise Tidy3dNotImplementedError("AnisotropicMedium has one pole-residue model per component")


@_register
@dataclass
class FullyAnisotropicMedium(_AbstractMedium):
    """Lossless medium with a full (symmetric, positive-definite, eigenvalues >= 1) permittivity tensor (ref medium.py:5058).  The
    sweep advances every E component with the diagonal of eps^-1; the off-diagonal coupling is added by a patch behind it
    (spec.AnisoSet).  A conductivity tensor is not supported."""

    permittivity: Any = ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))
    conductivity: Any = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None
    allow_gain: Optional[bool] = None

    def __post_init__(self):
        eps = np.asarray(self.permittivity, float).reshape(3, 3)
        if not np.allclose(eps, eps.T, atol=1e-6):
            raise ValidationError("Provided permittivity tensor is not symmetric.")
        if np.any(np.linalg.eigvalsh(0.5 * (eps + eps.T)) < 1 - 1e-6):
            raise ValidationError("Main diagonal of provided permittivity tensor is not >= 1.")
        self.permittivity = tuple(tuple(float(v) for v in row) for row in 0.5 * (eps + eps.T))
        self.conductivity = tuple(tuple(float(v) for v in row) for row in np.asarray(self.conductivity, float).reshape(3, 3))

    @classmethod
    def from_diagonal(cls, xx, yy, zz, rotation_matrix, **kw):
        """The diagonal medium diag(xx, yy, zz) (permittivities) in axes rotated by ``rotation_matrix`` (3 x 3; ref
        medium.py:5173: R eps R^T)."""
        R = np.asarray(rotation_matrix, float).reshape(3, 3)
        d = np.diag([float(getattr(m, "permittivity", m)) for m in (xx, yy, zz)])
        return cls(permittivity=R @ d @ R.T, **kw)

    @property
    def eps_tensor(self) -> np.ndarray:
        return np.asarray(self.permittivity, float)

    @property
    def inv_tensor(self) -> np.ndarray:
        return np.linalg.inv(self.eps_tensor)

    @property
    def is_lossless(self) -> bool:
        return not np.any(np.asarray(self.conductivity, float))

    def component(self, c: int):
        """The isotropic medium the sweep uses for E_c: permittivity 1 / [eps^-1]_cc."""
        return Medium(permittivity=float(1.0 / self.inv_tensor[c, c]))

    @property
    def n_cfl(self):
        """ref medium.py FullyAnisotropicMedium.n_cfl: sqrt of the smallest eigenvalue."""
        return float(np.sqrt(np.min(np.linalg.eigvalsh(self.eps_tensor))))

    def eps_model(self, frequency):
        """ref medium.py: the mean of the eigenvalues."""
        return float(np.mean(np.linalg.eigvalsh(self.eps_tensor))) + 0j * np.asarray(frequency, float)

    def eps_diagonal(self, frequency):
        return tuple(complex(v) for v in np.linalg.eigvalsh(self.eps_tensor))

    def pole_residue(self):
        raise Tidy3dNotImplementedError("FullyAnisotropicMedium has no scalar pole-residue model (see component(c) and spec.AnisoSet)")


@_register
@dataclass
class Medium2D(_AbstractMedium):
    """2-D material on a zero-thickness geometry (ref medium.py:6090): ``ss`` / ``tt`` describe the two in-plane components
    (in x, y, z order without the normal) as SHEET quantities — permittivity x thickness, sheet conductivity [S].  The front
    end gives the tangential E nodes on the sheet's plane (snapped to the nearest grid line, ref utils_2d.py:41-45) the
    volumetric equivalent of ref medium.py:6170-6238: the media either side averaged with the adjacent cell sizes, plus the
    sheet's contribution divided by the mean of those sizes (discretize.rasterize)."""

    ss: Any = None
    tt: Any = None
    name: Optional[str] = None
    frequency_range: Optional[Tuple[float, float]] = None
    allow_gain: Optional[bool] = None

    @property
    def is_pec(self):
        return False          # (per component; a PEC sheet has both of them PEC, ref medium.py:6137-6146)

    @property
    def is_pec_sheet(self):
        return bool(getattr(self.ss, "is_pec", False))

    @property
    def n_cfl(self):
        return 1.0            # ref medium.py Medium2D.n_cfl: the sheet does not enter the time step

    def eps_model(self, frequency):
        """ref medium.py Medium2D.eps_model: the mean of the in-plane components (sheet quantities)."""
        return np.mean([self.ss.eps_model(frequency), self.tt.eps_model(frequency)], axis=0)

    def pole_residue(self):
        raise Tidy3dNotImplementedError("Medium2D has no volumetric pole-residue model of its own (see discretize.rasterize)")


@_register
@dataclass
class CustomAnisotropicMedium(AnisotropicMedium):
    """Diagonally anisotropic medium whose components vary in space (ref medium.py:5300): each of xx / yy / zz a ``CustomMedium`` or a
    custom dispersive medium, rasterised at the nodes of its own E component."""

    interp_method: Optional[str] = None
    subpixel: bool = False

    def component(self, c: int):
        m = super().component(c)
        if self.interp_method is not None and hasattr(m, "interp_method"):      # ref medium.py:5365-5375: overrides the components' own
            m = dataclasses.replace(m, interp_method=self.interp_method)
        return m


@_register
@dataclass
class PerturbationMedium(Medium):
    """``Medium`` with heat / charge perturbation models attached (ref medium.py:5648).  An FDTD run sees the UNPERTURBED medium — the
    reference applies the models only through ``Simulation.perturbed_mediums_copy``, which hands over custom media — so the models are
    carried along and not used."""

    permittivity_perturbation: Any = None
    conductivity_perturbation: Any = None
    perturbation_spec: Any = None
    interp_method: str = "linear"
    subpixel: bool = True


@_register
@dataclass
class PerturbationPoleResidue(PoleResidue):
    """``PoleResidue`` with perturbation models attached (ref medium.py:5851); the FDTD run sees the unperturbed medium."""

    eps_inf_perturbation: Any = None
    poles_perturbation: Any = None
    perturbation_spec: Any = None
    interp_method: str = "linear"
    subpixel: bool = True


@_register
@dataclass
class Structure(_Model):
    """geometry + medium (ref components/structure.py:147)."""

    geometry: Any = None
    medium: Any = None
    name: Optional[str] = None


@_register
@dataclass
class LumpedResistor(_Model):
    """Rectangular lumped resistor (ref lumped_element.py:72): a planar box that enters the simulation as a ``Medium2D`` sheet
    of conductance  L_voltage / (L_lateral R)  (ref :150-168), behind the user's structures."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    resistance: float = 50.0
    voltage_axis: int = 0
    name: Optional[str] = None
    num_grid_cells: Optional[int] = 3

    def __post_init__(self):
        self.center = tuple(float(c) for c in self.center)
        self.size = tuple(float(v) for v in self.size)
        if sum(1 for v in self.size if v == 0.0) != 1:
            raise SetupError(f"lumped element '{self.name}' must be planar: exactly one zero-size dimension (ref lumped_element.py assert_plane)")
        if self.size.index(0.0) == int(self.voltage_axis):
            raise SetupError(f"'voltage_axis' must be in the plane of lumped element '{self.name}' (ref lumped_element.py:134-147)")
        if not self.resistance > 0:
            raise ValidationError("resistance must be positive")

    @property
    def normal_axis(self) -> int:
        return self.size.index(0.0)

    @property
    def sheet_conductance(self) -> float:
        lateral = 3 - int(self.voltage_axis) - self.normal_axis
        return self.size[int(self.voltage_axis)] / self.size[lateral] / float(self.resistance)

    def to_structure(self) -> "Structure":
        med = Medium(conductivity=self.sheet_conductance)
        return Structure(geometry=Box(center=self.center, size=self.size), medium=Medium2D(ss=med, tt=med), name=self.name)


@_register
@dataclass
class CoaxialLumpedResistor(_Model):
    """Coaxial lumped resistor (ref lumped_element.py:170): an annular ``Medium2D`` sheet between two concentric circles, of
    conductance  ln(r_out / r_in) / (2 pi R)  (ref :269-274)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    outer_diameter: float = 1.0
    inner_diameter: float = 0.5
    normal_axis: int = 2
    resistance: float = 50.0
    name: Optional[str] = None
    num_grid_cells: Optional[int] = 3

    def __post_init__(self):
        self.center = tuple(float(c) for c in self.center)
        if not self.resistance > 0:
            raise ValidationError("resistance must be positive")
        if not 0 < self.inner_diameter < self.outer_diameter:
            raise ValidationError(f"The 'inner_diameter' {self.inner_diameter} of a coaxial lumped element must be less than its "
                                  f"'outer_diameter' {self.outer_diameter}.")

    @property
    def sheet_conductance(self) -> float:
        return float(np.log(self.outer_diameter / self.inner_diameter) / (2 * np.pi * self.resistance))

    def to_structure(self) -> "Structure":
        med = Medium(conductivity=self.sheet_conductance)
        disks = [Cylinder(axis=int(self.normal_axis), radius=0.5 * d, length=0.0, center=self.center)
                 for d in (self.outer_diameter, self.inner_diameter)]
        return Structure(geometry=ClipOperation(operation="difference", geometry_a=disks[0], geometry_b=disks[1]),
                         medium=Medium2D(ss=med, tt=med), name=self.name)


# --------------------------------------------------------------------------------------
# grid specification  (ref components/grid/grid_spec.py)
# --------------------------------------------------------------------------------------

@_register
@dataclass
class UniformGrid(_Model):
    """ref grid_spec.py:212; coords :237-269 (N = ceil(size/dl), dl snapped to tile size)."""

    dl: float = 0.1

    def make_coords_initial(self, center: float, size: float) -> np.ndarray:
        num_cells = max(int(np.ceil(size / self.dl)), 1)
        dl_snapped = size / num_cells if size > 0 else self.dl
        return center - size / 2 + np.arange(num_cells + 1) * dl_snapped


@_register
@dataclass
class CustomGrid(_Model):
    """ref grid_spec.py:316; coords :350-385 + _postprocess_unaligned_grid :137-210."""

    dl: Tuple[float, ...] = ()
    custom_offset: Optional[float] = None

    def make_coords_initial(self, center: float, size: float) -> np.ndarray:
        b = np.append(0.0, np.cumsum(np.array(self.dl, float)))
        if self.custom_offset is None:
            b = b + (center - b[-1] / 2)
        else:
            b = b + self.custom_offset
        relax = self.custom_offset is not None
        bmin = np.nextafter(np.float32(center - size / 2), np.float32(-inf), dtype=np.float32)
        bmax = np.nextafter(np.float32(center + size / 2), np.float32(inf), dtype=np.float32)
        if bmax < b[0] or bmin > b[-1]:
            raise SetupError("Simulation domain does not overlap with the provided grid.")
        if size == 0:
            ind = min(int(np.searchsorted(b, center, side="right")), len(b) - 1)
            return b[ind - 1: ind + 1]
        b = b[b <= bmax]
        b = b[b >= bmin]
        dl_min, dl_max = b[1] - b[0], b[-1] - b[-2]
        while b[0] - dl_min >= bmin:
            b = np.insert(b, 0, b[0] - dl_min)
        while b[-1] + dl_max <= bmax:
            b = np.append(b, b[-1] + dl_max)
        if relax:
            if np.isclose(b[0] - dl_min, bmin):
                b = np.insert(b, 0, b[0] - dl_min)
            if np.isclose(b[-1] + dl_max, bmax):
                b = np.append(b, b[-1] + dl_max)
        return b


@_register
@dataclass
class CustomGridBoundaries(_Model):
    """ref grid_spec.py:272: explicit boundary coordinates."""

    coords: Tuple[float, ...] = ()

    def make_coords_initial(self, center: float, size: float) -> np.ndarray:
        b = np.array(self.coords, float)
        return CustomGrid(dl=tuple(np.diff(b)), custom_offset=float(b[0])).make_coords_initial(
            center, size)


@_register
@dataclass
class AutoGrid(_Model):
    """Graded non-uniform grid (ref grid_spec.py:386); meshed by ``tidy3d_amd.autogrid``."""

    min_steps_per_wvl: float = 10.0
    max_scale: float = 1.4
    dl_min: float = 0.0
    mesher: Any = None


@_register
@dataclass
class MeshOverrideStructure(_Model):
    """Box that overrides the step inside it (ref structure.py MeshOverrideStructure)."""

    geometry: Any = None
    dl: Tuple[Optional[float], Optional[float], Optional[float]] = (None, None, None)
    enforce: bool = False
    name: Optional[str] = None


@_register
@dataclass
class GridSpec(_Model):
    """ref grid_spec.py:520; ``make_grid`` :670; ``uniform`` classmethod :745; ``auto`` :757."""

    grid_x: Any = None
    grid_y: Any = None
    grid_z: Any = None
    wavelength: Optional[float] = None
    override_structures: Tuple[Any, ...] = ()
    snapping_points: Tuple[Any, ...] = ()

    def __post_init__(self):
        # the reference's default GridSpec is AutoGrid on all three axes (ref grid_spec.py:562-581)
        for name in ("grid_x", "grid_y", "grid_z"):
            if getattr(self, name) is None:
                setattr(self, name, AutoGrid())

    @classmethod
    def auto(cls, wavelength: Optional[float] = None, min_steps_per_wvl: float = 10.0, max_scale: float = 1.4,
             override_structures=(), snapping_points=(), dl_min: float = 0.0) -> "GridSpec":
        g = dict(min_steps_per_wvl=min_steps_per_wvl, max_scale=max_scale, dl_min=dl_min)
        return cls(grid_x=AutoGrid(**g), grid_y=AutoGrid(**g), grid_z=AutoGrid(**g), wavelength=wavelength,
                   override_structures=tuple(override_structures), snapping_points=tuple(snapping_points))

    @classmethod
    def uniform(cls, dl: float) -> "GridSpec":
        return cls(grid_x=UniformGrid(dl=dl), grid_y=UniformGrid(dl=dl), grid_z=UniformGrid(dl=dl))

    @property
    def grids_1d(self):
        return [self.grid_x, self.grid_y, self.grid_z]


# --------------------------------------------------------------------------------------
# boundaries  (ref components/boundary.py)
# --------------------------------------------------------------------------------------

@_register
@dataclass
class PMLParams(_Model):
    """ref boundary.py:195-227; defaults = DefaultPMLParameters :233-243. sigma/alpha in
    units of 2*EPSILON_0/dt (ref constants.py:120)."""

    sigma_order: int = 3
    sigma_min: float = 0.0
    sigma_max: float = 1.5
    kappa_order: int = 3
    kappa_min: float = 1.0
    kappa_max: float = 3.0
    alpha_order: int = 1
    alpha_min: float = 0.0
    alpha_max: float = 0.0


DefaultPMLParameters = PMLParams()
DefaultStablePMLParameters = PMLParams(sigma_order=3, sigma_min=0.0, sigma_max=1.0,
                                       kappa_order=3, kappa_min=1.0, kappa_max=5.0,
                                       alpha_order=1, alpha_min=0.0, alpha_max=0.9)


@_register
@dataclass
class PML(_Model):
    """ref boundary.py:275 (12 layers :379)."""

    num_layers: int = 12
    parameters: PMLParams = field(default_factory=PMLParams)
    name: Optional[str] = None


@_register
@dataclass
class StablePML(_Model):
    """ref boundary.py:392 (40 layers, DefaultStablePMLParameters :244)."""

    num_layers: int = 40
    parameters: PMLParams = field(default_factory=lambda: dataclasses.replace(
        DefaultStablePMLParameters))
    name: Optional[str] = None


@_register
@dataclass
class AbsorberParams(_Model):
    """ref boundary.py:166-192; sigma in units of 2*EPSILON_0/dt (ref constants.py:120)."""

    sigma_order: int = 3
    sigma_min: float = 0.0
    sigma_max: float = 1.5


@_register
@dataclass
class Absorber(_Model):
    """Adiabatic absorber: layers of polynomially rising electric conductivity in front of a PEC
    wall (ref boundary.py:427-476; 40 layers, DefaultAbsorberParameters sigma_max 6.4 :232)."""

    num_layers: int = 40
    parameters: AbsorberParams = field(default_factory=lambda: AbsorberParams(sigma_max=6.4))
    name: Optional[str] = None


@_register
@dataclass
class PECBoundary(_Model):
    """ref boundary.py:40."""
    name: Optional[str] = None


@_register
@dataclass
class PMCBoundary(_Model):
    """ref boundary.py:45."""
    name: Optional[str] = None


@_register
@dataclass
class Periodic(_Model):
    """ref boundary.py:27."""
    name: Optional[str] = None


@_register
@dataclass
class BlochBoundary(_Model):
    """ref boundary.py:55-160: F(r + L) = bloch_phase F(r), bloch_vec in units of 2 pi / L."""

    bloch_vec: float = 0.0
    name: Optional[str] = None

    @property
    def bloch_phase(self) -> complex:
        """ref boundary.py:75-79."""
        return complex(np.exp(1j * 2.0 * np.pi * self.bloch_vec))

    @classmethod
    def from_source(cls, source, domain_size: float, axis: int, medium=None):
        """Bloch vector of an angled source at its centre frequency (ref boundary.py:81-160)."""
        if axis == source.injection_axis:
            raise SetupError("Bloch boundary axis must be orthogonal to the injection axis of 'source'.")
        freq0 = source.source_time.freq0
        eps = 1.0 if medium is None else complex(np.asarray(medium.eps_model(freq0)).ravel()[0])
        from .constants import EPSILON_0, MU_0
        kmag = float(np.real(freq0 * np.sqrt(eps * EPSILON_0 * MU_0)))
        theta, phi = float(source.angle_theta), float(source.angle_phi)
        if theta == 0:
            return cls(bloch_vec=0)
        if source.direction == "-":
            theta += np.pi
        k_local = [kmag * np.sin(theta) * np.cos(phi), kmag * np.sin(theta) * np.sin(phi), kmag * np.cos(theta)]
        k_global = [k_local[0], k_local[1]]
        k_global.insert(source.injection_axis, k_local[2])          # unpop_axis
        return cls(bloch_vec=float(domain_size * k_global[axis]))


@_register
@dataclass
class Boundary(_Model):
    """Pair of edges along one axis (ref boundary.py:492); default PML both (:520-531)."""

    plus: Any = field(default_factory=PML)
    minus: Any = field(default_factory=PML)

    def __post_init__(self):
        per = [isinstance(e, Periodic) for e in (self.plus, self.minus)]
        if any(per) and not all(per):
            raise SetupError("Periodic boundaries must be applied on both sides of an axis "
                             "(ref boundary.py:536-560).")
        blo = [isinstance(e, BlochBoundary) for e in (self.plus, self.minus)]
        if any(blo) and not (all(blo) and self.plus.bloch_vec == self.minus.bloch_vec):
            raise SetupError("Bloch boundaries must be applied on both sides of an axis with the same "
                             "Bloch vector (ref boundary.py:536-575).")

    @classmethod
    def pml(cls, num_layers: int = 12, parameters: PMLParams = None):
        p = parameters or PMLParams()
        return cls(plus=PML(num_layers=num_layers, parameters=p),
                   minus=PML(num_layers=num_layers, parameters=p))

    @classmethod
    def stable_pml(cls, num_layers: int = 40, parameters: PMLParams = None):
        p = parameters or dataclasses.replace(DefaultStablePMLParameters)
        return cls(plus=StablePML(num_layers=num_layers, parameters=p),
                   minus=StablePML(num_layers=num_layers, parameters=p))

    @classmethod
    def absorber(cls, num_layers: int = 40, parameters: AbsorberParams = None):
        """ref boundary.py:676-700."""
        p = parameters or AbsorberParams(sigma_max=6.4)
        return cls(plus=Absorber(num_layers=num_layers, parameters=p),
                   minus=Absorber(num_layers=num_layers, parameters=p))

    @classmethod
    def pec(cls):
        return cls(plus=PECBoundary(), minus=PECBoundary())

    @classmethod
    def pmc(cls):
        return cls(plus=PMCBoundary(), minus=PMCBoundary())

    @classmethod
    def periodic(cls):
        return cls(plus=Periodic(), minus=Periodic())

    @classmethod
    def bloch(cls, bloch_vec: float):
        """ref boundary.py:592-607."""
        return cls(plus=BlochBoundary(bloch_vec=bloch_vec), minus=BlochBoundary(bloch_vec=bloch_vec))

    @classmethod
    def bloch_from_source(cls, source, domain_size: float, axis: int, medium=None):
        """ref boundary.py:609-640."""
        b = BlochBoundary.from_source(source, domain_size, axis, medium)
        return cls(plus=b, minus=dataclasses.replace(b))


@_register
@dataclass
class BoundarySpec(_Model):
    """ref boundary.py:732."""

    x: Boundary = field(default_factory=Boundary)
    y: Boundary = field(default_factory=Boundary)
    z: Boundary = field(default_factory=Boundary)

    @classmethod
    def all_sides(cls, boundary):
        mk = lambda: Boundary(plus=dataclasses.replace(boundary), minus=dataclasses.replace(boundary))
        return cls(x=mk(), y=mk(), z=mk())

    @classmethod
    def pml(cls, x: bool = False, y: bool = False, z: bool = False):
        """PML along the flagged axes, periodic elsewhere (ref boundary.py:803-831)."""
        mk = lambda f: Boundary.pml() if f else Boundary.periodic()
        return cls(x=mk(x), y=mk(y), z=mk(z))

    @classmethod
    def pec(cls, x: bool = False, y: bool = False, z: bool = False):
        mk = lambda f: Boundary.pec() if f else Boundary.periodic()
        return cls(x=mk(x), y=mk(y), z=mk(z))

    @property
    def to_list(self):
        return [(self.x.minus, self.x.plus), (self.y.minus, self.y.plus),
                (self.z.minus, self.z.plus)]


# --------------------------------------------------------------------------------------
# source time dependence  (ref components/source.py:60-260, components/time.py)
# --------------------------------------------------------------------------------------

class _SourceTime(_Model):

    @property
    def twidth(self) -> float:
        return 1.0 / (2 * np.pi * self.fwidth)

    def frequency_range(self, num_fwidth: float = 4.0):
        """ref source.py:133-153."""
        w = num_fwidth * self.fwidth
        return (max(0.0, self.freq0 - w), self.freq0 + w)

    def spectrum(self, times, freqs, dt, complex_fields: bool = False):
        """dt/sqrt(2 pi) * sum_n Re[amp(t_n)] e^{+i 2 pi f t_n}, times cut where the relative
        amplitude is below DFT_CUTOFF (ref time.py:46-105).  ``complex_fields`` (simulations with Bloch
        boundaries): the complex amplitude itself is injected, so the sum runs over amp(t_n), which
        is twice the positive-frequency part of the real signal; data normalised by it equal those of a
        real-field run."""
        times = np.asarray(times, float)
        freqs = np.atleast_1d(np.asarray(freqs, float))
        amps = self.amp_time(times) if complex_fields else np.real(self.amp_time(times))
        if np.all(amps == 0.0):
            return np.zeros(len(freqs), complex)
        rel = np.where(np.abs(amps) / np.amax(np.abs(amps)) > DFT_CUTOFF)[0]
        lo, hi = rel[0], rel[-1] + 1
        amps, tcut = amps[lo:hi], times[lo:hi]
        if tcut.size == 0:
            return np.zeros(len(freqs), complex)
        # blocked exact evaluation of the DTFT sum (the reference uses a running product)
        out = np.zeros(len(freqs), complex)
        for s in range(0, tcut.size, 8192):
            ph = np.exp(2j * np.pi * freqs[:, None] * tcut[None, s:s + 8192])
            out += ph @ amps[s:s + 8192]
        return dt * out / np.sqrt(2 * np.pi)


@_register
@dataclass
class GaussianPulse(_SourceTime):
    """ref source.py:155; ``amp_time`` :174-193; ``end_time`` :195-202."""

    freq0: float = 1.0
    fwidth: float = 1.0
    offset: float = 5.0
    amplitude: float = 1.0
    phase: float = 0.0
    remove_dc_component: bool = True

    def __post_init__(self):
        if self.offset < 2.5:
            raise ValidationError("GaussianPulse.offset must be >= 2.5 (ref source.py:127).")

    def amp_time(self, time):
        time = np.asarray(time, float)
        omega0 = 2 * np.pi * self.freq0
        ts = time - self.offset * self.twidth
        amp = (np.exp(1j * self.phase) * np.exp(-1j * omega0 * time)
               * np.exp(-(ts ** 2) / 2 / self.twidth ** 2) * self.amplitude)
        if self.remove_dc_component:
            return amp * (1j + ts / self.twidth ** 2 / omega0)
        return amp * 1j

    def end_time(self):
        return self.offset * self.twidth + END_TIME_FACTOR_GAUSSIAN * self.twidth

    @property
    def amp_complex(self):
        return self.amplitude * np.exp(1j * self.phase)


@_register
@dataclass
class ContinuousWave(_SourceTime):
    """ref source.py:226; ``amp_time`` :239-252."""

    freq0: float = 1.0
    fwidth: float = 1.0
    offset: float = 5.0
    amplitude: float = 1.0
    phase: float = 0.0

    def amp_time(self, time):
        time = np.asarray(time, float)
        omega0 = 2 * np.pi * self.freq0
        ts = time - self.offset * self.twidth
        return (np.exp(1j * self.phase) * np.exp(-1j * omega0 * time)
                / (1 + np.exp(-ts / self.twidth)) * self.amplitude)

    def end_time(self):
        return None

    @property
    def amp_complex(self):
        return self.amplitude * np.exp(1j * self.phase)


@_register
@dataclass
class TimeDataset(_Model):
    """ref data/dataset.py TimeDataset: ``values`` = TimeDataArray over t."""

    values: Any = None


@_register
@dataclass
class CustomSourceTime(_SourceTime):
    """ref source.py:259-435: amp_time(t) = amplitude e^{i phase - 2 pi i freq0 t} envelope(t - offset twidth),
    envelope = linear interpolation of the dataset, its end values outside."""

    freq0: float = 1.0
    fwidth: float = 1.0
    offset: float = 0.0
    amplitude: float = 1.0
    phase: float = 0.0
    source_time_dataset: Any = None

    def _data(self):
        v = getattr(self.source_time_dataset, "values", None)
        if v is None or isinstance(v, str):
            raise SetupError("CustomSourceTime: the JSON form carries no data; load the simulation from its "
                             ".hdf5 file (Simulation.from_file) or pass a TimeDataArray.")
        t = np.asarray(v.coords["t"], float).ravel()
        order = np.argsort(t)
        return t[order], np.asarray(v.values).ravel()[order]

    def amp_time(self, time):
        time = np.asarray(time, float)
        t, env = self._data()
        ts = time - self.offset * self.twidth
        e = np.interp(ts, t, env.real) + 1j * np.interp(ts, t, env.imag)        # np.interp clamps to the end values
        return np.exp(1j * self.phase) * np.exp(-2j * np.pi * self.freq0 * time) * self.amplitude * e

    def end_time(self):
        """ref source.py:423-435: last time with a non-zero envelope."""
        t, env = self._data()
        nz = ~np.isclose(np.abs(env), 0)
        return float(np.max(t[nz])) if nz.any() else None

    @property
    def amp_complex(self):
        return self.amplitude * np.exp(1j * self.phase)


# --------------------------------------------------------------------------------------
# sources  (ref components/source.py)
# --------------------------------------------------------------------------------------

class _Source(_Model):

    @property
    def geometry(self) -> Box:
        return Box(center=self.center, size=self.size)


@_register
@dataclass
class UniformCurrentSource(_Source):
    """ref source.py:585."""

    source_time: Any = None
    polarization: str = "Ez"
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    interpolate: bool = True
    confine_to_bounds: bool = False
    name: Optional[str] = None


@_register
@dataclass
class PointDipole(_Source):
    """ref source.py:600 (size fixed to (0,0,0) :624)."""

    source_time: Any = None
    polarization: str = "Ez"
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    interpolate: bool = True
    confine_to_bounds: bool = False
    name: Optional[str] = None

    def __post_init__(self):
        if tuple(self.size) != (0, 0, 0):
            raise ValidationError("PointDipole.size must be (0, 0, 0).")


@_register
@dataclass
class PlaneWave(_Source):
    """ref source.py:1090; polarisation vector :966-990 (pol_angle = 0 -> E along the first
    tangential axis ... for z injection: Ex)."""

    source_time: Any = None
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (inf, inf, 0.0)
    direction: str = "+"
    angle_theta: float = 0.0
    angle_phi: float = 0.0
    pol_angle: float = 0.0
    num_freqs: int = 1
    name: Optional[str] = None

    @property
    def injection_axis(self) -> int:
        zd = [d for d, s in enumerate(self.size) if s == 0]
        if len(zd) != 1:
            raise SetupError("PlaneWave must have exactly one zero-size dimension.")
        return zd[0]


@_register
@dataclass
class FieldDataset(_Model):
    """ref data/dataset.py FieldDataset: Ex .. Hz ScalarFieldDataArrays over (x, y, z, f)."""

    Ex: Any = None
    Ey: Any = None
    Ez: Any = None
    Hx: Any = None
    Hy: Any = None
    Hz: Any = None

    @property
    def field_components(self):
        return {k: getattr(self, k) for k in ("Ex", "Ey", "Ez", "Hx", "Hy", "Hz") if getattr(self, k) is not None}


def _need_data(ds, what: str):
    if ds is None or any(isinstance(v, str) for v in ds.field_components.values()):
        raise SetupError(f"{what}: the JSON form carries no data; load the simulation from its .hdf5 file "
                         "(Simulation.from_file) or pass ScalarFieldDataArrays.")
    if not ds.field_components:
        raise SetupError(f"{what}: the dataset holds no field component.")
    return ds


@_register
@dataclass
class CustomFieldSource(_Source):
    """Planar source from E / H data through the equivalence principle (ref source.py:781-900): coordinates
    relative to the source centre, tangential components only, normal = +axis."""

    source_time: Any = None
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    field_dataset: Any = None
    name: Optional[str] = None
    direction = "+"

    @property
    def injection_axis(self) -> int:
        zd = [d for d, s in enumerate(self.size) if s == 0]
        if len(zd) != 1:
            raise SetupError("CustomFieldSource must have exactly one zero-size dimension.")
        return zd[0]


@_register
@dataclass
class CustomCurrentSource(_Source):
    """Current densities J (Ex .. Ez) and M (Hx .. Hz) from a dataset (ref source.py:632-700), coordinates
    relative to the source centre."""

    source_time: Any = None
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    current_dataset: Any = None
    interpolate: bool = True
    confine_to_bounds: bool = False
    name: Optional[str] = None


@_register
@dataclass
class GaussianBeam(PlaneWave):
    """ref source.py:1109-1154: Gaussian beam on a finite plane; a positive ``waist_distance`` puts the
    waist behind the source plane (w.r.t. the propagation direction)."""

    waist_radius: float = 1.0
    waist_distance: float = 0.0


@_register
@dataclass
class AstigmaticGaussianBeam(PlaneWave):
    """ref source.py:1157-1201: simple astigmatic Gaussian beam (two waist sizes / distances)."""

    waist_sizes: Tuple[float, float] = (1.0, 1.0)
    waist_distances: Tuple[float, float] = (0.0, 0.0)


@_register
@dataclass
class TFSF(_Source):
    """Total-field/scattered-field box source (ref source.py:1204-1257): plane wave
    of 1 W/um^2 along ``injection_axis`` inside the box, nothing outside."""

    source_time: Any = None
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    injection_axis: int = 2
    direction: str = "+"
    angle_theta: float = 0.0
    angle_phi: float = 0.0
    pol_angle: float = 0.0
    num_freqs: int = 1
    name: Optional[str] = None


@_register
@dataclass
class ModeSpec(_Model):
    """ref components/mode.py:18-209."""

    num_modes: int = 1
    target_neff: Optional[float] = None
    num_pml: Tuple[int, int] = (0, 0)
    filter_pol: Optional[str] = None
    angle_theta: float = 0.0
    angle_phi: float = 0.0
    precision: str = "single"
    bend_radius: Optional[float] = None
    bend_axis: Optional[int] = None
    track_freq: Optional[str] = "central"
    group_index_step: Any = False


@_register
@dataclass
class ModeSource(_Source):
    """ref source.py:993-1085: injects mode ``mode_index`` carrying 1 W at freq0."""

    source_time: Any = None
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (inf, inf, 0.0)
    direction: str = "+"
    mode_spec: ModeSpec = field(default_factory=ModeSpec)
    mode_index: int = 0
    num_freqs: int = 1
    name: Optional[str] = None

    @property
    def injection_axis(self) -> int:
        zd = [d for d, s in enumerate(self.size) if s == 0]
        if len(zd) != 1:
            raise SetupError("ModeSource must be planar.")
        return zd[0]


# --------------------------------------------------------------------------------------
# monitors  (ref components/monitor.py)
# --------------------------------------------------------------------------------------

@_register
@dataclass
class ApodizationSpec(_Model):
    """ref components/apodization.py:8-102 (Gaussian ramps :87-94)."""

    start: Optional[float] = None
    end: Optional[float] = None
    width: Optional[float] = None

    def window(self, times) -> np.ndarray:
        times = np.asarray(times, float)
        amp = np.ones_like(times)
        if self.start is not None:
            m = times < self.start
            amp[m] *= np.exp(-0.5 * ((times[m] - self.start) / self.width) ** 2)
        if self.end is not None:
            m = times > self.end
            amp[m] *= np.exp(-0.5 * ((times[m] - self.end) / self.width) ** 2)
        return amp


class _Monitor(_Model):

    @property
    def geometry(self) -> Box:
        return Box(center=self.center, size=self.size)


_ALL_FIELDS = ("Ex", "Ey", "Ez", "Hx", "Hy", "Hz")


class _TimeMonitorMixin:

    def time_inds(self, tmesh) -> Tuple[int, int]:
        """ref monitor.py:187-214."""
        tmesh = np.asarray(tmesh)
        beg, end = 0, 0
        if tmesh.size == 0:
            return beg, end
        t_stop = self.stop
        if t_stop is None:
            end = int(tmesh.size)
            t_stop = tmesh[-1]
        else:
            tend = np.nonzero(tmesh <= t_stop)[0]
            if tend.size > 0:
                end = int(tend[-1] + 1)
        dt = 1e-20 if tmesh.size < 2 else tmesh[1] - tmesh[0]
        if np.abs(self.start - t_stop) < dt and self.start <= tmesh[-1]:
            beg = max(end - 1, 0)
        else:
            tbeg = np.nonzero(tmesh[:end] >= self.start)[0]
            beg = int(tbeg[0]) if tbeg.size > 0 else end
        return beg, end

    def num_steps(self, tmesh) -> int:
        """ref monitor.py:216-220."""
        b, e = self.time_inds(tmesh)
        return int((e - b) / self.interval)


@_register
@dataclass
class FieldMonitor(_Monitor):
    """Frequency-domain field monitor (ref monitor.py:363)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    name: str = "field"
    freqs: Tuple[float, ...] = ()
    fields: Tuple[str, ...] = _ALL_FIELDS
    interval_space: Tuple[int, int, int] = (1, 1, 1)
    colocate: bool = True
    apodization: ApodizationSpec = field(default_factory=ApodizationSpec)

    def frequency_range(self):
        return (min(self.freqs), max(self.freqs))


@_register
@dataclass
class FieldTimeMonitor(_Monitor, _TimeMonitorMixin):
    """Time-domain field monitor (ref monitor.py:403)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    name: str = "field_time"
    start: float = 0.0
    stop: Optional[float] = None
    interval: Optional[int] = None
    fields: Tuple[str, ...] = _ALL_FIELDS
    interval_space: Tuple[int, int, int] = (1, 1, 1)
    colocate: bool = True

    def __post_init__(self):
        if self.interval is None:      # ref monitor.py:151-170: None -> 1 (with a warning)
            self.interval = 1


@_register
@dataclass
class FluxMonitor(_Monitor):
    """Frequency-domain power flux through a plane or out of a box (ref monitor.py:569)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    name: str = "flux"
    freqs: Tuple[float, ...] = ()
    normal_dir: Optional[str] = None
    exclude_surfaces: Optional[Tuple[str, ...]] = None
    interval_space: Tuple[int, int, int] = (1, 1, 1)
    colocate: bool = True
    apodization: ApodizationSpec = field(default_factory=ApodizationSpec)

    def __post_init__(self):
        nz = sum(1 for s in self.size if s == 0)
        if nz == 1 and self.normal_dir is None:
            self.normal_dir = "+"      # ref monitor.py:237-247 (planar default)

    def frequency_range(self):
        return (min(self.freqs), max(self.freqs))


@_register
@dataclass
class FluxTimeMonitor(_Monitor, _TimeMonitorMixin):
    """Time-domain power flux (ref monitor.py:602)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    name: str = "flux_time"
    start: float = 0.0
    stop: Optional[float] = None
    interval: Optional[int] = None
    normal_dir: Optional[str] = None
    exclude_surfaces: Optional[Tuple[str, ...]] = None
    interval_space: Tuple[int, int, int] = (1, 1, 1)
    colocate: bool = True

    def __post_init__(self):
        if self.interval is None:
            self.interval = 1
        nz = sum(1 for s in self.size if s == 0)
        if nz == 1 and self.normal_dir is None:
            self.normal_dir = "+"


@_register
@dataclass
class ModeMonitor(_Monitor):
    """Modal decomposition on a plane (ref monitor.py:631)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    name: str = "mode"
    freqs: Tuple[float, ...] = ()
    mode_spec: ModeSpec = field(default_factory=ModeSpec)
    interval_space: Tuple[int, int, int] = (1, 1, 1)
    colocate: bool = False
    store_fields_direction: Optional[str] = None
    apodization: ApodizationSpec = field(default_factory=ApodizationSpec)

    def frequency_range(self):
        return (min(self.freqs), max(self.freqs))


@_register
@dataclass
class ModeSolverMonitor(ModeMonitor):
    """Stores the modes of its plane themselves (ref monitor.py:712-760): ModeSolverData with the six field
    components per frequency and mode index, computed by the mode solver the FDTD run uses
    (tidy3d_amd/plugins/mode.py)."""

    name: str = "mode_solver"
    colocate: bool = True
    direction: str = "+"


@_register
@dataclass
class PermittivityMonitor(_Monitor):
    """Diagonal of the complex relative permittivity at the Yee locations of E (ref monitor.py:447);
    never colocated (ref monitor.py:469-474)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    name: str = "eps"
    freqs: Tuple[float, ...] = ()
    interval_space: Tuple[int, int, int] = (1, 1, 1)
    colocate: bool = False
    apodization: ApodizationSpec = field(default_factory=ApodizationSpec)

    def frequency_range(self):
        return (min(self.freqs), max(self.freqs))


@_register
@dataclass
class FieldProjectionAngleMonitor(_Monitor):
    """Near-to-far projection to points (r, theta, phi) (ref monitor.py:930-1040); the near fields
    are recorded on the monitor's surfaces like a flux box and projected after the run
    (tidy3d_amd/projection.py; far-field approximation only)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    name: str = "proj"
    freqs: Tuple[float, ...] = ()
    theta: Tuple[float, ...] = ()
    phi: Tuple[float, ...] = ()
    proj_distance: float = 1e6
    normal_dir: Optional[str] = None
    exclude_surfaces: Optional[Tuple[str, ...]] = None
    custom_origin: Optional[Tuple[float, float, float]] = None
    far_field_approx: bool = True
    interval_space: Tuple[int, int, int] = (1, 1, 1)
    window_size: Tuple[float, float] = (0.0, 0.0)
    medium: Any = None
    colocate: bool = True
    apodization: ApodizationSpec = field(default_factory=ApodizationSpec)

    def __post_init__(self):
        self.theta = tuple(float(v) for v in np.atleast_1d(self.theta))
        self.phi = tuple(float(v) for v in np.atleast_1d(self.phi))

    def frequency_range(self):
        return (min(self.freqs), max(self.freqs))

    @property
    def local_origin(self):
        return tuple(self.center) if self.custom_origin is None else tuple(self.custom_origin)


@_register
@dataclass
class FieldProjectionCartesianMonitor(FieldProjectionAngleMonitor):
    """Projection to points (x, y) of a plane at ``proj_distance`` along ``proj_axis`` (ref monitor.py:1043)."""

    x: Tuple[float, ...] = ()
    y: Tuple[float, ...] = ()
    proj_axis: int = 2

    def __post_init__(self):
        self.x = tuple(float(v) for v in np.atleast_1d(self.x))
        self.y = tuple(float(v) for v in np.atleast_1d(self.y))


@_register
@dataclass
class FieldProjectionKSpaceMonitor(FieldProjectionAngleMonitor):
    """Projection to directions (ux, uy) around ``proj_axis`` (ref monitor.py:1173)."""

    ux: Tuple[float, ...] = ()
    uy: Tuple[float, ...] = ()
    proj_axis: int = 2

    def __post_init__(self):
        self.ux = tuple(float(v) for v in np.atleast_1d(self.ux))
        self.uy = tuple(float(v) for v in np.atleast_1d(self.uy))


@_register
@dataclass
class DiffractionMonitor(_Monitor):
    """Diffraction orders of a periodic structure through an infinite plane (ref monitor.py:1353-1407):
    the near fields of one period are recorded like a projection surface and decomposed into the
    allowed orders after the run (tidy3d_amd/projection.py ``diffraction``)."""

    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    name: str = "diffraction"
    interval_space: Tuple[int, int, int] = (1, 1, 1)
    colocate: bool = False
    freqs: Tuple[float, ...] = ()
    apodization: ApodizationSpec = field(default_factory=ApodizationSpec)
    normal_dir: str = "+"

    # what the projection core reads from a projection monitor (not part of the reference's fields)
    medium = None
    exclude_surfaces = None
    custom_origin = None

    def __post_init__(self):
        if [np.isinf(_to_float(v)) for v in self.size].count(True) != 2:
            raise SetupError("A 'DiffractionMonitor' must have a size of 'td.inf' along both transverse "
                             f"directions, given size={self.size}.")          # ref monitor.py:1390-1398
        if [float(_to_float(v)) == 0 for v in self.size].count(True) != 1:
            raise SetupError("A 'DiffractionMonitor' must be planar.")

    def frequency_range(self):
        return (min(self.freqs), max(self.freqs))

    @property
    def local_origin(self):
        return tuple(self.center)


@_register
@dataclass
class RunTimeSpec(_Model):
    """ref components/run_time_spec.py; evaluated in Simulation._run_time (simulation.py:3677)."""

    quality_factor: float = 1.0
    source_factor: float = 3.0


@_register
@dataclass
class Staircasing(_Model):
    """ref components/subpixel_spec.py (selection only; courant_ratio == 1)."""
    courant_ratio = 1.0


@_register
@dataclass
class SubpixelSpec(_Model):
    """ref components/subpixel_spec.py:117.  ``dielectric``: PolarizedAveraging (default) /
    VolumetricAveraging / Staircasing are honoured by the rasteriser (discretize._subpixel_average —
    the published methods; the reference's own code is server-side); ``metal`` / ``pec`` interfaces
    are staircased; ``courant_ratio`` follows :148."""

    dielectric: Any = None
    metal: Any = None
    pec: Any = None

    def courant_ratio(self, contain_pec_structures: bool) -> float:
        if contain_pec_structures and self.pec is not None:
            raw = getattr(self.pec, "raw", None)
            if isinstance(raw, dict) and raw.get("type") == "PECConformal":
                # PECConformal.courant_ratio = 1 - timestep_reduction (default 0.3)
                return 1.0 - float(raw.get("timestep_reduction", 0.3))
            return float(getattr(self.pec, "courant_ratio", 1.0))
        return 1.0


# --------------------------------------------------------------------------------------
# Simulation
# --------------------------------------------------------------------------------------

@_register
@dataclass
class Simulation(_Model):
    """The problem statement (ref components/simulation.py:1580; fields: base_sim/simulation.py
    :28-99, simulation.py:1683-2199).  Derived discretisation quantities live in
    ``tidy3d_amd.discretize`` (grid, dt, tmesh, nyquist_step, monitor index spans)."""

    size: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    run_time: Any = 1e-12
    medium: Any = field(default_factory=Medium)
    structures: Tuple[Structure, ...] = ()
    symmetry: Tuple[int, int, int] = (0, 0, 0)
    sources: Tuple[Any, ...] = ()
    boundary_spec: BoundarySpec = field(default_factory=BoundarySpec)
    monitors: Tuple[Any, ...] = ()
    grid_spec: GridSpec = field(default_factory=GridSpec)
    courant: float = 0.99
    normalize_index: Optional[int] = 0
    shutoff: float = 1e-5
    subpixel: Any = True
    lumped_elements: Tuple[Any, ...] = ()
    post_norm: Any = 1.0                # (ref simulation.py post_norm: a factor on the recorded fields, used by the adjoint pipeline; only 1 here)

    def __post_init__(self):
        self.size = tuple(float(s) for s in self.size)
        self.center = tuple(float(c) for c in self.center)
        self.structures = tuple(self.structures)
        self.lumped_elements = tuple(self.lumped_elements or ())
        self.sources = tuple(self.sources)
        self.monitors = tuple(self.monitors)
        names = [m.name for m in self.monitors if not isinstance(m, Unsupported)]
        if len(set(names)) != len(names):
            raise SetupError("Monitor names must be unique (ref base_sim/simulation.py:122).")
        if not (0 < self.courant <= 1):
            raise ValidationError("courant must be in (0, 1].")

    @classmethod
    def from_dict(cls, d: dict) -> "Simulation":
        sim = parse(d)
        if not isinstance(sim, Simulation):
            raise SetupError("dict does not describe a tidy3d Simulation")
        sim._source_dict = d          # the caller's own JSON form: written back verbatim to .hdf5 files
        return sim

    @classmethod
    def from_file(cls, fname: str) -> "Simulation":
        """.json, or .hdf5 / .h5 in the reference's layout (datasets included; ref base.py:364-420)."""
        import json
        if str(fname).endswith((".hdf5", ".h5")):
            from .hdf5io import load_simulation
            return load_simulation(fname)
        with open(fname) as f:
            return cls.from_dict(json.load(f))

    @property
    def geometry(self) -> Box:
        return Box(center=self.center, size=self.size)

    @property
    def complex_fields(self) -> bool:
        """ref simulation.py:4396-4411: complex time-stepping fields with Bloch boundaries."""
        return any(isinstance(e, BlochBoundary) for pair in self.boundary_spec.to_list for e in pair)

    @property
    def all_structures(self) -> Tuple[Structure, ...]:
        """The user's structures followed by the lumped elements as structures (ref simulation.py:1283-1289)."""
        extra = []
        for le in self.lumped_elements:
            if isinstance(le, Unsupported):
                le.fail()
            extra.append(le.to_structure())
        return tuple(self.structures) + tuple(extra)

    @property
    def mediums(self):
        """Distinct media, background first (ref scene.py:192)."""
        out = [self.medium]
        for s in self.structures:
            if not any(s.medium is m or s.medium == m for m in out):
                out.append(s.medium)
        return out

    def validate_pre_upload(self, source_required: bool = True):
        """ref simulation.py:3341-3361: a run needs at least one source."""
        if source_required and len(self.sources) == 0:
            raise SetupError("No sources in simulation.")
