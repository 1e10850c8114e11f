"""TEST INFRASTRUCTURE — a small FUNCTIONAL stand-in for ``xarray`` (not installed in this image).

The reference's data containers (``tidy3d.components.data``) subclass ``xarray.DataArray``; with an inert
stub they cannot be constructed, so ``tidy3d_amd.adapter.to_tidy3d`` — the code that turns a solve into a
genuine ``tidy3d.SimulationData`` — could never run here.  This module implements the slice of the
``DataArray`` interface those containers and their pydantic validators use (labelled dims and coords,
``transpose`` / ``sel`` / ``isel`` / ``interp`` on 1-D coordinate axes, elementwise arithmetic, attrs), on top
of plain NumPy.  It is installed as ``sys.modules['xarray']`` by ``oracle/tidy3d_ref_loader.py`` only when
the real package is absent.  Nothing in the product path imports it."""
from __future__ import annotations

import numpy as np


class Variable:
    pass


class _Coord:
    """One coordinate axis: behaves like a 1-D DataArray over its own dimension."""

    def __init__(self, name, values, attrs=None):
        self.name = name
        self.values = np.asarray(values)
        self.attrs = dict(attrs or {})
        self.dims = (name,)

    @property
    def data(self):
        return self.values

    @property
    def size(self):
        return self.values.size

    @property
    def shape(self):
        return self.values.shape

    @property
    def dtype(self):
        return self.values.dtype

    def __len__(self):
        return len(self.values)

    def __iter__(self):
        return iter(self.values)

    def __getitem__(self, i):
        return self.values[i]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def to_numpy(self):
        return self.values

    def tolist(self):
        return self.values.tolist()

    def item(self):
        return self.values.item()

    def copy(self):
        return _Coord(self.name, self.values.copy(), self.attrs)

    def __eq__(self, other):
        return self.values == np.asarray(other)

    def min(self):
        return self.values.min()

    def max(self):
        return self.values.max()


class _Coords:
    def __init__(self, owner):
        self._o = owner

    @property
    def dims(self):
        return self._o.dims

    def __getitem__(self, k):
        return self._o._mx_coords[k]

    def __contains__(self, k):
        return k in self._o._mx_coords

    def __iter__(self):
        return iter(self._o._mx_coords)

    def keys(self):
        return self._o._mx_coords.keys()

    def values(self):
        return self._o._mx_coords.values()

    def items(self):
        return self._o._mx_coords.items()

    def get(self, k, default=None):
        return self._o._mx_coords.get(k, default)

    def __len__(self):
        return len(self._o._mx_coords)

    def to_index(self):
        raise NotImplementedError


def _values_of(x):
    return x._mx_values if isinstance(x, DataArray) else (x.values if isinstance(x, _Coord) else x)


class DataArray:
    __slots__ = ("_mx_values", "_mx_dims", "_mx_coords", "attrs", "name")
    __array_priority__ = 50

    def __init__(self, data=None, coords=None, dims=None, name=None, attrs=None, **_):
        if isinstance(data, DataArray):
            self._mx_values = data._mx_values
            self._dims_set(data.dims if dims is None else tuple(dims))
            self._mx_coords = {k: c.copy() for k, c in data._mx_coords.items()}
            self.attrs = dict(data.attrs if attrs is None else attrs)
            self.name = data.name if name is None else name
            if coords is not None:
                for k, v in dict(coords).items():
                    self._mx_coords[k] = _Coord(k, _values_of(v), getattr(v, "attrs", None))
            return
        self._mx_values = np.asarray(data)
        if dims is None:
            if coords is not None and not isinstance(coords, dict):
                coords = dict(coords)
            dims = tuple(coords.keys()) if coords is not None else tuple(f"dim_{i}" for i in range(self._mx_values.ndim))
        self._dims_set(tuple(dims))
        if len(self.dims) != self._mx_values.ndim:
            raise ValueError(f"different number of dimensions on data ({self._mx_values.ndim}) and dims {self.dims}")
        self._mx_coords = {}
        for k, v in (dict(coords) if coords is not None else {}).items():
            c = _Coord(k, np.atleast_1d(np.asarray(_values_of(v))), getattr(v, "attrs", None))
            if k in self.dims and c.values.shape[0] != self._mx_values.shape[self.dims.index(k)]:
                raise ValueError(f"conflicting sizes for dimension '{k}': length {self._mx_values.shape[self.dims.index(k)]} "
                                 f"on the data but length {c.values.shape[0]} on coordinate '{k}'")
            self._mx_coords[k] = c
        self.attrs = dict(attrs or {})
        self.name = name

    def _dims_set(self, d):
        object.__setattr__(self, "_mx_dims", tuple(d))

    # ---- basic properties -------------------------------------------------------------
    @property
    def dims(self):
        return self._mx_dims

    @property
    def coords(self):
        return _Coords(self)

    @property
    def values(self):
        return self._mx_values

    @values.setter
    def values(self, v):
        self._mx_values = np.asarray(v)

    data = values

    @property
    def shape(self):
        return self._mx_values.shape

    @property
    def dtype(self):
        return self._mx_values.dtype

    @property
    def size(self):
        return self._mx_values.size

    @property
    def ndim(self):
        return self._mx_values.ndim

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    @property
    def real(self):
        return self._like(self._mx_values.real)

    @property
    def imag(self):
        return self._like(self._mx_values.imag)

    @property
    def T(self):
        return self.transpose(*reversed(self.dims))

    def __len__(self):
        return len(self._mx_values)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._mx_values, dtype=dtype)

    def __getattr__(self, k):            # coordinate access as attribute (da.x, da.f)
        if k.startswith("_") or k in DataArray.__slots__:
            raise AttributeError(k)
        c = object.__getattribute__(self, "_mx_coords")
        if k in c:
            return c[k]
        raise AttributeError(k)

    def to_numpy(self):
        return self._mx_values

    def item(self):
        return self._mx_values.item()

    def _like(self, values, dims=None, drop=()):
        dims = self.dims if dims is None else tuple(dims)
        out = object.__new__(type(self) if np.ndim(values) == len(dims) else DataArray)
        out._mx_values = np.asarray(values)
        out._dims_set(dims)
        out._mx_coords = {k: c.copy() for k, c in self._mx_coords.items() if k not in drop}
        out.attrs = dict(self.attrs)
        out.name = self.name
        return out

    def copy(self, deep=True, data=None):
        return self._like(self._mx_values.copy() if data is None else np.asarray(data))

    def astype(self, dtype):
        return self._like(self._mx_values.astype(dtype))

    def conj(self):
        return self._like(np.conj(self._mx_values))

    def transpose(self, *dims):
        if not dims:
            dims = tuple(reversed(self.dims))
        if tuple(dims) == self.dims:
            return self
        return self._like(np.transpose(self._mx_values, [self.dims.index(d) for d in dims]), dims)

    def assign_coords(self, coords=None, **kw):
        out = self._like(self._mx_values)
        for k, v in {**(coords or {}), **kw}.items():
            out._mx_coords[k] = _Coord(k, np.atleast_1d(np.asarray(_values_of(v))), getattr(v, "attrs", None))
        return out

    def assign_attrs(self, **kw):
        out = self._like(self._mx_values)
        out.attrs.update(kw)
        return out

    def squeeze(self, dim=None, drop=False):
        dims = [d for d, n in zip(self.dims, self.shape) if n == 1 and (dim is None or d == dim or d in np.atleast_1d(dim))]
        idx = tuple(0 if d in dims else slice(None) for d in self.dims)
        out = self._like(self._mx_values[idx], [d for d in self.dims if d not in dims], drop=dims if drop else ())
        return out

    # ---- selection ----------------------------------------------------------------------
    def isel(self, indexers=None, drop=False, **kw):
        kw = {**(indexers or {}), **kw}
        idx, new_dims, out_coords = [], [], {k: c.copy() for k, c in self._mx_coords.items()}
        for d in self.dims:
            i = kw.get(d, slice(None))
            if isinstance(i, (int, np.integer)):
                idx.append(int(i))
                if d in out_coords:
                    if drop:
                        del out_coords[d]
                    else:
                        out_coords[d] = _Coord(d, np.asarray(out_coords[d].values[int(i)]), out_coords[d].attrs)
            else:
                i = np.asarray(i) if not isinstance(i, slice) else i
                idx.append(i)
                new_dims.append(d)
                if d in out_coords:
                    out_coords[d] = _Coord(d, out_coords[d].values[i], out_coords[d].attrs)
        vals = self._mx_values
        for ax in range(len(idx) - 1, -1, -1):            # one axis at a time: outer indexing
            vals = np.take(vals, idx[ax], axis=ax) if not isinstance(idx[ax], slice) else vals[(slice(None),) * ax + (idx[ax],)]
        out = self._like(vals, new_dims)
        out._mx_coords = out_coords
        return out

    def sel(self, indexers=None, method=None, drop=False, **kw):
        kw = {**(indexers or {}), **kw}
        isel = {}
        for d, v in kw.items():
            c = self._mx_coords[d].values
            scalar = np.ndim(v) == 0
            want = np.atleast_1d(np.asarray(_values_of(v)))
            if method == "nearest":
                ii = np.array([int(np.argmin(np.abs(c - w))) for w in want])
            else:
                ii = []
                for w in want:
                    hit = np.nonzero(c == w)[0]
                    if hit.size == 0:
                        raise KeyError(f"{w!r} not found on coordinate '{d}'")
                    ii.append(int(hit[0]))
                ii = np.array(ii)
            isel[d] = int(ii[0]) if scalar else ii
        return self.isel(isel, drop=drop)

    def interp(self, coords=None, method="linear", kwargs=None, assume_sorted=False, **kw):
        """Separable interpolation along 1-D coordinate axes (linear / nearest); outside the range:
        ``fill_value`` of ``kwargs`` ("extrapolate" or a number), NaN by default — like xarray."""
        kw = {**(coords or {}), **kw}
        fill = (kwargs or {}).get("fill_value", np.nan)
        out = self
        for d, new in kw.items():
            ax = out.dims.index(d)
            x = out._mx_coords[d].values.astype(float)
            scalar = np.ndim(_values_of(new)) == 0
            xn = np.atleast_1d(np.asarray(_values_of(new), dtype=float))
            v = np.moveaxis(out._mx_values, ax, -1)
            if len(x) == 1:
                res = np.repeat(v, len(xn), axis=-1)
            elif method == "nearest":
                res = v[..., np.array([int(np.argmin(np.abs(x - w))) for w in xn])]
            else:
                order = np.argsort(x)
                xs, vs = x[order], v[..., order]
                i1 = np.clip(np.searchsorted(xs, xn, side="right"), 1, len(xs) - 1)
                i0 = i1 - 1
                t = (xn - xs[i0]) / (xs[i1] - xs[i0])
                res = vs[..., i0] * (1 - t) + vs[..., i1] * t
                if not (isinstance(fill, str) and fill == "extrapolate"):
                    outside = (xn < xs[0]) | (xn > xs[-1])
                    if outside.any():
                        res = np.where(outside, fill, res)
            res = np.moveaxis(res, -1, ax)
            nxt = out._like(res)
            nxt._mx_coords[d] = _Coord(d, xn, out._mx_coords[d].attrs)
            out = nxt.isel({d: 0}) if scalar else nxt
        return out

    # ---- reductions / arithmetic ----------------------------------------------------------
    def _reduce(self, fn, dim=None, **kw):
        if dim is None:
            return DataArray(fn(self._mx_values, **kw))
        dims = [dim] if isinstance(dim, str) else list(dim)
        axes = tuple(self.dims.index(d) for d in dims)
        return self._like(fn(self._mx_values, axis=axes, **kw), [d for d in self.dims if d not in dims], drop=dims)

    def sum(self, dim=None, **kw):
        return self._reduce(np.sum, dim)

    def mean(self, dim=None, **kw):
        return self._reduce(np.mean, dim)

    def max(self, dim=None, **kw):
        return self._reduce(np.max, dim)

    def min(self, dim=None, **kw):
        return self._reduce(np.min, dim)

    def integrate(self, coord):
        ax = self.dims.index(coord)
        x = self._mx_coords[coord].values
        vals = np.trapz(self._mx_values, x=x, axis=ax) if len(x) > 1 else np.zeros(np.delete(self.shape, ax), self.dtype)
        return self._like(vals, [d for d in self.dims if d != coord], drop=[coord])

    def _binary(self, other, fn, reflexive=False):
        if isinstance(other, DataArray):
            # align by dimension NAME (broadcast over the union of dims, no coordinate alignment)
            dims = list(self.dims) + [d for d in other.dims if d not in self.dims]

            def expand(a):
                v = a._mx_values
                src = [d for d in dims if d in a.dims]
                v = np.transpose(v, [a.dims.index(d) for d in src])
                return v.reshape([a.sizes[d] if d in a.dims else 1 for d in dims])
            a, b = expand(self), expand(other)
            out = self._like(fn(b, a) if reflexive else fn(a, b), dims)
            for k, c in other._mx_coords.items():
                out._mx_coords.setdefault(k, c.copy())
            return out
        o = _values_of(other)
        return self._like(fn(o, self._mx_values) if reflexive else fn(self._mx_values, o))

    def __add__(self, o): return self._binary(o, np.add)
    def __radd__(self, o): return self._binary(o, np.add, True)
    def __sub__(self, o): return self._binary(o, np.subtract)
    def __rsub__(self, o): return self._binary(o, np.subtract, True)
    def __mul__(self, o): return self._binary(o, np.multiply)
    def __rmul__(self, o): return self._binary(o, np.multiply, True)
    def __truediv__(self, o): return self._binary(o, np.true_divide)
    def __rtruediv__(self, o): return self._binary(o, np.true_divide, True)
    def __pow__(self, o): return self._binary(o, np.power)
    def __neg__(self): return self._like(-self._mx_values)
    def __abs__(self): return self._like(np.abs(self._mx_values))
    def __lt__(self, o): return self._binary(o, np.less)
    def __gt__(self, o): return self._binary(o, np.greater)
    def __le__(self, o): return self._binary(o, np.less_equal)
    def __ge__(self, o): return self._binary(o, np.greater_equal)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__":
            return NotImplemented
        das = [x for x in inputs if isinstance(x, DataArray)]
        res = ufunc(*[_values_of(x) for x in inputs], **kwargs)
        return das[0]._like(res) if np.shape(res) == das[0].shape else res

    def __getitem__(self, key):
        if isinstance(key, str):
            return self._mx_coords[key]
        if isinstance(key, dict):
            return self.isel(key)
        key = key if isinstance(key, tuple) else (key,)
        return self.isel({d: k for d, k in zip(self.dims, key)})

    def __setitem__(self, key, value):
        self._mx_values[key] = _values_of(value)

    def equals(self, other):
        return (isinstance(other, DataArray) and self.dims == other.dims and np.array_equal(self._mx_values, other._mx_values)
                and all(np.array_equal(self._mx_coords[k].values, other._mx_coords[k].values) for k in self._mx_coords))

    identical = equals

    def __eq__(self, other):
        return self._binary(other, np.equal)

    __hash__ = None

    def __repr__(self):
        return f"<mini-xarray {type(self).__name__} dims={self.dims} shape={self.shape} dtype={self.dtype}>"

    def to_hdf5(self, *a, **k):
        raise NotImplementedError("mini-xarray has no hdf5 IO (tidy3d_amd.hdf5io writes the reference's layout)")


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.data_vars = dict(data_vars or {})
        self.attrs = dict(attrs or {})

    def __getitem__(self, k):
        return self.data_vars[k]

    def __getattr__(self, k):
        dv = self.__dict__.get("data_vars", {})
        if k in dv:
            return dv[k]
        raise AttributeError(k)

    def keys(self):
        return self.data_vars.keys()

    def __iter__(self):
        return iter(self.data_vars)


def zeros_like(da):
    return da._like(np.zeros_like(da.values))


def ones_like(da):
    return da._like(np.ones_like(da.values))


def concat(objs, dim):
    objs = list(objs)
    ax = objs[0].dims.index(dim)
    out = objs[0]._like(np.concatenate([o.values for o in objs], axis=ax))
    out._mx_coords[dim] = _Coord(dim, np.concatenate([o._mx_coords[dim].values for o in objs]), objs[0]._mx_coords[dim].attrs)
    return out


def apply_ufunc(fn, *args, **kw):
    da = [a for a in args if isinstance(a, DataArray)][0]
    return da._like(fn(*[_values_of(a) for a in args]))


def install():
    """Register this module as ``xarray`` (plus the ``xarray.core.*`` names the reference imports)."""
    import sys
    import types
    from unittest.mock import MagicMock
    me = sys.modules[__name__]
    xr = types.ModuleType("xarray")
    xr.__path__ = []
    for k in ("DataArray", "Dataset", "Variable", "zeros_like", "ones_like", "concat", "apply_ufunc"):
        setattr(xr, k, getattr(me, k))
    xr.__getattr__ = lambda name: MagicMock(name=f"xarray.{name}")
    xr.__mini__ = True
    sys.modules["xarray"] = xr
    return xr
