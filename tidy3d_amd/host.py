"""ctypes binding of tidy3d_amd/libfdtd_host.so (include/fdtd_host.h): the rasteriser's whole-volume passes on a pool of native
threads.  Host code of the set-up phase — no GPU involved; `python -m tidy3d_amd.build` builds it beside libfdtd_hip.so.  Where the
library has not been built the NumPy statements it replaces run instead (tidy3d_amd/discretize.py; same results, tests/test_host_raster.py)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(HERE, "libfdtd_host.so")
SYMBOLS = ("fdtd_host_fill_u16", "fdtd_host_interface_nodes", "fdtd_host_interface_nodes_take", "fdtd_host_sample_media", "fdtd_host_free")
_lib = None
_tried = False


def threads() -> int:
    return max(1, min(32, os.cpu_count() or 1))


def load() -> Optional[C.CDLL]:
    """The helper library, or None where it has not been built ($TIDY3D_AMD_NO_HOST_LIB=1: the NumPy passes, for A/B and tests)."""
    global _lib, _tried
    if os.environ.get("TIDY3D_AMD_NO_HOST_LIB") == "1":
        return None
    if _tried:
        return _lib
    _tried = True
    if not os.path.exists(HOST_LIB):
        return None
    lib = C.CDLL(HOST_LIB)
    for s in SYMBOLS:
        getattr(lib, s)                       # (AttributeError: a stale build)
    lib.fdtd_host_fill_u16.argtypes = [C.c_void_p, C.c_int64, C.c_uint16, C.c_int]
    lib.fdtd_host_fill_u16.restype = None
    lib.fdtd_host_interface_nodes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.POINTER(C.c_void_p), C.c_int]
    lib.fdtd_host_interface_nodes.restype = C.c_int64
    lib.fdtd_host_interface_nodes_take.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.fdtd_host_interface_nodes_take.restype = None
    lib.fdtd_host_sample_media.argtypes = [C.c_int64, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + [C.c_uint16, C.c_void_p, C.c_int]
    lib.fdtd_host_sample_media.restype = C.c_int
    lib.fdtd_host_free.argtypes = [C.c_void_p]
    lib.fdtd_host_free.restype = None
    _lib = lib
    return lib


def fill_u16(out: np.ndarray, value: int) -> bool:
    lib = load()
    if lib is None or out.dtype != np.uint16 or not out.flags.c_contiguous:
        return False
    lib.fdtd_host_fill_u16(out.ctypes.data, out.size, int(value), threads())
    return True


def interface_nodes(m: np.ndarray, zflag: np.ndarray, plain: np.ndarray) -> Optional[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
    """(kk, jj, ii, ax_bits) of the interface nodes of m[nz][ny][nx] (include/fdtd_host.h), or None without the library."""
    lib = load()
    if lib is None or m.dtype != np.uint16 or not m.flags.c_contiguous:
        return None
    nz, ny, nx = m.shape
    zf = np.ascontiguousarray(zflag[:nz], np.uint8)
    pl = np.ascontiguousarray(plain, np.uint8)
    scan = C.c_void_p()
    n = lib.fdtd_host_interface_nodes(m.ctypes.data, nz, ny, nx, zf.ctypes.data, pl.ctypes.data, int(pl.size), C.byref(scan), threads())
    if n < 0:
        raise MemoryError("fdtd_host_interface_nodes failed")
    a = np.empty((3, n), np.int64)
    b = np.empty(n, np.uint8)
    lib.fdtd_host_interface_nodes_take(scan, a.ctypes.data, b.ctypes.data, threads())
    return a[0], a[1], a[2], b


class StructTable:
    """The structures of a simulation as fdtd_host_sample_media takes them; `ok` False where one of them is of a kind the native
    pass does not evaluate (PolySlab, slanted cylinders, groups, meshes ...): the caller keeps its NumPy pass then."""

    def __init__(self, structs, bounds):
        from . import schema as td
        n = len(structs)
        self.type = np.zeros(n, np.int32)
        self.par = np.zeros((n, 8))
        self.bounds = np.zeros((n, 6))
        self.mi = np.zeros(n, np.uint16)
        self.ok = load() is not None
        for g, ((geo, mi), (g0, g1)) in enumerate(zip(structs, bounds)):
            self.mi[g] = mi
            self.bounds[g, :3], self.bounds[g, 3:] = g0, g1
            if type(geo) is td.Box:
                self.type[g] = 0
                self.par[g, :3] = geo.center
                self.par[g, 3:6] = [s_ / 2 for s_ in geo.size]
            elif type(geo) is td.Sphere:
                self.type[g] = 1
                self.par[g, :3] = geo.center
                self.par[g, 3] = geo.radius ** 2
            elif type(geo) is td.Cylinder and np.isclose(geo.sidewall_angle, 0):
                self.type[g] = 2
                self.par[g, :3] = geo.center
                self.par[g, 3] = geo.radius
                self.par[g, 4] = geo._finite_length / 2
                self.par[g, 7] = geo.axis
            else:
                self.ok = False

    def sample(self, line: bool, lo, hi, which, background: int = 1) -> np.ndarray:
        """idx[n][8 | 64]: the media of the samples of n nodes with control volumes lo[3][n] .. hi[3][n]"""
        lib = load()
        lo = np.ascontiguousarray(np.stack(lo), float)
        hi = np.ascontiguousarray(np.stack(hi), float)
        n = lo.shape[1]
        wh = np.ascontiguousarray(which, np.uint8) if which is not None else np.zeros(max(n, 1), np.uint8)
        idx = np.empty((n, 8 if line else 64), np.uint16)
        rc = lib.fdtd_host_sample_media(n, int(bool(line)), lo.ctypes.data, hi.ctypes.data, wh.ctypes.data, len(self.type), self.type.ctypes.data,
                                        self.par.ctypes.data, self.bounds.ctypes.data, self.mi.ctypes.data, background, idx.ctypes.data, threads())
        if rc != 0:
            raise RuntimeError("fdtd_host_sample_media: unsupported structure")
        return idx
