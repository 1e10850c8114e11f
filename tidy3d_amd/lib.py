"""ctypes binding of libfdtd_hip.so (C ABI: include/fdtd_hip.h).

The library is the product: there is no CPU fallback.  ``load_library()`` raises
``SolverLibraryError`` when the shared object is missing or cannot be loaded, and
``FdtdLib.check`` turns every negative status into an exception carrying
``fdtd_last_error``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .exceptions import SolverLibraryError

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libfdtd_hip.so")



def _prefer_hw_queues() -> None:
    """The solver overlaps work on TWO streams per engine (boundary chunks + exchange / interior sweep; the edge and
    interior launches of a CPML step).  The HIP runtime multiplexes a process's streams onto $GPU_MAX_HW_QUEUES hardware
    queues (default 4); once an engine's two streams share one, their launches serialise (measured: a 64-plane z-slab step
    0.83 ms instead of 0.28 ms with six more streams alive in the process, 0.28 ms again with 8 queues: profiles/
    r04u_probe_hw_queues.jsonl).  The library does not rely on this variable: every engine MEASURES whether its two streams
    overlap before the first run that uses both, tries fresh streams if not, and otherwise falls back to one stream and
    says so (``FdtdStats.stream_overlap``, include/fdtd_hip.h).  Asking the runtime for 8 queues merely makes the good
    case the common one.  It is done when the product library is loaded — not at import — only if the user has not chosen
    a value, and it is without effect when the HIP runtime of this process is already initialised."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


# every symbol include/fdtd_hip.h declares (tests check the library exports all of them)
SYMBOLS = (
    "fdtd_last_error", "fdtd_device_count", "fdtd_create", "fdtd_destroy", "fdtd_set_steps",
    "fdtd_set_media", "fdtd_set_material", "fdtd_set_material16", "fdtd_set_pml", "fdtd_set_absorber", "fdtd_add_ade",
    "fdtd_add_point_source", "fdtd_add_tfsf", "fdtd_add_monitor", "fdtd_get_monitor",
    "fdtd_set_field", "fdtd_get_field", "fdtd_set_shutoff", "fdtd_comm_unique_id",
    "fdtd_comm_init", "fdtd_run", "fdtd_run_bloch", "fdtd_get_stats", "fdtd_reset", "fdtd_set_option",
    "fdtd_far_field", "fdtd_set_mirror_plus", "fdtd_add_aniso",
)

BC_PEC, BC_PMC, BC_PERIODIC, BC_NEIGHBOR = 0, 1, 2, 3
MON_TIME, MON_DFT = 0, 1
VARIANT_AUTO, VARIANT_SIMPLE, VARIANT_ZMARCH, VARIANT_FUSED = 0, 1, 2, 3
FLAG_TIME_KERNELS = 1
OPT_FLAGS, OPT_VARIANT, OPT_ZCHUNK, OPT_ROWS, OPT_XCD_REMAP, OPT_FUSED_LB, OPT_PML_FUSED, OPT_BND_PLANES, OPT_AUTOTUNE, OPT_PML_SPLIT, OPT_LDS_PAD, OPT_MEM_HINTS, OPT_PLACEMENT_TRIES, OPT_TBLOCK, OPT_EDGE_ZCHUNK, OPT_GRAPH, OPT_TWOSTEP, OPT_SHELL_PAIRS, OPT_STRIP, OPT_SHELL2, OPT_SHELL2_SHAPE, OPT_DEBUG_SYNC, OPT_TILE_SPLIT, OPT_DISP, OPT_WHATIF, OPT_SRC_PAGED, OPT_SLAB_BOXES_FIRST = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26
# FdtdStats.fused2_off_reason (include/fdtd_hip.h FDTD_F2_OFF_*)
F2_OFF_REASONS = {0: "", 1: "switched off", 2: "grid too small", 3: "z-slab rank", 4: "CPML (shell pairs not possible)",
                  5: "dispersive media not confined to a few planes along z", 6: "TFSF box while it injects", 7: "Bloch / PMC-plus faces (or rows not a multiple of 4 cells)", 8: "magnetic sources with absorber layers",
                  9: "magnetic source node on a tile seam", 10: "too many source nodes", 11: "two-pass kernels",
                  12: "CPML shell too large a part of the grid", 13: "no device memory for the third field set"}


class FdtdConfig(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("bc", C.c_int32 * 6), ("device", C.c_int32), ("variant", C.c_int32),
                ("flags", C.c_int32), ("z_chunk", C.c_int32), ("ch", C.c_float),
                ("reserved", C.c_int32 * 6)]


class FdtdStats(C.Structure):
    _fields_ = [("steps_done", C.c_int64), ("diverged", C.c_int32), ("stopped_early", C.c_int32),
                ("field_decay", C.c_double), ("run_ms", C.c_double), ("h_kernel_ms", C.c_double),
                ("e_kernel_ms", C.c_double), ("h_kernel_launches", C.c_int64),
                ("e_kernel_launches", C.c_int64), ("device_bytes", C.c_int64),
                ("fused_kernel_ms", C.c_double), ("fused_kernel_launches", C.c_int64),
                ("tile_rows", C.c_int32), ("tile_zchunk", C.c_int32),
                ("tile_order", C.c_int32), ("placement", C.c_int32),
                ("placement_ms_first", C.c_float), ("placement_ms_kept", C.c_float),
                ("stream_overlap", C.c_int32), ("stream_retries", C.c_int32),
                ("comm_ranks", C.c_int32), ("comm_rank", C.c_int32),
                ("two_step_pairs", C.c_int64), ("tblock_planes", C.c_int32), ("reserved0", C.c_int32),
                ("graph_pairs", C.c_int64), ("fused2_pairs", C.c_int64), ("fused2_shape", C.c_int64),
                ("shell_pairs", C.c_int64), ("shell_kernel_ms", C.c_double), ("shell_kernel_launches", C.c_int64),
                ("shell2_pairs", C.c_int64),
                ("fused2_off_reason", C.c_int32), ("struct_bytes", C.c_int32), ("disp_pairs", C.c_int64),
                ("single_step_reason", C.c_int32), ("src_paged_pairs", C.c_int32),
                ("seam_kernel_ms", C.c_double), ("seam_kernel_launches", C.c_int64)]


PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_int64, C.c_double, C.c_double, C.c_void_p)


class FdtdLib:
    def __init__(self, path: str):
        self.path = path
        try:
            self.dll = C.CDLL(path)      # RTLD_LOCAL: never interpose another library's symbols
        except OSError as e:
            raise SolverLibraryError(f"cannot load the HIP solver library '{path}': {e}") from e
        missing = [s for s in SYMBOLS if not hasattr(self.dll, s)]
        if missing:
            raise SolverLibraryError(f"'{path}' does not export: {', '.join(missing)}")
        d = self.dll
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        d.fdtd_last_error.restype = C.c_char_p
        d.fdtd_last_error.argtypes = [vp]
        d.fdtd_device_count.restype = C.c_int
        d.fdtd_create.argtypes = [C.POINTER(FdtdConfig), C.POINTER(vp)]
        d.fdtd_destroy.argtypes = [vp]
        d.fdtd_destroy.restype = None
        d.fdtd_set_steps.argtypes = [vp, C.c_int, vp, vp, C.c_int]
        d.fdtd_set_media.argtypes = [vp, vp, vp, C.c_int]
        d.fdtd_set_material.argtypes = [vp, vp, C.c_size_t]
        d.fdtd_set_material16.argtypes = [vp, vp, C.c_size_t]
        d.fdtd_set_pml.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int]
        d.fdtd_set_absorber.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
        d.fdtd_set_mirror_plus.argtypes = [vp, C.c_int, C.c_int]
        d.fdtd_add_ade.argtypes = [vp, C.c_int, i64, vp, C.c_int, vp, vp, f32]
        d.fdtd_add_aniso.argtypes = [vp, C.c_int, i64, vp, vp, vp, vp]
        d.fdtd_add_point_source.argtypes = [vp, i64, vp, vp, vp, vp, i64, vp, vp]
        d.fdtd_add_tfsf.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, i64, vp,
                                    i64, vp, vp, vp, vp, i64, vp, vp, vp, vp]
        d.fdtd_add_monitor.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, i64, vp, C.c_int, vp, vp]
        d.fdtd_get_monitor.argtypes = [vp, C.c_int, vp, C.c_size_t]
        d.fdtd_set_field.argtypes = [vp, C.c_int, vp, C.c_size_t]
        d.fdtd_get_field.argtypes = [vp, C.c_int, vp, C.c_size_t]
        d.fdtd_set_shutoff.argtypes = [vp, C.c_int, C.c_double, i64]
        d.fdtd_comm_unique_id.argtypes = [vp]
        d.fdtd_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
        d.fdtd_run.argtypes = [vp, i64, PROGRESS_FN, vp]
        d.fdtd_run_bloch.argtypes = [vp, vp, i64, vp, vp, PROGRESS_FN, vp]
        d.fdtd_get_stats.argtypes = [vp, C.POINTER(FdtdStats)]
        d.fdtd_reset.argtypes = [vp]
        d.fdtd_set_option.argtypes = [vp, C.c_int, C.c_int]
        f64 = C.c_double
        d.fdtd_far_field.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, f64, f64, f64, C.c_int, vp, vp, vp, vp]

    def far_field(self, u, v, wu, wv, currents, w0: float, k: complex, r_u, r_v, r_w, device: int = 0):
        """Surface integrals of the near -> far projection on the device (``fdtd_far_field``): ``currents`` [4, n_u, n_v]
        complex (J_u, J_v, M_u, M_v), directions as cosines along u, v and the surface normal -> complex [n_dir, 4]."""
        import numpy as np
        u, v, wu, wv = (np.ascontiguousarray(a, dtype=np.float64) for a in (u, v, wu, wv))
        cur = np.ascontiguousarray(currents, dtype=np.complex128)
        assert cur.shape == (4, u.size, v.size), cur.shape
        r_u, r_v, r_w = (np.ascontiguousarray(a, dtype=np.float64) for a in (r_u, r_v, r_w))
        out = np.empty((r_u.size, 4), dtype=np.complex128)
        p = lambda a: a.ctypes.data_as(C.c_void_p)           # noqa: E731
        self.check(self.dll.fdtd_far_field(int(device), int(u.size), int(v.size), p(u), p(v), p(wu), p(wv), p(cur), float(w0),
                                           float(np.real(k)), float(np.imag(k)), int(r_u.size), p(r_u), p(r_v), p(r_w), p(out)),
                   None, "fdtd_far_field")
        return out

    def error(self, handle) -> str:
        msg = self.dll.fdtd_last_error(handle)
        return msg.decode("utf-8", "replace") if msg else ""

    def check(self, status: int, handle=None, what: str = ""):
        if status < 0:
            raise SolverLibraryError(f"{what or 'libfdtd_hip'} failed: {self.error(handle)}")
        return status


_cached: Optional[FdtdLib] = None


def load_library(path: Optional[str] = None) -> FdtdLib:
    """Load the HIP library (in-tree build).  Fails loudly — never substitutes a CPU path."""
    global _cached
    if path is None and os.environ.get("TIDY3D_AMD_LIBRARY"):
        path = os.environ["TIDY3D_AMD_LIBRARY"]          # another build of the same library (A/B of compiler flags)
        if _cached is not None and getattr(_cached, "_path", None) == path:
            return _cached
        _prefer_hw_queues()
        _cached = FdtdLib(path)         # (checks that every symbol of include/fdtd_hip.h is exported; HipEngine.stats checks the struct layout)
        _cached._path = path
        import sys
        print(f"tidy3d_amd: solver library overridden by $TIDY3D_AMD_LIBRARY: {path}", file=sys.stderr)
        return _cached
    if path is None:
        if _cached is not None:
            return _cached
        if not os.path.exists(DEFAULT_LIB):
            raise SolverLibraryError(
                f"{DEFAULT_LIB} not found: build it with `python -m tidy3d_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback.")
        _prefer_hw_queues()
        _cached = FdtdLib(DEFAULT_LIB)
        return _cached
    return FdtdLib(path)
