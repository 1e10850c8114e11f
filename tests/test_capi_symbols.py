"""The C-ABI shared library loads (no GPU needed) and exports every symbol include/fdtd_hip.h
declares; the ctypes structures mirror the C ones."""
import ctypes
import os
import re

from tidy3d_amd import lib as L

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    src = open(os.path.join(ROOT, "include", "fdtd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fdtd_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert set(_declared()) == set(L.SYMBOLS)


def test_product_library_exports_every_symbol():
    from tidy3d_amd import build
    path = build.build()                      # hipcc cross-compiles for gfx950 without a GPU
    dll = ctypes.CDLL(path)
    for s in _declared():
        assert hasattr(dll, s), s
    assert L.load_library().path == path


def test_struct_layouts():
    assert ctypes.sizeof(L.FdtdConfig) == 4 * (3 + 6 + 4) + 4 + 4 * 6
    assert ctypes.sizeof(L.FdtdStats) == 8 + 4 + 4 + 8 * 4 + 8 * 3 + 8 * 2 + 4 * 6 + 4 * 4 + 8 + 4 * 2 + 8 + 8 + 8 + 8 * 3 + 8 + 4 * 2 + 8 + 4 * 2 + 8 * 2


def test_create_rejects_bad_config_without_touching_a_gpu(emu_lib):
    cfg = L.FdtdConfig()
    cfg.nx, cfg.ny, cfg.nz = 0, 4, 4
    h = ctypes.c_void_p()
    assert emu_lib.dll.fdtd_create(ctypes.byref(cfg), ctypes.byref(h)) < 0
    assert "bad grid" in emu_lib.error(None)
    cfg.nx = 4
    cfg.bc[1] = L.BC_PMC
    assert emu_lib.dll.fdtd_create(ctypes.byref(cfg), ctypes.byref(h)) < 0
    assert "PMC" in emu_lib.error(None)
