"""The rasteriser's native host passes (include/fdtd_host.h, tidy3d_amd/libfdtd_host.so: interface-node scan, media of the sub-pixel
samples, threaded fills) against the NumPy statements they replace (tidy3d_amd/discretize.py keeps them; $TIDY3D_AMD_NO_HOST_LIB=1
selects them): the same material indices, node for node, and the same material table, entry for entry."""
import ctypes
import os
import re

import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import host
from tidy3d_amd.discretize import discretize

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_library_exports_every_symbol_of_the_header():
    assert os.path.exists(host.HOST_LIB), "python -m tidy3d_amd.build builds tidy3d_amd/libfdtd_host.so"
    text = open(os.path.join(ROOT, "include", "fdtd_host.h")).read()
    names = sorted(set(re.findall(r"\b(fdtd_host_\w+)\s*\(", text)))
    assert len(names) == 5 and set(names) == set(host.SYMBOLS), names
    lib = ctypes.CDLL(host.HOST_LIB)
    for n in names:
        getattr(lib, n)


def both(sim, monkeypatch, **kw):
    monkeypatch.setenv("TIDY3D_AMD_NO_HOST_LIB", "0")
    assert host.load() is not None
    a = discretize(sim, **kw).spec
    monkeypatch.setenv("TIDY3D_AMD_NO_HOST_LIB", "1")
    assert host.load() is None
    b = discretize(sim, **kw).spec
    monkeypatch.setenv("TIDY3D_AMD_NO_HOST_LIB", "0")
    assert a.shape == b.shape and len(a.media) == len(b.media)
    assert np.array_equal(a.mat_idx, b.mat_idx)
    for m, n in zip(a.media, b.media):
        assert (m.eps_inf, m.sigma, m.poles, m.pec, m.name) == (n.eps_inf, n.sigma, n.poles, n.pec, n.name)
    return a


def bodies(rng, n, kinds):
    out = []
    for q in range(n):
        c = tuple(rng.uniform(-0.5, 0.5, 3))
        kind = kinds[q % len(kinds)]
        if kind == "box":
            geo = td.Box(center=c, size=tuple(rng.uniform(0.15, 0.7, 3)))
        elif kind == "slab":
            size = [td.inf, td.inf, td.inf]
            size[int(rng.integers(3))] = float(rng.uniform(0.1, 0.4))
            geo = td.Box(center=c, size=tuple(size))
        elif kind == "sphere":
            geo = td.Sphere(center=c, radius=float(rng.uniform(0.1, 0.45)))
        elif kind == "cyl":
            geo = td.Cylinder(center=c, radius=float(rng.uniform(0.1, 0.4)), length=float(rng.uniform(0.1, 0.6)), axis=int(rng.integers(3)))
        elif kind == "cone":
            geo = td.Cylinder(center=c, radius=0.3, length=0.4, axis=2, sidewall_angle=0.2)
        else:
            geo = td.PolySlab(vertices=[(c[0] - 0.3, c[1] - 0.2), (c[0] + 0.25, c[1] - 0.25), (c[0] + 0.1, c[1] + 0.3)], slab_bounds=(c[2] - 0.15, c[2] + 0.2), axis=2)
        med = [td.Medium(permittivity=float(rng.uniform(1.5, 12.0))), td.Medium(permittivity=4.0, conductivity=0.5), td.PECMedium(),
               td.Lorentz(eps_inf=2.0, coeffs=[(1.0, 4e14, 1e13)])][int(rng.choice(4, p=[0.7, 0.1, 0.1, 0.1]))]
        out.append(td.Structure(geometry=geo, medium=med))
    return out


@pytest.mark.parametrize("seed,kinds,subpixel", [
    (0, ("box", "sphere", "cyl"), True), (1, ("slab", "sphere"), True), (2, ("cyl", "box", "slab"), True), (3, ("sphere",), True),
    (4, ("box", "sphere", "cyl"), td.parse({"type": "SubpixelSpec", "dielectric": {"type": "VolumetricAveraging"}})),
    (5, ("box", "poly", "sphere"), True), (6, ("cone", "box"), True), (7, ("box", "sphere", "cyl", "slab"), False)])
def test_native_passes_equal_the_numpy_passes(seed, kinds, subpixel, monkeypatch):
    """random overlapping bodies (dielectric, lossy, PEC, dispersive — the latter three keep the staircase and switch the averaging
    off around them) on a NON-uniform grid; structures the native sampler does not evaluate (PolySlab, slanted cylinder) leave the
    sampling to NumPy while the node scan stays native"""
    rng = np.random.default_rng(seed)
    grid = td.GridSpec(grid_x=td.UniformGrid(dl=0.031), grid_y=td.CustomGrid(dl=tuple(rng.uniform(0.02, 0.05, 40))), grid_z=td.UniformGrid(dl=0.043))
    sim = td.Simulation(size=(1.5, 1.4, 1.3), grid_spec=grid, run_time=1e-13, structures=bodies(rng, 7, kinds), subpixel=subpixel,
                        medium=td.Medium(permittivity=float(rng.choice([1.0, 2.25]))),
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    spec = both(sim, monkeypatch, n_steps=2)
    if subpixel is not False:
        assert any(m.name.startswith("subpixel_") for m in spec.media)


def test_interface_node_scan_on_a_random_volume():
    """the scan alone, on noise: every flagged plane, every kind of neighbourhood, plain and non-plain media, array edges"""
    rng = np.random.default_rng(11)
    nz, ny, nx = 9, 13, 17
    m = rng.integers(0, 5, size=(nz, ny, nx)).astype(np.uint16)
    m[3:6, 2:9, 4:12] = 2
    plain = np.array([0, 1, 1, 0, 1, 0, 0], np.uint8).astype(bool)
    zflag = rng.random(nz) < 0.7
    kk, jj, ii, bits = host.interface_nodes(m, zflag, plain)
    want = []
    for k in range(nz):
        if not zflag[k]:
            continue
        for j in range(ny):
            for i in range(nx):
                me, b, keep = m[k, j, i], 0, plain[m[k, j, i]]
                for ax, (dk, dj, di) in enumerate(((0, 0, 1), (0, 1, 0), (1, 0, 0))):
                    for s in (-1, 1):
                        k2, j2, i2 = k + s * dk, j + s * dj, i + s * di
                        if 0 <= k2 < nz and 0 <= j2 < ny and 0 <= i2 < nx and m[k2, j2, i2] != me:
                            b |= 1 << ax
                            keep = keep and plain[m[k2, j2, i2]]
                if b and keep:
                    want.append((k, j, i, b))
    assert len(want) > 30
    assert [tuple(int(v) for v in r) for r in zip(kk, jj, ii, bits)] == want


def test_threaded_fill():
    a = np.empty(1 << 20, np.uint16)
    assert host.fill_u16(a, 7) and (a == 7).all()
