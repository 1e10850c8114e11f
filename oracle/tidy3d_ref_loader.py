"""TEST INFRASTRUCTURE — import the reference package from /root/reference in this container.

``import tidy3d`` fails here (h5py, xarray, shapely, autograd, ... are not installed).  This
loader registers inert stand-ins for those third-party modules so that the reference's *own*
schema code (pydantic models, grid/dt/tmesh arithmetic, source waveforms, pole-residue
conversions, monitor index logic) and its CPU mode solver run unmodified.  It contains no solver
logic.  Used only by tests/golden/make_golden.py (to generate committed fixtures) and by tests
that are skipped when /root/reference is absent (it does not exist on the GPU box).

Caveat recorded with the fixtures: the container has numpy 2.x while tidy3d pins numpy<2; the one
place where that changes behaviour on our path is ``size + fp_eps`` with ``fp_eps`` a
``np.float32`` (ref simulation.py:1049): NEP-50 promotion makes it a no-op under numpy 2.
``load_tidy3d(numpy1_semantics=True)`` rebinds ``fp_eps`` to a Python float in the two modules
that use it that way, restoring the numpy-1 (intended) behaviour.
"""
from __future__ import annotations

import importlib
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "tidy3d"))


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return MagicMock(name=f"{self.__name__}.{name}")


def _stub(name):
    m = _Stub(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _mesher_shims():
    """The reference's mesher (components/grid/mesher.py) makes three third-party calls that are not
    installed here: a 2-D box, an R-tree query over boxes (shapely) and a bracketed root find
    (pyroots.Brentq).  Minimal functional stand-ins so that the REFERENCE'S OWN mesher code runs and
    AutoGrid goldens can be generated from it (tests/golden/make_golden.py); nothing here is mesher
    logic.  The root finder is scipy's brentq at the tolerance the reference passes."""
    class _Box:
        def __init__(self, minx, miny, maxx, maxy):
            self.bounds = (float(minx), float(miny), float(maxx), float(maxy))

    class _STRtree:
        def __init__(self, geoms):
            self.geoms = list(geoms)

        def query(self, geom):
            a = geom.bounds
            return [i for i, g in enumerate(self.geoms)
                    if not (g.bounds[2] < a[0] or g.bounds[0] > a[2] or g.bounds[3] < a[1] or g.bounds[1] > a[3])]

    class _Result:
        def __init__(self, x0, converged):
            self.x0, self.converged = x0, converged

    class _Brentq:
        def __init__(self, raise_on_fail=True, epsilon=1e-6, **kw):
            self.epsilon, self.raise_on_fail = epsilon, raise_on_fail

        def __call__(self, f, a, b):
            from scipy.optimize import brentq
            try:
                x0 = brentq(f, a, b, xtol=self.epsilon * 1e-3, rtol=8.9e-16, maxiter=500)
                return _Result(x0, True)
            except Exception:       # noqa: BLE001 - no sign change etc.: "not converged", as pyroots reports it
                if self.raise_on_fail:
                    raise
                return _Result(None, False)

    sys.modules["shapely.geometry"].box = _Box
    sys.modules["shapely.strtree"].STRtree = _STRtree
    sys.modules["shapely.errors"].ShapelyDeprecationWarning = type("ShapelyDeprecationWarning", (Warning,), {})
    sys.modules["pyroots"].Brentq = _Brentq


_loaded = None


def load_tidy3d(numpy1_semantics: bool = True):
    """Return the reference ``tidy3d`` module (schema + plugins.mode usable; data containers,
    hdf5 IO and the web client are stubs)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise ImportError("/root/reference is not available")
    import numpy
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    for n in ["h5py", "shapely", "shapely.geometry", "shapely.geometry.base", "shapely.validation",
              "shapely.strtree", "shapely.errors", "pyroots", "dask", "dask.array", "h5netcdf",
              "toml", "boto3", "responses", "jwt", "boto3.s3", "boto3.s3.transfer", "botocore",
              "botocore.exceptions", "botocore.client", "botocore.config", "boto3.session",
              "xarray.core", "xarray.core.utils", "xarray.core.indexes", "xarray.core.indexing",
              "xarray.core.variable", "xarray.core.types", "xarray.core.options"]:
        if n not in sys.modules:
            _stub(n)
    _mesher_shims()
    if "xarray" not in sys.modules:
        # a small FUNCTIONAL DataArray (oracle/mini_xarray.py): the reference's data containers and their
        # validators run on the output of tidy3d_amd.adapter.to_tidy3d (tests/test_adapter_reference.py)
        from oracle import mini_xarray
        mini_xarray.install()
    if "autograd" not in sys.modules:
        ag = types.ModuleType("autograd")
        ag.__path__ = []
        sys.modules["autograd"] = ag
        agnp = types.ModuleType("autograd.numpy")
        agnp.__path__ = []
        agnp.__getattr__ = lambda name: getattr(numpy, name)
        sys.modules["autograd.numpy"] = agnp
        ag.numpy = agnp
        for n in ["autograd.builtins", "autograd.extend", "autograd.tracer",
                  "autograd.differential_operators", "autograd.scipy", "autograd.scipy.signal",
                  "autograd.numpy.numpy_boxes", "autograd.scipy.special", "autograd.core",
                  "autograd.numpy.fft", "autograd.scipy.ndimage", "autograd.wrap_util",
                  "autograd.misc", "autograd.misc.optimizers", "autograd.numpy.linalg",
                  "autograd.test_util", "autograd.util"]:
            _stub(n)

        class Box:
            pass
        sys.modules["autograd.tracer"].Box = Box
        sys.modules["autograd.tracer"].isbox = lambda x: False
        sys.modules["autograd.tracer"].getval = lambda x: x
        sys.modules["autograd.extend"].Box = Box
        sys.modules["autograd.extend"].defvjp = lambda *a, **k: None
        sys.modules["autograd.extend"].primitive = lambda f: f
        sys.modules["autograd.builtins"].dict = dict
    td = importlib.import_module("tidy3d")
    try:
        td.config.logging_level = "ERROR"
    except Exception:
        pass
    if numpy1_semantics:
        import tidy3d.components.simulation as simmod
        simmod.fp_eps = float(simmod.fp_eps)
        # same in the mesher: ``max_scale - fp_eps`` (mesher.py:702,707) would be rounded to float32
        import tidy3d.components.grid.mesher as meshmod
        meshmod.fp_eps = float(meshmod.fp_eps)
    _loaded = td
    return td


def load_mode_solver():
    """The reference's CPU eigenmode arithmetic (ref plugins/mode/solver.py) — works with or
    without the full package stubs."""
    td = load_tidy3d()
    from tidy3d.plugins.mode import solver
    return td, solver
