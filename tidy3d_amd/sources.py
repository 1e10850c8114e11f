"""Source discretisation: tidy3d source objects -> SolverSpec source lists, plus the
frequency-domain normalisation spectrum of each source (ref sim_data.py:931-953)."""
from __future__ import annotations

import numpy as np

from . import schema as td
from .discretize import current_source
from .exceptions import Tidy3dNotImplementedError


def _spectrum_fn(src, tmesh, dt, cplx=False):
    """Normalisation spectrum: the DTFT (ref time.py:46-105) of the samples that are actually
    injected.  Currents enter the E-update at t_n + dt/2, so the time axis is shifted by dt/2;
    monitor DFTs use the true sample times as well, hence no spurious frequency-dependent phase
    between fields and source (CHANGELOG:1292)."""
    st = src.source_time

    def fn(freqs):
        return st.spectrum(tmesh + dt / 2, np.asarray(freqs, float), dt, complex_fields=cplx)
    return fn


def build_sources(disc, mt) -> None:
    sim, spec, tmesh = disc.sim, disc.spec, disc.tmesh
    disc.source_norm = []
    for src in sim.sources:
        if isinstance(src, td.Unsupported):
            src.fail()
        if isinstance(src, (td.PointDipole, td.UniformCurrentSource)):
            spec.sources.append(current_source(spec, mt, src, tmesh))
            disc.source_norm.append(_spectrum_fn(src, tmesh, spec.dt, spec.bloch is not None))
        elif isinstance(src, td.CustomCurrentSource):
            # current densities from a dataset, coordinates relative to the source centre (ref source.py:632-700)
            ds = td._need_data(src.current_dataset, "CustomCurrentSource")
            for name, arr in ds.field_components.items():
                f_sel = float(np.asarray(arr.coords["f"])[np.argmin(np.abs(np.asarray(arr.coords["f"], float)
                                                                             - src.source_time.freq0))]) if "f" in arr.dims else None

                def profile(x, y, z, arr=arr, f_sel=f_sel):
                    pts = {"x": x - src.center[0], "y": y - src.center[1], "z": z - src.center[2]}
                    a = arr if f_sel is None else arr.sel(f=f_sel)
                    return td.interp_dataset(a, pts, "linear")
                spec.sources.append(current_source(spec, mt, src, tmesh, polarization=name, profile=profile))
            disc.source_norm.append(_spectrum_fn(src, tmesh, spec.dt, spec.bloch is not None))
        elif isinstance(src, td.CustomFieldSource):
            from .planewave import build_custom_field_source
            disc.source_norm.append(build_custom_field_source(disc, mt, src))
        elif isinstance(src, (td.GaussianBeam, td.AstigmaticGaussianBeam)):
            from .planewave import build_gaussian_beam
            disc.source_norm.append(build_gaussian_beam(disc, mt, src))
        elif isinstance(src, (td.PlaneWave, td.TFSF)):
            from .planewave import build_planewave
            disc.source_norm.append(build_planewave(disc, mt, src))
        elif isinstance(src, td.ModeSource):
            from .modesource import build_mode_source
            disc.source_norm.append(build_mode_source(disc, mt, src))
        else:
            raise Tidy3dNotImplementedError(f"source type '{src.type}' is not supported")
