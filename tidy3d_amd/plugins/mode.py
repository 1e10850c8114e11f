"""``ModeSolver`` façade (ref plugins/mode/mode_solver.py:97): the cross-section eigenmodes of a plane
of a Simulation, on the SAME Yee grid, permittivity raster (incl. sub-pixel averaging), symmetry walls,
mode-plane PML, bends and polarisation filter the FDTD run uses (tidy3d_amd/mode_solver.py — held to
the reference's ``compute_modes`` by direct comparison, tests/test_mode_solver.py).  For hosts where
tidy3d itself is not installed; with tidy3d present its own CPU ModeSolver works unchanged."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import numpy as np

from .. import schema as td
from ..data import DataArray
from ..discretize import discretize
from ..exceptions import SetupError
from ..modesource import colocated_mode, mode_profile


@dataclass
class ModeSolverData:
    """Mirror of tidy3d ModeSolverData: ``n_complex`` (f, mode_index) and the six field components
    (x, y, z, f, mode_index), modes normalised to unit power flux along the plane normal."""
    monitor: object
    n_complex: DataArray = None
    Ex: Optional[DataArray] = None
    Ey: Optional[DataArray] = None
    Ez: Optional[DataArray] = None
    Hx: Optional[DataArray] = None
    Hy: Optional[DataArray] = None
    Hz: Optional[DataArray] = None
    grid_expanded: Optional[Dict[str, np.ndarray]] = None

    @property
    def n_eff(self) -> DataArray:
        return DataArray(np.real(self.n_complex.values), self.n_complex.coords)

    @property
    def k_eff(self) -> DataArray:
        return DataArray(np.imag(self.n_complex.values), self.n_complex.coords)

    @property
    def field_components(self) -> Dict[str, DataArray]:
        return {k: getattr(self, k) for k in ("Ex", "Ey", "Ez", "Hx", "Hy", "Hz") if getattr(self, k) is not None}


class ModeSolver:
    def __init__(self, simulation, plane, mode_spec, freqs: Sequence[float], direction: str = "+", colocate: bool = True):
        self.simulation, self.plane, self.mode_spec = simulation, plane, mode_spec
        self.freqs = [float(f) for f in np.atleast_1d(freqs)]
        self.direction, self.colocate = direction, colocate
        zd = [a for a in range(3) if plane.size[a] == 0]
        if len(zd) != 1:
            raise SetupError("a mode plane needs exactly one zero-size dimension")
        self.normal_axis = zd[0]

    def solve(self, spec=None) -> ModeSolverData:
        """``spec``: an already discretised simulation (the FDTD run's own SolverSpec)."""
        sim = self.simulation
        if any(sim.symmetry):
            raise NotImplementedError("ModeSolver façade: run on the symmetry-free simulation")
        if spec is None:
            spec = discretize(sim, n_steps=1).spec
        p = self.normal_axis
        nm = int(self.mode_spec.num_modes)
        sign = 1 if self.direction == "+" else -1
        n_complex = np.zeros((len(self.freqs), nm), complex)
        names = "xyz"
        fields = None
        for i_f, f in enumerate(self.freqs):
            plane = mode_profile(spec, self.plane, self.mode_spec, f)
            u, v = plane.u, plane.v
            b = spec.boundaries
            ub = np.asarray(b[u][plane.lo[0]:plane.hi[0] + 1])
            vb = np.asarray(b[v][plane.lo[1]:plane.hi[1] + 1])
            n_complex[i_f] = plane.result.n_complex
            if fields is None:
                shape = [1, 1, 1]
                shape[u], shape[v] = len(ub) - 1, len(vb) - 1
                fields = {k: np.zeros(tuple(shape) + (len(self.freqs), nm), complex)
                          for k in ("Ex", "Ey", "Ez", "Hx", "Hy", "Hz")}
                coords = {names[u]: ub[:-1], names[v]: vb[:-1], names[p]: np.array([self.plane.center[p]])}
            r = plane.result
            for m in range(nm):
                if self.colocate:
                    tang = colocated_mode(plane, spec, m, sign, ub[:-1], vb[:-1])
                    comp = {"E" + names[u]: tang["Eu"], "E" + names[v]: tang["Ev"], "H" + names[u]: tang["Hu"],
                            "H" + names[v]: tang["Hv"], "E" + names[p]: r.Ew[:, :, m], "H" + names[p]: sign * r.Hw[:, :, m]}
                else:
                    comp = {"E" + names[u]: r.Eu[:, :, m], "E" + names[v]: r.Ev[:, :, m], "E" + names[p]: r.Ew[:, :, m],
                            "H" + names[u]: sign * r.Hu[:, :, m], "H" + names[v]: sign * r.Hv[:, :, m],
                            "H" + names[p]: r.Hw[:, :, m]}
                for k, arr in comp.items():
                    a3 = np.expand_dims(arr if (u < v) else arr.T, axis=p)       # (x, y, z) order
                    fields[k][..., i_f, m] = a3
        cf = {"f": np.asarray(self.freqs), "mode_index": np.arange(nm)}
        data = {k: DataArray(v, {**{d: coords[d] for d in "xyz"}, **cf}) for k, v in fields.items()}
        return ModeSolverData(monitor=self.plane, n_complex=DataArray(n_complex, cf), **data)

    data = property(lambda self: self.solve())
