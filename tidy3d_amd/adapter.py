"""Bridge to the real ``tidy3d`` package (lazy import; optional).

Input side:  a ``tidy3d.Simulation`` enters through its own JSON form
(``Simulation.json()``, ref components/base.py:364-420), i.e. the same dict layout as the
fixture tests/sims/simulation_sample.json, and is parsed by ``tidy3d_amd.schema.parse``.

Output side: ``to_tidy3d`` rebuilds genuine ``tidy3d.SimulationData`` /
``FieldData`` / ``FieldTimeData`` / ``FluxData`` objects from the mirror containers, following
the construction in the reference's fake backend line by line in *structure* (ref
tests/utils.py:880-1035: ``symmetry=(0,0,0)``, ``symmetry_center=simulation.center``,
``grid_expanded=simulation.discretize_monitor(monitor)``, one ``ScalarField*DataArray`` per
component with coords (x, y, z, f|t)), so everything downstream of ``web.run`` (plugins,
``FieldData.dot``, ``.flux``, ``to_file``) works on our results unchanged.

tidy3d itself is not installed in the build container nor on the GPU box (h5py, xarray, shapely,
autograd missing).  tests/test_adapter_reference.py runs this module against the reference checkout
(/root/reference) with a functional xarray stand-in: the reference's own container classes and their
validators accept what ``to_tidy3d`` builds, for every monitor-data type converted below.
"""
from __future__ import annotations

import json

from . import schema as mirror
from .data import FieldData, FieldTimeData, FluxData, FluxTimeData, SimulationData
from .exceptions import Tidy3dNotImplementedError


def tidy3d_available() -> bool:
    try:
        import tidy3d  # noqa: F401
        import xarray  # noqa: F401
        return True
    except Exception:
        return False


def from_tidy3d(simulation) -> mirror.Simulation:
    """tidy3d.Simulation (or its dict / JSON string) -> mirror Simulation."""
    if isinstance(simulation, mirror.Simulation):
        return simulation
    if isinstance(simulation, str):
        return mirror.Simulation.from_dict(json.loads(simulation))
    if isinstance(simulation, dict):
        return mirror.Simulation.from_dict(simulation)
    return mirror.Simulation.from_dict(json.loads(simulation.json()))


def to_tidy3d(sim_data: SimulationData, td_simulation=None):
    """Mirror SimulationData -> real tidy3d.SimulationData (needs an importable tidy3d)."""
    try:
        import tidy3d as td
    except Exception as e:
        raise Tidy3dNotImplementedError(
            "returning tidy3d.SimulationData needs an importable tidy3d package "
            f"({type(e).__name__}: {e}); call run(..., return_tidy3d=False) for the mirror "
            "containers.") from e
    if td_simulation is None:
        td_simulation = td.Simulation.parse_obj(sim_data.simulation.dict())
    by_name = {m.name: m for m in td_simulation.monitors}
    out = []
    for d in sim_data.data:
        mon = by_name[d.monitor.name]
        if isinstance(d, (FieldData, FieldTimeData)):
            is_time = isinstance(d, FieldTimeData)
            arr_cls = td.ScalarFieldTimeDataArray if is_time else td.ScalarFieldDataArray
            cls = td.FieldTimeData if is_time else td.FieldData
            comps = {k: arr_cls(v.values, coords={dim: v.coords[dim] for dim in v.dims})
                     for k, v in d.field_components.items()}
            out.append(cls(monitor=mon, symmetry=(0, 0, 0), symmetry_center=td_simulation.center,
                           grid_expanded=td_simulation.discretize_monitor(mon), **comps))
        elif isinstance(d, FluxData):
            out.append(td.FluxData(monitor=mon, flux=td.FluxDataArray(
                d.flux.values, coords={"f": d.flux.coords["f"]})))
        elif isinstance(d, FluxTimeData):
            out.append(td.FluxTimeData(monitor=mon, flux=td.FluxTimeDataArray(
                d.flux.values, coords={"t": d.flux.coords["t"]})))
        elif type(d).__name__ == "PermittivityData":
            comps = {k: td.ScalarFieldDataArray(v.values, coords={dim: v.coords[dim] for dim in v.dims})
                     for k, v in d.field_components.items()}
            out.append(td.PermittivityData(monitor=mon, symmetry=(0, 0, 0), symmetry_center=td_simulation.center,
                                           grid_expanded=td_simulation.discretize_monitor(mon), **comps))
        elif type(d).__name__ == "ModeData":
            out.append(td.ModeData(
                monitor=mon,
                amps=td.ModeAmpsDataArray(d.amps.values, coords={dim: d.amps.coords[dim] for dim in d.amps.dims}),
                n_complex=td.ModeIndexDataArray(d.n_complex.values,
                                                coords={dim: d.n_complex.coords[dim] for dim in d.n_complex.dims})))
        elif type(d).__name__ in ("FieldProjectionAngleData", "FieldProjectionCartesianData", "FieldProjectionKSpaceData"):
            # the three projection containers differ in their coordinates only (r, theta, phi | x, y, z | ux, uy, r)
            kind = type(d).__name__[len("FieldProjection"):-len("Data")]
            arr_cls = getattr(td, f"FieldProjection{kind}DataArray")
            comps = {k: arr_cls(v.values, coords={dim: v.coords[dim] for dim in v.dims})
                     for k, v in d.field_components.items()}
            out.append(getattr(td, type(d).__name__)(monitor=mon, projection_surfaces=mon.projection_surfaces,
                                                     medium=mon.medium or td_simulation.medium, **comps))
        elif type(d).__name__ == "ModeSolverData":
            comps = {k: td.ScalarModeFieldDataArray(v.values, coords={dim: v.coords[dim] for dim in v.dims})
                     for k, v in d.field_components.items()}
            out.append(td.ModeSolverData(
                monitor=mon, symmetry=(0, 0, 0), symmetry_center=td_simulation.center,
                grid_expanded=td_simulation.discretize_monitor(mon),
                n_complex=td.ModeIndexDataArray(d.n_complex.values,
                                                coords={dim: d.n_complex.coords[dim] for dim in d.n_complex.dims}), **comps))
        elif type(d).__name__ == "DiffractionData":
            comps = {k: td.DiffractionDataArray(v.values, coords={dim: v.coords[dim] for dim in v.dims})
                     for k, v in d.field_components.items()}
            out.append(td.DiffractionData(monitor=mon, sim_size=tuple(d.sim_size), bloch_vecs=tuple(d.bloch_vecs),
                                          medium=td_simulation.medium if d.structure_index < 0
                                          else td_simulation.structures[d.structure_index].medium, **comps))
        else:
            raise Tidy3dNotImplementedError(f"no tidy3d conversion for {type(d).__name__}")
    return td.SimulationData(simulation=td_simulation, data=tuple(out), log=sim_data.log,
                             diverged=bool(sim_data.diverged))


def install(td_module=None):
    """Monkey-patch ``tidy3d.web.run`` with the local solver — the same seam the reference's
    tests use for ``run_emulated`` (ref tests/test_plugins/test_adjoint.py:95)."""
    if td_module is None:
        import tidy3d as td_module
    from .web import Batch, Job, run
    td_module.web.run = run
    # the containers the plugins and user scripts reach for (ref web/api/container.py)
    td_module.web.Job, td_module.web.Batch = Job, Batch
    return run
