"""MI355X-native FDTD time-stepper behind the tidy3d API."""
from . import schema  # noqa: F401
from .exceptions import (SetupError, Tidy3dError, Tidy3dNotImplementedError,  # noqa: F401
                         ValidationError, DataError, SolverLibraryError)

__version__ = "0.1.0"


def run(simulation, task_name=None, folder_name="default", path=None, **kwargs):
    """Drop-in for ``tidy3d.web.run`` (ref web/api/webapi.py:49) — see tidy3d_amd.web.run."""
    from .web import run as _run
    return _run(simulation, task_name=task_name, folder_name=folder_name, path=path, **kwargs)


def load(path):
    """Read a SimulationData .hdf5 / .npz written by ``run(..., path=...)`` (or by the reference, for the data
    types the mirror has) — see tidy3d_amd.web.load."""
    from .web import load as _load
    return _load(path)
