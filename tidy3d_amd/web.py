"""``run()``: local, MI355X-native drop-in for ``tidy3d.web.run``.

Mirrors the signature and behaviour of reference web/api/webapi.py:49-155 (via
web/api/autograd/autograd.py:86): takes the Simulation positionally, tolerates and ignores the
cloud-only keyword arguments, validates (``validate_pre_upload``, simulation.py:3341), runs the
solve — here on the local GPU through ``libfdtd_hip.so`` instead of upload/start/monitor/
download — and returns a ``SimulationData``; the post-run warnings of
``Tidy3dStubData.postprocess`` (web/api/tidy3d_stub.py:219-233) are replicated.  The test-suite
seam of the reference is ``monkeypatch.setattr(td.web, "run", run_emulated)``
(tests/utils.py:880, tests/test_plugins/test_adjoint.py:95); this function plugs into the same
seam.
"""
from __future__ import annotations

import json
import logging
import os
import time
from typing import Optional

import numpy as np

from . import schema as td
from .data import SimulationData, assemble
from .discretize import discretize
from .exceptions import SetupError, SolverLibraryError

log = logging.getLogger("tidy3d_amd")


def _as_mirror(simulation):
    """Accept a mirror Simulation, a real tidy3d.Simulation (anything with .json()/.dict()), a
    dict in tidy3d's JSON form, or a path to a .json file."""
    if isinstance(simulation, td.Simulation):
        return simulation, False
    if isinstance(simulation, str):
        return td.Simulation.from_file(simulation), False
    if isinstance(simulation, dict):
        return td.Simulation.from_dict(simulation), False
    if hasattr(simulation, "json") and callable(simulation.json):
        text = simulation.json()
        if 'DataArray"' in text and hasattr(simulation, "to_file"):
            # dataset-defined objects (custom media / sources, triangle meshes): the JSON form holds only
            # placeholders, the data go through the reference's own .hdf5 writer (ref base.py:364-420)
            import os
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                fname = os.path.join(tmp, "simulation.hdf5")
                simulation.to_file(fname)
                return td.Simulation.from_file(fname), True
        return td.Simulation.from_dict(json.loads(text)), True
    raise SetupError(f"cannot interpret {type(simulation)!r} as a tidy3d Simulation")


def schedule_line(stats, n_cells: float, solve_s: float) -> str:
    """Which path the run took: the schedule, the step pairs taken of the steps run, why the rest (or all)
    went out as single steps, and the rate — a user cannot otherwise tell which of 190 / 140 / 125 / 95 Gcells/s a simulation gets."""
    from .lib import F2_OFF_REASONS
    steps = int(stats.steps_done)
    pairs, shell, shell2 = int(stats.fused2_pairs), int(stats.shell_pairs), int(stats.shell2_pairs)
    disp = int(getattr(stats, "disp_pairs", 0))
    rate = n_cells * steps / max(solve_s, 1e-9) / 1e9
    if not pairs:
        why = int(stats.fused2_off_reason)
        reason = F2_OFF_REASONS.get(why, f"reason {why}") if why else "fewer than two steps"
        return f"Schedule: one time step per sweep ({reason}); {rate:.1f} Gcells/s."
    if shell2:
        form = "two-step sweep over the bulk, the CPML shell two steps per sweep beside it"
    elif shell:
        form = "two-step sweep over the bulk, the CPML / periodic shell by single steps beside it"
    else:
        form = "two-step sweep over the whole grid"
    parts = [f"Schedule: two time steps per sweep ({form}) for {2 * pairs} of {steps} steps"]
    if disp:
        parts.append(f"dispersive cells advanced inside {disp} of the {pairs} pairs")
    single = steps - 2 * pairs
    if single > 0:
        why = int(getattr(stats, "single_step_reason", 0))
        parts.append(f"{single} single steps" + (f" ({F2_OFF_REASONS.get(why, f'reason {why}')})" if why and single > 2 else ""))
    return "; ".join(parts) + f"; {rate:.1f} Gcells/s."


def _log_line(step: int, n_steps: int, t: float, decay: float) -> str:
    """Format pinned by ref tests/test_data/test_sim_data.py:69,201-204."""
    perc = int(100 * step / max(n_steps, 1))
    mant = f"{decay:.3e}"
    return f"- Time step {step:6d} / time {t:.2e}s ({perc:3d} % done), field decay: {mant}"


def run(simulation, task_name: Optional[str] = None, folder_name: str = "default",
        path: Optional[str] = None, callback_url: Optional[str] = None, verbose: bool = True,
        progress_callback_upload=None, progress_callback_download=None,
        solver_version: Optional[str] = None, worker_group: Optional[str] = None,
        simulation_type: str = "tidy3d", parent_tasks=None, local_gradient: bool = False,
        *, device: int = 0, n_steps: Optional[int] = None, lib=None,
        return_tidy3d: Optional[bool] = None, devices=None, _dist_options: Optional[dict] = None,
        mode_grid_dispersion: Optional[bool] = None) -> SimulationData:
    """Solve ``simulation`` on the local MI355X and return its ``SimulationData``.

    Cloud-only arguments (``folder_name``, ``callback_url``, ``progress_callback_*``,
    ``solver_version``, ``worker_group``, ``parent_tasks``, ``local_gradient``) are accepted and
    ignored.  ``path``: when given, the data is written there — ``*.hdf5`` in the reference's file
    layout (loadable with ``tidy3d.SimulationData.from_file``), any other name as ``.npz``.  Extra keyword-only arguments select the GPU
    (``device``), override the number of time steps (``n_steps``, tests/benchmarks) or pass an
    explicitly loaded library (``lib``, tests).  ``devices=[0, 1, ...]``: one worker process per listed GPU, the
    grid split into z-slabs with RCCL ghost-plane exchange (``tidy3d_amd.dist``) — the whole multi-GPU run is this
    one call.  ``mode_grid_dispersion``: False launches mode sources with the continuum mode instead of the one the Yee
    grid propagates (``discretize.MODE_SOURCE_GRID_DISPERSION``, default True; one-GPU runs)."""
    from .engine import HipEngine

    sim, was_tidy3d = _as_mirror(simulation)
    sim.validate_pre_upload(source_required=True)
    if devices is not None and len(devices) > 1:
        sim_data = _run_on_devices(sim, [int(d) for d in devices], n_steps, verbose, _dist_options or {})
        want_td = was_tidy3d if return_tidy3d is None else return_tidy3d
        if want_td:
            from .adapter import to_tidy3d
            out = to_tidy3d(sim_data, simulation if was_tidy3d else None)
            if path:
                out.to_file(path)
            return out
        if path:
            save(sim_data, path)
        return sim_data
    if devices is not None and len(devices) == 1:
        device = int(devices[0])
    t_setup = time.perf_counter()
    disc = discretize(sim, n_steps=n_steps, mode_grid_dispersion=mode_grid_dispersion)
    spec = disc.spec
    lines = [f"Simulation domain Nx, Ny, Nz: {list(spec.shape)}",
             f"Applied symmetries: {tuple(sim.symmetry)}",
             f"Number of computational grid points: {spec.n_cells:.4e}.",
             f"Number of time steps: {spec.n_steps:.4e}",
             f"Time step size (dt): {spec.dt:.4e}s", "",
             "Running solver for %d time steps..." % spec.n_steps]

    def progress(step, t, decay):
        ln = _log_line(step, spec.n_steps, t, decay)
        lines.append(ln)
        if verbose:
            print(ln, flush=True)
        return False

    with HipEngine(spec, lib=lib, device=device) as eng:
        used_lib = eng.lib                                  # the library that ran the solve also integrates the projections
        setup_s = time.perf_counter() - t_setup
        t0 = time.perf_counter()
        stats = eng.run(progress=progress if spec.decay_every else None)
        solve_s = time.perf_counter() - t0
        raw = eng.results()
        steps_done = int(stats.steps_done)
        diverged = bool(stats.diverged)
        if stats.stopped_early:
            lines.append(f"Field decay smaller than shutoff factor, exiting solver "
                         f"(time step {steps_done}).")
        if diverged:
            lines.append("WARNING: field divergence detected, exiting solver.")
        lines.append(schedule_line(stats, spec.n_cells, solve_s))
    lines += ["", f"Setup time (s):  {setup_s:.4f}", f"Solver time (s): {solve_s:.4f}",
              f"Time-stepping speed (cells/s): {spec.n_cells * steps_done / max(solve_s, 1e-9):.2e}"]
    sim_data = assemble(disc, raw, log="\n".join(lines), diverged=diverged, n_steps_run=steps_done, device_lib=used_lib, device=device)

    # post-run warnings, ref web/api/tidy3d_stub.py:219-233
    if diverged:
        log.warning("The simulation has diverged! For more information, check 'SimulationData.log'.")
    elif sim.shutoff != 0 and spec.decay_every and sim_data.final_decay_value > sim.shutoff:
        log.warning(f"Simulation final field decay value of {sim_data.final_decay_value} is greater "
                    f"than the simulation shutoff threshold of {sim.shutoff}. Consider running the "
                    "simulation again with a larger 'run_time' duration for more accurate results.")

    want_td = was_tidy3d if return_tidy3d is None else return_tidy3d
    if want_td:
        from .adapter import to_tidy3d
        out = to_tidy3d(sim_data, simulation if was_tidy3d else None)
        if path:
            out.to_file(path)
        return out
    if path:
        save(sim_data, path)
    return sim_data


def _run_on_devices(sim, devices, n_steps, verbose: bool, opt: dict) -> SimulationData:
    """Spawn one ``python -m tidy3d_amd.dist_main`` per GPU (ranks in the order of ``devices``; rendezvous on
    127.0.0.1), wait, and load what rank 0 wrote."""
    import pickle
    import socket
    import subprocess
    import sys
    import tempfile
    import time as _time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    timeout_s = float(opt.get("timeout", 0) or 0)                 # whole multi-GPU call; 0 = none
    rdv_timeout = int(opt.get("rendezvous_timeout", 300))         # init_process_group of every rank
    with tempfile.TemporaryDirectory(prefix="tidy3d_amd_") as tmp:
        f_sim, f_out = os.path.join(tmp, "sim.pkl"), os.path.join(tmp, "data.pkl")
        with open(f_sim, "wb") as f:
            pickle.dump(sim, f, protocol=pickle.HIGHEST_PROTOCOL)
        # rank 0 binds the rendezvous port itself (MASTER_PORT=0 is not portable across torch versions), so the port is
        # chosen here — but held open until just before the ranks start, and a start that loses the race is retried
        for attempt in range(3):
            with socket.socket() as sck:
                sck.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                sck.bind(("127.0.0.1", 0))
                port = sck.getsockname()[1]
            procs, logs = [], []
            for rank, dev in enumerate(devices):
                env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(len(devices)), LOCAL_RANK=str(dev),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TIDY3D_AMD_RDV_TIMEOUT=str(rdv_timeout))
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (what this driver supports) unless the user chose
                env["PYTHONPATH"] = os.pathsep.join([root] + [p for p in (opt.get("pythonpath") or []) if p]
                                                    + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
                cmd = [sys.executable, "-m", "tidy3d_amd.dist_main", "--sim", f_sim, "--out", f_out,
                       "--backend", opt.get("backend", "nccl")]
                if n_steps is not None:
                    cmd += ["--n-steps", str(int(n_steps))]
                if opt.get("lib"):
                    cmd += ["--lib", opt["lib"]]
                if opt.get("hook"):
                    cmd += ["--hook", opt["hook"]]
                # stderr goes to a FILE per rank: a pipe nobody drains blocks its rank after ~64 KiB (HIP / RCCL warnings,
                # AMD_LOG_LEVEL) — and with it every sibling waiting for that rank in a collective
                lf = open(os.path.join(tmp, f"rank{rank}.attempt{attempt}.err"), "w+")
                logs.append(lf)
                procs.append(subprocess.Popen(cmd, env=env, stdout=None if verbose else subprocess.DEVNULL, stderr=lf, text=True))
            # poll ALL ranks: the first one that fails takes its siblings down (they would otherwise sit in
            # init_process_group or in an RCCL send / recv for ever); an overall timeout does the same
            t0 = _time.monotonic()
            failed = None
            while True:
                codes = [p.poll() for p in procs]
                bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
                if bad:
                    failed = (bad[0], codes[bad[0]])
                    break
                if all(c == 0 for c in codes):
                    break
                if timeout_s and _time.monotonic() - t0 > timeout_s:
                    failed = (-1, None)
                    break
                _time.sleep(0.05)
            if failed is not None:
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                for p in procs:
                    try:
                        p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        p.wait()
            tails = []
            for lf in logs:
                lf.seek(0)
                tails.append(lf.read()[-2000:])
                lf.close()
            if failed is None:
                break
            r, rc = failed
            lost_port = r >= 0 and ("eaddrinuse" in tails[r].lower() or "address already in use" in tails[r].lower())
            if lost_port and attempt < 2:
                continue                                   # lost the race for the port: once more with another one
            if r < 0:
                raise SolverLibraryError(f"multi-GPU run: no result after {timeout_s:.0f} s (timeout); ranks terminated. "
                                         f"rank 0 stderr: {tails[0]}")
            raise SolverLibraryError(f"multi-GPU run: rank {r} exited with status {rc} (its siblings were terminated): {tails[r]}")
        with open(f_out, "rb") as f:
            return pickle.load(f)


def save(sim_data: SimulationData, path: str) -> None:
    """``.hdf5``: the reference's own file layout, written through the HDF5 C library
    (tidy3d_amd/hdf5io.py) — loadable with ``tidy3d.SimulationData.from_file``; anything else: .npz."""
    if path.endswith(".hdf5"):
        from .hdf5io import write_simulation_data
        write_simulation_data(sim_data, path)
    else:
        save_npz(sim_data, path)


def load(path: str) -> SimulationData:
    """Counterpart of ``tidy3d.web.load`` for a local file written by ``run(..., path=...)``."""
    from .hdf5io import load_simulation_data
    return load_simulation_data(path)


def save_npz(sim_data: SimulationData, path: str) -> None:
    """Plain .npz dump of every DataArray (values + coords) plus the log; the tidy3d hdf5
    container needs h5py and is written through the real package (adapter.to_tidy3d)."""
    blobs = {"log": np.array(sim_data.log or ""), "diverged": np.array(sim_data.diverged)}
    for d in sim_data.data:
        name = d.monitor.name
        arrays = getattr(d, "field_components", None)
        if arrays is None and hasattr(d, "amps"):
            arrays = {"amps": d.amps, "n_complex": d.n_complex}
            if getattr(d, "mode_power", None) is not None:
                arrays["mode_power"] = d.mode_power
        if arrays is None:
            arrays = {"flux": d.flux}
        for k, v in arrays.items():
            blobs[f"{name}/{k}"] = v.values
            for dim, c in v.coords.items():
                blobs[f"{name}/{k}/{dim}"] = c
    np.savez(path if path.endswith(".npz") else path + ".npz", **blobs)


# ----------------------------------------------------------------------------------------------
# Job / Batch façade (ref web/api/container.py): the cloud life cycle upload -> start -> monitor ->
# download -> load collapses to one local solve; names, arguments and return types are kept so that
# scripts written against tidy3d.web.Job / Batch run unchanged.
# ----------------------------------------------------------------------------------------------

DEFAULT_DATA_PATH = "simulation_data.hdf5"       # ref web/api/webapi.py / container.py defaults
DEFAULT_DATA_DIR = "."


class Job:
    """ref container.py:35 — one simulation; ``run(path)`` returns its SimulationData."""

    def __init__(self, simulation, task_name: str, folder_name: str = "default", callback_url: Optional[str] = None,
                 solver_version: Optional[str] = None, verbose: bool = True, simulation_type: str = "tidy3d",
                 parent_tasks=None, **run_kwargs):
        self.simulation, self.task_name, self.folder_name = simulation, task_name, folder_name
        self.verbose, self.run_kwargs = verbose, run_kwargs
        self.task_id = f"local-{task_name}"
        self._data: Optional[SimulationData] = None
        self._path: Optional[str] = None

    def upload(self) -> None:                       # nothing leaves this machine
        pass

    def start(self) -> None:
        if self._data is None:
            self._data = run(self.simulation, task_name=self.task_name, folder_name=self.folder_name,
                             verbose=self.verbose, **self.run_kwargs)

    def monitor(self) -> None:
        pass

    @property
    def status(self) -> str:
        return "success" if self._data is not None else "draft"

    def download(self, path: str = DEFAULT_DATA_PATH) -> None:
        self.start()
        data = self._data
        if hasattr(data, "to_file"):               # a genuine tidy3d.SimulationData
            data.to_file(path)
        else:
            save(data, path)
        self._path = path

    def load(self, path: str = DEFAULT_DATA_PATH):
        if self._data is None or self._path != path:
            self.download(path)
        return self._data

    def run(self, path: str = DEFAULT_DATA_PATH):
        """ref container.py:190-207."""
        self.upload()
        self.start()
        self.monitor()
        return self.load(path=path)

    def delete(self) -> None:
        self._data = None

    def estimate_cost(self, verbose: bool = True) -> float:
        return 0.0                                   # no FlexCredits on your own GPU

    real_cost = estimate_cost


class BatchData:
    """ref container.py:342 — maps task names to data files and loads them one at a time."""

    def __init__(self, task_paths: dict, task_ids: dict, verbose: bool = True, _cache: Optional[dict] = None):
        self.task_paths, self.task_ids, self.verbose = dict(task_paths), dict(task_ids), verbose
        self._cache = _cache or {}

    def load_sim_data(self, task_name: str):
        if task_name in self._cache:
            return self._cache[task_name]
        return load(self.task_paths[task_name])

    def items(self):
        for task_name in self.task_paths:
            yield task_name, self.load_sim_data(task_name)

    def __getitem__(self, task_name: str):
        return self.load_sim_data(task_name)

    def __len__(self):
        return len(self.task_paths)


class Batch:
    """ref container.py:426 — several simulations; ``run(path_dir)`` returns a BatchData.  The jobs
    run one after the other on the local GPU (``device=`` in ``run_kwargs`` selects it)."""

    def __init__(self, simulations: dict, folder_name: str = "default", verbose: bool = True,
                 solver_version: Optional[str] = None, callback_url: Optional[str] = None,
                 simulation_type: str = "tidy3d", parent_tasks=None, **run_kwargs):
        self.simulations, self.folder_name, self.verbose = dict(simulations), folder_name, verbose
        self.run_kwargs = run_kwargs
        self.jobs = {name: Job(sim, task_name=name, folder_name=folder_name, verbose=verbose, **run_kwargs)
                     for name, sim in self.simulations.items()}

    @property
    def num_jobs(self) -> int:
        return len(self.jobs)

    def upload(self) -> None:
        pass

    def start(self) -> None:
        for job in self.jobs.values():
            job.start()

    def monitor(self) -> None:
        pass

    @staticmethod
    def _job_data_path(task_id: str, path_dir: str = DEFAULT_DATA_DIR) -> str:
        return os.path.join(path_dir, f"{task_id}.hdf5")          # ref container.py:762-778

    def download(self, path_dir: str = DEFAULT_DATA_DIR) -> None:
        os.makedirs(path_dir, exist_ok=True)
        for job in self.jobs.values():
            job.download(self._job_data_path(job.task_id, path_dir))

    def load(self, path_dir: str = DEFAULT_DATA_DIR) -> BatchData:
        paths = {name: self._job_data_path(job.task_id, path_dir) for name, job in self.jobs.items()}
        for name, job in self.jobs.items():
            if job._path != paths[name]:
                job.download(paths[name])
        return BatchData(task_paths=paths, task_ids={n: j.task_id for n, j in self.jobs.items()},
                         verbose=self.verbose, _cache={n: j._data for n, j in self.jobs.items()})

    def run(self, path_dir: str = DEFAULT_DATA_DIR) -> BatchData:
        """ref container.py:523-557."""
        os.makedirs(path_dir, exist_ok=True)
        self.upload()
        self.start()
        self.monitor()
        self.download(path_dir=path_dir)
        return self.load(path_dir=path_dir)

    def delete(self) -> None:
        for job in self.jobs.values():
            job.delete()

    def estimate_cost(self, verbose: bool = True) -> float:
        return 0.0

    real_cost = estimate_cost
