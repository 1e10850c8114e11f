"""``run(..., path="x.hdf5")`` writes the reference's own file layout (ref
components/base.py:691-738, data/data_array.py:248-281) through the HDF5 C library.  Pins:
(1) round trip through the library itself; (2) the reference's READ RECIPE (``dict_from_hdf5`` /
``DataArray.from_hdf5``: ``f["JSON_STRING"][()]``, ``np.array(group[name])``) executed by a real h5py
where one exists in the image (/opt/conda python3.9); (3) structure equality with a file the same
h5py writes the way the reference's ``to_hdf5`` does."""
import json
import os
import subprocess

import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import hdf5io
from tidy3d_amd.web import load, run

try:
    hdf5io.Hdf5Library.get()
    HAVE_LIB = True
except Exception:       # noqa: BLE001 - any failure to bind means "not available here"
    HAVE_LIB = False
pytestmark = pytest.mark.skipif(not HAVE_LIB, reason="HDF5 C library not present")

H5PY_PYTHON = "/opt/conda/bin/python3.9"
HAVE_H5PY = os.path.exists(H5PY_PYTHON) and subprocess.run(
    [H5PY_PYTHON, "-c", "import h5py"], capture_output=True).returncode == 0

DL = 0.05
PULSE = td.GaussianPulse(freq0=3e14, fwidth=1e14)


@pytest.fixture(scope="module")
def written(emu_lib, tmp_path_factory):
    sim = td.Simulation(
        size=(16 * DL, 12 * DL, 10 * DL), grid_spec=td.GridSpec.uniform(dl=DL), run_time=4e-14,
        structures=[td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=0.15), medium=td.Medium(permittivity=3.0))],
        sources=[td.PointDipole(center=(0, 0, 0), source_time=PULSE, polarization="Ez")],
        monitors=[td.FieldMonitor(center=(0, 0, 0.1), size=(0.4, 0.3, 0), freqs=[2.5e14, 3e14], name="f"),
                  td.FieldTimeMonitor(center=(0.1, 0, 0), size=(0, 0.2, 0.2), name="t", interval=4, fields=["Ez", "Hx"]),
                  td.FluxMonitor(center=(0, 0, 0.1), size=(0.4, 0.3, 0), freqs=[2.5e14, 3e14], name="fl"),
                  td.FluxTimeMonitor(center=(0, 0, 0.1), size=(0.4, 0.3, 0), name="flt", interval=8),
                  td.PermittivityMonitor(center=(0, 0, 0), size=(0.4, 0.3, 0), freqs=[3e14], name="eps")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=4)))
    path = str(tmp_path_factory.mktemp("h5") / "sim_data.hdf5")
    sd = run(sim, task_name="h5", verbose=False, lib=emu_lib, n_steps=60, path=path)
    return sim, sd, path


def test_round_trip_through_the_library(written):
    sim, sd, path = written
    back = load(path)
    assert [d.monitor.name for d in back.data] == [d.monitor.name for d in sd.data]
    assert back.log == sd.log and back.diverged == sd.diverged
    assert back.simulation.size == pytest.approx(sim.size)
    for a, b in zip(sd.data, back.data):
        assert type(a).__name__ == type(b).__name__
        fa = getattr(a, "field_components", None) or {"flux": a.flux}
        fb = getattr(b, "field_components", None) or {"flux": b.flux}
        for k, v in fa.items():
            if v is None:
                continue
            assert np.array_equal(np.asarray(v.values), np.asarray(fb[k].values)), (a.monitor.name, k)
            assert fb[k].values.dtype == np.asarray(v.values).dtype
            for dim in v.dims:
                assert np.array_equal(np.asarray(v.coords[dim]), np.asarray(fb[k].coords[dim]))


def test_layout_is_the_references(written):
    _, sd, path = written
    tree = hdf5io.read_tree(path)
    model = json.loads(tree["/JSON_STRING"])
    assert model["type"] == "SimulationData" and model["simulation"]["type"] == "Simulation"
    assert [e["type"] for e in model["data"]] == ["FieldData", "FieldTimeData", "FluxData", "FluxTimeData",
                                                  "PermittivityData"]
    assert model["data"][0]["Ex"] == "ScalarFieldDataArray" and model["data"][2]["flux"] == "FluxDataArray"
    assert model["data"][1]["Ez"] == "ScalarFieldTimeDataArray" and "Ex" not in model["data"][1]
    assert tree["/data/0/Ex/__xarray_dataarray_variable__"].dtype == np.complex64
    assert tree["/data/0/Ex/__xarray_dataarray_variable__"].shape == sd["f"].Ex.shape
    assert tree["/data/1/Ez/__xarray_dataarray_variable__"].dtype == np.float32
    assert set(k for k in tree if k.startswith("/data/0/Ex/") and not k.endswith("/")) == {
        "/data/0/Ex/__xarray_dataarray_variable__", "/data/0/Ex/x", "/data/0/Ex/y", "/data/0/Ex/z", "/data/0/Ex/f"}
    assert tree["/data/2/flux/f"].dtype == np.float64


_READ_RECIPE = r'''
import json, sys
import numpy as np
import h5py
fname = sys.argv[1]
# ref base.py:572-580 _json_string_from_hdf5
with h5py.File(fname, "r") as f:
    n = len([k for k in f.keys() if "JSON_STRING" in k])
    js = b""
    for ind in range(n):
        js += f["JSON_STRING" if ind == 0 else f"JSON_STRING_{ind}"][()]
model = json.loads(js)
out = {"types": [e["type"] for e in model["data"]], "arrays": {}}
DIMS = {"ScalarFieldDataArray": "xyzf", "ScalarFieldTimeDataArray": "xyzt", "FluxDataArray": "f", "FluxTimeDataArray": "t"}
with h5py.File(fname, "r") as f:
    for i, e in enumerate(model["data"]):
        for k, v in e.items():
            if isinstance(v, str) and v in DIMS:
                g = f[f"/data/{i}/{k}"]                      # ref data_array.py:269-279 from_hdf5
                vals = np.array(g["__xarray_dataarray_variable__"])
                coords = {d: np.array(g[d]) for d in DIMS[v] if d in g}
                out["arrays"][f"{i}/{k}"] = {"dtype": str(vals.dtype), "shape": list(vals.shape),
                                             "sum_re": float(np.real(vals).astype(np.float64).sum()),
                                             "sum_im": float(np.imag(vals).astype(np.float64).sum()),
                                             "coords": {d: [float(c.min()), float(c.max()), len(c)] for d, c in coords.items()}}
print(json.dumps(out))
'''


@pytest.mark.skipif(not HAVE_H5PY, reason="no python with h5py in this image")
def test_h5py_reads_it_with_the_references_recipe(written, tmp_path):
    _, sd, path = written
    script = tmp_path / "read.py"
    script.write_text(_READ_RECIPE)
    r = subprocess.run([H5PY_PYTHON, str(script), path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout)
    assert got["types"] == ["FieldData", "FieldTimeData", "FluxData", "FluxTimeData", "PermittivityData"]
    ex = got["arrays"]["0/Ex"]
    assert ex["dtype"] == "complex64" and ex["shape"] == list(sd["f"].Ex.shape)
    v = np.asarray(sd["f"].Ex.values)
    assert ex["sum_re"] == pytest.approx(float(v.real.astype(np.float64).sum()), rel=1e-12, abs=1e-12)
    assert ex["sum_im"] == pytest.approx(float(v.imag.astype(np.float64).sum()), rel=1e-12, abs=1e-12)
    assert ex["coords"]["f"] == [2.5e14, 3e14, 2]
    assert got["arrays"]["1/Ez"]["dtype"] == "float32"
    assert got["arrays"]["2/flux"]["shape"] == [2] and got["arrays"]["3/flux"]["dtype"] == "float32"
    assert got["arrays"]["4/eps_xx"]["dtype"] == "complex128"


_WRITE_RECIPE = r'''
import sys
import numpy as np
import h5py
# what the reference's to_hdf5 does (ref base.py:707-712, data_array.py:259-267)
with h5py.File(sys.argv[1], "w") as f:
    f["JSON_STRING"] = "{\"k\": 1}"
    g = f.create_group("/data/0/Ex")
    g["__xarray_dataarray_variable__"] = (np.arange(6).reshape(1, 2, 3, 1) * (1 + 2j)).astype(np.complex64)
    g["x"] = np.array([0.5]); g["f"] = np.array([2e14])
    g2 = f.create_group("/data/1/amps")
    g2["direction"] = ["+", "-"]
    g2["__xarray_dataarray_variable__"] = np.zeros((2, 1, 1), complex)
'''


@pytest.mark.skipif(not HAVE_H5PY, reason="no python with h5py in this image")
def test_reads_a_file_written_by_h5py_and_matches_its_structure(tmp_path):
    """The other direction: a file h5py writes the reference's way is read by this module, and the
    same content written here has the same h5dump header (types, spaces, string charset)."""
    ref = str(tmp_path / "ref.hdf5")
    script = tmp_path / "write.py"
    script.write_text(_WRITE_RECIPE)
    r = subprocess.run([H5PY_PYTHON, str(script), ref], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    tree = hdf5io.read_tree(ref)
    assert tree["/JSON_STRING"] == '{"k": 1}'
    assert tree["/data/0/Ex/__xarray_dataarray_variable__"].dtype == np.complex64
    assert tree["/data/0/Ex/__xarray_dataarray_variable__"][0, 1, 2, 0] == np.complex64(5 + 10j)
    assert tree["/data/1/amps/direction"] == ["+", "-"]
    mine = str(tmp_path / "mine.hdf5")
    with hdf5io.H5Writer(mine) as w:
        w.string("/JSON_STRING", '{"k": 1}')
        w.group("/data/0/Ex")
        w.array("/data/0/Ex/__xarray_dataarray_variable__", tree["/data/0/Ex/__xarray_dataarray_variable__"])
        w.array("/data/0/Ex/x", np.array([0.5]))
        w.array("/data/0/Ex/f", np.array([2e14]))
        w.group("/data/1/amps")
        w.strings("/data/1/amps/direction", ["+", "-"])
        w.array("/data/1/amps/__xarray_dataarray_variable__", np.zeros((2, 1, 1), complex))
    dump = "/opt/conda/bin/h5dump"
    if os.path.exists(dump):
        a = subprocess.run([dump, "-H", ref], capture_output=True, text=True).stdout.replace(ref, "F")
        b = subprocess.run([dump, "-H", mine], capture_output=True, text=True).stdout.replace(mine, "F")
        norm = lambda s: sorted(line.strip() for line in s.splitlines())      # creation order may differ
        assert norm(a) == norm(b), (a, b)


def test_job_and_batch_facade(emu_lib, tmp_path):
    """ref web/api/container.py: Job.run(path) / Batch.run(path_dir) -> BatchData[task_name]."""
    from tidy3d_amd.web import Batch, BatchData, Job
    mk = lambda eps: td.Simulation(
        size=(12 * DL, 10 * DL, 8 * DL), grid_spec=td.GridSpec.uniform(dl=DL), run_time=2e-14,
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.2, 0.2, 0.2)), medium=td.Medium(permittivity=eps))],
        sources=[td.PointDipole(center=(0, 0, 0), source_time=PULSE, polarization="Ez")],
        monitors=[td.FluxMonitor(center=(0, 0, 0.1), size=(0.3, 0.3, 0), freqs=[3e14], name="fl")],
        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    job = Job(mk(2.0), task_name="one", lib=emu_lib, n_steps=30, verbose=False)
    sd = job.run(path=str(tmp_path / "one.hdf5"))
    assert job.status == "success" and (tmp_path / "one.hdf5").exists() and sd["fl"].flux.shape == (1,)
    batch = Batch({"a": mk(2.0), "b": mk(6.0)}, lib=emu_lib, n_steps=30, verbose=False)
    bd = batch.run(path_dir=str(tmp_path / "batch"))
    assert isinstance(bd, BatchData) and len(bd) == 2 and batch.num_jobs == 2
    names = [n for n, _ in bd.items()]
    assert names == ["a", "b"]
    assert np.array_equal(bd["a"]["fl"].flux.values, sd["fl"].flux.values)
    assert not np.array_equal(bd["b"]["fl"].flux.values, sd["fl"].flux.values)
    # a fresh BatchData reads the files back
    again = BatchData(task_paths=bd.task_paths, task_ids=bd.task_ids)
    assert np.array_equal(again["b"]["fl"].flux.values, bd["b"]["fl"].flux.values)


def test_projection_data_round_trip(tmp_path):
    """FieldProjection{Angle,Cartesian,KSpace}Data through the .hdf5 layout (ref DATA_ARRAY_MAP names)."""
    from oracle.fdtd_numpy import OracleFdtd
    from tidy3d_amd.data import assemble
    from tidy3d_amd.discretize import discretize
    from tidy3d_amd.web import save
    common = dict(center=(0, 0, 0.2), size=(0.5, 0.4, 0), freqs=[3e14], normal_dir="+")
    sim = td.Simulation(
        size=(12 * DL, 10 * DL, 10 * DL), grid_spec=td.GridSpec.uniform(dl=DL), run_time=2e-14,
        sources=[td.PointDipole(center=(0, 0, 0), source_time=PULSE, polarization="Ex")],
        monitors=[td.FieldProjectionAngleMonitor(theta=[0.2, 0.5], phi=[0.0], proj_distance=1e3, name="a", **common),
                  td.FieldProjectionCartesianMonitor(x=[10.0, 20.0], y=[5.0], proj_distance=1e3, name="c", **common),
                  td.FieldProjectionKSpaceMonitor(ux=[0.1, 0.2], uy=[0.0, 0.3], proj_distance=1e3, name="k", **common)],
        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()))
    disc = discretize(sim, n_steps=40)
    sd = assemble(disc, OracleFdtd(disc.spec).run())
    path = str(tmp_path / "proj.hdf5")
    save(sd, path)
    tree = hdf5io.read_tree(path)
    model = json.loads(tree["/JSON_STRING"])
    assert [e["type"] for e in model["data"]] == ["FieldProjectionAngleData", "FieldProjectionCartesianData",
                                                  "FieldProjectionKSpaceData"]
    assert model["data"][1]["Etheta"] == "FieldProjectionCartesianDataArray"
    assert model["data"][0]["projection_surfaces"][0]["monitor"]["name"] == "a"
    back = load(path)
    for name in ("a", "c", "k"):
        for comp in ("Etheta", "Hphi"):
            assert np.array_equal(getattr(back[name], comp).values, getattr(sd[name], comp).values)
            assert getattr(back[name], comp).dims == getattr(sd[name], comp).dims


def test_mode_power_survives_the_file(emu_lib, tmp_path):
    """``ModeData.mode_power`` (what makes |amps|^2 comparable with a FluxMonitor) was dropped by every
    save / load.  It is stored as an extra DataArray group beside the two the JSON model names (the reference's loader
    ignores it) and in the .npz dump."""
    from tidy3d_amd.web import save_npz
    plane = (td.inf, td.inf, 0)
    ms = td.ModeSpec(num_modes=2, target_neff=2.0)
    sim = td.Simulation(
        size=(24 * DL, 0, 28 * DL), grid_spec=td.GridSpec.uniform(dl=DL), run_time=4e-14, medium=td.Medium(permittivity=2.0),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.3, td.inf, td.inf)), medium=td.Medium(permittivity=6.0))],
        sources=[td.ModeSource(center=(0, 0, -0.4), size=plane, source_time=PULSE, direction="+", mode_spec=ms, mode_index=0)],
        monitors=[td.ModeMonitor(center=(0, 0, 0.3), size=plane, freqs=[2.8e14, 3e14], mode_spec=ms, name="mm")],
        boundary_spec=td.BoundarySpec(x=td.Boundary.pml(num_layers=4), y=td.Boundary.periodic(), z=td.Boundary.pml(num_layers=4)))
    path = str(tmp_path / "mode.hdf5")
    sd = run(sim, task_name="mp", verbose=False, lib=emu_lib, n_steps=40, path=path)
    mp = np.asarray(sd["mm"].mode_power.values)
    assert mp.shape == (2, 2, 2) and np.all(np.isfinite(mp)) and np.abs(mp).max() > 0
    back = load(path)
    np.testing.assert_array_equal(np.asarray(back["mm"].mode_power.values), mp)
    np.testing.assert_array_equal(np.asarray(back["mm"].amps.values), np.asarray(sd["mm"].amps.values))
    assert "mode_power" not in json.loads(hdf5io.read_tree(path)["/JSON_STRING"])["data"][0]       # the model stays the reference's
    save_npz(sd, str(tmp_path / "mode.npz"))
    z = np.load(str(tmp_path / "mode.npz"))
    np.testing.assert_array_equal(z["mm/mode_power"], mp)
