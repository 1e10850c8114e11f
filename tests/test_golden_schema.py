"""The host-side mirror of the reference schema against golden values produced by the
reference's own code (tests/golden/make_golden.py imports /root/reference and records what
tidy3d computes).  Pins: grid boundaries incl. PML cells, dt, tmesh length, nyquist step,
monitor index spans / time indices, source waveforms and spectra, pole-residue conversions,
eps_model, n_cfl, geometry.inside, apodisation windows."""
import json
import os

import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import discretize as D

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "schema_golden.json")) as f:
    GOLD = json.load(f)


def _c(v):
    return complex(v[0], v[1])


@pytest.mark.parametrize("name", sorted(GOLD["simulations"]))
def test_simulation_discretisation_matches_reference(name):
    rec = GOLD["simulations"][name]
    exp = rec["expected"]
    sim = td.Simulation.from_dict(rec["json"])          # the reference's own JSON form
    b = D.make_boundaries(sim)
    assert [len(x) - 1 for x in b] == exp["num_cells"]
    for a, d in enumerate("xyz"):
        np.testing.assert_allclose(b[a], exp["boundaries"][d], rtol=1e-13, atol=1e-13)
    assert [list(x) for x in D.num_pml_layers(sim)] == exp["num_pml_layers"]
    dt = D.compute_dt(sim, b)
    assert dt == pytest.approx(exp["dt"], rel=1e-13)
    rt = D.run_time(sim)
    assert rt == pytest.approx(exp["run_time"], rel=1e-12)
    tmesh = D.make_tmesh(rt, dt)
    assert len(tmesh) == exp["num_time_steps"]
    assert tmesh[-1] == pytest.approx(exp["tmesh_last"], rel=1e-12)
    assert D.nyquist_step(sim, dt) == exp["nyquist_step"]
    assert list(D.frequency_range(sim)) == pytest.approx(exp["frequency_range"])
    assert sorted(m.n_cfl for m in sim.mediums) == pytest.approx(exp["mediums_n_cfl"], rel=1e-12)
    for m in sim.monitors:
        e = exp["monitors"][m.name]
        assert D.discretize_inds_monitor(b, m).tolist() == e["span"], m.name
        if "time_inds" in e:
            assert list(m.time_inds(tmesh)) == e["time_inds"]
            assert m.num_steps(tmesh) == e["num_steps"]


@pytest.mark.parametrize("i", range(len(GOLD["source_times"])))
def test_source_time_matches_reference(i):
    rec = GOLD["source_times"][i]
    st = td.parse(rec["json"])
    amp = st.amp_time(np.array(rec["t"]))
    ref = np.array([_c(a) for a in rec["amp"]])
    np.testing.assert_allclose(amp, ref, rtol=1e-12, atol=1e-300)
    if rec["end_time"] is None:
        assert st.end_time() is None
    else:
        assert st.end_time() == pytest.approx(rec["end_time"], rel=1e-14)
    assert list(st.frequency_range()) == pytest.approx(rec["frequency_range"])
    tm = np.arange(0, 3e-13, rec["spec_dt"])
    assert len(tm) == rec["spec_n"]
    sp = st.spectrum(tm, rec["spec_freqs"], rec["spec_dt"])
    ref = np.array([_c(s) for s in rec["spectrum"]])
    # the reference accumulates a running product exp(i w dt)^n; we evaluate the sum directly
    np.testing.assert_allclose(sp, ref, rtol=2e-9, atol=1e-12 * np.abs(ref).max())


def test_gaussian_pulse_has_no_dc_component():
    """ref tests/test_components/test_source.py:57-77."""
    st = td.GaussianPulse(freq0=1e14, fwidth=5e13)
    tm = np.arange(0, 1e-12, 1e-16)
    dc = st.spectrum(tm, [0.0], 1e-16)
    peak = st.spectrum(tm, [1e14], 1e-16)
    assert abs(dc[0]) < 1e-5 * abs(peak[0])
    st2 = td.GaussianPulse(freq0=1e14, fwidth=5e13, remove_dc_component=False)
    assert abs(st2.spectrum(tm, [0.0], 1e-16)[0]) > 1e3 * abs(dc[0])


@pytest.mark.parametrize("i", range(len(GOLD["media"])))
def test_medium_pole_residue_matches_reference(i):
    rec = GOLD["media"][i]
    med = td.parse(rec["json"])
    assert not isinstance(med, td.Unsupported), rec["json"]["type"]
    eps = med.eps_model(np.array(rec["freqs"]))
    np.testing.assert_allclose(eps, [_c(e) for e in rec["eps_model"]], rtol=1e-10)
    assert med.n_cfl == pytest.approx(rec["n_cfl"], rel=1e-12)
    if "poles" in rec:
        eps_inf, sigma, poles = med.pole_residue()
        assert eps_inf == pytest.approx(rec["eps_inf"])
        ref = [(_c(a), _c(c)) for a, c in rec["poles"]]
        assert len(poles) == len(ref)
        for (a, c), (ra, rc) in zip(poles, ref):
            assert a == pytest.approx(ra, rel=1e-12)
            assert c == pytest.approx(rc, rel=1e-12)


def test_ncfl_known_answers():
    """ref tests/test_components/test_medium.py:333-354."""
    assert td.Medium(permittivity=4.0).n_cfl == pytest.approx(2.0)
    assert td.Lorentz(eps_inf=0.16, coeffs=[(1, 2e14, 1e13)]).n_cfl == pytest.approx(0.4)
    assert td.PoleResidue(eps_inf=0.04, poles=[(-1 + 2j, 1 + 3j)]).n_cfl == pytest.approx(0.2)
    assert td.PEC.n_cfl == 1.0


@pytest.mark.parametrize("i", range(len(GOLD["geometry"])))
def test_geometry_inside_matches_reference(i):
    rec = GOLD["geometry"][i]
    g = td.parse(rec["json"])
    p = np.array(rec["points"])
    assert list(g.inside(p[:, 0], p[:, 1], p[:, 2])) == rec["inside"]
    np.testing.assert_allclose(np.array(g.bounds), np.array(rec["bounds"]))


@pytest.mark.parametrize("i", range(len(GOLD["apodization"])))
def test_apodization_window(i):
    rec = GOLD["apodization"][i]
    w = td.ApodizationSpec(**rec["spec"]).window(np.array(rec["t"]))
    np.testing.assert_allclose(w, rec["window"], rtol=1e-14)


def test_reference_grid_known_answers():
    """ref tests/test_components/test_grid.py:255-273: size 4, dl 1, 2 PML layers -> -4..4."""
    sim = td.Simulation(size=(4, 4, 4), grid_spec=td.GridSpec.uniform(dl=1.0), run_time=1e-12,
                        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=2)))
    for b in D.make_boundaries(sim):
        np.testing.assert_allclose(b, np.arange(-4, 5))


REF_SAMPLE = "/root/reference/tests/sims/simulation_sample.json"


@pytest.mark.skipif(not os.path.exists(REF_SAMPLE), reason="reference checkout not present")
def test_full_reference_fixture_parses():
    """The reference's kitchen-sink fixture (every source/monitor/medium type, ref
    tests/utils.py:400) is read in place (not copied): parsing never fails, unsupported types
    become Unsupported placeholders that raise only when used."""
    with open(REF_SAMPLE) as f:
        d = json.load(f)
    sim = td.Simulation.from_dict(d)
    assert len(sim.sources) == len(d["sources"])
    assert len(sim.monitors) == len(d["monitors"])
    kinds = {type(s).__name__ for s in sim.sources}
    assert {"UniformCurrentSource", "PointDipole", "ModeSource", "PlaneWave", "TFSF"} <= kinds
    assert {"GaussianBeam", "AstigmaticGaussianBeam", "CustomFieldSource", "CustomCurrentSource"} <= kinds
    assert not any(isinstance(s, td.Unsupported) for s in sim.sources)
    # (round 4: Medium2D and the custom dispersive media parse into their own classes; nothing in the sample's structures is left
    #  as a placeholder)
    left = {i: st.medium.type for i, st in enumerate(sim.structures) if isinstance(st.medium, td.Unsupported)}
    assert left == {28: "Medium with 'nonlinear_spec'", 29: "Medium with 'nonlinear_spec'"}       # (physics this solver does not model: raises when used)
    assert {"Medium2D", "CustomDrude", "CustomLorentz", "CustomDebye", "CustomPoleResidue", "CustomSellmeier"} <= {st.medium.type for st in sim.structures}
    with pytest.raises(Exception, match="nonlinear_spec"):
        sim.structures[28].medium.fail()
    from tidy3d_amd.exceptions import SetupError, Tidy3dNotImplementedError
    with pytest.raises(SetupError, match="hdf5"):
        D.make_boundaries(sim)        # the AutoGrid axis asks the TriangleMesh for its bounds: no data in JSON
    h5 = REF_SAMPLE[:-5] + ".h5"
    full = td.Simulation.from_file(h5)          # the same simulation with its datasets
    assert isinstance(full.structures[8].geometry, td.TriangleMesh) and full.structures[8].geometry.bounds[0][0] == -1.5
    with pytest.raises(Tidy3dNotImplementedError):
        D.discretize(full, n_steps=2)           # data on unstructured grids (structures 22 - 27): named when used


GEO = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "geometry_golden.json")))


@pytest.mark.parametrize("i", range(len(GEO["geometry"])))
def test_slanted_cylinder_and_transformed_inside_match_reference(i):
    """Slanted ``Cylinder`` (ref primitives.py:600-633, :720-737) and ``Transformed`` (ref geometry/
    base.py:2495-2632): ``inside`` and ``bounds`` recorded from the reference's own classes
    (tests/golden/make_geometry_golden.py)."""
    rec = GEO["geometry"][i]
    g = td.parse(rec["json"])
    p = np.array(GEO["points"])
    got = "".join("1" if b else "0" for b in np.asarray(g.inside(p[:, 0], p[:, 1], p[:, 2])).reshape(-1))
    assert got == rec["inside"]
    assert 0 < got.count("1") < len(got)
    np.testing.assert_allclose(np.array(g.bounds), np.array(rec["bounds"]), atol=1e-12)
    # broadcastable inputs (what the rasteriser passes) give the same answer as flat ones
    X, Y, Z = p[:7, 0][None, None, :], p[:5, 1][None, :, None], p[:3, 2][:, None, None]
    Xf, Yf, Zf = np.broadcast_arrays(X, Y, Z)
    assert np.array_equal(np.broadcast_to(g.inside(X, Y, Z), Xf.shape), g.inside(Xf.ravel(), Yf.ravel(), Zf.ravel()).reshape(Xf.shape))
