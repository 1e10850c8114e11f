"""K7 parity: the field-decay value, the step a run shuts off at and the divergence flag of the HIP library
against the oracle's (``OracleFdtd.decay / stopped_at / diverged``), on the emulator build here and on the
real library under -m gpu.  Reference behaviour: Simulation.shutoff (ref simulation.py:2089-2096), the
(perc_done, field_decay) progress pair (ref web/core/task_core.py:537), SimulationData.diverged
(ref sim_data.py:909).  W = sum|E|^2 + (mu0/eps0) sum|H|^2, evaluated every ``decay_every`` steps."""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from cases import CASES, DL, PULSE
from oracle.fdtd_numpy import OracleFdtd
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine


def _both(spec, lib, **kw):
    hist_o, hist_g = [], []
    o = OracleFdtd(spec)
    o.run(progress=lambda n, t, d: hist_o.append((n, d)))
    with HipEngine(spec, lib=lib, **kw) as e:
        st = e.run(progress=lambda n, t, d: hist_g.append((n, d)) and False)
    return o, st, np.array(hist_o), np.array(hist_g)


def _check_history(ho, hg, rtol):
    n = min(len(ho), len(hg))
    assert n > 2
    assert np.array_equal(ho[:n, 0], hg[:n, 0])
    assert np.allclose(hg[:n, 1], ho[:n, 1], rtol=rtol, atol=0), np.max(np.abs(hg[:n, 1] / ho[:n, 1] - 1))


def _shutoff_case(name, shutoff, n_steps=400, every=8):
    spec = discretize(CASES[name](), n_steps=n_steps).spec
    spec.shutoff, spec.decay_every = shutoff, every
    spec.decay_ref_step = 60           # the test pulse is over by then
    return spec


def check_shutoff(lib, name, shutoff, **kw):
    spec = _shutoff_case(name, shutoff)
    o, st, ho, hg = _both(spec, lib, **kw)
    assert o.stopped_at is not None and not o.diverged, "the case must reach its shutoff level"
    _check_history(ho, hg, 2e-4)
    assert st.stopped_early == 1 and st.diverged == 0
    assert st.steps_done == o.stopped_at                       # the same step, not merely a close one
    assert abs(st.field_decay / o.decay - 1) < 2e-4
    return o, st


def check_divergence(lib, **kw):
    """A time step 1.6 x above the Courant limit: the fields grow by orders of magnitude per evaluation and
    overflow fp32 long before fp64 — so the oracle runs in fp32 arithmetic here, and the two must raise the flag at
    the same evaluation (the energy sums differ in rounding only)."""
    spec = discretize(CASES["pec_box_vec"](), n_steps=600).spec
    spec.dt *= 1.6
    spec.shutoff, spec.decay_every, spec.decay_ref_step = 1e-5, 8, 40
    for s in spec.sources:                                    # (waveforms were sampled with the stable dt: irrelevant here)
        pass
    hist_g = []
    o = OracleFdtd(spec, dtype=np.float32)
    with np.errstate(all="ignore"):
        o.run()
    with HipEngine(spec, lib=lib, **kw) as e:
        st = e.run(progress=lambda n, t, d: hist_g.append((n, d)) and False)
    assert o.diverged and st.diverged == 1 and st.stopped_early == 0
    assert abs(st.steps_done - o.stopped_at) <= spec.decay_every, (st.steps_done, o.stopped_at)
    assert st.steps_done < spec.n_steps
    return o, st


# ---------------------------------------------------------------- CPU: the HIP sources under the emulator
@pytest.mark.parametrize("name,shutoff", [("pml_box", 1e-2), ("media_mix", 3e-1), ("absorber_mix", 3e-2), ("drude_in_pml", 3e-2)])
def test_emu_shutoff_matches_oracle(name, shutoff, emu_lib):
    check_shutoff(emu_lib, name, shutoff)


def test_emu_divergence_flag(emu_lib):
    check_divergence(emu_lib)


def test_emu_decay_is_repeatable_and_counts_h(emu_lib):
    """Two runs give the same bits (fixed-order reduction, no atomics); in a lossless PEC cavity the decay stays
    level once the source is off because the magnetic energy is counted (the electric energy alone of a few
    standing modes swings widely twice per period)."""
    spec = _shutoff_case("pec_box_vec", 0.0, n_steps=200, every=4)
    runs = []
    for _ in range(2):
        h = []
        with HipEngine(spec, lib=emu_lib) as e:
            e.run(progress=lambda n, t, d: h.append(d) and False)
        runs.append(np.array(h))
    assert np.array_equal(runs[0], runs[1])
    late = runs[0][len(runs[0]) // 2:]
    assert late.max() / late.min() < 1.25


# ---------------------------------------------------------------- GPU: the product library
@pytest.mark.gpu
@pytest.mark.parametrize("name,shutoff", [("pml_box", 1e-2), ("media_mix", 3e-1), ("absorber_mix", 3e-2), ("drude_in_pml", 3e-2)])
def test_gpu_shutoff_matches_oracle(name, shutoff, hip_lib):
    check_shutoff(hip_lib, name, shutoff)


@pytest.mark.gpu
def test_gpu_divergence_flag(hip_lib):
    check_divergence(hip_lib)


@pytest.mark.gpu
def test_gpu_tfsf_and_shutoff_bitwise_repeatable(hip_lib):
    """TFSF box (edge and corner nodes receive several corrections: summed per node in a fixed order, no
    atomics) + CPML + shutoff: two runs of a fresh engine agree bit for bit in every monitor, in the final
    fields, in the decay history and in the step the run stops at."""
    spec = discretize(CASES["tfsf_box"](), n_steps=500).spec
    spec.shutoff, spec.decay_every, spec.decay_ref_step = 2e-2, 8, 60
    res = []
    for _ in range(2):
        h = []
        with HipEngine(spec, lib=hip_lib) as e:
            st = e.run(progress=lambda n, t, d: h.append((n, d)) and False)
            res.append((st.steps_done, np.array(h), e.results(), [e.get_field(c) for c in range(6)]))
    assert res[0][0] == res[1][0] and res[0][0] < spec.n_steps
    assert np.array_equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
    for a, b in zip(res[0][3], res[1][3]):
        assert np.array_equal(a, b)
