"""The code path every LARGE CPML run takes, under bit-level test on real hardware ("weak" 1 and 2).

For >= 2^24 cells a step with CPML goes out as three concurrent launches on two streams (interior tiles: x-only
instantiation on the main stream; z-slab planes and y-edge tile rows: all-axes instantiation on the second stream).
The small parity cases never reach it: they are one workgroup column wide (tile_x == 0 always) and default to ONE launch.
Here:

  (a) a 520 x 72 x 72 grid — three x-tiles (256 + 256 + 8 cells), odd x layer counts 5 / 3, CPML on all six faces,
      a Lorentz sphere, a lossy box and a PEC box — forced onto the three-launch split with the recursions
      inside the sweep (mask 7), y / z inside and x as slab kernels (6), all as slab kernels (0), as ONE launch
      (the small-grid default) and as the two-pass kernels: all bit-identical, and <= 2e-5 from the fp64 oracle;
  (b) the bench V2 spec at BASELINE's full 512^3 (materials + 12 CPML layers on six faces, random initial fields):
      the default path (three launches, two streams) three times over — run-to-run identical bits: no stream race —
      and identical to the slab-kernel placement and to the two-pass kernels;
  (c) the bench V0 spec (random +-1e-3 initial fields, PEC walls, a dipole) at 128^3 x 40 steps against the oracle
      <= 1e-5, and at 512^3 x 12 steps fused == two-pass bit for bit.
"""
import numpy as np
import pytest

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

from cases import DL, PULSE, rel_err

pytestmark = pytest.mark.gpu


def two_x_tile_cpml_sim(N=(512, 64, 64)):
    sx, sy, sz = (n * DL for n in N)
    structures = [
        td.Structure(geometry=td.Sphere(center=(0.35 * sx, 0.05, 0.1), radius=0.9),
                     medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13)])),
        # a lossy bar that runs through the x tile boundary at cell 256 and into the low-x CPML
        td.Structure(geometry=td.Box(center=(-0.2 * sx, -0.4, -0.3), size=(0.7 * sx, 0.8, 0.6)),
                     medium=td.Medium(permittivity=3.0, conductivity=0.02)),
        td.Structure(geometry=td.Box(center=(0.1 * sx, 0.6, 0.5), size=(0.5, 0.4, 0.3)), medium=td.PEC)]
    bspec = td.BoundarySpec(x=td.Boundary(minus=td.PML(num_layers=5), plus=td.PML(num_layers=3)),
                            y=td.Boundary.pml(num_layers=4),
                            z=td.Boundary(minus=td.PML(num_layers=3), plus=td.PML(num_layers=5)))
    # dipoles next to the x tile boundaries (cells 256 and 512 of the 520) and next to both x slabs
    sources = [td.PointDipole(center=(-0.5 * sx + 251.3 * DL, 0.13, 0.07), source_time=PULSE, polarization="Ez"),
               td.PointDipole(center=(-0.5 * sx + 3.2 * DL, -0.21, 0.33), source_time=PULSE, polarization="Hy"),
               td.PointDipole(center=(0.5 * sx - 6.6 * DL, 0.4, -0.5), source_time=PULSE, polarization="Ex"),
               td.PointDipole(center=(0.0, -1.2, 1.1), source_time=PULSE, polarization="Hz")]
    monitors = [td.FieldTimeMonitor(center=(0, 0, 0), size=(td.inf, 0.4, 0), name="t", colocate=False, interval=9),
                td.FieldMonitor(center=(0, 0, 0.2), size=(td.inf, td.inf, 0), freqs=[2.5e14, 3e14], name="f")]
    return td.Simulation(size=(sx, sy, sz), grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12,
                         structures=structures, sources=sources, monitors=monitors, boundary_spec=bspec, shutoff=0)


def _run(spec, lib, variant=L.VARIANT_AUTO, split=None, mask=None, init=None, steps=None, rows=None, zc=0):
    with HipEngine(spec, lib=lib, variant=variant, axis_shift=0, z_chunk=zc) as e:
        if rows:
            e.set_option(L.OPT_ROWS, rows)
        if split is not None:
            e.set_option(L.OPT_PML_SPLIT, split)
        if mask is not None:
            e.set_option(L.OPT_PML_FUSED, mask)
        if init is not None:
            for c in range(6):
                e.set_field(c, init[c])
        e.run(steps)
        return [e.get_field(c) for c in range(6)], e.results()


def test_three_launch_cpml_two_x_tiles_bit_identical_and_oracle(hip_lib):
    from oracle.fdtd_numpy import OracleFdtd
    disc = discretize(two_x_tile_cpml_sim(), n_steps=48)
    spec = disc.spec
    assert spec.shape == (520, 72, 72), spec.shape
    ref_f, ref_m = _run(spec, hip_lib, variant=L.VARIANT_ZMARCH, rows=4, zc=2)
    runs = {"split_mask7": dict(variant=L.VARIANT_FUSED, split=1, mask=7),
            "split_mask6": dict(variant=L.VARIANT_FUSED, split=1, mask=6),
            "slab_kernels": dict(variant=L.VARIANT_FUSED, split=1, mask=0),
            "one_launch": dict(variant=L.VARIANT_FUSED, split=0, mask=7),
            "split_mask7_rows7": dict(variant=L.VARIANT_FUSED, split=1, mask=7, rows=7, zc=5),
            "default": dict()}
    for name, kw in runs.items():
        f, m = _run(spec, hip_lib, **kw)
        for c in range(6):
            assert np.array_equal(f[c], ref_f[c]), (name, c, float(np.abs(f[c] - ref_f[c]).max()))
        for k in ref_m:
            assert np.array_equal(m[k], ref_m[k]), (name, k)
    o = OracleFdtd(spec)
    om = o.run()
    for k in om:
        assert rel_err(ref_m[k], om[k]) < 2e-5, (k, rel_err(ref_m[k], om[k]))
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.H))
    for c in range(3):
        assert np.linalg.norm(ref_f[c] - o.E[c]) / en < 2e-5, c
        assert np.linalg.norm(ref_f[3 + c] - o.H[c]) / hn < 2e-5, c


def _bench_init(n, seed_offset=0):
    out = []
    for c in range(6):
        arr = np.empty((n, n, n), dtype=np.float32)
        for k in range(n):
            arr[k] = np.random.default_rng(c * 100003 + k + seed_offset).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
        out.append(arr)
    return out


def test_bench_v2_spec_512_default_path_repeatable_and_equal_to_slab_and_two_pass(hip_lib):
    """bench.py's `workloads.v2` at full size: what the CPML throughput number is measured on."""
    from bench import build_spec
    n, steps = 512, 12
    spec = build_spec(n, steps + 4, "v2")
    init = _bench_init(n)
    ref, _ = _run(spec, hip_lib, init=init, steps=steps)                       # default: three launches, two streams
    assert all(np.isfinite(x).all() for x in ref) and max(np.abs(x).max() for x in ref) > 0
    for rep in range(2):                                                        # stream races would show as differing bits
        got, _ = _run(spec, hip_lib, init=init, steps=steps)
        for c in range(6):
            assert np.array_equal(got[c], ref[c]), ("repeat", rep, c)
    got, _ = _run(spec, hip_lib, init=init, steps=steps, mask=0)               # CPML as slab kernels around the sweep
    for c in range(6):
        assert np.array_equal(got[c], ref[c]), ("slab kernels", c)
    del got
    got, _ = _run(spec, hip_lib, init=init, steps=steps, variant=L.VARIANT_ZMARCH)
    for c in range(6):
        assert np.array_equal(got[c], ref[c]), ("two-pass", c)


def test_bench_v0_spec_vs_oracle_128_and_two_pass_512(hip_lib):
    """bench.py's headline workload: random initial fields, PEC walls, a dipole."""
    from bench import build_spec
    from oracle.fdtd_numpy import OracleFdtd
    n, steps = 128, 40
    spec = build_spec(n, steps + 4, "v0")
    init = _bench_init(n)
    got, _ = _run(spec, hip_lib, init=init, steps=steps)
    o = OracleFdtd(spec)
    for c in range(3):
        o.E[c][...] = init[c]
        o.H[c][...] = init[3 + c]
    for _ in range(steps):
        o.step()
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.H))
    for c in range(3):
        assert np.linalg.norm(got[c] - o.E[c]) / en < 1e-5, c
        assert np.linalg.norm(got[3 + c] - o.H[c]) / hn < 1e-5, c
    n, steps = 512, 12
    spec = build_spec(n, steps + 4, "v0")
    init = _bench_init(n)
    a, _ = _run(spec, hip_lib, init=init, steps=steps)
    b, _ = _run(spec, hip_lib, init=init, steps=steps, variant=L.VARIANT_ZMARCH)
    for c in range(6):
        assert np.array_equal(a[c], b[c]), c


_STREAMS_SCRIPT = r"""
import json, sys
import torch                                  # FIRST: libfdtd_hip.so then binds to the HIP runtime torch ships (one runtime per process)
idle = [torch.cuda.Stream() for _ in range(8)]
for s_ in idle:                               # make the runtime really create them
    with torch.cuda.stream(s_):
        torch.zeros(16, device="cuda").add_(1.0)
torch.cuda.synchronize()
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from cases import pipelined_slab_case
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine
from tidy3d_amd.lib import load_library
lib = load_library()
disc = discretize(pipelined_slab_case(), n_steps=40)
with HipEngine(disc.spec, lib=lib) as e:
    e.run()
    ref = [e.get_field(c) for c in range(6)]
    ref_m = e.results()
with HipEngine(disc.spec, lib=lib, force_comm=True) as e:
    e.comm_init(e.unique_id())
    st = e.run()
    got = [e.get_field(c) for c in range(6)]
    got_m = e.results()
same = all(np.array_equal(a, b) for a, b in zip(ref, got)) and all(np.array_equal(ref_m[k], got_m[k]) for k in ref_m)
print(json.dumps({"stream_overlap": int(st.stream_overlap), "stream_retries": int(st.stream_retries),
                  "comm_ranks": int(st.comm_ranks), "comm_rank": int(st.comm_rank), "same_bits": bool(same)}))
"""


def test_two_streams_overlap_is_measured_with_idle_torch_streams_alive():
    """with more live streams in the process than the runtime has hardware queues an engine's two
    streams can land on ONE queue and silently serialise (3x slower z-slab steps).  The engine now measures the overlap
    before its first two-stream run, tries fresh streams, and falls back to one stream — flagged in FdtdStats — if none
    overlaps.  Eight idle torch streams alive (own process: torch must initialise HIP before the library does): whatever
    the outcome, it is reported, and the results are the same bits as the single-stream run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, "-c", _STREAMS_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(out)
    assert out["stream_overlap"] in (1, -1), out            # probed: verified, or the flagged fallback
    assert out["comm_ranks"] == 1 and out["comm_rank"] == 0 and out["same_bits"], out


@pytest.mark.parametrize("tb", [8, 16, 4096 + 16])
def test_two_step_slab_schedule_bit_identical_256(hip_lib, tb):
    """FDTD_OPT_TBLOCK on real hardware (single stream, and A / B chains on the two streams): bench V1-like spec at 256^3
    (dielectric sphere, random initial fields, a dipole) == single steps, bit for bit."""
    from bench import build_spec
    n, steps = 256, 14
    spec = build_spec(n, steps + 4, "v1")
    init = _bench_init(n)

    def run(tblock):
        with HipEngine(spec, lib=hip_lib, axis_shift=0) as e:
            e.set_option(L.OPT_TBLOCK, tblock)
            for c in range(6):
                e.set_field(c, init[c])
            st = e.run(steps)
            return [e.get_field(c) for c in range(6)], int(st.two_step_pairs)
    ref, p0 = run(0)
    got, p1 = run(tb)
    assert p0 == 0 and p1 == steps // 2
    for c in range(6):
        assert np.array_equal(got[c], ref[c]), c


@pytest.mark.parametrize("name", ["media_mix", "drude_in_pml", "tfsf_box", "periodic_box", "absorber_mix"])
def test_captured_step_pairs_bit_identical(hip_lib, name):
    """FDTD_OPT_GRAPH: runs of steps without monitor records or decay checks replayed as captured hipGraphs of two steps
    (source kernels read the step counter from device memory) == direct launches, bit for bit: CPML (one launch of the
    all-axes instantiation), ADE, TFSF incident grid, periodic ghost copies, absorber layers, monitors every 7th step,
    decay checks every 16th, two run() calls."""
    from cases import CASES
    fn = CASES[name]
    sim = fn((52, 36, 40)) if name == "periodic_box" else fn(tuple(int(n * 3) for n in fn.__defaults__[0]))
    disc = discretize(sim, n_steps=90)
    disc.spec.decay_every = 16

    def run(graph):
        with HipEngine(disc.spec, lib=hip_lib, axis_shift=0) as e:
            e.set_option(L.OPT_GRAPH, graph)
            st = e.run(41)
            pairs = int(st.graph_pairs)
            st = e.run(49)
            return [e.get_field(c) for c in range(6)], e.results(), pairs + int(st.graph_pairs), int(st.reserved0)
    ref_f, ref_m, p0, _ = run(0)
    got_f, got_m, p1, status = run(1)
    # (pairs are taken on the fused path only: a row length that is not a multiple of 4 runs the two-pass kernels, status 0)
    assert p0 == 0 and (p1 >= 15 if disc.spec.shape[0] % 4 == 0 else status in (0, 1)), (p0, p1, status)
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.array_equal(got_m[k], ref_m[k]), k


def test_captured_step_pairs_small_grid_speed(hip_lib):
    """What the graphs were meant for: grids bound by dependent launches.  Vacuum PEC cubes with a dipole, 2000 steps each
    way; prints us per step.  On this stack (ROCm 7.2) replaying a captured pair is NOT faster than launching its kernels
    (64^3: 20.1 vs 17.4 us per step, profiles/r3i) — which is why FDTD_OPT_GRAPH defaults to off; the test only guards
    against a gross regression of either path."""
    import time
    out = {}
    for n in (64, 128, 200):
        sim = td.Simulation(size=(n * DL,) * 3, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12,
                            sources=[td.PointDipole(center=(0.1, 0, 0), source_time=PULSE, polarization="Ez")], monitors=[],
                            boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()), shutoff=0)
        disc = discretize(sim, n_steps=4400)
        disc.spec.decay_every = 0
        t = {}
        for graph in (0, 1):
            with HipEngine(disc.spec, lib=hip_lib, axis_shift=0) as e:
                e.set_option(L.OPT_GRAPH, graph)
                e.run(200)
                t0 = time.perf_counter()
                e.run(2000)
                t[graph] = (time.perf_counter() - t0) / 2000 * 1e6
        out[n] = t
    print("\n[graphs] us per step, direct vs captured pairs:", {n: (round(v[0], 1), round(v[1], 1)) for n, v in out.items()})
    for n, v in out.items():
        assert v[1] < 1.5 * v[0] and v[0] < 1.5 * v[1], (n, v)


@pytest.mark.parametrize("workload", ["v0", "v1", "v2"])
def test_store_hints_and_address_space_paths_change_nothing_on_the_device(hip_lib, workload):
    """non-temporal stores, global-address-space loads and the deferred E / psi stores exist on the
    device only (the emulator compiles them away).  FDTD_OPT_MEM_HINTS = 0 selects the instantiations without them: same
    bits on the bench specs at 256^3 (V2 forced onto the three-launch split), 12 steps from random fields."""
    from bench import build_spec
    n, steps = 256, 12
    spec = build_spec(n, steps + 4, workload)
    init = _bench_init(n, seed_offset=7)

    def run(hints):
        with HipEngine(spec, lib=hip_lib, axis_shift=0) as e:
            e.set_option(L.OPT_MEM_HINTS, hints)
            e.set_option(L.OPT_PML_SPLIT, 1)
            for c in range(6):
                e.set_field(c, init[c])
            e.run(steps)
            return [e.get_field(c) for c in range(6)]
    a, b = run(1), run(0)
    for c in range(6):
        assert np.array_equal(a[c], b[c]), c


@pytest.mark.parametrize("w,zc,n", [(16, 32, 512), (12, 16, 512), (8, 64, 512), (16, 32, 320), (-1, 0, 256), (-1, 0, 512)])
def test_two_steps_per_sweep_bit_identical_bench_v0(hip_lib, w, zc, n):
    """FDTD_OPT_TWOSTEP on real hardware: bench.py's headline workload (random initial fields, PEC walls, a dipole; two x tiles
    -> the seam kernels) advanced by fused2_step_kernel == single sweeps, bit for bit; odd step count -> one single step."""
    from bench import build_spec
    steps = 13
    spec = build_spec(n, steps + 4, "v0")
    init = _bench_init(n)

    def run(twostep):
        with HipEngine(spec, lib=hip_lib, axis_shift=0) as e:
            e.set_option(L.OPT_TWOSTEP, twostep)
            e.set_option(L.OPT_PLACEMENT_TRIES, 0)
            for c in range(6):
                e.set_field(c, init[c])
            st = e.run(steps)
            return [e.get_field(c) for c in range(6)], int(st.fused2_pairs)
    ref, p0 = run(0)
    got, p1 = run(w + 64 * zc if w > 0 else -1)          # (-1: the library's default, shape by grid size)
    assert p0 == 0 and p1 == steps // 2
    for c in range(6):
        assert np.array_equal(got[c], ref[c]), c


@pytest.mark.parametrize("variant", [10, 11, 12, 13, 14])
def test_prefetch_and_row_exchange_instantiations_on_the_device(hip_lib, variant):
    """The measuring-aid instantiations of round 6 that give CORRECT results (FDTD_OPT_WHATIF = 10 ... 14, csrc/fdtd_kernels2.hpp): part of the
    next plane through LDS by LDS-DMA (10 - 12), E_x of the row above from the wave above (13: the default's form), the sweep without it
    (14).  The LDS-DMA ones raced on the device in their first form — a DMA that hit in L2 landed before the wave's queued ds_read of the
    slice had returned; the emulator cannot see that — so they are held to the normal sweep here, three runs each, 320^3 bench V0."""
    from bench import build_spec
    n, steps = 320, 12
    spec = build_spec(n, steps + 4, "v0")
    init = _bench_init(n)
    with HipEngine(spec, lib=hip_lib, axis_shift=0) as e:
        e.set_option(L.OPT_TWOSTEP, 16 + 64 * 32)
        e.set_option(L.OPT_PLACEMENT_TRIES, 0)

        def run(whatif):
            e.reset()
            e.set_option(L.OPT_WHATIF, whatif)
            for c in range(6):
                e.set_field(c, init[c])
            st = e.run(steps)
            assert int(st.fused2_pairs) == steps // 2
            return [e.get_field(c) for c in range(6)]
        ref = run(0)
        for _ in range(3):
            got = run(variant)
            for c in range(6):
                assert np.array_equal(got[c], ref[c]), (variant, c)
        e.set_option(L.OPT_WHATIF, 0)


@pytest.mark.parametrize("w,zc,n", [(16, 32, 512), (8, 32, 512), (-1, 0, 256)])
def test_two_steps_per_sweep_with_materials_bit_identical_bench_v1(hip_lib, w, zc, n):
    """bench.py's V1 workload (a sub-pixel-averaged dielectric sphere: 93 media; PEC walls, a dipole, random initial fields) on
    the two-step sweep's materials instantiation == single sweeps, bit for bit; prints both speeds."""
    import time
    from bench import build_spec
    steps = 41
    spec = build_spec(n, steps + 4, "v1")
    init = _bench_init(n)

    def run(twostep):
        with HipEngine(spec, lib=hip_lib, axis_shift=0) as e:
            e.set_option(L.OPT_TWOSTEP, twostep)
            e.set_option(L.OPT_PLACEMENT_TRIES, 0)
            for c in range(6):
                e.set_field(c, init[c])
            e.run(1)
            t0 = time.perf_counter()
            st = e.run(steps - 1)
            dt = time.perf_counter() - t0
            return [e.get_field(c) for c in range(6)], int(st.fused2_pairs), dt / (steps - 1) * 1e3
    ref, p0, ms0 = run(0)
    got, p1, ms1 = run(w + 64 * zc if w > 0 else -1)
    print(f"[V1 {n}^3] single sweeps {ms0:.4f} ms per step, two steps per sweep {ms1:.4f} ms ({p1} pairs)")
    assert p0 == 0 and p1 == (steps - 1) // 2
    for c in range(6):
        assert np.array_equal(got[c], ref[c]), c


@pytest.mark.parametrize("workload,n", [("va", 320), ("v1a", 256), ("v4a", 256)])
def test_two_steps_per_sweep_with_absorber_layers_bit_identical(hip_lib, workload, n):
    """An open problem on the two-step sweep: Absorber boundaries (40 layers on all faces; vacuum / a dielectric sphere), damped in
    registers == single sweeps with the damping launches around them, bit for bit; prints both speeds."""
    import time
    from bench import build_spec
    steps = 41
    spec = build_spec(n, steps + 4, workload)
    init = _bench_init(n)

    def run(twostep):
        with HipEngine(spec, lib=hip_lib, axis_shift=0) as e:
            e.set_option(L.OPT_TWOSTEP, twostep)
            e.set_option(L.OPT_PLACEMENT_TRIES, 0)
            for c in range(6):
                e.set_field(c, init[c])
            e.run(1)
            t0 = time.perf_counter()
            st = e.run(steps - 1)
            dt = time.perf_counter() - t0
            return [e.get_field(c) for c in range(6)], int(st.fused2_pairs), dt / (steps - 1) * 1e3, e.results()
    ref, p0, ms0, ref_m = run(0)
    got, p1, ms1, got_m = run(-1)
    print(f"[{workload} {n}^3] single sweeps {ms0:.4f} ms per step, two steps per sweep {ms1:.4f} ms ({p1} pairs)")
    assert p0 == 0 and p1 == (steps - 1) // 2
    for c in range(6):
        assert np.array_equal(got[c], ref[c]), c
    for k in ref_m:                                        # v4a: the flux box's running DFT
        assert np.array_equal(got_m[k], ref_m[k]), k


def test_two_steps_per_sweep_config2_probe_records_bit_identical(hip_lib):
    """BASELINE config[1] (200^3 PEC cavity, dipole, a point FieldTimeMonitor recording EVERY step): the two-step sweep copies the
    probe's samples of the middle step out on the way — same records, same fields as single steps; prints both speeds."""
    import time
    import tidy3d_amd.schema as td
    from tidy3d_amd.discretize import discretize
    n, dl = 200, 0.05
    pulse = td.GaussianPulse(freq0=3.5e13, fwidth=1.2e13)
    sim = td.Simulation(size=(n * dl,) * 3, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                        sources=[td.PointDipole(center=(1.3, -0.7, 2.1), source_time=pulse, polarization="Ez")],
                        monitors=[td.FieldTimeMonitor(center=(-2.1, 1.2, -0.6), size=(0, 0, 0), name="t", colocate=False)],
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()), shutoff=0)
    steps = 2001
    disc = discretize(sim, n_steps=steps)
    disc.spec.decay_every = 0

    def run(twostep):
        with HipEngine(disc.spec, lib=hip_lib) as e:
            e.set_option(L.OPT_TWOSTEP, twostep)
            e.run(1)
            t0 = time.perf_counter()
            st = e.run(steps - 1)
            dt = time.perf_counter() - t0
            return [e.get_field(c) for c in range(6)], e.results(), int(st.fused2_pairs), dt / (steps - 1) * 1e6
    ref, ref_m, p0, us0 = run(0)
    got, got_m, p1, us1 = run(-1)
    print(f"[config2 200^3 + probe] single steps {us0:.1f} us per step, two steps per sweep {us1:.1f} us ({p1} pairs)")
    assert p0 == 0 and p1 == (steps - 1) // 2
    assert np.abs(ref_m["t"]).max() > 0
    for c in range(6):
        assert np.array_equal(got[c], ref[c]), c
    assert np.array_equal(got_m["t"], ref_m["t"])


@pytest.mark.parametrize("w,zc", [(16, 32), (8, 5), (5, 3)])
def test_two_steps_per_sweep_everything_at_once_on_the_device(hip_lib, w, zc):
    """The device-only pieces of the two-step sweep (DPP wave shifts, the vote-based node-table walk, readlane) with everything
    it covers at once: three x tiles (two seams), ragged rows and chunks, PMC walls on the min faces, a lossy bar through a
    seam + sub-pixel sphere + PEC box, electric and magnetic dipoles next to the seams and the walls, probes (E and H components, one in the column
    left of a seam, one on a source node) at intervals 1 / 2 — same fields and records as single sweeps."""
    import tidy3d_amd.schema as td
    from tidy3d_amd.discretize import discretize
    import test_emu_fused2 as T
    N = (520, 37, 45)
    size = tuple(n * T.DL for n in N)
    sim = T._sim(N, monitors=False, structures=T.MEDIA_WIDE, bspec=T.PMC_MIN)
    srcs = list(sim.sources) + [td.PointDipole(center=(-0.5 * size[0] + 0.6 * T.DL, -0.5 * size[1] + 0.4 * T.DL, -0.5 * size[2] + 1.2 * T.DL),
                                               source_time=T.PULSE, polarization="Ey"),
                                td.PointDipole(center=(0.12, 0.03, -0.02), source_time=T.PULSE, polarization="Hy"),
                                td.PointDipole(center=(-0.5 * size[0] + 257.5 * T.DL, 0.02, 0.0), source_time=T.PULSE, polarization="Hz")]
    mons = [td.FieldTimeMonitor(center=(-0.5 * size[0] + 255.4 * T.DL, 0.0, 0.05), size=(0, 0, 0), name="seam", interval=1,
                                fields=["Ex", "Hy", "Hz"], colocate=False),
            td.FieldTimeMonitor(center=(-0.5 * size[0] + 1.1 * T.DL, -0.5 * size[1] + 1.2 * T.DL, -0.5 * size[2] + 0.9 * T.DL), size=(0, 0, 0),
                                name="corner", interval=2, colocate=False),
            td.FieldTimeMonitor(center=(0.3 * size[0] - 0.5 * size[0] + 0.02, 0.01, 0.03), size=(0, 0, 0), name="src", interval=1,
                                fields=["Ez"], colocate=False)]
    mons += [td.FieldMonitor(center=(-0.5 * size[0] + 255.4 * T.DL, 0, 0), size=(0, td.inf, td.inf), freqs=[2.8e14], name="dft_x",
                             fields=["Hx", "Hz", "Ey"]),
             td.FluxMonitor(center=(0, 0, -0.1), size=(td.inf, td.inf, 0), freqs=[3e14, 3.1e14], name="flux")]
    disc = discretize(sim.updated_copy(sources=srcs, monitors=mons), n_steps=61)
    disc.spec.decay_every = 16

    def run(twostep):
        with HipEngine(disc.spec, lib=hip_lib, variant=L.VARIANT_FUSED) as e:
            e.set_option(L.OPT_TWOSTEP, twostep)
            pairs = 0
            for r in (25, 36):
                pairs += int(e.run(r).fused2_pairs)
            return [e.get_field(c) for c in range(6)], e.results(), pairs
    ref_f, ref_m, p0 = run(0)
    got_f, got_m, p1 = run(w + 64 * zc)
    assert p0 == 0 and p1 >= 20, p1
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.abs(ref_m[k]).max() > 0 and np.array_equal(got_m[k], ref_m[k]), k
    # ... and the two-step sweep against the fp64 oracle directly, in every tile shape (fields, probe records and the DFT / flux
    # spectra; fp32 round-off over 61 steps)
    om, o = _oracle_of(disc.spec)
    for k in om:
        assert rel_err(got_m[k], om[k]) < 2e-5, (k, rel_err(got_m[k], om[k]))
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.H))
    for c in range(3):
        assert np.linalg.norm(got_f[c] - o.E[c]) / en < 2e-5, c
        assert np.linalg.norm(got_f[3 + c] - o.H[c]) / hn < 2e-5, c


_ORACLE_CACHE = {}


def _oracle_of(spec):
    """The fp64 oracle's run of a spec, once per test session (the three tile shapes share it)."""
    from oracle.fdtd_numpy import OracleFdtd
    key = id(spec)
    hit = _ORACLE_CACHE.get("last")
    if hit is not None and hit[0] == (spec.shape, spec.n_steps, len(spec.sources), len(spec.monitors), len(spec.media)):
        return hit[1], hit[2]
    o = OracleFdtd(spec)
    om = o.run()
    _ORACLE_CACHE["last"] = ((spec.shape, spec.n_steps, len(spec.sources), len(spec.monitors), len(spec.media)), om, o)
    return om, o


def test_two_steps_per_sweep_absorber_layers_and_flux_dft_vs_oracle(hip_lib):
    """An open problem the two-step sweep covers end to end — Absorber layers on all six faces (damped in registers), a dielectric
    sphere with sub-pixel surface cells, a closed flux box with a running DFT and a probe: pairs == single steps bit for bit, and
    the pairs run <= 2e-5 from the fp64 oracle directly (fields, probe, flux spectra)."""
    import tidy3d_amd.schema as td
    from tidy3d_amd.discretize import discretize
    from cases import DL, PULSE
    N = (300, 44, 40)
    size = tuple(n * DL for n in N)
    bspec = td.BoundarySpec.all_sides(td.Absorber(num_layers=6))
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12,
                        structures=[td.Structure(geometry=td.Sphere(center=(0.3, 0.05, 0.0), radius=0.45), medium=td.Medium(permittivity=3.0))],
                        sources=[td.PointDipole(center=(-0.4, 0.0, 0.05), source_time=PULSE, polarization="Ez"),
                                 td.PointDipole(center=(-0.5 * size[0] + 254.6 * DL, 0.2, -0.1), source_time=PULSE, polarization="Ey")],
                        monitors=[td.FluxMonitor(center=(0.3, 0.05, 0.0), size=(1.6, 1.3, 1.2), freqs=[2.5e14, 3e14, 3.5e14], name="flux"),
                                  td.FieldTimeMonitor(center=(0.9, 0.1, 0.1), size=(0, 0, 0), name="probe", interval=1, colocate=False)],
                        boundary_spec=bspec, shutoff=0)
    disc = discretize(sim, n_steps=70)
    disc.spec.decay_every = 0

    def run(twostep):
        with HipEngine(disc.spec, lib=hip_lib, variant=L.VARIANT_FUSED, axis_shift=0) as e:
            e.set_option(L.OPT_TWOSTEP, twostep)
            st = e.run()
            return [e.get_field(c) for c in range(6)], e.results(), int(st.fused2_pairs)
    ref_f, ref_m, p0 = run(0)
    got_f, got_m, p1 = run(16 + 64 * 16)
    assert p0 == 0 and p1 >= 30, p1
    for c in range(6):
        assert np.array_equal(got_f[c], ref_f[c]), c
    for k in ref_m:
        assert np.abs(ref_m[k]).max() > 0 and np.array_equal(got_m[k], ref_m[k]), k
    from oracle.fdtd_numpy import OracleFdtd
    o = OracleFdtd(disc.spec)
    om = o.run()
    for k in om:
        assert rel_err(got_m[k], om[k]) < 2e-5, (k, rel_err(got_m[k], om[k]))
    en = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.E))
    hn = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in o.H))
    for c in range(3):
        assert np.linalg.norm(got_f[c] - o.E[c]) / en < 2e-5, c
        assert np.linalg.norm(got_f[3 + c] - o.H[c]) / hn < 2e-5, c


def test_randomised_two_step_self_check_on_the_device(hip_lib):
    """scripts/fuzz_twostep.py inside the driver-run suite: 60 seeded random cases — grid shapes with 1 - 3 x tiles, tile shapes,
    walls (PEC, PMC, absorber layers, CPML -> shell pairs), media, dispersive bodies and plane waves across a unit cell (-> z holes
    in the bulk), electric / magnetic dipoles, probes, DFT and flux monitors, decay checks — step pairs == single steps, bit for
    bit (fields and every record)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_twostep", os.path.join(os.path.dirname(__file__), "..", "scripts", "fuzz_twostep.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad, taken = fz.run_cases(60, seed=4, lib=hip_lib, quiet=True)
    assert bad == 0
    assert taken >= 42, taken          # (cases whose random features keep single steps are few)


def test_randomised_shell2_pairs_on_the_device(hip_lib):
    """scripts/fuzz_shell2.py inside the driver-run suite: 40 seeded random CPML-walled simulations (1 - 3 x tiles of the bulk, CPML /
    StablePML of random thickness, PEC, PMC on min faces, bodies through the layers, random initial fields, dipoles deep inside the
    bulk, monitors inside it, split runs, the three launch forms and random tile shapes of the boxes) — the shell advanced by
    shell2_step_kernel (two steps per sweep, psi carried and ping-ponged) == single steps, bit for bit (fields and every record)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_shell2", os.path.join(os.path.dirname(__file__), "..", "scripts", "fuzz_shell2.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad, taken = fz.run_cases(40, seed=5, lib=hip_lib, quiet=True)
    assert bad == 0
    assert taken >= 36, taken


def test_randomised_variant_cross_check_on_the_device(hip_lib):
    """scripts/fuzz_variants.py inside the driver-run suite: 40 seeded random simulations (walls of every kind per face incl. PMC on
    plus faces, CPML / StablePML / absorber layers, periodic axes; dielectric, lossy, PEC, Lorentz and Drude bodies; dipoles of both
    kinds; time / DFT / flux monitors; decay checks; runs cut in two; a second x tile now and then) — the fused sweep == the
    two-pass kernels == (periodic z) a z-slab rank exchanging with itself over RCCL, bit for bit, and <= 2e-5 from the fp64 oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_variants", os.path.join(os.path.dirname(__file__), "..", "scripts", "fuzz_variants.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad, far, worst = fz.run_cases(40, seed=7, lib=hip_lib, quiet=True, big=True)
    assert bad == 0 and far == 0, (bad, far, worst)


@pytest.mark.gpu
def test_randomised_metasurface_cells_in_shell2_pairs_on_the_device(hip_lib):
    """scripts/fuzz_cell.py on the device (its unattended runs: 480 of 480, profiles/r5/r5zj): BASELINE config 5's kind of problem in
    small — periodic x and / or y (the boxes wrap x through halo lanes, the rows next to a y wrap take single steps), CPML on z, a
    Drude / Lorentz body whose planes are z holes of the bulk, a plane wave or a current sheet, flux planes through the wrap — in
    shell2 pairs against single steps of the same library: fields and records, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
    import fuzz_cell
    bad, taken = fuzz_cell.run_cases(24, seed=5, lib=hip_lib, quiet=True)
    assert bad == 0, bad
    assert taken >= 16, taken
