"""GPU parity tests proper (run with -m gpu on a real MI355X): the product library
libfdtd_hip.so, called through its C ABI, against the fp64 oracle on seeded small cases, and —
at BASELINE.json's full 512^3 size — through size-independent bit-exact properties
(translation invariance under periodic boundaries, linearity under power-of-two scaling,
independence of the launch geometry).

Tolerance: fp32 GPU vs fp64 oracle, rel-L2 <= 2e-5 over 60..100 steps (north_star: "stated fp32
tolerance"; SURVEY.md section 7 step 3 asks <= 1e-5 over 200 steps on the vacuum case, checked in
test_vacuum_200_steps)."""
import numpy as np
import pytest

from cases import CASES, DL, PULSE, rel_err, run_case

import tidy3d_amd.schema as td
from tidy3d_amd import lib as L
from tidy3d_amd.constants import C_0
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine

pytestmark = pytest.mark.gpu
TOL = 2e-5


def test_library_is_native_and_sees_a_gpu(hip_lib):
    assert hip_lib.path.endswith("libfdtd_hip.so")
    assert hip_lib.dll.fdtd_device_count() >= 1


@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_oracle_small(name, hip_lib):
    worst, disc = run_case(name, hip_lib, n_steps=60)
    assert worst < TOL, (name, disc.spec.shape, worst)


@pytest.mark.parametrize("name", ["pec_box_vec", "media_mix", "drude_in_pml", "pml_box", "absorber_mix", "bloch_xy_pml_z",
                                  "bloch_planewave"])
def test_gpu_matches_oracle_medium(name, hip_lib):
    """~3x larger grids (several workgroups per axis, several z-chunks), 100 steps."""
    worst, disc = run_case(name, hip_lib, n_steps=100, scale=3, z_chunk=8)
    assert worst < TOL, (name, disc.spec.shape, worst)


def test_vacuum_200_steps(hip_lib):
    """<= 1e-5 rel-L2 over 200 steps, vacuum + PEC + dipole."""
    sim = td.Simulation(size=(64 * DL,) * 3, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12,
                        sources=[td.PointDipole(center=(0.1, 0, 0), source_time=PULSE, polarization="Ez")],
                        monitors=[td.FieldTimeMonitor(center=(0, 0, 0), size=(1.0, 1.0, 0), name="t",
                                                      interval=20, colocate=False)],
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()), shutoff=0)
    disc = discretize(sim, n_steps=200)
    from oracle.fdtd_numpy import OracleFdtd
    ref = OracleFdtd(disc.spec).run()["t"]
    with HipEngine(disc.spec, lib=hip_lib) as e:
        e.run()
        got = e.results()["t"]
    assert rel_err(got, ref) < 1e-5


def _periodic_spec(n, src_cell, n_steps, amplitude=1.0):
    dl = 0.0625      # power of two: every boundary / step is exact, so the grid is bitwise uniform
    # 0.1 dl off the Ez node (x, y on boundaries, z on a centre): the nearest node is unambiguous
    c = ((src_cell[0] + 0.1) * dl - n * dl / 2, (src_cell[1] + 0.1) * dl - n * dl / 2,
         (src_cell[2] + 0.6) * dl - n * dl / 2)
    pulse = td.GaussianPulse(freq0=2e14, fwidth=1e14, amplitude=amplitude)
    sim = td.Simulation(size=(n * dl,) * 3, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                        sources=[td.PointDipole(center=c, source_time=pulse, polarization="Ez",
                                                interpolate=False)],
                        monitors=[], shutoff=0,
                        boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(),
                                                      z=td.Boundary.periodic()))
    d = discretize(sim, n_steps=n_steps)
    assert d.spec.shape == (n, n, n)
    return d.spec


def _final_fields(spec, lib, **kw):
    with HipEngine(spec, lib=lib, **kw) as e:
        e.run()
        return [e.get_field(c) for c in range(6)]


def test_translation_invariance_512_bit_exact(hip_lib):
    """512^3 periodic: moving the source by (67, 5, 37) cells moves every field by exactly the same
    shift, bit for bit — any indexing slip at wavefront, workgroup, z-chunk or ghost-plane edges
    would break this."""
    n, steps, shift = 512, 24, (67, 5, 37)
    a = _final_fields(_periodic_spec(n, (100, 200, 300), steps), hip_lib)
    p2 = tuple((p + s) % n for p, s in zip((100, 200, 300), shift))
    b = _final_fields(_periodic_spec(n, p2, steps), hip_lib)
    assert max(np.abs(x).max() for x in a) > 0
    for fa, fb in zip(a, b):
        assert np.array_equal(np.roll(fa, (shift[2], shift[1], shift[0]), axis=(0, 1, 2)), fb)


def test_linearity_power_of_two_512_bit_exact(hip_lib):
    """512^3: scaling the source by 4 scales every field by exactly 4 (fp32 exponent shift)."""
    n, steps = 512, 16
    a = _final_fields(_periodic_spec(n, (256, 256, 256), steps, amplitude=1.0), hip_lib)
    b = _final_fields(_periodic_spec(n, (256, 256, 256), steps, amplitude=4.0), hip_lib)
    for fa, fb in zip(a, b):
        assert np.array_equal(4.0 * fa, fb)


@pytest.mark.parametrize("zchunk,rows", [(1, 1), (7, 2), (64, 8), (512, 4)])
def test_launch_geometry_bit_identical(hip_lib, zchunk, rows):
    from cases import media_mix
    sim = media_mix((52, 44, 36))
    disc = discretize(sim, n_steps=40)
    outs = []
    for zc, r in ((32, 4), (zchunk, rows)):
        with HipEngine(disc.spec, lib=hip_lib, z_chunk=zc) as e:
            e.set_option(L.OPT_ROWS, r)
            e.run()
            outs.append([e.get_field(c) for c in range(6)])
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


def test_config2_vacuum_200_cube_cavity_resonances(hip_lib):
    """BASELINE config[1]: vacuum 200^3 Yee cells, PointDipole + PEC walls + FieldTimeMonitor,
    curl stencil only.  The spectrum of the probe must peak at the eigenfrequencies of the
    *discrete* PEC cavity,  sin^2(w dt/2)/(c dt)^2 = sum_i sin^2(k_i d/2)/d^2, k_i = m_i pi/L."""
    n, dl = 200, 0.05
    pulse = td.GaussianPulse(freq0=3.5e13, fwidth=1.2e13)
    sim = td.Simulation(size=(n * dl,) * 3, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                        sources=[td.PointDipole(center=(1.3, -0.7, 2.1), source_time=pulse, polarization="Ez")],
                        monitors=[td.FieldTimeMonitor(center=(-2.1, 1.2, -0.6), size=(0, 0, 0), name="t",
                                                      fields=["Ez"], colocate=False)],
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()), shutoff=0)
    steps = 12000
    disc = discretize(sim, n_steps=steps)
    with HipEngine(disc.spec, lib=hip_lib) as e:
        st = e.run()
        raw = e.results()["t"]
    assert not st.diverged
    sig = raw[:, 0].reshape(steps, -1)[:, 0].astype(np.float64)
    dt = disc.spec.dt
    pad = 8 * steps
    spec_ = np.abs(np.fft.rfft(sig * np.hanning(steps), pad))
    f = np.fft.rfftfreq(pad, dt)
    Lc = n * dl
    modes = []
    for m in range(0, 4):
        for q in range(0, 4):
            for p in range(0, 4):
                if (m > 0) + (q > 0) + (p > 0) < 2:
                    continue
                k = np.pi * np.array([m, q, p]) / Lc
                s = np.sum(np.sin(k * dl / 2) ** 2 / dl ** 2)
                modes.append(2 / dt * np.arcsin(C_0 * dt * np.sqrt(s)) / 2 / np.pi)
    modes = np.array(sorted(set(np.round(modes, 3))))
    from scipy.signal import find_peaks
    pk, _ = find_peaks(spec_, height=spec_.max() * 0.05)
    assert len(pk) >= 3
    res = 1.0 / (steps * dt)
    for x in f[pk]:
        assert np.min(np.abs(modes - x)) < 1.5 * res, (x, modes[:8])


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_rccl_self_exchange_matches_ghost_copy(hip_lib):
    """The RCCL ghost-plane path and the two-stream boundary/interior schedule on real hardware
    with the one GPU a test box has: a 1-rank communicator exchanges the periodic-z planes with
    itself (ncclSend/ncclRecv to self) and must reproduce the plain ghost-copy run bit for bit."""
    from cases import periodic_box, media_mix
    import tidy3d_amd.schema as tds
    sim = periodic_box((40, 36, 32))
    disc = discretize(sim, n_steps=50)
    with HipEngine(disc.spec, lib=hip_lib) as e:
        e.run()
        ref = [e.get_field(c) for c in range(6)]
        ref_m = e.results()
    with HipEngine(disc.spec, lib=hip_lib, force_comm=True) as e:
        e.comm_init(e.unique_id())
        e.run()
        got = [e.get_field(c) for c in range(6)]
        got_m = e.results()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    for k in ref_m:
        assert np.array_equal(ref_m[k], got_m[k])


@pytest.mark.parametrize("schedule", ["cpml_three_launches", "shell_pairs_r4", "shell2_pairs", "slab_rank_fused", "slab_rank_fused_pml_in_sweep",
                                      "slab_rank_pairs", "slab_rank_cpml_pairs", "slab_rank_cpml_pairs_boxes_behind", "slab_rank_cpml_pairs_boxes_in_front",
                                      "slab_rank_cpml_pairs_boxes_third_stream", "slab_rank_two_pass"])
def test_schedules_do_not_depend_on_stream_timing(schedule, hip_lib):
    """every schedule that splits a step between the two streams, run normally (three times: a race shows as
    run-to-run differences too) and with FDTD_OPT_DEBUG_SYNC — a device-wide synchronisation in front of and behind every launch
    group, so that no two launches ever overlap.  A schedule with a missing cross-stream edge (the two races the round-4 device
    fuzz found were of that kind) gives different bits in the two modes; here: the same, fields and records."""
    from cases import pipelined_slab_case
    opts, comm = {}, False
    # (round 6: where the shell's boxes of a CPML slab-rank pair go — FDTD_OPT_SLAB_BOXES_FIRST 0 / 1 / 2; the default takes the third stream
    #  on this slab of 132 planes.  Under FDTD_OPT_DEBUG_SYNC the boxes stay on the main stream.)
    boxes = {"slab_rank_cpml_pairs_boxes_behind": 0, "slab_rank_cpml_pairs_boxes_in_front": 1, "slab_rank_cpml_pairs_boxes_third_stream": 2}.get(schedule)
    if boxes is not None:
        schedule = "slab_rank_cpml_pairs"
    if schedule in ("cpml_three_launches", "shell_pairs_r4", "shell2_pairs"):
        N = (96, 72, 64)
        size = tuple(n * DL for n in N)
        sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=DL), run_time=1e-12, shutoff=0,
                            structures=[td.Structure(geometry=td.Sphere(center=(0.4, 0.1, -0.2), radius=0.7), medium=td.Medium(permittivity=2.5)),
                                        td.Structure(geometry=td.Box(center=(0, -1.0, 0), size=(td.inf, 0.5, 0.6)), medium=td.Medium(permittivity=3.0, conductivity=0.02))],
                            sources=[td.PointDipole(center=(0.05, 0.02, 0.03), source_time=PULSE, polarization="Ez"),
                                     td.PointDipole(center=(-0.1, 0.06, -0.05), source_time=PULSE, polarization="Hy")],
                            monitors=[td.FieldTimeMonitor(center=(0.1, 0.1, 0.05), size=(0, 0, 0), name="probe", interval=3, colocate=False),
                                      td.FieldMonitor(center=(0, 0, 0.1), size=(1.0, 0.8, 0), freqs=[2.5e14, 3e14], name="f")],
                            boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=6)))
        disc = discretize(sim, n_steps=100)
        disc.spec.decay_every = 0
        opts = {"cpml_three_launches": {L.OPT_TWOSTEP: 0, L.OPT_PML_SPLIT: 1},
                "shell_pairs_r4": {L.OPT_TWOSTEP: 8 + 64 * 8, L.OPT_SHELL_PAIRS: 1, L.OPT_SHELL2: 0},
                "shell2_pairs": {L.OPT_TWOSTEP: 8 + 64 * 8, L.OPT_SHELL_PAIRS: 1, L.OPT_SHELL2: 1}}[schedule]
    else:
        comm = True
        if schedule == "slab_rank_pairs":
            from cases import slab_pairs_box
            disc = discretize(slab_pairs_box((72, 60, 132), periodic_z=True), n_steps=80)
            disc.spec.decay_every = 0
        elif schedule == "slab_rank_cpml_pairs":      # round 5: a rank with x / y layers in shell2 pairs, the planes next to the cuts as z holes
            from cases import slab_pairs_pml_box
            disc = discretize(slab_pairs_pml_box((72, 60, 132), layers=(6, 5, 0), periodic_z=True), n_steps=80)
            disc.spec.decay_every = 0
        else:
            disc = discretize(pipelined_slab_case(), n_steps=90)
            disc.spec.decay_every = 16
        opts = {"slab_rank_fused": {}, "slab_rank_fused_pml_in_sweep": {L.OPT_PML_FUSED: 7, L.OPT_BND_PLANES: 3},
                "slab_rank_pairs": {L.OPT_TWOSTEP: 8 + 64 * 8}, "slab_rank_cpml_pairs": {L.OPT_TWOSTEP: 8 + 64 * 8, L.OPT_PML_FUSED: 7},
                "slab_rank_two_pass": {}}[schedule]
        if boxes is not None:
            opts[L.OPT_SLAB_BOXES_FIRST] = boxes

    def run(debug_sync):
        kw = dict(force_comm=True) if comm else dict(axis_shift=0)
        if schedule == "slab_rank_two_pass":
            kw["variant"] = L.VARIANT_ZMARCH
        with HipEngine(disc.spec, lib=hip_lib, **kw) as e:
            if comm:
                e.comm_init(e.unique_id())
            for k, v in opts.items():
                e.set_option(k, v)
            e.set_option(L.OPT_DEBUG_SYNC, debug_sync)
            st = e.run()
            return [e.get_field(c) for c in range(6)], e.results(), int(st.fused2_pairs), int(st.shell2_pairs)
    ref_f, ref_m, p_ref, q_ref = run(1)
    assert max(float(np.abs(f).max()) for f in ref_f) > 0
    if schedule in ("shell_pairs_r4", "shell2_pairs", "slab_rank_pairs", "slab_rank_cpml_pairs"):
        assert p_ref > 10, p_ref
    if schedule in ("shell2_pairs", "slab_rank_cpml_pairs"):
        assert q_ref > 10, q_ref
    if schedule == "slab_rank_cpml_pairs":        # ... and the same bits as the plain one-GPU run of the same problem (periodic z, no exchange)
        with HipEngine(disc.spec, lib=hip_lib, axis_shift=0) as e:
            e.set_option(L.OPT_TWOSTEP, 0)
            e.run()
            for c in range(6):
                assert np.array_equal(e.get_field(c), ref_f[c]), c
    for rep in range(3):
        f, m, p, q = run(0)
        assert (p, q) == (p_ref, q_ref)
        for c in range(6):
            assert np.array_equal(f[c], ref_f[c]), (schedule, rep, c, float(np.abs(f[c] - ref_f[c]).max()))
        for k in ref_m:
            assert np.array_equal(np.asarray(m[k]), np.asarray(ref_m[k])), (schedule, rep, k)


@pytest.mark.parametrize("bnd,pml_fused", [(0, -1), (3, -1), (0, 7), (5, 7)])
def test_pipelined_slab_schedule_with_corrections_on_real_streams(hip_lib, bnd, pml_fused):
    """The pipelined z-slab schedule under REAL stream concurrency with everything it splits
    between the two streams: x/y CPML slabs, ADE (Lorentz sphere), a lossy box, electric and
    magnetic dipoles, a plane wave (TFSF machinery: the replica of the 1-D incident grid on the comm
    stream), time and DFT monitors, field-decay checks (joined tails).  Periodic z, one rank,
    RCCL exchange with itself == the plain single-stream run, bit for bit.  pml_fused = 7: the CPML recursions inside
    the sweeps of the slab rank (the H-side psi of the ghost plane exchanged with the ghost fields, edge tiles on a
    third stream) instead of slab kernels."""
    from cases import pipelined_slab_case
    sim = pipelined_slab_case()
    disc = discretize(sim, n_steps=90)
    disc.spec.decay_every = 16
    assert disc.spec.tfsf, "the plane wave must go through the incident-grid machinery"
    with HipEngine(disc.spec, lib=hip_lib) as e:
        e.run()
        ref = [e.get_field(c) for c in range(6)]
        ref_m = e.results()
    with HipEngine(disc.spec, lib=hip_lib, force_comm=True) as e:
        assert e.variant == L.VARIANT_FUSED
        e.comm_init(e.unique_id())
        if bnd:
            e.set_option(L.OPT_BND_PLANES, bnd)
        if pml_fused >= 0:
            e.set_option(L.OPT_PML_FUSED, pml_fused)
        e.run(40)
        e.run(50)                      # a second call re-primes the pipeline
        got = [e.get_field(c) for c in range(6)]
        got_m = e.results()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    for k in ref_m:
        assert np.array_equal(ref_m[k], got_m[k]), k


def test_two_pass_slab_rank_records_before_the_comm_stream_moves_on(hip_lib):
    """A race scripts/fuzz_variants.py found on the device (round 4; the CPU emulator runs the streams in issue order and cannot
    see it): in the two-pass z-slab schedule the comm stream advanced H of the slab's top plane — H-side corrections and
    update, waiting for the E interior of the LAST step only — while the main stream's monitor record of the new step was still
    reading H^{n-1/2} of that plane.  A volume time monitor that reaches the top plane came back with that plane's H
    half-sample from the wrong side of the update in one run out of a few.  The record now posts an event the comm stream waits
    for.  Periodic box (x periodic and not a multiple of four cells: the two-pass kernels), H and E dipoles, records every 4th
    step, six runs == the plain run, bit for bit."""
    import tidy3d_amd.schema as tds
    from cases import DL, PULSE
    N = (30, 35, 18)
    size = tuple(n * DL for n in N)
    per = tds.Boundary.periodic()
    sim = tds.Simulation(
        size=size, grid_spec=tds.GridSpec.uniform(dl=DL), run_time=1e-12, shutoff=0,
        structures=[tds.Structure(geometry=tds.Sphere(center=(0.2, 0.1, 0.1), radius=0.3), medium=tds.Drude(eps_inf=1.5, coeffs=[(6e14, 5e13)]))],
        sources=[tds.PointDipole(center=(0.1, -0.2, 0.3), source_time=PULSE, polarization="Hx"),
                 tds.PointDipole(center=(-0.3, 0.2, -0.1), source_time=PULSE, polarization="Ey")],
        monitors=[tds.FieldTimeMonitor(center=(0.4, -0.5, 0.25), size=(0.7, 0.6, 0.5), name="vol", interval=4, colocate=False),
                  tds.FieldTimeMonitor(center=(0, 0, -0.3), size=(0, 0, 0), name="probe", interval=3, colocate=False)],
        boundary_spec=tds.BoundarySpec(x=per, y=tds.Boundary(minus=tds.PECBoundary(), plus=tds.PECBoundary()), z=per))
    disc = discretize(sim, n_steps=41)
    vol = [m for m in disc.spec.monitors if m.name == "vol"][0]
    assert vol.hi[2] == disc.spec.shape[2]                      # the monitor reaches the slab's top plane
    with HipEngine(disc.spec, lib=hip_lib, variant=L.VARIANT_ZMARCH, axis_shift=0) as e:
        e.run()
        ref_m = e.results()
    assert float(np.abs(ref_m["vol"]).max()) > 0
    for rep in range(6):
        with HipEngine(disc.spec, lib=hip_lib, variant=L.VARIANT_ZMARCH, force_comm=True) as e:
            assert e.variant == L.VARIANT_ZMARCH
            e.comm_init(e.unique_id())
            e.run(24)
            e.run(17)
            got_m = e.results()
        for k in ref_m:
            assert np.array_equal(ref_m[k], got_m[k]), (rep, k)


@pytest.mark.parametrize("variant", ["fused", "two_pass"])
def test_pmc_plus_face_on_a_slab_rank_on_real_streams(hip_lib, variant):
    """PMC on a PLUS face (x) of a z-slab rank: the mirror images beyond the wall are refreshed plane range by plane range on
    the stream that owns the planes — boundary planes on the comm stream in front of their H-side corrections, the interior on
    the main one.  Periodic z, one rank, RCCL exchange with itself == the plain run, bit for bit (the 2-rank gloo run of
    tests/test_dist_gloo.py has the z wall on the last rank too)."""
    import tidy3d_amd.schema as tds
    from cases import DL, PULSE
    N = (44, 36, 40)
    size = tuple(n * DL for n in N)
    sim = tds.Simulation(
        size=size, grid_spec=tds.GridSpec.uniform(dl=DL), run_time=1e-12, shutoff=0,
        structures=[tds.Structure(geometry=tds.Sphere(center=(0.5 * size[0] - 0.1, 0, 0.1), radius=0.3), medium=tds.Medium(permittivity=2.5, conductivity=0.01))],
        # (a dipole within two cells of the wall, in the slab's top planes: its mirror image lies beyond the stored image cells, so
        #  refreshing them is more than a no-op there — the order of refresh and update is visible in the bits)
        sources=[tds.PointDipole(center=(0.5 * size[0] - 0.06, -0.07, 0.5 * size[2] - 0.03), source_time=PULSE, polarization="Ez"),
                 tds.PointDipole(center=(0.5 * size[0] - 0.16, 0.07, 0.01), source_time=PULSE, polarization="Ey"),
                 tds.PointDipole(center=(-0.1, 0.07, 0.5 * size[2] - 0.08), source_time=PULSE, polarization="Hx")],
        monitors=[tds.FieldTimeMonitor(center=(0.3, 0.05, 0.2), size=(0.5, 0.2, 0.4), name="t", colocate=False, interval=7)],
        boundary_spec=tds.BoundarySpec(x=tds.Boundary(minus=tds.PML(num_layers=4), plus=tds.PMCBoundary()), y=tds.Boundary.pml(num_layers=3),
                                       z=tds.Boundary.periodic()))
    disc = discretize(sim, n_steps=70)
    disc.spec.decay_every = 16
    assert disc.spec.mirror_plus[0] >= 0
    v = L.VARIANT_FUSED if variant == "fused" else L.VARIANT_ZMARCH
    with HipEngine(disc.spec, lib=hip_lib, variant=v, axis_shift=0) as e:
        e.run()
        ref = [e.get_field(c) for c in range(6)]
        ref_m = e.results()
    assert max(float(np.abs(f).max()) for f in ref) > 0
    N_wall = disc.spec.mirror_plus[0]
    for rep in range(4):                        # (a race shows in one run out of a few)
        with HipEngine(disc.spec, lib=hip_lib, variant=v, force_comm=True) as e:
            e.comm_init(e.unique_id())
            e.run(30)
            e.run(40)
            got = [e.get_field(c) for c in range(6)]
            got_m = e.results()
        for c, (a, b) in enumerate(zip(ref, got)):
            assert np.array_equal(a[:, :, :N_wall], b[:, :, :N_wall]), (rep, c)      # (inside the wall: the images are refreshed before they are read)
        for k in ref_m:
            assert np.array_equal(ref_m[k], got_m[k]), (rep, k)


@pytest.mark.parametrize("rows,zc", [(7, 16), (3, 5), (15, 64)])
def test_fused_sweep_equals_two_pass_bit_for_bit(hip_lib, rows, zc):
    """The fused single-sweep kernel and the two-pass kernels perform the same IEEE operations
    per cell (the library is built with -ffp-contract=off), whatever the tiling."""
    from cases import media_mix, periodic_box
    for sim in (media_mix((52, 44, 36)), periodic_box((264, 40, 36))):
        disc = discretize(sim, n_steps=30)
        assert disc.spec.shape[0] % 4 == 0
        outs = []
        for variant, r, z in ((L.VARIANT_ZMARCH, 4, 2), (L.VARIANT_FUSED, rows, zc)):
            with HipEngine(disc.spec, lib=hip_lib, variant=variant, z_chunk=z) as e:
                e.set_option(L.OPT_ROWS, r)
                e.run()
                outs.append([e.get_field(c) for c in range(6)] + list(e.results().values()))
        for x, y in zip(*outs):
            assert np.array_equal(x, y)


_MIE_CACHE = {}


def _mie_run(ppw, hip_lib):
    """BASELINE config[3] at ``ppw`` cells per vacuum wavelength (40 = the 512^3 grid): dielectric sphere (radius 1.4 lambda0, size
    parameter ~8.8: several Mie resonances in the band), PlaneWave TFSF + 12-layer PML, a closed flux box in the scattered-field
    region with a running DFT at 25 frequencies.  The run ends on the field decay (shutoff 1e-5), not on run_time: the resonances
    ring for ~220 periods, and a record cut at 70 periods (what this test did until round 4) moves sigma_sca by up to 4 % at the
    resonance flanks whatever the grid (profiles/r4/r4i_mie_*: 0.95 f0 read -2.8 % after 70 periods, +0.9 % after 140, +1.1 %
    after 280 at lambda0 / 30)."""
    import time
    from tidy3d_amd.analytic import mie_cross_sections
    from tidy3d_amd.data import assemble
    if ppw in _MIE_CACHE:
        return _MIE_CACHE[ppw]
    lam0 = 1.0
    f0 = C_0 / lam0
    dl = lam0 / ppw
    n = (512 - 24) * ppw // 40
    r, eps = 56 * lam0 / 40, 2.56
    freqs = [float(v) * f0 for v in np.linspace(0.85, 1.15, 25)]
    box = 2 * r + lam0
    sim = td.Simulation(
        size=(n * dl,) * 3, grid_spec=td.GridSpec.uniform(dl=dl), run_time=400 / f0,
        structures=[td.Structure(geometry=td.Sphere(radius=r), medium=td.Medium(permittivity=eps))],
        sources=[td.TFSF(center=(0, 0, 0), size=(box,) * 3, source_time=td.GaussianPulse(freq0=f0, fwidth=f0 / 6), injection_axis=2, direction="+")],
        monitors=[td.FluxMonitor(center=(0, 0, 0), size=(box + lam0,) * 3, freqs=freqs, name="sca")],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=12)), shutoff=1e-5)
    t0 = time.time()
    disc = discretize(sim)
    assert disc.spec.shape == (n + 24,) * 3
    t1 = time.time()
    with HipEngine(disc.spec, lib=hip_lib) as e:
        st = e.run()
        raw = e.results()
    t2 = time.time()
    assert not st.diverged and st.stopped_early            # ended on the field decay
    got = assemble(disc, raw, log="")["sca"].flux.values
    _, ana = mie_cross_sections(r, eps, freqs)
    N = disc.spec.shape[0]
    print(f"\n[mie {N}^3, lambda0/{ppw}] setup {t1 - t0:.1f}s, solve {t2 - t1:.1f}s ({st.steps_done} steps, "
          f"{N**3 * st.steps_done / (st.run_ms * 1e-3) / 1e6:.0f} Mcells/s, {int(st.fused2_pairs)} step pairs, reason {int(st.fused2_off_reason)}), "
          f"sigma_sca/analytic - 1 [%] = {np.round((got / ana - 1) * 100, 2)}")
    _MIE_CACHE[ppw] = (got / ana - 1, np.asarray(freqs) / f0)
    return _MIE_CACHE[ppw]


def test_config4_mie_sphere_512_cube(hip_lib):
    """BASELINE config[3]: Mie scattering, dielectric sphere, PlaneWave TFSF + PML, 512^3 cells on one MI355X; scattering
    cross-section from the flux through a box in the scattered-field region (normalised to 1 W/um^2 incident) vs the Mie series
    at 25 frequencies across the band (dl = lambda0/40, lambda/25 inside the sphere; sub-pixel averaged interface).  No fitted
    parameter: mean deviation <= 1.1 % (measured 1.06 %; round 5: tightened from 1.3 %), worst point (the steepest flank of the sharpest resonance in the band) <= 5 %, and at the
    five frequencies the earlier rounds quoted (0.85, 0.95, 1, 1.05, 1.15 f0) <= 2.5 %."""
    dev, f = _mie_run(40, hip_lib)
    assert np.abs(dev).mean() < 0.011, np.abs(dev).mean()
    assert np.abs(dev).max() < 0.05, np.abs(dev).max()
    five = [int(np.argmin(np.abs(f - v))) for v in (0.85, 0.95, 1.0, 1.05, 1.15)]
    assert np.abs(dev[five]).max() < 0.025, dev[five]


def test_mie_grid_refinement_second_order(hip_lib):
    """The physics pin that separates "interface representation error" from "bug" while the oracle stays unpinned by reference
    field data: the config-4 problem — same sphere, same physical box, same TFSF source, flux box and
    25 frequencies, both runs ended by the field decay — at lambda0/20 (268^3 cells) and lambda0/40 (512^3).  A second-order scheme
    brings sigma_sca / Mie series towards 1 by 4x per halving of dl: measured 3.65x on the mean deviation (3.87 % -> 1.06 %) and
    3.75x on the worst point (15.6 % -> 4.2 %) (profiles/r4/r4i_mie_converged_25f.jsonl); asserted >= 3.3x and >= 3x.  The one
    global frequency shift that fits each spectrum best — the grid's numerical dispersion — falls 0.35 % -> 0.18 % -> 0.11 % at
    lambda0 / 20, 30, 40 (scripts/probe_mie_refinement.py)."""
    d20, _ = _mie_run(20, hip_lib)
    d40, _ = _mie_run(40, hip_lib)
    m20, m40, w20, w40 = np.abs(d20).mean(), np.abs(d40).mean(), np.abs(d20).max(), np.abs(d40).max()
    print(f"[mie refinement] mean deviation {m20:.4f} -> {m40:.4f} (x {m20 / m40:.2f}), worst {w20:.4f} -> {w40:.4f} (x {w20 / w40:.2f})")
    assert m20 / m40 >= 3.3
    assert w20 / w40 >= 3.0


def test_config3_si_strip_waveguide_mode_launch(hip_lib):
    """BASELINE config[2]: Si strip waveguide, 400 x 200 x 800 cells (+12 PML cells per face =
    424 x 224 x 824), ModeSource + PML + FluxMonitor on one MI355X.  The launched field is compared
    with the CPU eigenmode (same solver that is pinned to the reference's compute_modes): modal
    purity |a+|^2 / flux, backward power, and n_eff from the phase advance between two planes."""
    import time
    from tidy3d_amd.data import assemble
    lam = 1.55
    f0 = C_0 / lam
    dl = 0.01
    pulse = td.GaussianPulse(freq0=f0, fwidth=f0 / 10)
    plane = (td.inf, td.inf, 0)
    sim = td.Simulation(
        size=(4.0, 2.0, 8.0), grid_spec=td.GridSpec.uniform(dl=dl), run_time=2.6e-13,
        medium=td.Medium(permittivity=1.44 ** 2),
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 0), size=(0.45, 0.22, td.inf)),
                                 medium=td.Medium(permittivity=3.48 ** 2))],
        sources=[td.ModeSource(center=(0, 0, -3.5), size=plane, source_time=pulse, direction="+",
                               mode_spec=td.ModeSpec(num_modes=1), mode_index=0)],
        monitors=[td.FluxMonitor(center=(0, 0, 3.0), size=plane, freqs=[f0], name="fwd"),
                  td.FluxMonitor(center=(0, 0, -3.8), size=plane, freqs=[f0], name="bwd"),
                  td.ModeMonitor(center=(0, 0, 3.0), size=plane, freqs=[f0], mode_spec=td.ModeSpec(num_modes=1), name="mm"),
                  td.FieldMonitor(center=(0, 0, 1.0), size=(0, 0, 0), freqs=[f0], name="p1", fields=["Ex"]),
                  td.FieldMonitor(center=(0, 0, 2.0), size=(0, 0, 0), freqs=[f0], name="p2", fields=["Ex"])],
        boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=12)), shutoff=1e-5)
    import tidy3d_amd.modesource as _ms
    eig, solve = [0.0], _ms.mode_profile

    def timed(*a, **k):                              # (the CPU eigen-solves of the mode plane: not the rasteriser's time)
        t = time.time()
        try:
            return solve(*a, **k)
        finally:
            eig[0] += time.time() - t
    t0 = time.time()
    _ms.mode_profile = timed
    try:
        disc = discretize(sim)
    finally:
        _ms.mode_profile = solve
    assert disc.spec.shape == (424, 224, 824)
    t1 = time.time()
    print(f"\n[config3] set-up without the eigen-solves {t1 - t0 - eig[0]:.2f}s (eigen-solves {eig[0]:.1f}s)")
    assert t1 - t0 - eig[0] < 1.5                    # VERDICT round 5, item 6 (native host passes of the rasteriser: include/fdtd_host.h)
    with HipEngine(disc.spec, lib=hip_lib) as e:
        st = e.run()
        raw = e.results()
    t2 = time.time()
    sd = assemble(disc, raw, log="")
    src_plane = list(disc.mode_planes.values())[0]
    beta = src_plane.beta[0].real                   # propagation constant of the mode ON THE GRID (modesource.mode_profile)
    neff = beta * C_0 / (2 * np.pi * f0)
    fwd, bwd = float(sd["fwd"].flux.values[0]), float(sd["bwd"].flux.values[0])
    a = sd["mm"].amps.values
    # power found in the CPU eigenmode / total power through the same plane, both as the monitors measure them
    purity = abs(a[0, 0, 0]) ** 2 * float(sd["mm"].mode_power.values[0, 0, 0]) / fwd
    dphi = np.angle(sd["p2"].Ex.values.ravel()[0] / sd["p1"].Ex.values.ravel()[0])
    dphi_ref = np.angle(np.exp(1j * beta * 1.0))
    print(f"\n[config3] setup {t1 - t0:.1f}s solve {t2 - t1:.1f}s ({st.steps_done} steps, {int(st.fused2_pairs)} step pairs (reason {int(st.fused2_off_reason)}), "
          f"{disc.spec.n_cells * st.steps_done / (st.run_ms * 1e-3) / 1e6:.0f} Mcells/s) neff={neff:.6f} "
          f"fwd={fwd:.6f} bwd={bwd:.3e} purity={purity:.7f} |a-|^2={abs(a[1, 0, 0])**2:.3e} "
          f"dphi={dphi:.5f} vs {dphi_ref:.5f}")
    assert not st.diverged
    assert 0.985 < fwd < 1.002                 # 1 W launched (x colocation factor of the flux measurement)
    assert abs(bwd) < 1e-6                     # one-way launch: backward power below -60 dB
    # north_star: "injected mode vs CPU ModeSolver to 1e-5" — the field 6.5 um downstream is the eigenmode:
    # all but 1e-5 of the power through the plane is found in it (k = 1; fp32 solve, 9000 steps)
    assert abs(1 - purity) < 1e-5
    assert abs(a[1, 0, 0]) ** 2 < 1e-6         # nothing comes back from the far end
    # phase advance over 1 um between two probes vs the propagation constant the grid dispersion predicts
    assert abs(np.angle(np.exp(1j * (dphi - dphi_ref)))) < 2e-3


def test_config5_au_nanoparticle_array_1024x1024x256(hip_lib):
    """BASELINE config[4] on ONE MI355X (the 8-GPU z-slab run of the same grid is the driver's):
    dispersive Au (Johnson & Christy, 5 pole pairs -> ADE) nano-disc array, 1024 x 1024 x 256 cells,
    periodic in x/y, CPML in z, plane wave.  Checks stability of the 5-pole ADE over the whole run, the 16-fold translation
    symmetry of the array, and — round 5 — R, T and A against the array's UNIT CELL (64 x 64 x 256, one disc) run through the fp64
    oracle for the whole run time (tests/golden/config5_unit_cell_oracle.json, made by scripts/make_config5_unit_cell_golden.py): to
    1e-3 of the incident power — a wrong ADE coefficient, a wrong CPML term or a wrong TFSF / flux normalisation cannot pass; the
    same unit cell on the HIP engine agrees with the oracle to 5e-5 and with the full array to 6e-4 (rim nodes of the staircase)."""
    import time
    from cases import gold_johnson_christy
    from tidy3d_amd.data import assemble
    dl = 0.005
    nxy, nz = 1024, 256 - 24
    pitch = 64 * dl
    au = gold_johnson_christy()
    f0 = 5e14
    pulse = td.GaussianPulse(freq0=f0, fwidth=1e14)
    L, Lz = nxy * dl, nz * dl
    discs = [td.Structure(geometry=td.Cylinder(center=(-L / 2 + (i + 0.5) * pitch, -L / 2 + (j + 0.5) * pitch, 0.0),
                                               radius=0.08, length=0.04, axis=2), medium=au)
             for i in range(16) for j in range(16)]
    slab = td.Structure(geometry=td.Box(center=(0, 0, -Lz / 4 - 0.02), size=(td.inf, td.inf, Lz / 2)),
                        medium=td.Medium(permittivity=2.1))
    plane = (td.inf, td.inf, 0)
    sim = td.Simulation(
        size=(L, L, Lz), grid_spec=td.GridSpec.uniform(dl=dl), run_time=6e-14,
        structures=[slab] + discs,
        sources=[td.PlaneWave(center=(0, 0, Lz / 2 - 0.1), size=plane, source_time=pulse, direction="-")],
        monitors=[td.FluxMonitor(center=(0, 0, Lz / 2 - 0.05), size=plane, freqs=[f0], name="R"),
                  td.FluxMonitor(center=(0, 0, -Lz / 2 + 0.1), size=plane, freqs=[f0], name="T"),
                  td.FieldMonitor(center=(0, 0, 0.03), size=(td.inf, 0, 0), freqs=[f0], name="line", fields=["Ex"],
                                  colocate=False)],
        boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml()),
        shutoff=1e-4)
    t0 = time.time()
    disc = discretize(sim)
    assert disc.spec.shape == (1024, 1024, 256)
    t1 = time.time()
    assert t1 - t0 < 2.0                             # (measured 0.8-1.1 s with the native host passes, 3.1 s before: profiles/r6/r6s_time_setup.jsonl)
    with HipEngine(disc.spec, lib=hip_lib) as e:
        st = e.run()
        raw = e.results()
    t2 = time.time()
    sd = assemble(disc, raw, log="")
    area = L * L
    R = float(sd["R"].flux.values[0]) / area           # above the one-way source: only the reflected (+z) wave
    T = -float(sd["T"].flux.values[0]) / area          # transmitted wave travels -z
    A = 1 - R - T
    line = sd["line"].Ex.values[:, 0, 0, 0]
    print(f"\n[config5] setup {t1 - t0:.1f}s solve {t2 - t1:.1f}s ({st.steps_done} steps, {int(st.fused2_pairs)} step pairs (reason {int(st.fused2_off_reason)}), "
          f"{disc.spec.n_cells * st.steps_done / (st.run_ms * 1e-3) / 1e6:.0f} Mcells/s) R={R:.4f} T={T:.4f} A={A:.4f}")
    assert not st.diverged
    assert 0.0 < R < 1.0 and 0.0 < T < 1.0 and 0.005 < A < 0.9
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "scripts"))
    from make_config5_unit_cell_golden import config5_sim, rta
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config5_unit_cell_oracle.json")))
    d1 = discretize(config5_sim(64, shutoff=1e-4))
    with HipEngine(d1.spec, lib=hip_lib) as e1:
        st1 = e1.run()
        R1, T1, A1 = rta(assemble(d1, e1.results(), log=""), 64 * dl)
    print(f"[config5] unit cell on the engine R={R1:.5f} T={T1:.5f} A={A1:.5f} ({st1.steps_done} steps); fp64 oracle R={gold['R']:.5f} T={gold['T']:.5f} A={gold['A']:.5f}")
    # the engine on the oracle's own problem: fp32 kernels against the fp64 restatement of the whole ADE + CPML + plane-wave chain
    assert abs(R1 - gold["R"]) < 5e-5 and abs(T1 - gold["T"]) < 5e-5 and abs(A1 - gold["A"]) < 5e-5, (R1, T1, A1, gold)
    # the array against its unit cell: the 256 discs sit at different absolute coordinates, so the rounding of the staircase's inside
    # test puts a few rim nodes (the radius is exactly 16 cells) on the other side for some of them — measured 2.8e-4 in R, 3e-5 in T
    assert abs(R - R1) < 6e-4 and abs(T - T1) < 6e-4, (R, R1, T, T1)
    assert abs(R - gold["R"]) < 1e-3 and abs(T - gold["T"]) < 1e-3 and abs(A - gold["A"]) < 1e-3, (R, T, A, gold)
    # 16 periods along x: the line scan repeats every 64 cells
    per = line[:1024].reshape(16, 64)
    assert np.max(np.abs(per - per[0])) < 2e-3 * np.max(np.abs(per))


def test_angled_mode_source_tilted_waveguide(hip_lib):
    """tests/test_mode_solver.py's angled launch on the GPU (2-D grid: one cell along y)."""
    from cases import check_tilted_launch, tilted_slab_waveguide
    from tidy3d_amd.data import assemble
    disc = discretize(tilted_slab_waveguide(0.2, dl=0.02))
    with HipEngine(disc.spec, lib=hip_lib) as e:
        st = e.run()
        raw = e.results()
    assert not st.diverged
    check_tilted_launch(assemble(disc, raw, log=""))


def test_config5_stack_au_film_vs_airy(hip_lib):
    """The quantitative pin of BASELINE config[4]'s physics: the same stack — vacuum, 40 nm of
    Johnson & Christy Au (5 pole pairs: the ADE kernel), glass half-space running into the CPML, plane wave from above,
    flux planes at the same places — with a continuous film instead of the discs, whose reflectance and transmittance
    are the Airy formula with ``PoleResidue.eps_model(f)`` (ref medium.py:2900-2913).  Periodic 16 x 16 x 256 cells on the
    GPU; tolerance 0.5 % of the incident power (dl = 5 nm: the film is 8 cells, interfaces on grid planes; measured
    with the fp64 oracle: |dR| <= 0.003, |dT| <= 0.002)."""
    from cases import gold_johnson_christy
    from tidy3d_amd.analytic import thin_film_RT
    from tidy3d_amd.data import assemble
    au = gold_johnson_christy()
    dl, nxy, nz, d = 0.005, 16, 256 - 24, 0.04
    L, Lz = nxy * dl, nz * dl
    freqs = np.array([4.2e14, 4.6e14, 5.0e14, 5.4e14, 5.8e14])
    pulse = td.GaussianPulse(freq0=5e14, fwidth=1e14)
    plane = (td.inf, td.inf, 0)
    sim = td.Simulation(
        size=(L, L, Lz), grid_spec=td.GridSpec.uniform(dl=dl), run_time=1.2e-13,
        structures=[td.Structure(geometry=td.Box(center=(0, 0, -Lz / 4 - d / 2 - Lz), size=(td.inf, td.inf, Lz / 2 + 2 * Lz)),
                                 medium=td.Medium(permittivity=2.1)),
                    td.Structure(geometry=td.Box(center=(0, 0, 0), size=(td.inf, td.inf, d)), medium=au)],
        sources=[td.PlaneWave(center=(0, 0, Lz / 2 - 0.1), size=plane, source_time=pulse, direction="-")],
        monitors=[td.FluxMonitor(center=(0, 0, Lz / 2 - 0.05), size=plane, freqs=list(freqs), name="R"),
                  td.FluxMonitor(center=(0, 0, -Lz / 2 + 0.1), size=plane, freqs=list(freqs), name="T")],
        boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(), z=td.Boundary.pml()), shutoff=1e-5)
    disc = discretize(sim)
    assert disc.spec.shape == (16, 16, 256)
    with HipEngine(disc.spec, lib=hip_lib) as e:
        st = e.run()
        raw = e.results()
    sd = assemble(disc, raw, log="")
    R = sd["R"].flux.values / (L * L)
    T = -sd["T"].flux.values / (L * L)
    Ra, Ta = thin_film_RT(au.eps_model(freqs), d, 2.1, freqs)
    print(f"\n[config5 film] R={np.round(R, 4)} Airy {np.round(Ra, 4)}  T={np.round(T, 4)} Airy {np.round(Ta, 4)}")
    assert not st.diverged
    assert np.max(np.abs(R - Ra)) < 0.005 and np.max(np.abs(T - Ta)) < 0.005
    assert np.all(1 - R - T > 0.02)                       # the film absorbs: A = 0.04 ... 0.34 over the band


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["media_mix", "bloch_xy_pml_z", "absorber_mix"])
def test_axis_renaming_on_the_gpu(case, hip_lib):
    """Cyclic renaming of the axes (engine.permute_spec, the default layout choice for narrow grids): the renamed
    runs reproduce the plain one on the real kernels (different summation order of the corrections: fp32 rounding)."""
    from tidy3d_amd.discretize import discretize
    from tidy3d_amd.engine import HipEngine
    fn = CASES[case]
    sim = fn(tuple(int(n * 3) for n in fn.__defaults__[0]))
    disc = discretize(sim, n_steps=80)
    outs = []
    for s in (0, 1, 2):
        with HipEngine(disc.spec, lib=hip_lib, axis_shift=s) as e:
            e.run()
            outs.append(([e.get_field(c) for c in range(6)], e.results()))
    for s in (1, 2):
        for c in range(6):
            assert np.abs(outs[s][0][c] - outs[0][0][c]).max() <= 2e-6 * max(np.abs(outs[0][0][c]).max(), 1e-30), (s, c)
        for k, v in outs[0][1].items():
            assert np.abs(outs[s][1][k] - v).max() <= 2e-6 * max(np.abs(v).max(), 1e-30), (s, k)


def test_near_to_far_projection_from_gpu_fields(hip_lib):
    """the near fields a FieldProjectionAngleMonitor needs are accumulated ON the device
    (running DFT on the six surfaces of its box, K6) and the N / L integrals over those surfaces are taken ON the
    device as well (``fdtd_far_field``, K9 — what ``web.run`` does).  Same analytic pins as the oracle-driven test
    (sin(theta) pattern, far-sphere power = near-box flux, E / H = eta0); the device integrals against the NumPy sums
    over the same near fields to rounding; the far field against the oracle's to fp32 accuracy."""
    from test_projection import _dipole_sim, check_dipole_far_field
    from oracle.fdtd_numpy import OracleFdtd
    from tidy3d_amd.data import assemble
    sim, theta, phi = _dipole_sim()
    disc = discretize(sim)
    with HipEngine(disc.spec, lib=hip_lib) as e:
        st = e.run()
        raw = e.results()
    sd = assemble(disc, raw, n_steps_run=int(st.steps_done), device_lib=hip_lib)
    check_dipole_far_field(sd, theta, phi)
    host = assemble(disc, raw, n_steps_run=int(st.steps_done))
    for comp, arr in host["far"].field_components.items():
        a, b = np.asarray(arr.values), np.asarray(sd["far"].field_components[comp].values)
        assert np.abs(a - b).max() <= 1e-11 * max(np.abs(a).max(), 1e-300), comp
    ref = assemble(disc, OracleFdtd(disc.spec).run())
    a, b = sd["far"].Etheta.values, ref["far"].Etheta.values
    assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max()


def test_far_field_kernel_many_directions(hip_lib):
    """K9 at a size the NumPy loop takes minutes for: 96 x 80 lattice points, 4096 directions, complex k.  Checked
    against a direct fp64 evaluation of a sample of directions, and for repeatability (fixed reduction tree)."""
    rng = np.random.default_rng(5)
    nu, nv, nd = 96, 80, 4096
    u, v = np.sort(rng.uniform(-2, 2, nu)), np.sort(rng.uniform(-1.5, 1.5, nv))
    wu, wv = np.gradient(u), np.gradient(v)
    cur = rng.normal(size=(4, nu, nv)) + 1j * rng.normal(size=(4, nu, nv))
    th, ph = rng.uniform(0, np.pi, nd), rng.uniform(0, 2 * np.pi, nd)
    ru, rv, rw = np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)
    k, w0 = 2 * np.pi / 0.5 * (1.0 + 0.002j), 0.7
    got = hip_lib.far_field(u, v, wu, wv, cur, w0, k, ru, rv, rw)
    assert np.array_equal(got, hip_lib.far_field(u, v, wu, wv, cur, w0, k, ru, rv, rw))
    for d in range(0, nd, 257):
        phase = np.exp(-1j * k * (u[:, None] * ru[d] + v[None, :] * rv[d] + w0 * rw[d])) * wu[:, None] * wv[None, :]
        want = (cur * phase[None]).sum(axis=(1, 2))
        np.testing.assert_allclose(got[d], want, rtol=0, atol=1e-11 * np.abs(want).max())


def test_placement_probe_is_invisible_in_the_results(hip_lib):
    """A run of 2^24 cells samples three alternative placements of its field arrays before the first step
    (fdtd_capi.hip ``probe_placement``) and may move the fields, set before the run, to another set of allocations:
    bit-identical to the run that keeps its first allocations."""
    n = 256
    sim = td.Simulation(size=(n * 0.05 - 1e-9,) * 3, grid_spec=td.GridSpec.uniform(dl=0.05), run_time=1e-12,
                        structures=[td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=2.0), medium=td.Medium(permittivity=4.0))],
                        sources=[td.PointDipole(center=(0, 0, 0), source_time=td.GaussianPulse(freq0=2e14, fwidth=2e13), polarization="Ez")],
                        monitors=[td.FieldTimeMonitor(center=(1.0, 0.5, 0.2), size=(0, 0, 0), name="p")],
                        boundary_spec=td.BoundarySpec.all_sides(td.PECBoundary()), shutoff=0)
    disc = discretize(sim, n_steps=24)
    assert disc.spec.shape == (n, n, n)
    rng = np.random.default_rng(11)
    init = [rng.uniform(-1e-3, 1e-3, (n, n, n)).astype(np.float32) for _ in range(6)]

    def run(tries):
        with HipEngine(disc.spec, lib=hip_lib) as e:
            e.set_option(L.OPT_PLACEMENT_TRIES, tries)
            for c in range(6):
                e.set_field(c, init[c])
            st = e.run()
            return int(st.placement) >> 8, [e.get_field(c) for c in range(6)], e.results()
    tried0, f0, m0 = run(0)
    tried3, f3, m3 = run(3)
    assert tried0 == 0 and tried3 == 3
    for a, b in zip(f0, f3):
        assert np.array_equal(a, b)
    for k in m0:
        assert np.array_equal(m0[k], m3[k])
