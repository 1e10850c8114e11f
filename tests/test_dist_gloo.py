"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the z-slab decomposition through the real
library code (emulated HIP + RCCL shim) must reproduce the single-slab result of the same library
bit for bit, and the oracle to fp32 tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest

from cases import CASES, rel_err
from tidy3d_amd.discretize import discretize
from tidy3d_amd.engine import HipEngine, split_slabs

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _launch(world, case, n_steps, out, port, slab_shift=None, placement_tries=None, pml_fused=None, twostep=None, slab_boxes=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if slab_boxes is not None:
        env["SLAB_BOXES"] = str(slab_boxes)
    if twostep is not None:
        env["TWOSTEP"] = str(twostep)
    if slab_shift is not None:
        env["SLAB_SHIFT"] = str(slab_shift)
    if placement_tries is not None:
        env["PLACEMENT_TRIES"] = str(placement_tries)
    if pml_fused is not None:
        env["PML_FUSED"] = str(pml_fused)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), case, str(n_steps), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]


def test_split_slabs():
    assert split_slabs(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert split_slabs(512, 8)[3] == (192, 256)


GLOO_CASES = [(2, "media_mix"), (2, "pml_box"), (3, "periodic_box"), (2, "drude_in_pml"),
                                        (2, "periodic_box_tall"), (4, "periodic_box_tall"), (2, "planewave_periodic"),
                                        (2, "tfsf_box"), (3, "au_array"), (3, "absorber_mix"),
                                        # complex fields on z-slabs: Bloch x / y with CPML in z; Bloch on all axes (the
                                        # wrap-around planes rank n-1 <-> rank 0 are rotated by exp(-+ i phi_z))
                                        (2, "bloch_xy_pml_z"), (2, "bloch_box"), (3, "bloch_box"),
                                        # PMC on plus faces: the x wall crosses every slab, the z wall is the last rank's
                                        (2, "pmc_plus_mix"),
                                        # graded cells along z: a rank needs its neighbours' cell sizes next to the cuts (the ghost
                                        # entries of the step arrays; replicas of its own until round 4's variant fuzz noticed)
                                        (2, "nonuniform_grid"), (3, "nonuniform_grid")]


@pytest.mark.parametrize("world,case", GLOO_CASES)
def test_two_rank_run_matches_single_slab(world, case, emu_lib, tmp_path):
    n_steps = 60 if case == "planewave_periodic" else 30     # let the injected wave reach the monitors
    out = str(tmp_path / "dist.npz")
    _launch(world, case, n_steps, out, 29511 + GLOO_CASES.index((world, case)))      # one port per case (parallel runs)
    got = np.load(out)
    disc = discretize(CASES[case](), n_steps=n_steps)
    disc.spec.decay_every = 10
    # complex (Bloch) fields run the two-pass kernels on z-slabs: same bits as the single-slab two-pass run (the fused
    # sweep rotates a Bloch-z ghost plane BEFORE updating it — one rounding apart, tests/test_emu_fused.py)
    from tidy3d_amd import lib as L
    with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_ZMARCH if disc.spec.bloch is not None else L.VARIANT_AUTO) as e:
        st = e.run()
        ref = e.results()
        fields = [e.get_field(c) for c in range(6)]
    for c in range(6):
        assert np.array_equal(got[f"field{c}"], fields[c]), f"component {c} differs from the single-slab run"
    for k, v in ref.items():
        assert np.array_equal(got[f"mon_{k}"], v), k
    assert float(got["decay"]) == pytest.approx(st.field_decay, rel=1e-6)
    from oracle.fdtd_numpy import OracleFdtd
    o = OracleFdtd(disc.spec)
    oref = o.run()
    scale = max(np.linalg.norm(v) / np.sqrt(v.size) for v in oref.values())
    for k, v in oref.items():           # (scale-aware: see cases.run_case)
        den = max(np.linalg.norm(v), 0.5 * scale * np.sqrt(v.size))
        assert np.linalg.norm(got[f"mon_{k}"] - v) / den < 2e-5


IN_SWEEP_CASES = [(2, "pml_box", 7), (3, "pml_box", 7), (2, "drude_in_pml", 7), (2, "media_mix", 7), (3, "au_array", 7), (2, "pml_box", 6)]


@pytest.mark.parametrize("world,case,mask", IN_SWEEP_CASES)
def test_cpml_inside_the_sweeps_of_slab_ranks(world, case, mask, emu_lib, tmp_path):
    """FDTD_OPT_PML_FUSED > 0 on z-slab ranks: the recursions run inside the sweeps as on one GPU; the H-side psi of a
    rank's ghost plane -1 (x, y axes) arrives with the ghost fields.  Same arithmetic and summation order: same bits as
    the single-slab run (whose CPML is inside the sweep too)."""
    out = str(tmp_path / "dist.npz")
    _launch(world, case, 30, out, 29581 + IN_SWEEP_CASES.index((world, case, mask)), pml_fused=mask)      # one port per case
    got = np.load(out)
    disc = discretize(CASES[case](), n_steps=30)
    disc.spec.decay_every = 10
    with HipEngine(disc.spec, lib=emu_lib) as e:
        e.run()
        ref = e.results()
        fields = [e.get_field(c) for c in range(6)]
    for c in range(6):
        assert np.array_equal(got[f"field{c}"], fields[c]), c
    for k, v in ref.items():
        assert np.array_equal(got[f"mon_{k}"], v), k


SLAB_PAIR_CASES = [(2, "slab_pairs_box", 8 + 64 * 6), (3, "slab_pairs_box", 16 + 64 * 32), (2, "slab_pairs_box_periodic", 6 + 64 * 5),
                   (4, "slab_pairs_box_periodic", 4 + 64 * 3)]


@pytest.mark.parametrize("world,case,twostep", SLAB_PAIR_CASES)
def test_step_pairs_on_slab_ranks(world, case, twostep, emu_lib, tmp_path):
    """z-slab ranks advance two time steps per sweep: the planes two or more away from a cut by the two-step sweep, the two
    planes next to it by two single steps on the comm stream through the third field set, shipping their planes after each —
    the messages of two single steps.  Same bits as the single-slab run (which takes single steps here: the grid is small),
    with dipoles of both kinds next to the cuts, materials through them, monitors whose records end pairs."""
    import cases
    out = str(tmp_path / "dist.npz")
    _launch(world, case, 46, out, 29701 + SLAB_PAIR_CASES.index((world, case, twostep)), twostep=twostep)
    got = np.load(out)
    assert (got["pairs"] >= 8).all(), got["pairs"]
    disc = discretize(getattr(cases, case)(), n_steps=46)
    disc.spec.decay_every = 10
    with HipEngine(disc.spec, lib=emu_lib) as e:
        st = e.run()
        assert int(st.fused2_pairs) == 0
        ref = e.results()
        fields = [e.get_field(c) for c in range(6)]
    assert max(float(np.abs(f).max()) for f in fields) > 0
    for c in range(6):
        assert np.array_equal(got[f"field{c}"], fields[c]), c
    for k, v in ref.items():
        assert np.array_equal(got[f"mon_{k}"], v), k
    assert float(got["decay"]) == pytest.approx(st.field_decay, rel=1e-6)


CPML_PAIR_CASES = [(2, "slab_pairs_pml_box", 8 + 64 * 6), (2, "slab_pairs_pml_box", 16 + 64 * 8), (3, "slab_pairs_pml_box_tall", 8 + 64 * 5),
                   (3, "slab_pairs_pml_box_near_cut", 8 + 64 * 4)]       # (the last one: a dipole next to a cut — its rank takes single steps beside ranks in pairs)


@pytest.mark.parametrize("world,case,twostep,boxes", [c + (None,) for c in CPML_PAIR_CASES] + [CPML_PAIR_CASES[0] + (2,), CPML_PAIR_CASES[2] + (0,)])
def test_step_pairs_on_slab_ranks_that_carry_cpml(world, case, twostep, boxes, emu_lib, tmp_path):
    """Round 5: z-slab ranks with CPML (x / y layers on every rank, z layers on the end ranks) advance in shell2 pairs — bulk and boxes
    as on one GPU over the planes two or more away from a cut, the two planes next to a cut as a z hole that takes two single steps
    and ships its planes (and the H-side psi of the top plane) after each.  Same bits as the single-slab run, materials through the
    cuts, monitor records ending pairs; every rank takes pairs.  `boxes`: FDTD_OPT_SLAB_BOXES_FIRST — the shell's boxes behind the bulk
    (0), in front of it (default) or on a third stream beside it (2: round 6)."""
    import cases
    out = str(tmp_path / "dist.npz")
    _launch(world, case, 46, out, 29741 + CPML_PAIR_CASES.index((world, case, twostep)) + (0 if boxes is None else 10 + boxes), twostep=twostep,
            pml_fused=7, slab_boxes=boxes)
    got = np.load(out)
    if case == "slab_pairs_pml_box_near_cut":
        assert got["pairs"].max() >= 8 and got["pairs"].min() == 0, got["pairs"]
    else:
        assert (got["pairs"] >= 8).all(), got["pairs"]
    disc = discretize(getattr(cases, case)(), n_steps=46)
    disc.spec.decay_every = 10
    with HipEngine(disc.spec, lib=emu_lib) as e:
        st = e.run()
        ref = e.results()
        fields = [e.get_field(c) for c in range(6)]
    assert max(float(np.abs(f).max()) for f in fields) > 0
    for c in range(6):
        assert np.array_equal(got[f"field{c}"], fields[c]), c
    for k, v in ref.items():
        assert np.array_equal(got[f"mon_{k}"], v), k
    assert float(got["decay"]) == pytest.approx(st.field_decay, rel=1e-6)


@pytest.mark.parametrize("seed,index", [(1, 0), (1, 2), (1, 4), (1, 5), (1, 7), (2, 0), (2, 3)])
def test_random_cpml_boxes_in_step_pairs_on_two_to_four_ranks(seed, index, emu_lib, tmp_path):
    """cases.random_slab_pml_box (scripts/fuzz_slab_cpml.py runs it unattended: 70 of 70 clean): layers absent on an axis, PMC min
    walls, StablePML, bodies through cuts and layers, dipoles that keep single ranks in single steps, 3 - 4 ranks, four tile shapes."""
    import cases
    sim, world, twostep, steps = cases.random_slab_pml_box(seed, index)
    out = str(tmp_path / "dist.npz")
    _launch(world, f"slabfuzz:{seed}:{index}", steps, out, 29761 + 10 * seed + index, twostep=twostep, pml_fused=7)
    got = np.load(out)
    assert got["pairs"].max() >= 6, got["pairs"]
    disc = discretize(sim, n_steps=steps)
    disc.spec.decay_every = 10
    with HipEngine(disc.spec, lib=emu_lib) as e:
        e.run()
        ref = e.results()
        fields = [e.get_field(c) for c in range(6)]
    for c in range(6):
        assert np.array_equal(got[f"field{c}"], fields[c]), c
    for k, v in ref.items():
        assert np.array_equal(got[f"mon_{k}"], v), k


def test_in_sweep_cpml_is_asked_for_only_where_the_whole_problem_allows_pairs():
    """dist.cpml_pairs_possible: decided from the whole (renamed) problem, so that every rank posts the same messages."""
    import cases
    from tidy3d_amd.dist import cpml_pairs_possible
    yes = discretize(cases.slab_pairs_pml_box(), n_steps=4).spec
    assert cpml_pairs_possible(yes)
    # ... and only where every rank's slab can take them (2^20 cells, a bulk of eight planes): else the slab kernels stay
    assert not cpml_pairs_possible(yes, [(0, 25), (25, 50)])
    import dataclasses
    big = dataclasses.replace(yes, shape=(512, 512, 128))
    assert cpml_pairs_possible(big, [(0, 64), (64, 128)]) and not cpml_pairs_possible(big, [(0, 12), (12, 128)])
    for name in ("slab_pairs_box", "media_mix", "au_array", "drude_in_pml", "absorber_mix"):
        sim = CASES[name]() if name in CASES else getattr(cases, name)()
        assert not cpml_pairs_possible(discretize(sim, n_steps=4).spec), name


FUZZ_RANK_CASES = [(2, 31, 0), (3, 31, 1), (2, 31, 2), (3, 31, 3), (2, 31, 4), (2, 31, 5), (3, 31, 6), (2, 31, 7)]


@pytest.mark.parametrize("world,seed,index", FUZZ_RANK_CASES)
def test_random_simulations_on_two_and_three_ranks(world, seed, index, emu_lib, tmp_path):
    """Random simulations of scripts/fuzz_variants.py (walls of every kind, graded cells, dispersive / anisotropic bodies, dipoles of
    both kinds, plane waves, monitors across the cuts, decay checks) cut into 2 or 3 z-slabs at whatever planes the split puts
    them: fields inside the walls and every record == the single-slab run of the same library, bit for bit.  (On the device
    the same cases run as a slab rank exchanging with itself over RCCL, tests/test_gpu_production_path.py; real cuts — unequal
    slabs, a neighbour's cell sizes, records gathered from several ranks — only exist here.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_variants
    from tidy3d_amd import lib as L
    rng = np.random.default_rng(seed)
    for _ in range(index + 1):
        disc, steps, per_z, desc, *_ = fuzz_variants.draw(rng, False)
    if disc.spec.shape[2] < 4 * world:
        pytest.skip("too few planes for %d slabs: %s" % (world, desc))
    out = str(tmp_path / "dist.npz")
    _launch(world, f"fuzz:{seed}:{index}", steps, out, 29751 + FUZZ_RANK_CASES.index((world, seed, index)))
    got = np.load(out)
    if "refused" in got:
        pytest.skip("the split is refused on every rank: %s (%s)" % (got["refused"], desc))
    sl = fuzz_variants.inside(disc)
    with HipEngine(disc.spec, lib=emu_lib, variant=L.VARIANT_ZMARCH if disc.spec.bloch is not None else L.VARIANT_AUTO, axis_shift=0) as e:
        st = e.run()
        ref = e.results()
        fields = [e.get_field(c) for c in range(6)]
    for c in range(6):
        assert np.array_equal(got[f"field{c}"][sl], fields[c][sl]), (desc, c)
    for k, v in ref.items():
        assert np.array_equal(got[f"mon_{k}"], v), (desc, k)


@pytest.mark.parametrize("world,case", [(2, "media_mix"), (3, "periodic_box")])
def test_placement_probe_on_slabs_changes_nothing(world, case, emu_lib, tmp_path):
    """Every rank samples alternative placements of its slab's field arrays before the first step (forced on for
    these small grids): same bits as the single-slab run."""
    out = str(tmp_path / "dist.npz")
    _launch(world, case, 30, out, 29721 + world, placement_tries=102)
    got = np.load(out)
    disc = discretize(CASES[case](), n_steps=30)
    disc.spec.decay_every = 10
    with HipEngine(disc.spec, lib=emu_lib) as e:
        e.run()
        ref = e.results()
        fields = [e.get_field(c) for c in range(6)]
    for c in range(6):
        assert np.array_equal(got[f"field{c}"], fields[c]), c
    for k, v in ref.items():
        assert np.array_equal(got[f"mon_{k}"], v), k


def test_balanced_slabs_follow_the_cost_model():
    """Equal modelled cost, not equal plane counts: a dispersive block in the upper half and z-PML
    at both ends shift the cuts; the partition stays contiguous, complete and PML-clear."""
    from tidy3d_amd.engine import balanced_slabs, plane_costs
    import tidy3d_amd.schema as td
    sim = td.Simulation(
        size=(1.0, 1.0, 4.0), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=1e-13,
        structures=[td.Structure(geometry=td.Box(center=(0, 0, 1.0), size=(td.inf, td.inf, 1.6)),
                                 medium=td.Lorentz(eps_inf=2.0, coeffs=[(1.5, 4e14, 3e13), (0.5, 6e14, 3e13)]))],
        sources=[td.PointDipole(center=(0, 0, -1), source_time=td.GaussianPulse(freq0=3e14, fwidth=3e13),
                                polarization="Ex")],
        monitors=[], boundary_spec=td.BoundarySpec(x=td.Boundary.periodic(), y=td.Boundary.periodic(),
                                                   z=td.Boundary.pml(num_layers=8)))
    spec = discretize(sim, n_steps=2).spec
    nz = spec.shape[2]
    cost = plane_costs(spec)
    for world in (2, 3, 4):
        slabs = balanced_slabs(spec, world)
        assert slabs[0][0] == 0 and slabs[-1][1] == nz
        assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
        assert all(z1 - z0 >= 4 for z0, z1 in slabs)
        assert all(8 + 2 <= z0 <= nz - 8 - 2 for z0, _ in slabs[1:])
        bal = [cost[a:b].sum() for a, b in slabs]
        uni = [cost[a:b].sum() for a, b in split_slabs(nz, world)]
        assert max(bal) <= max(uni) + 1e-9
        assert HipEngine._fused_slabs_ok(spec, world, slabs)
    assert max(cost[a:b].sum() for a, b in balanced_slabs(spec, 2)) < 0.9 * max(
        cost[a:b].sum() for a, b in split_slabs(nz, 2))


def test_balanced_slabs_price_planes_as_step_pairs_do_where_the_ranks_take_them():
    """Round 6: ranks that advance in shell2 step pairs (dist.cpml_pairs_possible) pay 2.2 x a bulk plane for a plane inside the z layers,
    not the 3.6 x of the single-step kernels — the end ranks get more planes than the single-step model hands them (measured on the
    device: scripts/probe_end_rank.py, profiles/r6/r6er_end_rank_split_fresh_processes.jsonl), the partition stays contiguous, complete
    and clear of the layers, and make_engine's gate still lets every rank take pairs."""
    from tidy3d_amd.dist import cpml_pairs_possible
    from tidy3d_amd.engine import balanced_slabs, plane_costs
    import tidy3d_amd.schema as td
    sim = td.Simulation(
        size=(5.0, 5.0, 9.0), grid_spec=td.GridSpec.uniform(dl=0.05), run_time=1e-13,
        structures=[td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=1.0), medium=td.Medium(permittivity=4.0))],
        sources=[td.PointDipole(center=(0, 0, -1), source_time=td.GaussianPulse(freq0=3e14, fwidth=3e13), polarization="Ex")],
        monitors=[], boundary_spec=td.BoundarySpec.all_sides(td.PML(num_layers=10)))
    spec = discretize(sim, n_steps=2).spec
    nz = spec.shape[2]
    assert cpml_pairs_possible(spec)
    single, pair = plane_costs(spec), plane_costs(spec, pairs=True)
    assert 1.2 < pair[0] / pair[nz // 2] < 0.85 * single[0] / single[nz // 2]
    assert (pair[:12] == pair[0]).all() and pair[12] == pair[nz // 2] and (pair[nz - 12:] == pair[0]).all()      # (10 layers + the two-cell collar)
    for world in (2, 4, 8):
        a, b = balanced_slabs(spec, world), balanced_slabs(spec, world, pairs=True)
        assert b[0][0] == 0 and b[-1][1] == nz and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert all(10 + 2 <= z0 <= nz - 10 - 2 for z0, _ in b[1:])
        if world > 2:
            assert b[0][1] - b[0][0] > a[0][1] - a[0][0] and b[-1][1] - b[-1][0] > a[-1][1] - a[-1][0]
        cost = [pair[z0:z1].sum() for z0, z1 in b]
        assert max(cost) <= max(pair[z0:z1].sum() for z0, z1 in a) + 1e-9
        assert cpml_pairs_possible(spec, b) == (world == 2)           # (2^20 cells per rank: the library's threshold for pairs)


@pytest.mark.parametrize("shift,world,case", [(1, 2, "media_mix"), (2, 3, "drude_in_pml")])
def test_slab_axis_renaming_matches_single_slab(shift, world, case, emu_lib, tmp_path):
    """tidy3d_amd.dist.make_engine may rename the axes cyclically so that the slab axis is the one with the most
    planes (best_slab_shift): the stitched, renamed-back result equals the plain single-GPU run (fp32 rounding of the
    correction sums)."""
    from tidy3d_amd.dist import best_slab_shift
    assert best_slab_shift((1024, 1024, 256), 8) == 1 and best_slab_shift((512, 512, 512), 8) == 0
    assert best_slab_shift((60, 60, 400), 2) == 0              # already along the longest axis
    out = str(tmp_path / "dist.npz")
    _launch(world, case, 30, out, 29731 + shift, slab_shift=shift)
    got = np.load(out)
    disc = discretize(CASES[case](), n_steps=30)
    disc.spec.decay_every = 10
    with HipEngine(disc.spec, lib=emu_lib, axis_shift=0) as e:
        e.run()
        ref = e.results()
        fields = [e.get_field(c) for c in range(6)]
    for c in range(6):
        assert got[f"field{c}"].shape == fields[c].shape
        assert np.abs(got[f"field{c}"] - fields[c]).max() <= 1e-6 * max(np.abs(fields[c]).max(), 1e-30), c
    for k, v in ref.items():
        assert got[f"mon_{k}"].shape == v.shape
        assert np.abs(got[f"mon_{k}"] - v).max() <= 1e-6 * max(np.abs(v).max(), 1e-30), k


def test_distributed_run_entry_point(emu_lib, tmp_path):
    """``tidy3d_amd.dist.run`` (every rank calls it, rank 0 returns the SimulationData) == ``web.run`` on one slab."""
    from tidy3d_amd.web import run
    out = str(tmp_path / "run.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "tests", "dist_run_worker.py"), "pml_box", "40", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    sd = run(CASES["pml_box"](), verbose=False, lib=emu_lib, n_steps=40)
    n = 0
    for d in sd.data:
        comps = getattr(d, "field_components", None) or {"flux": d.flux}
        for k, v in comps.items():
            assert np.array_equal(got[f"{d.monitor.name}__{k}"], np.asarray(v.values)), (d.monitor.name, k)
            n += 1
    assert n >= 12


def test_automatic_slab_axis_choice(emu_lib, tmp_path):
    """128 x 128 x 16 cells on 2 ranks: make_engine picks the renaming by itself (8-plane slabs would become
    64-plane slabs); the stitched result equals the plain single-GPU run."""
    import cases
    from tidy3d_amd.dist import best_slab_shift
    disc = discretize(cases.wide_flat(), n_steps=6)
    assert disc.spec.shape == (128, 128, 16) and best_slab_shift(disc.spec.shape, 2) == 1
    out = str(tmp_path / "dist.npz")
    _launch(2, "wide_flat", 6, out, 29569)
    got = np.load(out)
    disc.spec.decay_every = 10
    with HipEngine(disc.spec, lib=emu_lib, axis_shift=0) as e:
        e.run()
        ref = e.results()
        fields = [e.get_field(c) for c in range(6)]
    for c in range(6):
        assert got[f"field{c}"].shape == fields[c].shape
        assert np.abs(got[f"field{c}"] - fields[c]).max() <= 1e-6 * max(np.abs(fields[c]).max(), 1e-30), c
    for k, v in ref.items():
        assert np.abs(got[f"mon_{k}"] - v).max() <= 1e-6 * max(np.abs(v).max(), 1e-30), k


def test_web_run_devices_spawns_the_ranks(emu_lib, tmp_path):
    """``web.run(sim, devices=[0, 1])`` is the whole multi-GPU run: it spawns one rank per device
    (tidy3d_amd.dist_main), the ranks split the grid into z-slabs and exchange ghost planes, rank 0 hands the
    SimulationData back.  Here: emulated library + gloo on CPU; monitors span the slab cut; equal to the
    single-process run of the same library."""
    from tidy3d_amd.web import run
    sim = CASES["au_array"]()
    opt = dict(backend="gloo", lib=emu_lib.path, hook="dist_hook:setup",
               pythonpath=[os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hipemu")])
    multi = run(sim, task_name="two", verbose=False, n_steps=40, devices=[0, 1], _dist_options=opt)
    single = run(sim, task_name="one", verbose=False, n_steps=40, lib=emu_lib)
    assert [d.monitor.name for d in multi.data] == [d.monitor.name for d in single.data]
    for a, b in zip(multi.data, single.data):
        comps = getattr(a, "field_components", None) or {"flux": a.flux}
        for k, v in comps.items():
            w = (getattr(b, "field_components", None) or {"flux": b.flux})[k]
            np.testing.assert_array_equal(np.asarray(v.values), np.asarray(w.values))


def test_web_run_devices_a_dying_rank_raises_instead_of_hanging(emu_lib):
    """A rank that floods stderr and dies: its siblings are terminated and ``run`` raises with that
    rank's status and the tail of its stderr — within seconds, not after a collective times out."""
    import time
    from tidy3d_amd.exceptions import SolverLibraryError
    from tidy3d_amd.web import run
    sim = CASES["au_array"]()
    opt = dict(backend="gloo", lib=emu_lib.path, hook="dist_hook:setup_rank1_dies_noisily", timeout=240,
               pythonpath=[os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hipemu")])
    t0 = time.monotonic()
    with pytest.raises(SolverLibraryError, match=r"(?s)rank 1 exited with status 7.*rccl warning line"):
        run(sim, task_name="two", verbose=False, n_steps=40, devices=[0, 1], _dist_options=opt)
    assert time.monotonic() - t0 < 120
