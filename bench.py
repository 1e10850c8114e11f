#!/usr/bin/env python
"""Benchmark of the FDTD hot path on MI355X: Mcells/s on a 512^3 Yee grid + HBM roofline.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line from
rank 0.  For N > 1 the driver launches one process per GPU with torch.distributed.run; the
512^3 grid is split into N z-slabs (strong scaling) with RCCL ghost-plane exchange.

Workload (BASELINE.json metric; SURVEY.md section 8(d) variant V0): 512 x 512 x 512 cells, uniform
dl = 0.05 um, vacuum, PEC walls, one Ez point dipole (GaussianPulse 200 THz), fields initialised
to uniform random +-1e-3 (rng seed 0), fp32, curl-stencil only.  A "step" = one full time step
(H pass + E pass) of all 6 components of every cell.  Inputs are resident in HBM before the
timed region.

On one GPU the library advances this workload TWO time steps per sweep (fused2_step_kernel, bit-identical
to single sweeps): K steps are K / 2 launches.  `single_steps` on the line is the same
engine advancing one step per sweep.

Extra objects on the JSON line:
  roofline     — dominant kernel: algorithmic bytes per launch (the kernel's OWN minimum: 48 B per cell per launch,
                 whether the launch advances one time step or two — see roofline_entry; the cell rate against the
                 throughput roofline of SURVEY.md 8(d)'s 72 B two-pass kernel is reported beside it under an explicit
                 name, throughput_vs_survey_8d_roofline: a ratio of rates, not a bandwidth fraction) / its average
                 launch duration measured with hipEvents on the launch stream inside the library
                 (FDTD_FLAG_TIME_KERNELS), against the 8 TB/s HBM peak; `traffic` = PMC bytes per launch
                 from profiles/pmc_traffic.json when they were measured on these kernel sources.
  cpu_baseline — the fp64->fp32 NumPy curl loop of oracle/fdtd_numpy.py (kind "port": the
                 reference has no solver to time), on a bounded sample (smaller grid, few
                 steps), rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12            # B/s, MI355X spec (MI355X_MICROARCH.md)
BYTES_PER_CELL_PASS = 36     # 3 reads curl source + 3 reads + 3 writes updated field, fp32


# SURVEY.md 8(d) throughput variants (each adds to the previous one)
WORKLOADS = {
    "v0": "vacuum, PEC walls, curl stencil only",
    "v1": "v0 + dielectric sphere r=100 cells eps=4 (per-component material indices)",
    "v2": "v1 + CPML 12 layers x 6 faces (inside the cell count)",
    "v3": "v2 + one Lorentz pole in the sphere (ADE)",
    "v4": "v3 + closed FluxMonitor box (300 cells), running DFT at 3 frequencies",
    "va": "v0 with Absorber boundaries, 40 layers x 6 faces (inside the cell count): an open problem without CPML",
    "v1a": "v1 with Absorber boundaries, 40 layers x 6 faces (inside the cell count)",
    "v4a": "v1a + closed FluxMonitor box (300 cells), running DFT at 3 frequencies: a complete open scattering problem",
}


def build_spec(n: int, n_steps: int, workload: str):
    import tidy3d_amd.schema as td
    from tidy3d_amd.discretize import discretize

    dl = 0.05
    size = (n * dl - 1e-6 * dl,) * 3
    pulse = td.GaussianPulse(freq0=2e14, fwidth=2e13)
    structures = []
    bspec = td.BoundarySpec.all_sides(td.PECBoundary())
    if workload in ("v1", "v2", "v3", "v4", "v1a", "v4a"):
        structures.append(td.Structure(geometry=td.Sphere(center=(0, 0, 0), radius=100 * dl * n / 512),
                                       medium=td.Medium(permittivity=4.0)))
    if workload in ("v3", "v4"):
        structures[0] = td.Structure(geometry=structures[0].geometry,
                                     medium=td.Lorentz(eps_inf=2.0, coeffs=[(2.0, 4e14, 2e13)]))
    if workload in ("v2", "v3", "v4"):
        size = ((n - 24) * dl - 1e-6 * dl,) * 3      # a hair under n - 24 cells: ceil(size / dl) must not round up
        bspec = td.BoundarySpec.all_sides(td.PML(num_layers=12))
    if workload in ("va", "v1a", "v4a"):
        size = ((n - 80) * dl - 1e-6 * dl,) * 3
        bspec = td.BoundarySpec.all_sides(td.Absorber(num_layers=40))
    monitors = []
    if workload in ("v4", "v4a"):      # closed flux box around the sphere, running DFT at 3 frequencies
        monitors.append(td.FluxMonitor(center=(0, 0, 0), size=(300 * dl * n / 512,) * 3, name="flux",
                                       freqs=[1.8e14, 2.0e14, 2.2e14]))
    sim = td.Simulation(size=size, grid_spec=td.GridSpec.uniform(dl=dl), run_time=1e-12,
                        structures=structures,
                        sources=[td.PointDipole(center=(0, 0, 0), source_time=pulse, polarization="Ez")],
                        monitors=monitors, boundary_spec=bspec, shutoff=0)
    disc = discretize(sim, n_steps=n_steps)
    disc.spec.decay_every = 0
    assert disc.spec.shape == (n, n, n), disc.spec.shape
    return disc.spec


def cpu_baseline(n: int = 320, steps: int = 18):
    """Naive NumPy curl loop (oracle, fp32 arrays) on a bounded sample of the same workload
    (320^3 x 20 steps ~ 10 s of single-core work on the GPU box's host)."""
    from oracle.fdtd_numpy import OracleFdtd
    spec = build_spec(n, steps + 2, "v0")
    o = OracleFdtd(spec, dtype=np.float32)
    rng = np.random.default_rng(0)
    for a in o.E + o.H:
        a[...] = rng.uniform(-1e-3, 1e-3, a.shape).astype(np.float32)
    o.step()
    o.step()
    t0 = time.perf_counter()
    for _ in range(steps):
        o.step()
    dt = time.perf_counter() - t0
    return {"value": n ** 3 * steps / dt / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
            "sample": f"{n}^3 vacuum PEC grid, {steps} steps, NumPy sliced curl loop fp32 "
                      f"(single thread; host has {os.cpu_count()} cores)"}


def source_hash() -> str:
    """sha256 over the kernel sources (every __global__ function lives in fdtd_kernels.hpp / fdtd_kernels2.hpp / fdtd_strip.hpp / fdtd_shell2.hpp): ties a
    PMC traffic figure to the kernel code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("fdtd_kernels.hpp", "fdtd_kernels2.hpp", "fdtd_fused2.hip", "fdtd_fused2c.hip", "fdtd_fused2d.hip", "fdtd_fused2s.hip", "fdtd_strip.hpp", "fdtd_shell2.hpp", "fdtd_shell2.hip"):
        h.update(open(os.path.join(ROOT, "tidy3d_amd/csrc", f), "rb").read())
    return h.hexdigest()[:16]


def min_bytes_per_cell(workload: str, spec) -> float:
    """The fused sweep's OWN minimum HBM traffic per cell-step: 6 field reads + 6 writes = 48 B, plus 32 B
    per cell and CPML-axis membership (4 psi values read and written) when the recursions run inside it.
    Material words cost nothing where a 256-cell row segment is uniform (row-segment words) and 4 B/cell
    elsewhere; they are left out, so the fraction is a lower bound."""
    b = 48.0
    n = np.array(spec.shape, dtype=np.float64)
    for a in range(3):
        lay = spec.pml[a][0].num_layers + spec.pml[a][1].num_layers
        b += 32.0 * lay / n[a]
    return b


def roofline_entry(st, kr, local_cells, cells, K, elapsed, world, workload, spec):
    h_ms = st.h_kernel_ms / max(1, st.h_kernel_launches)
    e_ms = st.e_kernel_ms / max(1, st.e_kernel_launches)
    f_ms = st.fused_kernel_ms / max(1, st.fused_kernel_launches)
    # when boundary planes are launched separately, normalise to the per-step duration
    h_step = st.h_kernel_ms / kr
    e_step = st.e_kernel_ms / kr
    f_step = st.fused_kernel_ms / kr
    survey_bytes = 2 * BYTES_PER_CELL_PASS * local_cells        # 72 B per cell-step, two passes
    two_step = int(getattr(st, "fused2_pairs", 0)) > 0
    steps_per_launch = 2 if two_step else 1
    if two_step:
        # ONE launch advances E and H by TWO time steps.  The events bracket fused2_step_kernel ALONE (what rocprofv3 reports for it;
        # until round 6 they took the seam kernel behind it in: `seam_kernel_ms` now, and part of `whole_step_frac`).  `frac` is
        # priced on what THIS launch must move at least — 6 reads + 6 writes per cell, once, for both steps — so it is a
        # fraction of the HBM peak by construction (<= 1).  How far temporal blocking carries the throughput past what a
        # one-step-per-pass kernel can reach is reported under explicit names (`throughput_vs_*`), not as a fraction.
        dom, dom_ms = "fused2_step_kernel", f_ms
        dom_bytes = 48.0 * local_cells
    elif st.fused_kernel_launches:
        # one launch advances E and H.  `frac` is priced against what THIS kernel must move at least
        # (48 B per cell-step + psi), so it cannot exceed 1
        dom, dom_ms = "fused_step_kernel", f_step
        dom_bytes = min_bytes_per_cell(workload, spec) * local_cells
    elif e_step >= h_step:
        dom, dom_ms, dom_bytes = "e_update_kernel", e_step, BYTES_PER_CELL_PASS * local_cells
        survey_bytes = dom_bytes
    else:
        dom, dom_ms, dom_bytes = "h_update_kernel", h_step, BYTES_PER_CELL_PASS * local_cells
        survey_bytes = dom_bytes
    achieved = dom_bytes / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
    cell_steps_per_s = local_cells * steps_per_launch / (dom_ms * 1e-3) if dom_ms > 0 else 0.0     # of the dominant kernel
    r = {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": None, "traffic_source": None,
         "traffic_frac": None, "traffic_over_minimum": None,
         # throughput of the dominant kernel against the throughput roofline of a ONE-step-per-pass kernel (HBM peak / bytes
         # per cell-step): ratios of cell rates, above 1 when temporal blocking moves fewer bytes per step than that kernel must
         "throughput_vs_single_sweep_minimum": cell_steps_per_s * 48.0 / HBM_PEAK,
         "throughput_vs_survey_8d_roofline": cell_steps_per_s * 2 * BYTES_PER_CELL_PASS / HBM_PEAK,
         "avg_launch_ms": {"h_update_kernel": h_ms, "e_update_kernel": e_ms,
                           ("fused2_step_kernel" if two_step else "fused_step_kernel"): f_ms},
         "per_step_ms": {"h_update_kernel": h_step, "e_update_kernel": e_step,
                         ("fused2_step_kernel" if two_step else "fused_step_kernel"): f_step},
         "seam_kernel_ms_per_launch": (st.seam_kernel_ms / max(1, st.seam_kernel_launches)) if two_step else None,
         "algorithmic_bytes_per_launch": dom_bytes,
         "algorithmic_bytes_per_cell": dom_bytes / local_cells,
         "time_steps_per_launch": steps_per_launch,
         # the whole timed region (every launch of a step, all ranks) priced on the same per-launch minimum
         "whole_step_frac": (dom_bytes / local_cells / steps_per_launch * cells * K / elapsed) / (HBM_PEAK * world),
         "whole_step_throughput_vs_survey_8d_roofline": (2 * BYTES_PER_CELL_PASS * cells * K / elapsed) / (HBM_PEAK * world)}
    if two_step:
        shape = int(st.fused2_shape)
        r["two_steps_per_sweep"] = {"waves_per_workgroup": shape & 63, "planes_per_chunk": shape >> 6,
                                    "pairs_in_this_run": int(st.fused2_pairs)}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and workload == "v0":
        try:
            rec = json.load(open(pmc))
            t512 = rec.get(dom)                        # PMC bytes of one 512^3 launch (separate rocprofv3 --pmc passes)
            # only a figure measured on THIS code counts (the record carries the hash of the kernel sources)
            if t512 is not None and rec.get("source_hash") == source_hash():
                r["traffic"] = t512 * local_cells / 512 ** 3      # per launch like `achieved`
                r["traffic_frac"] = (r["traffic"] / (dom_ms * 1e-3) / HBM_PEAK) if dom_ms > 0 else None
                r["traffic_over_minimum"] = r["traffic"] / dom_bytes
                r["traffic_source"] = {"file": rec.get("file"), "commit": rec.get("commit"),
                                       "source_hash": rec.get("source_hash")}
        except Exception:
            pass
    return r


def _workload_traffic(workload, cells, two_step):
    """PMC bytes of one time step of a secondary workload (all its launches; half a step pair's when it runs in pairs), from
    profiles/pmc_traffic.json when it was measured on THIS kernel source (hash), scaled from 512^3; else None."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if rec.get("source_hash") != source_hash():
            return None
        w = (rec.get("workloads") or {}).get(workload)
        if two_step and w:
            return w["bytes_per_pair"] / 2 * cells / 512 ** 3
        if not two_step and workload == "v2" and rec.get("v2_step_bytes") and not w:
            return rec["v2_step_bytes"] * cells / 512 ** 3
    except Exception:
        pass
    return None


_INIT = {}


def init_plane(c, k, n):
    """synthetic initial data: uniform random +-1e-3, one generator per (component, plane) -> the same global field at any N"""
    key = (c, k, n)
    if key not in _INIT:
        if len(_INIT) > 6 * 512:
            _INIT.clear()
        _INIT[key] = np.random.default_rng(c * 100003 + k).uniform(-1e-3, 1e-3, (n, n)).astype(np.float32)
    return _INIT[key]


MIN_TIMED_STEPS = 100        # a timed region shorter than this is raised to it (20 steps are 14 ms)


def secondary_workload(HipEngine, L, n, workload, device, args, steps=100, repeats=3):
    """One more single-GPU measurement on the same box: same grid, another SURVEY 8(d) variant (build_spec)."""
    spec = build_spec(n, steps * (repeats + 2) + 64, workload)
    eng = HipEngine(spec, device=device, variant=args.variant, z_chunk=args.zchunk)
    try:
        if args.rows:
            eng.set_option(L.OPT_ROWS, args.rows)
        if args.pml_fused >= 0:
            eng.set_option(L.OPT_PML_FUSED, args.pml_fused)
        for c in range(6):
            arr = np.empty((n, n, n), dtype=np.float32)
            for k in range(n):
                arr[k] = init_plane(c, k, n)
            eng.set_field(c, arr)
        import torch
        eng.run(10)
        samples = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(steps)
            torch.cuda.synchronize()
            samples.append(time.perf_counter() - t0)
        el = float(np.median(samples))
        eng.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
        eng.run(10)
        st = eng.stats()
        cells = n ** 3
        two_step = int(st.fused2_pairs) > 0
        single = None
        if two_step:                                   # the same engine advancing one step per sweep
            eng.set_option(L.OPT_FLAGS, 0)
            eng.set_option(L.OPT_TWOSTEP, 0)
            eng.run(10)
            ss = []
            for _ in range(repeats):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.run(steps)
                torch.cuda.synchronize()
                ss.append(time.perf_counter() - t0)
            single = float(np.median(ss))
        own = min_bytes_per_cell(workload, spec)
        survey = 72.0 + (own - 48.0) + 4.0          # two passes + psi + material word: the SURVEY.md 8(d) accounting
        if two_step:
            # two steps per sweep: the fields cross the HBM interface once per PAIR (24 B per cell-step); psi moves every step
            # (a lower bound for shell pairs too: their shell — CPML slabs + collar — still moves its fields every step)
            own = 24.0 + (own - 48.0)
            if int(getattr(st, "shell2_pairs", 0)) > 0:     # shell2 pairs: psi crosses the interface once per PAIR too (both sides ping-ponged)
                own = 24.0 + 0.5 * (own - 24.0)
        traffic = _workload_traffic(workload, cells, two_step)
        return {"workload": f"{workload}: {WORKLOADS[workload]}", "value": cells * steps / el / 1e6, "unit": "Mcells/s",
                "ms_per_step": el / steps * 1e3, "steps": steps, "repeats": repeats,
                "ms_per_step_samples": [e / steps * 1e3 for e in samples],
                # the step goes out as three CONCURRENT sweep launches (interior tiles on the main stream, z-slab planes and
                # y-edge tile rows on the second): their durations overlap, so the sum is not a time per step
                "sweep_launches_per_step": st.fused_kernel_launches / 10,
                "sweep_launch_ms_sum_concurrent": st.fused_kernel_ms / 10,
                "stream_overlap": int(st.stream_overlap),
                "two_steps_per_sweep": ({"pairs_in_10_steps": int(st.fused2_pairs), "waves_per_workgroup": int(st.fused2_shape) & 63,
                                         "planes_per_chunk": int(st.fused2_shape) >> 6,
                                         "single_steps_ms_per_step": single / steps * 1e3,
                                         "single_steps_value": cells * steps / single / 1e6} if two_step else None),
                "traffic_per_step": traffic, "traffic_frac": (traffic / (el / steps) / HBM_PEAK) if traffic else None,
                "traffic_over_minimum": (traffic / (own * cells)) if traffic else None,
                "shell_pairs_in_10_steps": int(st.shell_pairs),
                # of those: pairs whose shell (CPML slabs + collar) went out as shell2_step_kernel launches — two steps per sweep with
                # psi carried — instead of two single steps (round 5)
                "shell2_pairs_in_10_steps": int(getattr(st, "shell2_pairs", 0)),
                # pairs that advanced the dispersive (ADE) cells themselves: the sweep subtracts their paged memory terms, ade2_kernel
                # follows it (round 6; without it the planes of a dispersive body take single steps)
                "disp_pairs_in_10_steps": int(getattr(st, "disp_pairs", 0)),
                "bytes_per_cell_own_minimum": own,
                "whole_step_frac": own * cells * steps / el / HBM_PEAK,
                "bytes_per_cell_survey_8d": survey,
                "whole_step_throughput_vs_survey_8d_roofline": survey * cells * steps / el / HBM_PEAK}
    finally:
        eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", "--n", dest="n", type=int, default=512, help="cells per axis")
    ap.add_argument("--workload", default="v0", choices=["v0", "v1", "v2", "v3", "v4", "va", "v1a", "v4a"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--zchunk", type=int, default=0)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--pml-fused", type=int, default=-1, help="axis mask of the CPML recursions folded into the fused sweep (0, 6, 7)")
    ap.add_argument("--tile-order", type=int, default=-1, help="-1: the library times both tile orders on the first sweep (default); "
                    "0 / 1: plain / XCD-aware (profiling runs: keeps the probe sweeps out of the counters)")
    ap.add_argument("--tblock", type=int, default=-1, help="planes per slab of the two-step slab-interleaved schedule (FDTD_OPT_TBLOCK; "
                    "0 = single steps, -1 = library default)")
    ap.add_argument("--placement-tries", type=int, default=-1, help="FDTD_OPT_PLACEMENT_TRIES (0 = keep the first allocations: profiling "
                    "runs, so that the kernel statistics hold the timed sweeps only)")
    ap.add_argument("--opt", action="append", default=[], help="extra engine option NAME=VALUE (tidy3d_amd.lib.OPT_*), e.g. OPT_PML_POOL=0; repeatable")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--no-workloads", action="store_true", help="skip the secondary V2 (materials + CPML) measurement")
    ap.add_argument("--no-single-steps", action="store_true", help="skip the single-steps leg of a run that goes out in step pairs "
                    "(profiling runs: the counters then hold the launches of the pairs only)")
    ap.add_argument("--sweep", action="store_true", help="A/B kernel launch parameters (N=1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tidy3d_amd import lib as L
    from tidy3d_amd.engine import HipEngine, split_slabs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_SINGLE_DEVICE"):      # debugging aid: several ranks on one GPU
        local_rank = 0
    # BENCH_EMULATE=1 (tests/test_bench_dryrun.py): the same rank logic on CPU — gloo instead of RCCL for the
    # torch side, the HIP sources under the emulator with its RCCL shim.  Never set on a GPU box.
    emulate = bool(os.environ.get("BENCH_EMULATE"))
    lib = None
    if not emulate:
        L.load_library()          # before the first HIP call of this process (torch.cuda.set_device below): see lib._prefer_hw_queues
    if emulate:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
        import build_emu
        import dist_worker
        lib = L.load_library(build_emu.build())
        _cb = dist_worker.EXCHANGE_FN(dist_worker._exchange)
        lib.dll.hipemu_set_exchange(_cb, None)
        local_rank = 0
    if world > 1:
        if emulate:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    n, K_asked, W = args.n, args.steps, args.warmup
    # timed regions of at least MIN_TIMED_STEPS steps: a caller that asks for fewer gets that many, and the line says so
    K = max(K_asked, MIN_TIMED_STEPS) if not os.environ.get("BENCH_EMULATE") else K_asked
    spec = build_spec(n, K + W + 64, args.workload)
    slabs = split_slabs(n, world)
    eng = HipEngine(spec, lib=lib, device=local_rank, variant=args.variant, z_chunk=args.zchunk,
                    slab=slabs[rank], rank=rank, n_ranks=world)
    if args.rows:
        eng.set_option(L.OPT_ROWS, args.rows)
    if args.pml_fused >= 0:
        eng.set_option(L.OPT_PML_FUSED, args.pml_fused)
    if args.tile_order >= 0:
        eng.set_option(L.OPT_XCD_REMAP, args.tile_order)
    if args.tblock >= 0:
        eng.set_option(L.OPT_TBLOCK, args.tblock)
    if args.placement_tries >= 0:
        eng.set_option(L.OPT_PLACEMENT_TRIES, args.placement_tries)
    for kv in args.opt:
        name, val = kv.split("=")
        eng.set_option(getattr(L, name), int(val))
    if world > 1:
        uid = [eng.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(uid[0])
    # synthetic initial data (uniform random +-1e-3, rng seed 0 -> same global field at any N)
    z0, z1 = slabs[rank]
    for c in range(6):
        # generate only this slab's planes, reproducibly: one generator per (component, plane)
        arr = np.empty((z1 - z0, n, n), dtype=np.float32)
        for k in range(z0, z1):
            arr[k - z0] = init_plane(c, k, n)
        eng.set_field(c, arr)

    def dev_sync():
        if not emulate:
            torch.cuda.synchronize()

    def sync():
        dev_sync()
        if world > 1:
            dist.barrier()

    def timed(k):
        sync()
        t0 = time.perf_counter()
        eng.run(k)
        dev_sync()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0

    eng.run(W)
    # REPEATS timed regions of exactly K steps each (barrier + device sync on both sides, max over ranks);
    # the median is the reported number, the spread is reported beside it (median of 5)
    R = max(1, args.repeats)
    samples, rank_samples = [], []
    for _ in range(R):
        e = timed(K)
        if world > 1:
            # every rank's own clock around the same K steps: the max is the job's time, min / max per rank go on the line
            t = torch.tensor([e], dtype=torch.float64, device="cpu" if emulate else "cuda")
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per_rank = [float(x.item()) for x in allt]
            rank_samples.append(per_rank)
            e = max(per_rank)
        samples.append(e)
    elapsed = float(np.median(samples))
    cells = n ** 3
    value = cells * K / elapsed / 1e6

    # roofline of the dominant kernel: separate short run with per-launch hipEvents on the launch stream
    eng.set_option(L.OPT_FLAGS, L.FLAG_TIME_KERNELS)
    kr = min(K, 100)            # (round 6: 100 steps, not 20 — ten launches straight after the timed region read 2 - 5 % long)
    eng.run(kr)
    st = eng.stats()
    eng.set_option(L.OPT_FLAGS, 0)
    local_cells = (z1 - z0) * n * n
    out = {
        "metric": "Mcells/s on 512^3 Yee grid", "value": value, "unit": "Mcells/s",
        "n_gpus": world, "steps": K, "steps_requested": K_asked, "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]}; {n}^3 Yee cells, Ez point dipole "
                               "(GaussianPulse 200 THz), random +-1e-3 initial fields",
                   "grid": [n, n, n], "parallelism": f"z-slab x{world}",
                   "tile": {"rows": int(st.tile_rows), "zchunk": int(st.tile_zchunk), "xcd_order": int(st.tile_order),
                            "placement": {"tried": int(st.placement) >> 8, "kept": int(st.placement) & 255,
                                          "probe_ms_first": float(st.placement_ms_first), "probe_ms_kept": float(st.placement_ms_kept)},
                            "how": "library default" if not (args.rows or args.zchunk) else "flags"},
                   "bytes_per_cell_step": 2 * BYTES_PER_CELL_PASS,
                   "roofline_mcells_per_gpu": HBM_PEAK / (2 * BYTES_PER_CELL_PASS) / 1e6},
        # where the twelve field arrays landed in device memory decides up to 15 % of the step time on this part (DESIGN.md section 7:
        # "placement"): the engine times up to three alternative allocations on its first large run and keeps the fastest — the line
        # says how many it tried, which it kept and what the probe sweeps took (on the top level, not in config)
        "placement": {"tried": int(st.placement) >> 8, "kept": int(st.placement) & 255,
                      "probe_ms_first": float(st.placement_ms_first), "probe_ms_kept": float(st.placement_ms_kept),
                      "note": "box-to-box spread of the headline (181-198 Gcells/s over round 5's visits) is this effect, not the kernel"},
        "repeats": {"n": R, "statistic": "median", "ms_per_step": [e / K * 1e3 for e in samples],
                    "min_ms_per_step": min(samples) / K * 1e3, "max_ms_per_step": max(samples) / K * 1e3},
        "roofline": roofline_entry(st, kr, local_cells, cells, K, elapsed, world, args.workload, spec),
        "schedule": {"two_steps_per_sweep_pairs_in_roofline_run": int(st.fused2_pairs),
                     "two_step_pairs_in_roofline_run": int(st.two_step_pairs), "tblock_planes": int(st.tblock_planes),
                     "stream_overlap": int(st.stream_overlap), "stream_retries": int(st.stream_retries)},
    }
    if world > 1:
        # proof that the halo communicator (RCCL; the emulator's shim under BENCH_EMULATE) spans `world` ranks: every rank
        # reports what ncclCommCount / ncclCommUserRank return for ITS communicator
        mine = torch.tensor([int(st.comm_ranks), int(st.comm_rank), z1 - z0, int(st.fused2_pairs), int(st.fused2_shape), int(st.fused2_off_reason)],
                            dtype=torch.int64, device="cpu" if emulate else "cuda")
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        rows = [[int(v) for v in x.tolist()] for x in allc]
        med = np.median(np.asarray(rank_samples), axis=0) / K * 1e3
        out["rccl"] = {"rccl_ranks": sorted({r[0] for r in rows}), "comm_user_ranks": [r[1] for r in rows],
                       "planes_per_rank": [r[2] for r in rows],
                       # z-slab ranks advance two time steps per sweep too (the planes next to a cut by single steps on the comm
                       # stream, shipping after each): step pairs of every rank in the roofline run, their tile shape, and — where
                       # a rank took none — why (FDTD_F2_OFF_*)
                       "two_steps_per_sweep": {"steps_in_this_run": kr, "pairs_per_rank": [r[3] for r in rows],
                                               "waves_per_workgroup": [r[4] & 63 for r in rows], "planes_per_chunk": [r[4] >> 6 for r in rows],
                                               "off_reason_per_rank": [r[5] for r in rows]},
                       "ms_per_step_per_rank": [float(v) for v in med],
                       "ms_per_step_rank_min": float(med.min()), "ms_per_step_rank_max": float(med.max())}
    if world == 1 and int(st.fused2_pairs) > 0 and not args.no_single_steps:
        # the same engine (same placement of the arrays) advancing ONE step per sweep: what the headline was before
        # the two-step kernel, and what every run outside its scope still gets
        eng.set_option(L.OPT_TWOSTEP, 0)
        eng.run(10)
        ss = [timed(MIN_TIMED_STEPS) for _ in range(3)]
        eng.set_option(L.OPT_TWOSTEP, -1)
        el1 = float(np.median(ss))
        out["single_steps"] = {"value": cells * MIN_TIMED_STEPS / el1 / 1e6, "unit": "Mcells/s", "ms_per_step": el1 / MIN_TIMED_STEPS * 1e3,
                               "steps": MIN_TIMED_STEPS, "repeats": 3, "kernel": "fused_step_kernel"}
    if world == 1 and not emulate and args.workload == "v0" and not args.no_workloads:
        # the workload every real simulation resembles (materials + CPML on all faces), same grid, same box
        eng.close()
        eng = None
        out["workloads"] = {"v2": secondary_workload(HipEngine, L, n, "v2", local_rank, args),
                            # materials inside PEC walls: the two-step sweep's materials instantiation
                            "v1": secondary_workload(HipEngine, L, n, "v1", local_rank, args),
                            # an open problem without CPML: Absorber boundaries, damped inside the two-step sweep
                            "va": secondary_workload(HipEngine, L, n, "va", local_rank, args),
                            # + a Lorentz pole in the sphere — K4 (ADE) inside the step pairs since round 6
                            "v3": secondary_workload(HipEngine, L, n, "v3", local_rank, args),
                            # + a closed flux box with a running DFT at three frequencies — K6 inside the pairs
                            "v4": secondary_workload(HipEngine, L, n, "v4", local_rank, args)}

    if args.sweep and world == 1 and eng is not None:
        res = []
        for rows in (1, 2, 4, 8):
            for zc in (1, 4, 8, 16, 32, 64, 128):
                eng.set_option(L.OPT_ROWS, rows)
                eng.set_option(L.OPT_ZCHUNK, zc)
                eng.run(3)
                t = timed(20)
                res.append({"rows": rows, "zchunk": zc, "mcells": cells * 20 / t / 1e6})
                print(json.dumps(res[-1]), file=sys.stderr, flush=True)
        out["sweep"] = res

    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline()
    if eng is not None:
        eng.close()
    if rank == 0:
        # (RCCL's version banner — the GPU boxes export NCCL_DEBUG=VERSION — sits in the C library's stdout buffer since the communicators
        #  were made and would come out at exit, BEHIND the line: push it out first, so that the JSON line is the last one this rank prints)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
