"""Multi-GPU plumbing: one process per GPU (torchrun), z-slab decomposition, RCCL halo exchange.

``torch.distributed`` is used for *plumbing only*: broadcasting the RCCL unique id that the C
library's communicator is created from, barriers, and gathering monitor buffers to rank 0 after
the run.  The per-step ghost-plane exchange (ncclSend/ncclRecv over xGMI, overlapped with the
interior update on a second HIP stream) lives entirely inside ``libfdtd_hip.so``
(csrc/fdtd_capi.hip, ``exchange``); there is no data-path collective besides that
nearest-neighbour exchange and one scalar all-reduce per field-decay evaluation
(SURVEY.md section 8(e)).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

from .engine import (HipEngine, balanced_slabs, lane_efficiency, permute_spec, split_slabs,  # noqa: F401
                     unpermute_array)
from .spec import SolverSpec


def env_ranks() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def best_slab_shift(shape, world: int) -> int:
    """Cyclic axis renaming (engine.permute_spec) for a z-slab run: the slab axis should be the one with the most
    planes — thicker slabs (a 1024 x 1024 x 256 grid on 8 GPUs: 128 planes of 1024 x 256 cells per rank instead of 32
    planes of 1024 x 1024) mean 4 x smaller ghost planes and a better interior / boundary-chunk ratio — as long as
    the rows along the new x stay filled (within 20 % of the best).  Only when it at least doubles the planes of a
    grid of 2^18 cells or more.  Every rank derives the same answer."""
    if int(np.prod([int(n) for n in shape])) < (1 << 18):
        return 0
    eff = [lane_efficiency(int(shape[(0 + s) % 3])) for s in range(3)]
    best, best_nz = 0, 2 * int(shape[2]) - 1
    for s in (1, 2):
        nz = int(shape[(2 + s) % 3])
        if nz > best_nz and eff[s] >= 0.8 * max(eff) and nz >= 4 * world:
            best, best_nz = s, nz
    return best


def cpml_pairs_possible(spec: SolverSpec, slabs=None) -> bool:
    """May the z-slab ranks of this (already renamed) problem advance in step pairs although they carry CPML?  What the library's
    shell2 pairs need of the WHOLE problem: CPML on some face, nothing else in the shell (no periodic x / y face, no absorber layers,
    no dispersive or fully anisotropic media, no Bloch phases, no PMC-plus wall).  Every rank derives the same answer."""
    from .spec import BC_PERIODIC
    if not any(f.num_layers > 0 for ax in spec.pml for f in ax):
        return False
    if spec.absorber is not None or spec.bloch is not None or spec.aniso or spec.mirror_plus is not None:
        return False
    if any(m.poles for m in spec.media):
        return False
    if any(int(b) == BC_PERIODIC for ax in spec.bc[:2] for b in ax):
        return False
    if slabs is not None:
        # every rank must be able to take them — else the in-sweep recursions only cost (on thin slabs the slab kernels are 4 - 16 %
        # faster in single steps, profiles/r04p): 2^20 cells (the library's threshold for step pairs) and room for a bulk of eight
        # planes between the cuts' holes and the z layers
        lz = [int(f.num_layers) for f in spec.pml[2]]
        for q, (z0, z1) in enumerate(slabs):
            room = (z1 - z0) - (2 if q > 0 else lz[0] + 2) - (2 if q < len(slabs) - 1 else lz[1] + 1)
            if room < 8 or spec.shape[0] * spec.shape[1] * (z1 - z0) < (1 << 20):
                return False
    return True


def make_engine(spec: SolverSpec, lib=None, device: Optional[int] = None, axis_shift: Optional[int] = None,
                **kw) -> HipEngine:
    """This rank's slab engine with its RCCL communicator initialised (needs an initialised
    torch.distributed process group when WORLD_SIZE > 1)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return HipEngine(spec, lib=lib, device=device or 0, **kw)
    rank, world = dist.get_rank(), dist.get_world_size()
    shift = best_slab_shift(spec.shape, world) if axis_shift is None else int(axis_shift) % 3
    user_z = {m.name: (int(m.lo[2]), int(m.hi[2])) for m in spec.monitors}
    spec = permute_spec(spec, shift)          # the slab axis of the renamed problem is its z
    nz = spec.shape[2]
    if nz < 2 * world:
        raise ValueError(f"{nz} planes cannot be split into {world} slabs of >= 2 planes")
    slabs = balanced_slabs(spec, world)       # equal modelled cost, not equal plane counts
    if cpml_pairs_possible(spec):
        # the ranks of this problem will advance in shell2 step pairs: price the planes as the pairs do (a z-layer plane 2.2 x a bulk
        # plane, not 3.6 x) — where that split leaves every rank room for its pairs; else the single-step split stands
        paired = balanced_slabs(spec, world, pairs=True)
        if cpml_pairs_possible(spec, paired):
            slabs = paired
    if device is None:
        device = env_ranks()[2]
    eng = HipEngine(spec, lib=lib, device=device, slab=slabs[rank], rank=rank, n_ranks=world,
                    all_slabs=slabs, **kw)
    eng.slab_shift, eng.slab_user_z = shift, user_z     # gather_results renames the stitched boxes back
    if cpml_pairs_possible(spec, slabs):
        # CPML recursions inside the sweeps of EVERY rank (decided from the whole problem: the ranks then post the same messages):
        # the state step pairs of CPML-carrying slab ranks start from and end in (fdtd_capi.hip, Run::slab_shell2_pair).
        # The exchange ships the top plane's psi_H when 64 * (rows + 1) <= 512 holds on a rank: FDTD_OPT_ROWS must be the SAME on every
        # rank while this option is on (a rank with 8 or more rows per workgroup would post four messages fewer per cut and the run
        # would hang) — nothing in this package sets it per rank.  A rank whose own monitors or sources keep it from pairs (a time
        # monitor recording every step, a decay check every other step) pays the in-sweep recursions without them: 4 - 16 % on thin
        # slabs (profiles/r04p); cpml_pairs_possible judges the whole problem only.
        from . import lib as L
        eng.set_option(L.OPT_PML_FUSED, 7)
    uid = [eng.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    eng.comm_init(uid[0])
    return eng


def gather_results(eng: HipEngine) -> Optional[Dict[str, np.ndarray]]:
    """Collect every monitor's slab-local part on rank 0 and stitch along z.  Returns the
    full-box arrays (same layout as HipEngine.results()) on rank 0, None elsewhere."""
    import torch.distributed as dist
    local = eng.monitor_data()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {k: v[0] for k, v in local.items()}
    rank, world = dist.get_rank(), dist.get_world_size()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0)
    if rank != 0:
        return None
    out = {}
    for m in eng.spec.monitors:
        parts = [(g[m.name][1], g[m.name][0]) for g in gathered if m.name in g]
        if not parts:
            continue
        parts.sort(key=lambda p: p[0][0])
        out[m.name] = np.concatenate([p[1] for p in parts], axis=2)
        assert out[m.name].shape[2] == m.hi[2] - m.lo[2], (m.name, out[m.name].shape)
        out[m.name] = unpermute_array(out[m.name], getattr(eng, "slab_shift", 0))
    return out


def run(simulation, verbose: bool = True, n_steps: Optional[int] = None, lib=None, **kw):
    """Distributed counterpart of ``tidy3d_amd.web.run``: every rank calls it with the same
    Simulation; rank 0 returns the SimulationData, the others None."""
    import torch.distributed as dist
    from .data import assemble
    from .discretize import discretize
    from .web import _as_mirror
    sim, _ = _as_mirror(simulation)
    sim.validate_pre_upload(source_required=True)
    disc = discretize(sim, n_steps=n_steps)
    eng = make_engine(disc.spec, lib=lib, **kw)
    try:
        stats = eng.run()
        raw = gather_results(eng)
    finally:
        eng.close()
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank != 0:
        return None
    return assemble(disc, raw, log=f"distributed run over {eng.n_ranks} z-slabs", diverged=bool(stats.diverged),
                    n_steps_run=int(stats.steps_done))
