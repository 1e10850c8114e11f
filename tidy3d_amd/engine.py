"""HipEngine: drives libfdtd_hip.so for one SolverSpec (or one z-slab of it).

Python here is plumbing only: it casts the fp64 coefficient tables of
``tidy3d_amd.coeffs`` to fp32, slices them to this rank's z-slab, hands them to
the C ABI (include/fdtd_hip.h) and reads monitor buffers back.  All time stepping
happens inside ``fdtd_run`` on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from . import lib as L
from .coeffs import damping_tables, h_coeff, inv_steps, material_table, pml_axis
from .exceptions import SetupError, SolverLibraryError
from .spec import BC_PERIODIC, MonitorSpec, SolverSpec


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _cplx_f32(a) -> np.ndarray:
    """complex array -> float32 array with (re, im) interleaved on the last axis."""
    a = np.asarray(a, dtype=np.complex128)
    out = np.empty(a.shape + (2,), dtype=np.float32)
    out[..., 0] = a.real
    out[..., 1] = a.imag
    return np.ascontiguousarray(out)


def split_slabs(nz: int, n_ranks: int) -> List[Tuple[int, int]]:
    """Contiguous z-slabs, as equal as possible."""
    base, rem = divmod(nz, n_ranks)
    out, z = [], 0
    for r in range(n_ranks):
        n = base + (1 if r < rem else 0)
        out.append((z, z + n))
        z += n
    return out


# per cell and step PAIR on MI355X (fdtd_capi.hip kShell2*Ps, profiles/r5/r5d, r5e): the clipped two-step sweep over the bulk (with
# material words), a cell of the shell's wide boxes (y / z slabs), a cell of its x strips
PAIR_BULK_PS, PAIR_BULK_MAT_PS, PAIR_WIDE_PS, PAIR_STRIP_PS = 11.3, 12.9, 27.0, 41.0


def plane_costs(spec: SolverSpec, pairs: bool = False) -> np.ndarray:
    """Relative cost of one xy-plane per step, from the measured per-cell kernel times on MI355X
    (DESIGN.md section 5): fused sweep 9.3 ps/cell (+5 % with material words), CPML slab kernels
    24 ps per cell and PML-axis membership, ADE 21 ps per dispersive cell and pole.

    ``pairs``: the ranks advance in shell2 step pairs (dist.cpml_pairs_possible) — a plane then costs what the pair's launches take
    of it: the bulk's cells the clipped two-step sweep, the ring of x / y layers (+ the two-cell collar) the shell's strips and wide
    boxes, and a plane inside the z layers is wide-box cells throughout: 2.2 x a bulk plane, where the single-step model says 3.6 x
    (round 6: end ranks of a z-CPML problem were handed too few planes; scripts/probe_end_rank.py measures the split)."""
    nx, ny, nz = spec.shape
    if pairs:
        (xl, xh), (yl, yh), (zl, zh) = [(p[0].num_layers, p[1].num_layers) for p in spec.pml]
        ox = nx - (xl + 2 if xl else 0) - (xh + 2 if xh else 0)
        oy = ny - (yl + 2 if yl else 0) - (yh + 2 if yh else 0)
        bulk = PAIR_BULK_MAT_PS if spec.mat_idx is not None else PAIR_BULK_PS
        ring = (nx - ox) * ny * PAIR_STRIP_PS + ox * (ny - oy) * PAIR_WIDE_PS
        cost = np.full(nz, 0.5 * (ox * oy * bulk + ring))
        zpl = 0.5 * ((nx - ox) * ny * PAIR_STRIP_PS + ox * ny * PAIR_WIDE_PS)
        if zl:
            cost[:zl + 2] = zpl
        if zh:
            cost[nz - zh - 2:] = zpl
        return cost
    cost = np.full(nz, nx * ny * 9.3 * (1.05 if spec.mat_idx is not None else 1.0))
    (xl, xh), (yl, yh), (zl, zh) = [(p[0].num_layers, p[1].num_layers) for p in spec.pml]
    cost += 24.0 * ((xl + xh) * ny + (yl + yh) * nx)
    cost[:zl] += 24.0 * nx * ny
    if zh:
        cost[nz - zh:] += 24.0 * nx * ny
    if spec.absorber is not None:            # damping kernel: 48 B per cell of a layer
        (_, _, axl, axh), (_, _, ayl, ayh), (_, _, azl, azh) = spec.absorber
        cost += 8.0 * ((axl + axh) * ny + (ayl + ayh) * nx)
        cost[:azl] += 8.0 * nx * ny
        if azh:
            cost[nz - azh:] += 8.0 * nx * ny
    if spec.mat_idx is not None:
        npoles = np.array([len(m.poles) for m in spec.media], dtype=np.float64)
        if npoles.any():
            for c in range(3):
                cost += 7.0 * npoles[spec.mat_idx[c]].reshape(nz, -1).sum(axis=1)
    return cost


def balanced_slabs(spec: SolverSpec, n_ranks: int, min_planes: int = 4, pairs: bool = False) -> List[Tuple[int, int]]:
    """Contiguous z-slabs of (nearly) equal modelled cost instead of equal plane counts: PML z-slabs
    and dispersive regions make some planes heavier (SURVEY.md section 8(e) "load balance").  Cuts
    are kept two planes clear of the z-PML so that the fused z-slab schedule stays available; falls
    back to ``split_slabs`` when the constraints cannot be met.  Deterministic: every rank derives
    the same partition from the same spec.  ``pairs``: price the planes as shell2 step pairs do (plane_costs)."""
    nz = spec.shape[2]
    if n_ranks <= 1:
        return [(0, nz)]
    if nz < min_planes * n_ranks:
        return split_slabs(nz, n_ranks)
    cum = np.concatenate(([0.0], np.cumsum(plane_costs(spec, pairs))))
    zl, zh = spec.pml[2][0].num_layers, spec.pml[2][1].num_layers
    lo_ok = zl + 2 if zl else min_planes
    hi_ok = nz - zh - 2 if zh else nz - min_planes
    cuts = []
    prev = 0
    for r in range(1, n_ranks):
        z = int(np.argmin(np.abs(cum - cum[-1] * r / n_ranks)))
        z = max(z, prev + min_planes, lo_ok)
        z = min(z, nz - min_planes * (n_ranks - r), hi_ok)
        if z < prev + min_planes:
            return split_slabs(nz, n_ranks)
        cuts.append(z)
        prev = z
    edges = [0] + cuts + [nz]
    return [(edges[i], edges[i + 1]) for i in range(n_ranks)]


def lane_efficiency(n: int) -> float:
    """Busy fraction of the 256-cell row segments (64 lanes x float4) the main kernels give a wavefront."""
    return n / (256.0 * -(-n // 256))


def sweep_cost(shape, pml_layers=None) -> float:
    """Relative cost per cell of the fused sweep for a grid laid out as ``shape`` = (nx, ny, nz): idle lanes of the
    256-cell row segments, idle rows of the 3-row tiles, and the tiles that meet a y or z CPML slab — those run the
    all-axes instantiation at 2 waves per SIMD, about half the rate of the others; x layers put
    every tile on the x-CPML instantiation (+10 %).  Measured on
    424 x 224 x 824 with 12 layers per face (profiles/r03i_probe_c3_axis_shift.jsonl): x = 224 runs a step in
    0.82-0.99 ms, x = 424 in 1.08-1.11, x = 824 in 1.05; the model says 1.24 : 1.37 : 1.44."""
    nx, ny, nz = (int(n) for n in shape)
    lay = pml_layers if pml_layers is not None else ((0, 0),) * 3
    rows = 3.0 * -(-ny // 3) / ny
    fy = min(1.0, (lay[1][0] + lay[1][1]) / ny)
    fz = min(1.0, (lay[2][0] + lay[2][1]) / nz)
    fx = 1.10 if (lay[0][0] + lay[0][1]) > 0 else 1.0          # x layers: every tile runs the x-CPML instantiation (V2 vs V1, r03g)
    return rows / lane_efficiency(nx) * (2.0 - (1.0 - fy) * (1.0 - fz)) * fx


def source_sheet(spec: SolverSpec):
    """The long-lived source plane of a CPML-walled run, if it has one: (axis the plane is normal to, fraction of the run it injects
    for) — a mode plane, a current sheet or the injection plane of a plane wave: more nodes than the two-step sweep's node table
    takes (fdtd_fused2.hpp kMaxInj) on one grid plane (E and H nodes: two adjacent indices).  While such a list injects, its planes
    take single steps as a Z HOLE of the step pairs (fdtd_capi.hip, Run::shell2_pair) — which needs the plane normal to the
    device's z; laid out otherwise the whole grid takes single steps until the list ends.  None: no such plane."""
    # (grids below 2^20 cells take single steps whatever their layout: fdtd_capi.hip fused2_shape)
    if not any(int(f.num_layers) > 0 for ax in spec.pml for f in ax) or spec.n_steps <= 0 or spec.n_cells < (1 << 20):
        return None
    best = None
    lists = [(np.asarray(sc.ijk), max(len(sc.wave_e), len(sc.wave_h))) for sc in spec.sources]
    lists += [(np.concatenate([np.asarray(t.e_corr_ijk).reshape(-1, 3), np.asarray(t.h_corr_ijk).reshape(-1, 3)]), len(t.wave)) for t in spec.tfsf]
    for ijk, n_wave in lists:
        if len(ijk) <= 256:
            continue
        span = ijk.max(axis=0) - ijk.min(axis=0)
        flat = [a for a in range(3) if span[a] <= 1]
        if len(flat) != 1:
            continue                      # (a box of corrections — a TFSF volume — has no single plane)
        # (a run with a shutoff level seldom uses its whole run time: BASELINE config 3 stops after 1.41 x its source's length)
        n_run = min(spec.n_steps, int(1.5 * n_wave)) if spec.shutoff > 0 else spec.n_steps
        frac = min(1.0, n_wave / float(max(n_run, 1)))
        if best is None or frac > best[1]:
            best = (flat[0], frac)
    return best


def best_axis_shift(shape, pml_layers=None, sheet=None) -> int:
    """Cyclic axis shift s (new axis a holds old axis (a + s) % 3) with the lowest ``sweep_cost``; 0 unless that buys
    more than 5 % on a grid of at least 2^18 cells (below that a run is bound by launches, not lanes).
    ``pml_layers`` = ((minus, plus), ...) CPML layer counts per axis.  ``sheet`` = ``source_sheet(spec)``: a layout that does
    not put the sheet's normal along the device's z pays single steps (1.3 x the cost of step pairs, profiles/r5) for that
    fraction of the run — BASELINE config 3 (mode plane alive for 71 % of its steps): x = 224 99.8 Gcells/s over the run with
    29 % of its steps in pairs, x = 424 (the plane normal to z) 106.1 with all of them (profiles/r5/r5zc_c3_full_layouts.jsonl)."""
    if int(np.prod([int(n) for n in shape])) < (1 << 18):
        return 0
    lay = pml_layers if pml_layers is not None else ((0, 0),) * 3
    cost = [sweep_cost([shape[(a + s) % 3] for a in range(3)], [lay[(a + s) % 3] for a in range(3)]) for s in range(3)]
    if sheet is not None:
        p, frac = sheet
        cost = [c * (1.0 if (2 + s) % 3 == p else 1.0 + 0.3 * frac) for s, c in enumerate(cost)]
    s = int(np.argmin(cost))
    return s if cost[s] < 0.95 * cost[0] else 0


def permute_spec(spec: SolverSpec, s: int) -> SolverSpec:
    """The same problem with its axes cyclically renamed: new axis a = old axis (a + s) % 3, for fields,
    components, boundaries, sources and monitors alike.  Maxwell's curl is invariant under CYCLIC renaming
    (x -> y -> z -> x keeps the handedness), so the kernels solve the renamed problem unchanged; the host uses
    it to lay narrow grids out with their longest axis along x (see ``best_axis_shift``)."""
    import dataclasses
    if s % 3 == 0:
        return spec
    sig = [(a + s) % 3 for a in range(3)]                # old axis held by new axis a
    inv = [(p - s) % 3 for p in range(3)]                # new axis that holds old axis p
    tr = tuple(2 - sig[2 - q] for q in range(3))         # numpy axes (z', y', x') <- positions of old axes

    def comp_map(c):
        c = np.asarray(c)
        return (np.asarray(inv)[c % 3] + 3 * (c // 3)).astype(np.int32)

    def pick(seq):
        return tuple(seq[sig[a]] for a in range(3))
    mat = None
    if spec.mat_idx is not None:
        mat = np.ascontiguousarray(np.stack([np.transpose(spec.mat_idx[sig[c]], tr) for c in range(3)]))
    sources = [dataclasses.replace(sc, comp=comp_map(sc.comp), ijk=np.ascontiguousarray(np.asarray(sc.ijk)[:, sig]).astype(np.int32))
               for sc in spec.sources]
    tfsf = [dataclasses.replace(t, e_corr_comp=comp_map(t.e_corr_comp), h_corr_comp=comp_map(t.h_corr_comp),
                                e_corr_ijk=np.ascontiguousarray(np.asarray(t.e_corr_ijk)[:, sig]).astype(np.int32),
                                h_corr_ijk=np.ascontiguousarray(np.asarray(t.h_corr_ijk)[:, sig]).astype(np.int32))
            for t in spec.tfsf]
    monitors = [dataclasses.replace(m, comps=tuple(int(v) for v in comp_map(m.comps)), lo=pick(m.lo), hi=pick(m.hi))
                for m in spec.monitors]
    return dataclasses.replace(
        spec, shape=pick(spec.shape), boundaries=pick(spec.boundaries), bc=pick(spec.bc), pml=pick(spec.pml),
        mat_idx=mat, sources=sources, tfsf=tfsf, monitors=monitors,
        absorber=None if spec.absorber is None else list(pick(spec.absorber)),
        bloch=None if spec.bloch is None else pick(spec.bloch),
        mirror_plus=None if getattr(spec, "mirror_plus", None) is None else pick(spec.mirror_plus))


def unpermute_array(arr: np.ndarray, s: int) -> np.ndarray:
    """[..., z', y', x'] of the renamed problem -> [..., z, y, x]."""
    if s % 3 == 0:
        return arr
    inv = [(p - s) % 3 for p in range(3)]
    lead = arr.ndim - 3
    axes = tuple(range(lead)) + tuple(lead + 2 - inv[p] for p in (2, 1, 0))
    return np.ascontiguousarray(np.transpose(arr, axes))


def bloch_device_spec(spec: SolverSpec):
    """Device layout of a simulation with Bloch boundaries: every x / y axis with a non-trivial phase
    gets one ghost cell at each end (PEC faces for the kernels; the ghost cells are refilled every step
    with the rotated copy of the cell one period away, fdtd_run_bloch) — so the fused sweep and every
    other kernel run unchanged.  Returns (device spec, n_real per axis (0 = no ghost cells)); sources
    and monitors are shifted by the ghost offset, the ghost cells hold the background medium.  A Bloch
    z keeps its periodic faces: the z ghost planes the engine always had do the same job."""
    import dataclasses
    from .spec import BC_PEC
    n_real = [0, 0, 0]
    bounds = [np.asarray(b, float) for b in spec.boundaries]
    bc = [tuple(b) for b in spec.bc]
    mat = spec.mat_idx
    absorber = spec.absorber
    shape = list(spec.shape)
    for a in (0, 1):
        if spec.bloch[a] == 0.0 or spec.bc[a][0] != BC_PERIODIC or spec.shape[a] == 1:
            continue
        n_real[a] = spec.shape[a]
        b = bounds[a]
        bounds[a] = np.concatenate(([b[0] - (b[-1] - b[-2])], b, [b[-1] + (b[1] - b[0])]))
        bc[a] = (BC_PEC, BC_PEC)
        shape[a] += 2
        if mat is not None:
            pad = [(0, 0)] * 4
            pad[3 - a] = (1, 1)                          # mat_idx is [3, nz, ny, nx]
            mat = np.pad(mat, pad, mode="constant", constant_values=1)
        if absorber is not None:
            absorber = [((np.pad(sb, 1), np.pad(sc, 1), lo, hi) if ax == a else (sb, sc, lo, hi))
                        for ax, (sb, sc, lo, hi) in enumerate(absorber)]
    off = np.array([1 if n_real[a] else 0 for a in range(3)], dtype=np.int32)
    sources = [dataclasses.replace(sc, ijk=(np.asarray(sc.ijk) + off[None, :]).astype(np.int32)) for sc in spec.sources]
    monitors = [dataclasses.replace(m, lo=tuple(int(v) for v in np.asarray(m.lo) + off),
                                    hi=tuple(int(v) for v in np.asarray(m.hi) + off)) for m in spec.monitors]
    dev = dataclasses.replace(spec, shape=tuple(shape), boundaries=tuple(bounds), bc=tuple(bc), mat_idx=mat,
                              absorber=absorber, sources=sources, monitors=monitors)
    return dev, n_real


def _local_pml_counts(tables: List[np.ndarray]) -> Tuple[int, int]:
    """Leading/trailing run of planes whose CPML tables differ from identity (see engine notes):
    the library's slab ranges only need to be supersets of the true PML planes."""
    kinv_e, b_e, c_e, kinv_h, b_h, c_h = tables
    mask = (kinv_e != 1) | (b_e != 0) | (c_e != 0) | (kinv_h != 1) | (b_h != 0) | (c_h != 0)
    n = len(mask)
    if not mask.any():
        return 0, 0
    lead = int(np.argmin(mask)) if not mask.all() else n
    if lead == n:
        return n, 0
    trail = int(np.argmin(mask[::-1]))
    return lead, trail


class HipEngine:
    def __init__(self, spec: SolverSpec, lib: Optional[L.FdtdLib] = None, device: int = 0,
                 variant: int = L.VARIANT_AUTO, flags: int = 0, z_chunk: int = 0,
                 slab: Optional[Tuple[int, int]] = None, rank: int = 0, n_ranks: int = 1,
                 force_comm: bool = False, all_slabs: Optional[List[Tuple[int, int]]] = None, _bloch_twin=None,
                 axis_shift: Optional[int] = None):
        self.lib = lib or L.load_library()
        self.spec = spec
        self.rank, self.n_ranks = rank, n_ranks
        self.force_comm = bool(force_comm)
        self._all_slabs = list(all_slabs) if all_slabs is not None else None
        # Bloch boundaries: complex fields = this engine (real part) + a twin engine (imaginary part,
        # source weights times -i) on the ghost-cell device layout of bloch_device_spec, advanced together
        # by fdtd_run_bloch (one GPU)
        self.twin: Optional["HipEngine"] = None
        # narrow grids: cyclic renaming of the axes so that the best-filled one runs along x (single GPU)
        self.axis_shift = 0
        self.user_zrange = {m.name: (int(m.lo[2]), int(m.hi[2])) for m in spec.monitors}
        if getattr(spec, "aniso", None):
            axis_shift = 0                  # (the coupling lists are laid out for the user's axes)
        if _bloch_twin is None and n_ranks == 1 and slab is None and not force_comm and axis_shift != 0:
            layers = tuple((int(f0.num_layers), int(f1.num_layers)) for f0, f1 in spec.pml)
            # (round 6: lists the node table cannot hold go out as paged source terms inside the pairs — FDTD_OPT_SRC_PAGED — whatever the
            #  plane's orientation; only beside a periodic face do their planes still have to be z holes: the sheet's say in the layout)
            sheet = source_sheet(spec) if any(b == BC_PERIODIC for ax in spec.bc for b in ax) else None
            self.axis_shift = best_axis_shift(spec.shape, layers, sheet) if axis_shift is None else int(axis_shift) % 3
            spec = permute_spec(spec, self.axis_shift)
            self.spec = spec
        self.ghost = (0, 0, 0)            # ghost cells per axis in front of the real cells (Bloch device layout)
        self.user_shape = tuple(spec.shape)
        if spec.bloch is not None:
            if spec.tfsf:
                raise SolverLibraryError("TFSF sources cannot be combined with Bloch boundaries")
            if _bloch_twin is None:
                spec, self.n_real = bloch_device_spec(spec)
                self.spec = spec
            else:
                self.n_real = list(_bloch_twin)
            self.ghost = tuple(1 if n else 0 for n in self.n_real)
        nx, ny, nz = spec.shape
        self.z0, self.z1 = slab if slab is not None else (0, nz)
        self.nzl = self.z1 - self.z0
        if self.nzl < 1:
            raise SolverLibraryError("empty z-slab")
        self.handle = C.c_void_p()
        self._keep = []          # host arrays that must outlive a call
        d = self.lib.dll

        # Rows that are not a multiple of 4 cells are padded with PEC cells beyond the x-max wall —
        # exactly the PEC truncation the wall stands for (every PML is PEC-backed) — so the float4 /
        # fused kernels apply to any nx.  Periodic x cannot be padded (two-pass scalar kernels then).
        self.pad_x = (-nx) % 4 if (spec.bc[0][1] != BC_PERIODIC and variant != L.VARIANT_SIMPLE) else 0
        self.nxp = nx + self.pad_x
        cfg = L.FdtdConfig()
        cfg.nx, cfg.ny, cfg.nz = self.nxp, ny, self.nzl
        bc = [spec.bc[0][0], spec.bc[0][1], spec.bc[1][0], spec.bc[1][1], spec.bc[2][0], spec.bc[2][1]]
        per_z = spec.bc[2][0] == BC_PERIODIC
        if n_ranks > 1 or force_comm:    # force_comm: 1-rank RCCL self exchange (periodic z), a test aid
            if self.z0 > 0 or per_z:
                bc[4] = L.BC_NEIGHBOR
            if self.z1 < nz or per_z:
                bc[5] = L.BC_NEIGHBOR
        for i, b in enumerate(bc):
            cfg.bc[i] = int(b)
        if len(spec.media) > 1023:
            # a WIDE material table (more than 1022 media, ref scene.py:52 allows 65530): 16-bit indices, coefficients from global memory —
            # the two-pass kernels take it; the sweeps keep their 10-bit words and LDS table (fdtd_kernels.hpp MatP)
            if variant == L.VARIANT_FUSED:
                raise ValueError("more than 1022 media need the two-pass kernels (VARIANT_AUTO / VARIANT_ZMARCH)")
            variant = L.VARIANT_ZMARCH
        elif (n_ranks > 1 or force_comm) and spec.bloch is not None:
            variant = L.VARIANT_ZMARCH          # complex fields on z-slabs: two-pass kernels (fdtd_run_bloch)
        elif (n_ranks > 1 or force_comm) and variant in (L.VARIANT_AUTO, L.VARIANT_FUSED):
            # every rank takes the same decision (split_slabs is deterministic): fused z-slab schedule
            # iff float4-aligned rows, >= 4 planes in every slab and no slab cut inside the z-PML
            slabs = list(all_slabs) if all_slabs is not None else split_slabs(nz, n_ranks)
            ok = (self.nxp % 4 == 0 and min(b - a for a, b in slabs) >= 4
                  and self._fused_slabs_ok(spec, n_ranks, slabs))
            variant = L.VARIANT_FUSED if ok else L.VARIANT_ZMARCH
        self.variant = variant            # the schedule this engine (and every other rank) runs
        cfg.device, cfg.variant, cfg.flags, cfg.z_chunk = device, variant, flags, z_chunk
        cfg.ch = float(h_coeff(spec.dt))
        st = d.fdtd_create(C.byref(cfg), C.byref(self.handle))
        if st < 0:
            raise SolverLibraryError(f"fdtd_create failed: {self.lib.error(None)}")
        try:
            self._setup(spec)
            if spec.bloch is not None and _bloch_twin is None:
                import dataclasses
                rot = [dataclasses.replace(sc, w_re=np.asarray(sc.w_im, float), w_im=-np.asarray(sc.w_re, float))
                       for sc in spec.sources]                   # -i (w_re + i w_im) = w_im - i w_re
                self.twin = HipEngine(dataclasses.replace(spec, sources=rot), lib=self.lib, device=device,
                                      variant=variant, flags=flags, z_chunk=z_chunk, _bloch_twin=tuple(self.n_real),
                                      slab=slab, rank=rank, n_ranks=n_ranks, force_comm=force_comm, all_slabs=all_slabs)
                self.twin.user_shape = self.user_shape
        except Exception:
            self.close()
            raise

    @staticmethod
    def _fused_slabs_ok(spec: SolverSpec, n_ranks: int, slabs=None) -> bool:
        """The fused sweep keeps H^{n+1/2} of the plane below a slab in registers only, so the E-side
        CPML correction of a slab's first plane (which differentiates H along z) cannot be formed
        for it: z-slab cuts must fall outside the z-PML.  Otherwise the two-pass kernels are used."""
        nz = spec.shape[2]
        n_lo, n_hi = spec.pml[2][0].num_layers, spec.pml[2][1].num_layers
        cuts = [z0 for z0, _ in (slabs if slabs is not None else split_slabs(nz, n_ranks))][1:]
        if spec.bc[2][0] == BC_PERIODIC and n_ranks >= 1:
            cuts = cuts + [0]
        # ... and leave two planes between a cut and the z-PML: the boundary chunk next to a cut
        # is corrected on the comm stream before the neighbour's new planes arrive (fdtd_run)
        lo = n_lo + 2 if n_lo else 0
        hi = nz - n_hi - 2 if n_hi else nz
        return all(lo <= z <= hi for z in cuts)

    # ------------------------------------------------------------------ setup
    def _chk(self, st, what):
        return self.lib.check(st, self.handle, what)

    def _setup(self, spec: SolverSpec):
        d, h = self.lib.dll, self.handle
        nx0, ny, nz = spec.shape
        pad = self.pad_x
        nx = self.nxp                       # row length on the device
        z0, z1, nzl = self.z0, self.z1, self.nzl
        sxy = nx * ny
        ip, idl = inv_steps(spec)
        for a in range(3):
            p, q = _f32(ip[a]), _f32(idl[a])
            if a == 0 and pad:
                p, q = _f32(np.append(p, [p[-1]] * pad)), _f32(np.append(q, [q[-1]] * pad))
            if a == 2:
                if (z0, z1) != (0, nz) or self.n_ranks > 1 or self.force_comm:
                    # a z-slab: the steps of the planes beyond its cuts come along as ghost entries (the library only knows how to
                    # wrap / replicate its own) — the neighbour's cell below, the neighbour's cell above; across a periodic z
                    # the wrap; beyond an outer wall the replica
                    per = spec.bc[2][0] == BC_PERIODIC
                    lo_i = z0 - 1 if z0 > 0 else (nz - 1 if per else 0)
                    hi_i = z1 if z1 < nz else (0 if per else nz - 1)
                    p = _f32(np.concatenate(([p[lo_i]], p[z0:z1], [p[hi_i]])))
                    q = _f32(np.concatenate(([q[lo_i]], q[z0:z1], [q[hi_i]])))
                else:
                    p, q = _f32(p[z0:z1]), _f32(q[z0:z1])
            self._chk(d.fdtd_set_steps(h, a, _ptr(p), _ptr(q), len(p)), "fdtd_set_steps")
        self.mt = mt = material_table(spec.media, spec.dt)
        ca, cb = _f32(mt.ca), _f32(mt.cb)
        self._chk(d.fdtd_set_media(h, _ptr(ca), _ptr(cb), len(ca)), "fdtd_set_media")
        if spec.mat_idx is not None or pad:
            wide = len(ca) > 256               # more than 8 bits of medium index: the 16-bit entry point
            m = np.zeros((3, nzl, ny, nx), dtype=np.uint16 if wide else np.uint8)   # index 0 = PEC in the padding
            m[..., :nx0] = spec.mat_idx[:, z0:z1] if spec.mat_idx is not None else 1
            if wide:
                self._chk(d.fdtd_set_material16(h, _ptr(m), m.size), "fdtd_set_material16")
            else:
                self._chk(d.fdtd_set_material(h, _ptr(m), m.nbytes), "fdtd_set_material")
        # CPML
        for a in range(3):
            P = pml_axis(spec, a)
            if P.n_lo + P.n_hi == 0:
                continue
            tabs = [P.kinv_e, P.b_e, P.c_e, P.kinv_h, P.b_h, P.c_h]
            n_lo, n_hi = P.n_lo, P.n_hi
            if a == 0 and pad:          # identity in the padding; the hi slab range grows to cover it
                ident = [1.0, 0.0, 0.0, 1.0, 0.0, 0.0]
                tabs = [np.append(t, [v] * pad) for t, v in zip(tabs, ident)]
                n_hi = n_hi + pad if n_hi else 0
            if a == 2:
                tabs = [t[z0:z1] for t in tabs]
                n_lo, n_hi = _local_pml_counts(tabs)
                if n_lo + n_hi == 0:
                    continue
            t32 = [_f32(t) for t in tabs]
            self._chk(d.fdtd_set_pml(h, a, n_lo, n_hi, *[_ptr(t) for t in t32], len(t32[0])),
                      "fdtd_set_pml")
        # PMC on plus faces: the wall index of every mirrored axis (two ghost cells lie beyond it)
        mp = getattr(spec, "mirror_plus", None)
        if mp is not None and any(w >= 0 for w in mp):
            # (z-slab ranks: x / y walls cross every slab; a z wall and its two image planes belong to the last one)
            # (decided from the split alone, so that every rank raises — a lone rank giving up would leave the others waiting)
            if mp[2] >= 0 and (self.n_ranks > 1 or (z0, z1) != (0, nz)):
                last = (self._all_slabs or split_slabs(nz, self.n_ranks))[-1]
                if last[1] - last[0] < 6:
                    raise SetupError("a PMC plus face along z needs a last z-slab of at least 6 planes (the wall's two image planes and "
                                     f"the two they mirror inside its interior launch); it has {last[1] - last[0]}")
            for a, w in enumerate(mp):
                if w < 0 or (a == 2 and z1 != nz):
                    continue
                self._chk(d.fdtd_set_mirror_plus(h, a, int(w) - (z0 if a == 2 else 0)), "fdtd_set_mirror_plus")
        # absorber layers (damping tables, slab-local along z)
        dm = damping_tables(spec)
        if dm is not None:
            for a in range(3):
                D = dm[a]
                fb, fc, n_lo, n_hi = D.fb, D.fc, D.n_lo, D.n_hi
                if a == 0 and pad:      # identity in the PEC padding; the hi layers grow to cover it
                    fb, fc = np.append(fb, [1.0] * pad), np.append(fc, [1.0] * pad)
                    n_hi = n_hi + pad if n_hi else 0
                if a == 2:
                    fb, fc = fb[z0:z1], fc[z0:z1]
                    n_lo = int(np.clip(n_lo - z0, 0, nzl))
                    n_hi = int(np.clip(z1 - (nz - n_hi), 0, nzl))
                if n_lo + n_hi == 0:
                    continue
                fb32, fc32 = _f32(fb), _f32(fc)
                self._chk(d.fdtd_set_absorber(h, a, n_lo, n_hi, _ptr(fb32), _ptr(fc32), len(fb32)),
                          "fdtd_set_absorber")
        # ADE groups
        for c in range(3):
            for m in range(mt.n_media):
                if not mt.is_dispersive(m):
                    continue
                if spec.mat_idx is not None:
                    kk, jj, ii = np.nonzero(spec.mat_idx[c, z0:z1] == m)
                    idx = (kk.astype(np.int64) * sxy + jj.astype(np.int64) * nx + ii).astype(np.uint32)
                elif m == 1:
                    kk, jj, ii = np.meshgrid(np.arange(nzl), np.arange(ny), np.arange(nx0), indexing="ij")
                    idx = (kk.astype(np.int64) * sxy + jj * nx + ii).reshape(-1).astype(np.uint32)
                else:
                    continue
                if idx.size == 0:
                    continue
                kap, bet = _cplx_f32(mt.kap[m]), _cplx_f32(mt.bet[m])
                self._chk(d.fdtd_add_ade(h, c, idx.size, _ptr(idx), len(mt.kap[m]), _ptr(kap),
                                         _ptr(bet), float(mt.cc[m])), "fdtd_add_ade")
        # fully anisotropic bodies: the off-diagonal coupling lists (spec.AnisoSet) with the weights folded with the material
        # coefficients of the neighbour nodes — the curl the sweep applied at node j is (E^{n+1} - Ca E^n) / Cb there
        aniso = getattr(spec, "aniso", None) or []
        if aniso:
            if self.n_ranks > 1 or self.force_comm or (z0, z1) != (0, nz):
                from .exceptions import Tidy3dNotImplementedError
                raise Tidy3dNotImplementedError("FullyAnisotropicMedium is not available in z-slab (multi-GPU) runs")
            from .constants import EPSILON_0
            for st_ in aniso:
                n = len(st_.ijk)
                cells = (st_.ijk[:, 2] * sxy + st_.ijk[:, 1] * nx + st_.ijk[:, 0]).astype(np.uint32)
                nbr = np.full((n, 8), 0xFFFFFFFF, np.uint32)
                w_new = np.zeros((n, 8), np.float64)
                w_old = np.zeros((n, 8), np.float64)
                for slot in range(8):
                    b = st_.nbr_comp[slot]
                    j = st_.nbr_ijk[:, slot]
                    ok = j[:, 0] >= 0
                    if spec.mat_idx is not None:
                        mi = spec.mat_idx[b][j[ok, 2], j[ok, 1], j[ok, 0]]
                        ca_b, cb_b = np.asarray(mt.ca)[mi], np.asarray(mt.cb)[mi]
                    else:
                        ca_b, cb_b = np.full(int(ok.sum()), mt.ca[1]), np.full(int(ok.sum()), mt.cb[1])
                    live = cb_b != 0
                    wn = np.where(live, (spec.dt / EPSILON_0) * st_.g[ok, slot] / np.where(live, cb_b, 1.0), 0.0)
                    idx = (j[ok, 2] * sxy + j[ok, 1] * nx + j[ok, 0]).astype(np.uint32)
                    nbr[ok, slot] = np.where(wn != 0, idx, 0xFFFFFFFF)
                    w_new[ok, slot] = wn
                    w_old[ok, slot] = wn * ca_b
                c32, n32 = np.ascontiguousarray(cells), np.ascontiguousarray(nbr.reshape(-1))
                wn32, wo32 = _f32(w_new.reshape(-1)), _f32(w_old.reshape(-1))
                self._chk(d.fdtd_add_aniso(h, int(st_.comp), n, _ptr(c32), _ptr(n32), _ptr(wn32), _ptr(wo32)), "fdtd_add_aniso")
        # sources
        from .spec import BC_PEC
        for s in spec.sources:
            k = s.ijk[:, 2]
            keep = (k >= z0) & (k < z1)
            # an electric current on a PEC min wall, tangential to it, drives nothing: the wall holds that node at
            # zero (the kernels apply source terms after the wall condition, so such an entry would un-zero it)
            for a in range(3):
                if spec.bc[a][0] == BC_PEC:
                    keep &= ~((s.ijk[:, a] == 0) & (s.comp < 3) & (s.comp != a))
            if not keep.any():
                continue
            ijk = s.ijk[keep]
            cell = ((ijk[:, 2] - z0).astype(np.int64) * sxy + ijk[:, 1].astype(np.int64) * nx
                    + ijk[:, 0]).astype(np.uint32)
            comp = np.ascontiguousarray(s.comp[keep], dtype=np.int32)
            wre, wim = _f32(s.w_re[keep]), _f32(s.w_im[keep])
            we, wh = _cplx_f32(s.wave_e), _cplx_f32(s.wave_h)
            self._chk(d.fdtd_add_point_source(h, len(cell), _ptr(comp), _ptr(cell), _ptr(wre),
                                              _ptr(wim), len(s.wave_e), _ptr(we), _ptr(wh)),
                      "fdtd_add_point_source")
        for t in spec.tfsf:
            def loc(ijk, *arrs):
                k = ijk[:, 2]
                keep = (k >= z0) & (k < z1)
                ij = ijk[keep]
                cell = ((ij[:, 2] - z0).astype(np.int64) * sxy + ij[:, 1].astype(np.int64) * nx
                        + ij[:, 0]).astype(np.uint32)
                return [cell] + [np.ascontiguousarray(a[keep]) for a in arrs]
            ecell, ecomp, ew, eaux = loc(t.e_corr_ijk, t.e_corr_comp.astype(np.int32),
                                         t.e_corr_w.astype(np.float32), t.e_corr_aux.astype(np.int32))
            hcell, hcomp, hw, haux = loc(t.h_corr_ijk, t.h_corr_comp.astype(np.int32),
                                         t.h_corr_w.astype(np.float32), t.h_corr_aux.astype(np.int32))
            ae, be, ah, bh, wave = _f32(t.ae), _f32(t.be), _f32(t.ah), _f32(t.bh), _f32(t.wave)
            self._chk(d.fdtd_add_tfsf(h, t.n_aux, _ptr(ae), _ptr(be), _ptr(ah), _ptr(bh),
                                      int(t.src_cell), len(wave),
                                      _ptr(wave), len(ecell), _ptr(ecomp), _ptr(ecell), _ptr(ew),
                                      _ptr(eaux), len(hcell), _ptr(hcomp), _ptr(hcell), _ptr(hw),
                                      _ptr(haux)), "fdtd_add_tfsf")
        # monitors (intersected with the slab)
        self.mon_ids: List[Tuple[MonitorSpec, int, Tuple[int, int]]] = []
        for m in spec.monitors:
            lo2, hi2 = max(m.lo[2], z0), min(m.hi[2], z1)
            if hi2 <= lo2:
                self.mon_ids.append((m, -1, (0, 0)))
                continue
            comps = np.asarray(m.comps, dtype=np.int32)
            lo = np.asarray([m.lo[0], m.lo[1], lo2 - z0], dtype=np.int32)
            hi = np.asarray([m.hi[0], m.hi[1], hi2 - z0], dtype=np.int32)
            steps = np.ascontiguousarray(m.steps, dtype=np.int64)
            if m.kind == "dft":
                pe, ph = _cplx_f32(m.phase_e), _cplx_f32(m.phase_h)
                mid = d.fdtd_add_monitor(h, L.MON_DFT, len(comps), _ptr(comps), _ptr(lo), _ptr(hi),
                                         len(steps), _ptr(steps), len(m.freqs), _ptr(pe), _ptr(ph))
            else:
                mid = d.fdtd_add_monitor(h, L.MON_TIME, len(comps), _ptr(comps), _ptr(lo), _ptr(hi),
                                         len(steps), _ptr(steps), 0, None, None)
            self._chk(mid, "fdtd_add_monitor")
            self.mon_ids.append((m, mid, (lo2, hi2)))
        self._chk(d.fdtd_set_shutoff(h, int(spec.decay_every), float(spec.shutoff),
                                     int(spec.decay_ref_step)), "fdtd_set_shutoff")

    # ------------------------------------------------------------------ multi-GPU
    def unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self.lib.check(self.lib.dll.fdtd_comm_unique_id(buf), None, "fdtd_comm_unique_id")
        return buf.raw

    def comm_init(self, uid: bytes):
        buf = C.create_string_buffer(uid, 128)
        self._chk(self.lib.dll.fdtd_comm_init(self.handle, buf, self.rank, self.n_ranks),
                  "fdtd_comm_init")

    # ------------------------------------------------------------------ run / IO
    def run(self, n_steps: Optional[int] = None,
            progress: Optional[Callable[[int, float, float], bool]] = None) -> L.FdtdStats:
        n_steps = self.spec.n_steps if n_steps is None else n_steps
        dt = self.spec.dt

        def _cb(step, _t, decay, _user):
            try:
                return 1 if (progress and progress(int(step), step * dt, float(decay))) else 0
            except KeyboardInterrupt:
                return 1
        cb = L.PROGRESS_FN(_cb) if progress else C.cast(None, L.PROGRESS_FN)
        if self.twin is not None:
            ph = (C.c_double * 3)(*[float(v) for v in self.spec.bloch])
            nr = (C.c_int * 3)(*[int(v) for v in self.n_real])
            self._chk(self.lib.dll.fdtd_run_bloch(self.handle, self.twin.handle, int(n_steps), ph, nr, cb, None),
                      "fdtd_run_bloch")
        else:
            self._chk(self.lib.dll.fdtd_run(self.handle, int(n_steps), cb, None), "fdtd_run")
        return self.stats()

    def stats(self) -> L.FdtdStats:
        st = L.FdtdStats()
        self._chk(self.lib.dll.fdtd_get_stats(self.handle, C.byref(st)), "fdtd_get_stats")
        if int(st.struct_bytes) != C.sizeof(L.FdtdStats):      # a stale or foreign build ($TIDY3D_AMD_LIBRARY): its layout is not this binding's
            raise SolverLibraryError(f"'{self.lib.path}' fills an FdtdStats of {int(st.struct_bytes)} bytes, this binding expects "
                                     f"{C.sizeof(L.FdtdStats)}: rebuild the library (python -m tidy3d_amd.build)")
        return st

    def set_option(self, key: int, value: int):
        self._chk(self.lib.dll.fdtd_set_option(self.handle, key, value), "fdtd_set_option")

    def reset(self):
        self._chk(self.lib.dll.fdtd_reset(self.handle), "fdtd_reset")
        if self.twin is not None:
            self.twin.reset()

    def _dev_comp(self, comp: int) -> int:
        """Component id on the device (axes cyclically renamed by ``axis_shift``)."""
        return (comp % 3 - self.axis_shift) % 3 + 3 * (comp // 3)

    def get_field(self, comp: int) -> np.ndarray:
        dc = self._dev_comp(comp)
        a = self._get_field_real(dc)
        if self.twin is not None:
            a = a + 1j * self.twin._get_field_real(dc)
        return unpermute_array(a, self.axis_shift)

    def _get_field_real(self, comp: int) -> np.ndarray:
        nx, ny, _ = self.spec.shape
        out = np.empty((self.nzl, ny, self.nxp), dtype=np.float32)
        self._chk(self.lib.dll.fdtd_get_field(self.handle, comp, _ptr(out), out.nbytes),
                  "fdtd_get_field")
        if any(self.ghost):                 # Bloch device layout: strip the ghost cells
            gx, gy = self.ghost[0], self.ghost[1]
            ux, uy = self.user_shape[0], self.user_shape[1]
            return np.ascontiguousarray(out[:, gy:gy + uy, gx:gx + ux])
        return np.ascontiguousarray(out[..., :nx]) if self.pad_x else out

    def set_field(self, comp: int, arr: np.ndarray):
        if self.axis_shift:
            sig = [(a + self.axis_shift) % 3 for a in range(3)]
            arr = np.transpose(np.asarray(arr), tuple(2 - sig[2 - q] for q in range(3)))
            comp = self._dev_comp(comp)
            shift, self.axis_shift = self.axis_shift, 0          # the device-oriented array goes straight down
            try:
                return self.set_field(comp, arr)
            finally:
                self.axis_shift = shift
        if self.twin is not None:
            self.twin.set_field(comp, np.imag(arr))
            arr = np.real(arr)
        a = _f32(arr)
        if any(self.ghost):                 # ghost cells are filled by the library before the first step
            gx, gy = self.ghost[0], self.ghost[1]
            a = _f32(np.pad(a, ((0, 0), (gy, gy), (gx, gx))))
        if self.pad_x:
            a = _f32(np.pad(a, ((0, 0), (0, 0), (0, self.pad_x))))
        self._chk(self.lib.dll.fdtd_set_field(self.handle, comp, _ptr(a), a.nbytes),
                  "fdtd_set_field")

    def monitor_data(self) -> Dict[str, Tuple[np.ndarray, Tuple[int, int]]]:
        """name -> (array over the slab-local part of the box, (z_lo, z_hi) global plane range).
        time: float32 [n_rec, n_comps, bz, by, bx]; dft: complex64 [nf, n_comps, bz, by, bx]."""
        out = {}
        for m, mid, (lo2, hi2) in self.mon_ids:
            if mid < 0:
                continue
            bz, by, bx = hi2 - lo2, m.hi[1] - m.lo[1], m.hi[0] - m.lo[0]
            if m.kind == "dft":
                arr = np.empty((len(m.freqs), len(m.comps), bz, by, bx), dtype=np.complex64)
            else:
                arr = np.empty((len(m.steps), len(m.comps), bz, by, bx), dtype=np.float32)
            self._chk(self.lib.dll.fdtd_get_monitor(self.handle, mid, _ptr(arr), arr.nbytes),
                      "fdtd_get_monitor")
            out[m.name] = (arr, (lo2, hi2))
        if self.twin is not None:       # complex fields (ref simulation.py:4396 complex_fields): value = re + i im
            im = self.twin.monitor_data()
            for m, mid, _ in self.mon_ids:
                if mid >= 0:
                    out[m.name] = (out[m.name][0] + 1j * im[m.name][0], out[m.name][1])
        if self.axis_shift:                 # back to the user's axes (single slab: the z range is the box's own)
            out = {k: (unpermute_array(v[0], self.axis_shift), self.user_zrange[k]) for k, v in out.items()}
        return out

    def results(self) -> Dict[str, np.ndarray]:
        """Single-slab convenience: name -> full array (same layout as the oracle's results())."""
        return {k: v[0] for k, v in self.monitor_data().items()}

    def close(self):
        if getattr(self, "twin", None) is not None:
            self.twin.close()
            self.twin = None
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.dll.fdtd_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
