// Two time steps per sweep (in-kernel temporal blocking of the plain curl stencil), gfx950.
//
// fused2_step_kernel reads set `a` (E^n, H^{n-1/2}) and writes set `b` (E^{n+2}, H^{n+3/2}): the intermediate
// fields E^{n+1}, H^{n+1/2} live in registers and LDS only, so a pair of steps moves 6 reads + 6 writes per cell
// instead of 12 + 12.  Same update formulas (upd_h / upd_e), same operand order, same wall handling as
// fused_step_kernel -> the same bits as two single sweeps (tests/test_emu_fused2.py, tests/test_gpu_production_path.py).
//
// Workgroup = W waves = W rows of 256 cells: rows j0-2 .. j0+R, R = W - 3 of them written (j0 .. j0+R-1); the two rows
// below and the one above are recomputed halos.  The z-march is a two-stage pipeline with TWO barriers per plane:
//   iteration k:  S1  H1[k]   = H^{n+1/2}[k]      from E^n, H^{n-1/2}         (all rows)            publish H1_{x,z}
//                 ---- barrier ----
//                 S2  E1[k]   = E^{n+1}[k]        (+ the E-side point sources of step n)            publish E1_{x,z}
//                 S3  H2[k-1] = H^{n+3/2}[k-1]    from H1[k-1], E1[k-1], E1[k], E1[j+1][k-1]        publish H2_{x,z}
//                 ---- barrier ----
//                 S4  E2[k-1] = E^{n+2}[k-1]      from E1[k-1], H2[k-1], H2[k-2], H2[j-1][k-1]      store E2, H2
// A chunk [k0, k1) runs iterations k0-1 .. k1 (its first iteration is S1 + S2 only; a prologue supplies
// H1_{x,y}[k0-2]).  Along x a wave covers its 256 cells exactly as in fused_step_kernel for step one (edge lanes load /
// recompute the neighbouring column from set `a`); the second step would need the NEIGHBOUR tile's intermediate
// values on the seam: those five values per seam row are computed wrong here and repaired afterwards by
// seam_kernel from the intermediate values both tiles leave in a small scratch array.
//
// Scope (fdtd_capi.hip checks it): non-dispersive media (uniform, or packed medium words + (Ca, Cb) table), PEC walls (PMC allowed on
// the min faces: symmetry planes), absorber layers (damped in registers), point sources (<= kMaxInj nodes while they inject; the
// E-side ones of step n+1 are applied in S4, the H-side ones in S3), small time monitors (their samples of the middle step are
// copied out for pair_record_kernel) and DFT monitors, one box per launch.  Round 4: the CLIP instantiations (OPT bit 4) write one
// box only — the bulk of a grid whose shell takes single steps beside it (fdtd_capi.hip: CPML slabs + collar, the rows / planes
// next to periodic y / z faces, the boundary planes of a z-slab rank, z holes around dispersive cells and big source planes) —
// and wrap a periodic x axis inside the sweep.  Everything else takes single steps.
#pragma once
#include "fdtd_fused2.hpp"
// WHAT-IF instantiations (OPT bits 8 - 11; FDTD_OPT_WHATIF, fdtd_fused2w.hip): measuring aids that SKIP part of the work — their
// results are wrong, their times say where the sweep's time goes and bound what an optimisation could gain.  Switched inside one
// engine (same allocations, same clocks: the only A/B this part resolves), vacuum instantiation only:
//   1  E_y / H_y of a plane not loaded            2  no second barrier per plane        3  the three halo rows load nothing
//   4  every plane load reads the same (cached) row: no HBM reads, same instructions    5  no field stores
//   6  no barriers and no LDS exchange (own values instead of the neighbour rows')      7  loads + stores only (the copy floor of this tiling)
//   8  no barriers, LDS traffic kept
// PREFETCH instantiations (OPT bits 8 - 11 = 10 ... 12; CORRECT results, the same bits): part of plane k+1 travels global memory -> LDS
// by LDS-DMA (global_load_lds_dwordx4) while plane k is computed — the one way to keep a second batch of loads in flight that costs
// no registers (the sweep has 3 of 128 left).  The LDS for it comes from the E1 exchange arrays: published BEHIND the second barrier
// (read behind the next first barrier) they need no second buffer, which leaves 6 exchange arrays + 3 prefetch arrays = 144 KB.
//   10  H_x H_y H_z of the next plane       11  E_y H_y       12  E_x E_y (plane k+2) E_z (plane k+1)
//   13  (no DMA) E_x of the row above comes from the wave above through a ninth exchange array instead of a second global load of the
//       same line: published between the barriers of iteration k-1 (it is that wave's E_x[k], loaded a plane ahead), read in S1 of
//       iteration k; the top row of the workgroup and the first iteration of a chunk load it as before.  Measured - 2.1 % and made the
//       DEFAULT of every sixteen-wave instantiation (EXJ below);  14 = the sweep without it
//   15  13 + E_z through LDS too: a wave requests E_z[k+1] of its row behind S1 of iteration k (into registers S1 has just freed),
//       leaves it in a tenth... eighth array (E1 single-buffered as in 10 - 12) in front of the second barrier, and in S1 of iteration
//       k+1 takes its own E_z[k+1] and the row above's from there: one global load of E_z per row and plane instead of two, half a plane ahead

namespace fdtd {

// val[t] = w_re[t] * Re(wave[step]) - w_im[t] * Im(wave[step]): the term point_source_kernel adds, formed by the same
// operations
// What small time monitors record of a step pair (n, n+1), in ONE launch behind the sweep: E^n and H^{n-1/2} from the set the
// sweep read, the middle step from the samples it copied out, H^{n+3/2} from the set it wrote (behind the seam repair) — with
// the operations and in the order of time_record_multi_kernel's four launches (E: = 1 * E; H: += 0.5 * H per half-sample).
// One thread per (monitor, component, cell): blockIdx.y = monitor * 6 + component slot.
__global__ __launch_bounds__(256) void pair_record_kernel(PairRecP r, GridP g, FieldP a, FieldP b, const float* cap) {
  const int mi = blockIdx.y / 6, ic = blockIdx.y % 6;
  if (mi >= r.n_mon || ic >= r.nc[mi]) return;
  const BoxP bx = r.box[mi];
  const long long cells = (long long)bx.nx * bx.ny * bx.nz;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cells) return;
  const int c = r.comp[mi][ic];
  const int lx = (int)(t % bx.nx), ly = (int)((t / bx.nx) % bx.ny), lz = (int)(t / ((long long)bx.nx * bx.ny));
  const long long p = (long long)(bx.lo2 + lz) * g.sxy + (long long)(bx.lo1 + ly) * g.nx + bx.lo0 + lx;
  const long long idx = (long long)ic * cells + t;
  const float* fa = c == 0 ? a.ex : (c == 1 ? a.ey : (c == 2 ? a.ez : (c == 3 ? a.hx : (c == 4 ? a.hy : a.hz))));
  const float* fb = c == 0 ? b.ex : (c == 1 ? b.ey : (c == 2 ? b.ez : (c == 3 ? b.hx : (c == 4 ? b.hy : b.hz))));
  const float mid = cap[r.cap_off[mi] + idx];
  float* on = r.out_n[mi];
  float* om = r.out_m[mi];
  // (r.pre_done: E^n and the first H half-sample of a record at step n were taken in front of the sweep — H-side sources of
  //  step n have changed H^{n-1/2} in the read set since)
  if (c < 3) {
    if (on && !r.pre_done) on[idx] = 1.0f * fa[p];
    if (om) om[idx] = 1.0f * mid;
  } else {
    if (on) { float o = on[idx]; if (!r.pre_done) o = o + 0.5f * fa[p]; o = o + 0.5f * mid; on[idx] = o; }
    if (om) { float o = om[idx]; o = o + 0.5f * mid; o = o + 0.5f * fb[p]; om[idx] = o; }
  }
}

__global__ __launch_bounds__(256) void dft_record_dump_kernel(DftDumpP r, const float* dump, float2* acc, long long cells,
                                                              long long fstride, const float2* phase, int nf) {
  const int q = blockIdx.y;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cells || q >= r.n) return;
  const float v = dump[r.off[q] + t];
  float2* a0 = acc + (long long)r.slot[q] * cells;
  for (int k = 0; k < nf; ++k) {
    const float2 ph = phase[k];
    float2 a = a0[(long long)k * fstride + t];
    a.x += v * ph.x;
    a.y += v * ph.y;
    a0[(long long)k * fstride + t] = a;
  }
}

// the same for every step at once: tab[step * stride + off + t]
__global__ __launch_bounds__(256) void inject_table_kernel(float* tab, long long stride, long long off, const float* w_re,
                                                           const float* w_im, const float2* wave, long long n_steps, int n) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_steps * n) return;
  const long long step = g / n;
  const int t = (int)(g % n);
  const float2 a = wave[step];
  tab[step * stride + off + t] = w_re[t] * a.x - w_im[t] * a.y;
}

__global__ __launch_bounds__(256) void inject_values_kernel(float* val, const float* w_re, const float* w_im,
                                                            const float2* wave, long long step, int n) {
  const int t = threadIdx.x;
  if (t >= n) return;
  const float2 a = wave[step];
  val[t] = w_re[t] * a.x - w_im[t] * a.y;
}

__device__ __forceinline__ long long seam_at(const GridP& g, int seam, int arr, int j, int k) {
  return (((long long)seam * kSeamArrays + arr) * (g.nz + 2) + (k + 1)) * g.ny + j;
}

// OPT: bit 0 = non-temporal stores, bit 1 = materials (packed medium words + (Ca, Cb) table, as fused_step_kernel<MAT>),
// bit 3 = absorber layers (the damping of damp_kernel / damp4_kernel applied in registers),
// bit 2 = the node table holds monitor samples (8-wave workgroups run two per CU: LB = 512 asks for 4 waves per SIMD, i.e. <= 128 VGPRs)
// x neighbours across the wave: lane i takes the value of lane i+1 / i-1 (the last / first lane keeps its own, as __shfl_down /
// __shfl_up do).  One DPP move (wave_shl:1 / wave_shr:1, GFX9) instead of a ds_bpermute through the LDS crossbar with its
// address arithmetic and its lgkmcnt wait on the critical path of every stage.
__device__ __forceinline__ float lane_next(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int x = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x130, 0xf, 0xf, false));
#else
  return __shfl_down(v, 1);
#endif
}
__device__ __forceinline__ float lane_prev(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int x = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false));
#else
  return __shfl_up(v, 1);
#endif
}

// LDS-DMA: the wave's 64 lanes fetch 16 bytes each from base + off (off: the lane's byte offset) straight into LDS at
// dst[lane] — no registers, and, issued through inline asm, absent from hipcc's s_waitcnt bookkeeping: nothing drains it at the
// workgroup's barriers (a __builtin_amdgcn_global_load_lds in flight makes every __syncthreads wait for vmcnt(0)).  `dst` is the
// wave's own 64-entry slice (wave-uniform); lds_dma_wait() in front of reading it.  On the CPU emulator: a plain copy.
__device__ __forceinline__ void lds_dma16(const float* base, unsigned off, float4* dst, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(off), "s"(lds_addr), "s"(base) : "memory");
#else
  dst[lane] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
#endif
}
__device__ __forceinline__ void lds_dma_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// in front of re-using a slice: the wave's ds_reads of it have RETURNED (issued is not enough — with sixteen waves reading 3 KB each
// the LDS queue is ~400 cycles deep, and a DMA that hits in L2 lands before that: seen on the device as wrong values spreading from a
// seam column, where the neighbouring tile's edge loads had warmed the line; profiles/r6/r6u)
__device__ __forceinline__ void lds_dma_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

// A per-lane value the optimiser cannot see through.  Lane-constant predicates (first / last lane of a tile, cells beyond the row
// end) are re-derived from it inside the march: hoisted out of the loop each of them is a 64-bit lane mask that lives in an
// SGPR pair for the whole kernel — a dozen of them were a third of the SGPRs this kernel spills to VGPR lanes and reads back
// (v_readlane) inside every plane.
__device__ __forceinline__ int opaque_lane(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// lane l's value of v, l wave-uniform
__device__ __forceinline__ int lane_value(int v, int l) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readlane(v, l);
#else
  return __shfl(v, l);
#endif
}

// the sweep of tile t (logical index, y fastest) by its workgroup; fused2_step_kernel below picks t and the instantiation
template <int LB, int OPT>
__device__ __forceinline__ void fused2_step_tile(const GridP& g, const FieldP& a, const FieldP& b, const StepP& s, const MatP& m,
                                                 int zchunk, int nbx, int nby, int nbz, const InjP& inj,
                                                 float* __restrict__ seam, const DampT& dmp, const ClipP& clip, int t,
                                                 const DispP& dp, const SrcP& sr) {
  constexpr int V = 4;
  constexpr bool SRC = (OPT & 64) != 0;    // paged source terms (fdtd_fused2.hpp SrcP): TFSF boxes, mode planes, sheets while they inject
  constexpr bool NT = (OPT & 1) != 0, MAT = (OPT & 2) != 0, MON = (OPT & 4) != 0;     // MON: the node table may hold monitor samples
  // DISP (round 6): the grid holds dispersive cells and the pair advances them (K4 twice).  What the sweep needs of step n's ADE
  // update is E^{n+1} <- E^{n+1} - cc S(Q^n) at those cells — H^{n+3/2} differentiates it — and that memory term is known before
  // the sweep starts: it lies in paged storage (DispP::cs, one block per row segment that holds a dispersive cell, zero at the
  // segment's other cells), kept up to date by the ADE kernels.  S2 subtracts it last of all (as launch_ade follows the damping
  // launch) and leaves E^{n+1} of the rows it owns in DispP::e1; ade2_kernel behind the sweep forms Q^{n+1} from it, corrects
  // E^{n+2} and forms Q^{n+2} — ade_kernel's operations in its order, twice.
  constexpr bool DISP = (OPT & 32) != 0;
  constexpr int WHATIF = (OPT >> 8) & 15;
  constexpr int PF = (WHATIF >= 10 && WHATIF <= 12) ? WHATIF : 0;            // prefetch instantiation (correct results)
  // EXJ: E_x of the row above comes from the wave above through a ninth exchange array (fused2_xch_arrays) instead of a second global
  // load of a line the workgroup has already fetched — 512^3 vacuum inside one engine 0.690 -> 0.675 ms per step (profiles/r6/r6u): the
  // sweep is bound by requests between the CUs and L2 (11.6 B per cycle and CU), not by HBM alone; eight-wave workgroups: 256^3 - 1.5 %,
  // 320^3 - 3.1 % (r6u_exj_8waves).  Not in the eight-wave materials instantiations (two workgroups per CU: 2 x (72 + 8) KB of LDS would
  // leave one) nor with absorber layers (below); WHATIF 14 = the sweep without it, for A/B inside one engine.  Whole bench lines of two builds, three rounds
  // interleaved (profiles/r6/r6v): V1 + 3 %, V2 / V3 / V4 + 2 %, the absorber instantiation - 5 % (two more spilled registers) — left out there.
  constexpr bool EZL = WHATIF == 15;                                         // E_z through LDS (variant 15)
  constexpr bool EXJ = (WHATIF == 0 && (OPT & 4096) == 0) || WHATIF == 13 || EZL;      // (bit 12: the KERNEL's instantiation has no room for it — set by fused2_step_kernel, which sized the LDS)
  constexpr bool E1S = PF != 0 || EZL;                                       // E1 exchanged through ONE buffer (published behind the second barrier)
  constexpr int XE = E1S ? 6 : 8, XZ = 7;                                    // exchange arrays of E_x[k+1] (EXJ) and E_z[k+1] (EZL)
  [[maybe_unused]] constexpr int NPF = PF == 11 ? 2 : (PF ? 3 : 0);                           // arrays of the next plane that travel through LDS
  constexpr int NXCH = E1S ? 6 : 8;                                           // exchange arrays (PF: E1 single-buffered)
  static_assert((!PF && !EZL) || (OPT & 0xfe) == 0, "prefetch instantiations: the vacuum sweep only");
  constexpr bool DAMP = (OPT & 8) != 0;   // absorber layers: both fields of both steps are damped in registers (damp_kernel's factors)
  // CLIP: the launch covers the box `clip` only — the bulk of a grid whose shell (CPML slabs + a two-cell collar, the boundary
  // planes of a z-slab rank) is advanced by single steps beside it.  Tile rows and chunks start at the box's origin, nothing is
  // stored outside it, and the seam scratch is written for every row / plane this workgroup computes (the seam kernel
  // differentiates the row and plane below the box's first ones, which no workgroup owns).
  constexpr bool CLIP = (OPT & 16) != 0;
  // periodic x (clipped launches only; periodic y / z faces are part of the shell: the box stays two cells clear of them): the
  // first lane of a row takes column nx - 1 as its x-halo column, the last one column 0 as its right neighbour — step one is
  // exact — and the wrap is one more seam for step two (the row's last column | its first), repaired by seam_kernel
  const bool per_x = CLIP && g.bcx0 == BC_PERIODIC;
  // (Issuing the loads of plane k+1 behind the second barrier of plane k — the one way to overlap them with compute inside a wave —
  //  was measured: + 60 registers, slower at every workgroup size, profiles/r3q; taken out.)
  const int tile_y = t % nby;
  const int tile_x = (t / nby) % nbx;
  const int tile_z = t / (nby * nbx);
  HIP_DYNAMIC_SHARED(float4, xch)      // [8][W][64]: H1_x H1_z | H2_x H2_z | E1_x E1_z (buffer 0) | E1_x E1_z (buffer 1)
  const int tx = threadIdx.x;
  const int ty = __builtin_amdgcn_readfirstlane((int)threadIdx.y);     // one row per wave
  const int W = blockDim.y, R = W - 3;
  const int slot = W * 64;
  const int me = ty * 64 + tx;
  {
    const float4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NXCH; ++q) xch[q * slot + me] = z4;     // rows beyond the grid publish E = 0, H = 0
  }
  // absorber layers: the x factors of the tile's 256 cells (fb: cell boundaries, fc: cell centres; 1 outside the layers)
  [[maybe_unused]] float4* xdm = xch + (EXJ ? 9 : 8) * slot;             // [2][64]
  // the (Ca, Cb) table: behind them in the DYNAMIC allocation (fused2_lds_bytes).  As a static array of this function it was allocated
  // once per tile body of a kernel — three or four in the launches that dispatch by tile class, 24 - 32 KB beside 144 KB of exchange
  // arrays: HSA_STATUS_ERROR_INVALID_ALLOCATION on the device (round 6, the random cases of tests/test_fuzz_round6.py)
  [[maybe_unused]] float2* lut_s = reinterpret_cast<float2*>(xdm + (DAMP ? 2 * 64 : 0));
  if constexpr (MAT) {
    for (int q = me; q < m.n_media; q += slot) lut_s[q] = m.lut[q];
  }
  if constexpr (DAMP) {
    if (ty == 0) {
      const int ii = (tile_x * 64 + tx) * V;
      float4 b4 = {1.f, 1.f, 1.f, 1.f}, c4 = {1.f, 1.f, 1.f, 1.f};
      if (ii < g.nx) { b4 = *reinterpret_cast<const float4*>(dmp.fb[0] + ii); c4 = *reinterpret_cast<const float4*>(dmp.fc[0] + ii); }
      xdm[tx] = b4; xdm[64 + tx] = c4;
    }
  }
  __syncthreads();
  const float ca = m.ca1, cb = m.cb1;
  const int k0 = (CLIP ? clip.k0 : 0) + tile_z * zchunk;
  const int k1 = min(k0 + zchunk, CLIP ? clip.k1 : g.nz);
  const int kA = k0 > 0 ? k0 - 1 : 0;
  int j = (CLIP ? clip.j0 : 0) + tile_y * R + ty - 2;
  const bool row_ok = (j >= 0) && (j < g.ny);
  if (!row_ok) {                          // takes part in the barriers only
    if constexpr (WHATIF != 6 && WHATIF != 7 && WHATIF != 8)
      for (int k = kA; k <= k1; ++k) { __syncthreads(); if constexpr (WHATIF != 2) __syncthreads(); }
    return;
  }
  const int i0 = (tile_x * 64 + tx) * V;
  const unsigned ux = (unsigned)i0;
  const unsigned ub = ux * 4u;
  const unsigned ubc = (i0 < g.nx) ? ub : 0u;
  const bool act = (i0 < g.nx);
  const bool do_e1 = ty >= 1;                       // rows j0-1 .. j0+R
  const bool do_h2 = ty >= 1 && ty <= W - 2;        // rows j0-1 .. j0+R-1
  const bool own = ty >= 2 && ty <= W - 2 && (!CLIP || j < clip.j1);          // rows j0 .. j0+R-1: stored
  const float ch = g.ch;
  const bool last_x = (i0 + V >= g.nx);
  const bool first_x = (i0 == 0);
  const bool use_jp = (j + 1 < g.ny);
  const long long rowb = (long long)j * g.nx;
  const long long rowpb = use_jp ? rowb + g.nx : 0;
  const bool xh = act && (tx == 0) && (!first_x || per_x);     // the tile's first lane recomputes H1_{y,z} of column i0-1
  const int im = first_x ? (per_x ? g.nx - 1 : 0) : i0 - 1;
  // min faces: a PEC wall (tangential E = 0 on it) or a PMC one (H mirrored with the opposite sign behind it), as in
  // fused_step_kernel; max faces are PEC walls
  const bool pmc_x0 = g.bcx0 == BC_PMC, pmc_y0 = g.bcy0 == BC_PMC, pmc_z0 = !g.pec_z0;
  const bool wall_y = (j == 0) && !pmc_y0;

  // damping factors: w(H_x) = (bx cy) cz, w(H_y) = (cx by) cz, w(H_z) = (cx cy) bz;  w(E_x) = (cx by) bz, w(E_y) = (bx cy) bz,
  // w(E_z) = (bx by) cz — the products of damp_kernel, formed in its order; a field outside every layer is multiplied by 1
  [[maybe_unused]] float byv = 1.f, cyv = 1.f, bxm = 1.f, cxm = 1.f, bz_m = 1.f, cz_m = 1.f;
  if constexpr (DAMP) { byv = dmp.fb[1][j]; cyv = dmp.fc[1][j]; }
  float ipx[V], idx[V];
  zero<V>(ipx); zero<V>(idx);
  float ipx_m = 0.f;
  if (act) {
    ldv<V>(ipx, at(uni(s.ipx), ub));
    ldv<V>(idx, at(uni(s.idx), ub));
    ipx_m = s.ipx[im];
    if constexpr (DAMP) { bxm = dmp.fb[0][im]; cxm = dmp.fc[0][im]; }
  }
  const float ipy = s.ipy[j], idy = s.idy[j];

  // carried along the march
  float exk[V], eyk[V];                    // E^n_{x,y}[k]
  float h1x[V], h1y[V], h1z[V];            // H1[k-1]
  float e1x[V], e1y[V], e1z[V];            // E1[k-1]
  float h2xm[V], h2ym[V];                  // H2_{x,y}[k-2]
  zero<V>(h1x); zero<V>(h1y); zero<V>(h1z); zero<V>(e1x); zero<V>(e1y); zero<V>(e1z); zero<V>(h2xm); zero<V>(h2ym);
  float exk_m = 0.f;
  float ipz_m = 0.f, idz_m = 0.f;          // 1 / steps of plane k-1
  int qm0 = 0, qm1 = 0;                    // table rows of plane k-1
  [[maybe_unused]] uint32_t rw_m = kBgWord;                                        // material row-segment word of plane k-1 (S4)
  [[maybe_unused]] int ss_m = -1;                                                  // block of the segment's source terms, plane k-1 (S3, S4)
  {
    const long long p0 = (long long)kA * g.sxy + rowb;
    ldf<V, true>(exk, uni(a.ex + p0), ubc);
    ldf<V, true>(eyk, uni(a.ey + p0), ubc);
    if (xh) exk_m = a.ex[p0 + im];
  }
  // ---- prologue: H1_{x,y}[kA-1] (a chunk that starts on the z-min wall needs none: E1_{x,y}[0] = 0 there) ----------
  if (kA > 0 && do_e1) {
    const long long pb = (long long)(kA - 1) * g.sxy + rowb;
    float ezm[V], ezj[V], exm[V], eym[V], ho[V], hoy[V];
    zero<V>(ezj);
    ldf<V, true>(ezm, uni(a.ez + pb), ubc);
    ldf<V, true>(exm, uni(a.ex + pb), ubc);
    ldf<V, true>(eym, uni(a.ey + pb), ubc);
    if (use_jp) ldf<V, true>(ezj, uni(a.ez + (long long)(kA - 1) * g.sxy + rowpb), ubc);
    float ezx = lane_next(ezm[0]);
    if (act && (tx == 63 || last_x)) ezx = last_x ? (per_x ? a.ez[pb] : 0.f) : a.ez[pb + ux + V];
    const float ipz = s.ipz[kA - 1];
    ldf<V, true>(ho, uni(a.hx + pb), ubc);
    ldf<V, true>(hoy, uni(a.hy + pb), ubc);
    if constexpr (DAMP) {
      const float czp = dmp.fc[2][kA - 1];
      const float4 b4 = xdm[tx], c4 = xdm[64 + tx];
      const float bx[V] = {b4.x, b4.y, b4.z, b4.w}, cx[V] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
      for (int e = 0; e < V; ++e) { ho[e] *= bx[e] * cyv * czp; hoy[e] *= cx[e] * byv * czp; }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) h1x[e] = upd_h(ho[e], ch, ezj[e] - ezm[e], ipy, eyk[e] - eym[e], ipz);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float ez_ip = (e + 1 < V) ? ezm[(e + 1) % V] : ezx;
      h1y[e] = upd_h(hoy[e], ch, exk[e] - exm[e], ipz, ez_ip - ezm[e], ipx[e]);
    }
  }
  int cur = 0;
  const long long seam_arr = (long long)(g.nz + 2) * g.ny;               // one scratch array
  const long long seam_row = seam_at(g, tile_x, 0, j, 0);                // this row in array 0 of the seam to the right, plane 0
  // what one plane loads from set `a` (edge lanes: the neighbouring columns)
  struct Ld {
    float exn[V], eyn[V], ezk[V], exj[V], ezj[V], hxn[V], hyn[V], hzn[V];
    float eyx_g, ezx_g;                                  // E^n_{y,z} of column i0 + 256 (last lane of a tile with a right neighbour)
    float exn_m, ez_mm, ey_mm, ex_jm, hy_o, hz_o;        // column i0 - 1 (first lane of a tile with a left neighbour)
  };
  // PF: the arrays of plane k that travel through LDS, requested one plane ahead (behind S1 of plane k-1; the first one in front of the march)
  [[maybe_unused]] float4* pfb = xch + NXCH * slot;           // [NPF][W][64]
  [[maybe_unused]] auto pf_issue = [&](int k) __attribute__((always_inline)) {
    if constexpr (PF != 0) {
      const long long pb = (long long)k * g.sxy + rowb;
      const long long up = (k < g.nz) ? g.sxy : 0;
      lds_dma_fence();
      if constexpr (PF == 10) {
        lds_dma16(uni(a.hx + pb), ubc, pfb + 0 * slot + ty * 64, tx);
        lds_dma16(uni(a.hy + pb), ubc, pfb + 1 * slot + ty * 64, tx);
        lds_dma16(uni(a.hz + pb), ubc, pfb + 2 * slot + ty * 64, tx);
      } else if constexpr (PF == 11) {
        lds_dma16(uni(a.ey + pb + up), ubc, pfb + 0 * slot + ty * 64, tx);
        lds_dma16(uni(a.hy + pb), ubc, pfb + 1 * slot + ty * 64, tx);
      } else {
        lds_dma16(uni(a.ex + pb + up), ubc, pfb + 0 * slot + ty * 64, tx);
        lds_dma16(uni(a.ey + pb + up), ubc, pfb + 1 * slot + ty * 64, tx);
        lds_dma16(uni(a.ez + pb), ubc, pfb + 2 * slot + ty * 64, tx);
      }
    }
  };
  [[maybe_unused]] auto pf_take = [&](int q, float (&o)[V]) __attribute__((always_inline)) {
    const float4 t4 = pfb[q * slot + me];
    o[0] = t4.x; o[1] = t4.y; o[2] = t4.z; o[3] = t4.w;
  };
  auto issue = [&](int k, Ld& L) __attribute__((always_inline)) {
    if constexpr (WHATIF == 3) {      // (the three halo rows of a workgroup cost no loads — the upper bound of sharing them with the neighbouring workgroups)
      if (ty < 2 || ty == W - 1) {
        zero<V>(L.exn); zero<V>(L.eyn); zero<V>(L.ezk); zero<V>(L.exj); zero<V>(L.ezj); zero<V>(L.hxn); zero<V>(L.hyn); zero<V>(L.hzn);
        L.eyx_g = L.ezx_g = L.exn_m = L.ez_mm = L.ey_mm = L.ex_jm = L.hy_o = L.hz_o = 0.f;
        return;
      }
    }
    const int txo = (MAT || DAMP) ? tx : opaque_lane(tx);       // (the materials / absorber instantiations have no VGPR to spare for the re-derivation)
    const int i0o = (tile_x * 64 + txo) * V;
    const bool act = i0o < g.nx, last_x = i0o + V >= g.nx, first_x = i0o == 0;
    const bool xh = act && txo == 0 && (!first_x || per_x);
    [[maybe_unused]] const bool wall_x0 = first_x && !pmc_x0 && !per_x;
    // (WHATIF 4: every plane reads plane (k & 1), row 0 — the same instructions, all of them cache hits)
    const long long pb = WHATIF == 4 ? (long long)(k & 1) * g.sxy : (long long)k * g.sxy + rowb;
    const long long pjb = WHATIF == 4 ? (long long)(k & 1) * g.sxy + g.nx : (long long)k * g.sxy + rowpb;
    const long long up = WHATIF == 4 ? 0 : ((k < g.nz) ? g.sxy : 0);         // (iteration nz only needs E1_{x,y}[nz] = 0: it reads the ghost plane twice)
    if constexpr (PF != 12) ldf<V, true>(L.exn, uni(a.ex + pb + up), ubc);
    if constexpr (WHATIF == 1) zero<V>(L.eyn);      // (E_y / H_y not loaded at all — the upper bound of prefetching them)
    else if constexpr (PF != 11 && PF != 12) ldf<V, true>(L.eyn, uni(a.ey + pb + up), ubc);
    if constexpr (EZL) {
      if (k == kA) ldf<V, true>(L.ezk, uni(a.ez + pb), ubc);
      else { const float4 t4 = xch[XZ * slot + me]; L.ezk[0] = t4.x; L.ezk[1] = t4.y; L.ezk[2] = t4.z; L.ezk[3] = t4.w; }
    } else if constexpr (PF != 12) ldf<V, true>(L.ezk, uni(a.ez + pb), ubc);
    if (use_jp) {
      if constexpr (EXJ) {
        if (ty == W - 1 || k == kA) ldf<V, true>(L.exj, uni(a.ex + pjb), ubc);
        else { const float4 t4 = xch[XE * slot + me + 64]; L.exj[0] = t4.x; L.exj[1] = t4.y; L.exj[2] = t4.z; L.exj[3] = t4.w; }
      } else {
        ldf<V, true>(L.exj, uni(a.ex + pjb), ubc);
      }
      if constexpr (EZL) {
        if (ty == W - 1 || k == kA) ldf<V, true>(L.ezj, uni(a.ez + pjb), ubc);
        else { const float4 t4 = xch[XZ * slot + me + 64]; L.ezj[0] = t4.x; L.ezj[1] = t4.y; L.ezj[2] = t4.z; L.ezj[3] = t4.w; }
      } else {
        ldf<V, true>(L.ezj, uni(a.ez + pjb), ubc);
      }
    } else {
      zero<V>(L.exj); zero<V>(L.ezj);
    }
    if constexpr (PF != 10) ldf<V, true>(L.hxn, uni(a.hx + pb), ubc);
    if constexpr (WHATIF == 1) zero<V>(L.hyn);
    else if constexpr (PF != 10 && PF != 11) ldf<V, true>(L.hyn, uni(a.hy + pb), ubc);
    if constexpr (PF != 10) ldf<V, true>(L.hzn, uni(a.hz + pb), ubc);
    L.eyx_g = 0.f; L.ezx_g = 0.f;
    if (act && txo == 63 && !last_x) { L.eyx_g = a.ey[pb + ux + V]; L.ezx_g = a.ez[pb + ux + V]; }
    if (per_x && act && last_x) { L.eyx_g = a.ey[pb]; L.ezx_g = a.ez[pb]; }        // (the row's first column)
    L.exn_m = 0.f; L.ez_mm = 0.f; L.ey_mm = 0.f; L.ex_jm = 0.f; L.hy_o = 0.f; L.hz_o = 0.f;
    if (xh && do_e1) {
      const long long pm = pb + im;
      L.exn_m = a.ex[pm + up];
      L.ez_mm = a.ez[pm]; L.ey_mm = a.ey[pm];
      L.ex_jm = use_jp ? a.ex[pjb + im] : 0.f;
      L.hy_o = a.hy[pm]; L.hz_o = a.hz[pm];
    }
    if constexpr (PF != 0) {          // what came through LDS (requested a plane ago; every load of this plane is out by now)
      lds_dma_wait();
      if constexpr (PF == 10) { pf_take(0, L.hxn); pf_take(1, L.hyn); pf_take(2, L.hzn); }
      else if constexpr (PF == 11) { pf_take(0, L.eyn); pf_take(1, L.hyn); }
      else { pf_take(0, L.exn); pf_take(1, L.eyn); pf_take(2, L.ezk); }
    }
  };
  Ld LA, LB2;
  auto body = [&](int k, Ld& L) __attribute__((always_inline)) {
    const int txo = (MAT || DAMP) ? tx : opaque_lane(tx);       // (the materials / absorber instantiations have no VGPR to spare for the re-derivation)
    const int i0o = (tile_x * 64 + txo) * V;
    const bool act = i0o < g.nx, last_x = i0o + V >= g.nx, first_x = i0o == 0;
    const bool xh = act && txo == 0 && (!first_x || per_x);
    [[maybe_unused]] const bool wall_x0 = first_x && !pmc_x0 && !per_x;
    // (the left / right side of a seam: a tile edge with a neighbour, or — periodic x — the row's ends)
    const bool seam_l = (txo == 63 && !last_x) || (per_x && last_x), seam_r = txo == 0 && (tile_x > 0 || per_x);
    const long long seam_back = (tile_x > 0 ? -1 : (long long)(nbx - 1)) * kSeamArrays * seam_arr;     // from this tile's seam to the one on its left
    // (iteration k = nz, last chunk only: plane nz is the z-max wall, E1_{x,y}[nz] = 0 is all it contributes; its loads
    //  read the ghost plane, its other results are never used)
    const long long pb = (long long)k * g.sxy + rowb;
    const float ipz = s.ipz[k], idz = s.idz[k];          // (the step arrays carry one ghost entry at each end)
    [[maybe_unused]] float bzk = 1.f, czk = 1.f;
    if constexpr (DAMP) { bzk = dmp.fb[2][min(k, g.nz - 1)]; czk = dmp.fc[2][min(k, g.nz - 1)]; }
    int q0 = 0, q1 = 0;
    if (inj.n > 0) { q0 = inj.start[k]; q1 = inj.start[k + 1]; }     // ([nz + 2] entries: plane nz holds none)
    [[maybe_unused]] int d0 = 0, d1 = 0;
    if constexpr (MON) { if (inj.dstart) { d0 = inj.dstart[k]; d1 = inj.dstart[k + 1]; } }
    float hy_m = 0.f, hz_m = 0.f;
    float (&exn)[V] = L.exn, (&eyn)[V] = L.eyn, (&ezk)[V] = L.ezk, (&exj)[V] = L.exj, (&ezj)[V] = L.ezj;
    float (&hxn)[V] = L.hxn, (&hyn)[V] = L.hyn, (&hzn)[V] = L.hzn;
    // material row-segment word of this plane (scalar load); where the segment is mixed (6 % of them around a sphere) the
    // packed words are fetched where they are used: held from here they cost four registers the instantiation does not have
    [[maybe_unused]] uint32_t rw = kBgWord;
    if constexpr (MAT) {
      if (do_e1) rw = m.roww[((long long)min(k, g.nz - 1) * g.ny + j) * nbx + tile_x];     // (plane nz: the wall, E1 = 0 whatever the medium)
    }
    [[maybe_unused]] int ss = -1;              // block of the row segment's source terms (SRC), -1 = no source node in it
    if constexpr (SRC) {
      if (do_e1 && k < g.nz) ss = sr.sseg[((long long)k * g.ny + j) * nbx + tile_x];
    }
    [[maybe_unused]] int ds = -1;              // block of the row segment's memory terms (DISP), -1 = no dispersive cell in it
    if constexpr (DISP) {
      if (do_e1 && k < g.nz) ds = dp.dseg[((long long)k * g.ny + j) * nbx + tile_x];
    }
    if constexpr (WHATIF == 7) {          // loads + stores only: every array of the plane read as the sweep reads it, six written
      issue(k, L);
      if (own && k > k0 && act) {
        const long long po = pb - g.sxy + i0;
        float o0[V], o1[V], o2[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { o0[e] = L.exn[e] + L.exj[e]; o1[e] = L.ezk[e] + L.ezj[e]; o2[e] = L.eyn[e]; }
        stv_h<V, NT>(b.ex + po, o0); stv_h<V, NT>(b.ey + po, o2); stv_h<V, NT>(b.ez + po, o1);
        stv_h<V, NT>(b.hx + po, L.hxn); stv_h<V, NT>(b.hy + po, L.hyn); stv_h<V, NT>(b.hz + po, L.hzn);
      }
      return;
    }
    {
      issue(k, L);
      float eyx = lane_next(eyk[0]);
      float ezx = lane_next(ezk[0]);
      if (act && (txo == 63 || last_x)) {
        if (!last_x || per_x) { eyx = L.eyx_g; ezx = L.ezx_g; }
        else { eyx = 0.f; ezx = 0.f; }
      }
      if constexpr (DAMP) {                 // H^{n-1/2} is damped before step n advances it
        {
          const float4 b4 = xdm[tx];
          const float bx[V] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int e = 0; e < V; ++e) hxn[e] *= bx[e] * cyv * czk;
        }
        {
          const float4 c4 = xdm[64 + tx];
          const float cx[V] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
          for (int e = 0; e < V; ++e) { hyn[e] *= cx[e] * byv * czk; hzn[e] *= cx[e] * cyv * bzk; }
        }
        L.hy_o *= cxm * byv * czk;
        L.hz_o *= cxm * cyv * bzk;
      }
      // ---- S1: H1[k] ----
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float ey_ip = (e + 1 < V) ? eyk[(e + 1) % V] : eyx;
        const float ez_ip = (e + 1 < V) ? ezk[(e + 1) % V] : ezx;
        hxn[e] = upd_h(hxn[e], ch, ezj[e] - ezk[e], ipy, eyn[e] - eyk[e], ipz);
        hyn[e] = upd_h(hyn[e], ch, exn[e] - exk[e], ipz, ez_ip - ezk[e], ipx[e]);
        hzn[e] = upd_h(hzn[e], ch, ey_ip - eyk[e], ipx[e], exj[e] - exk[e], ipy);
      }
      // x-halo column: H1_{y,z} at i0-1 recomputed by the tile's first lane
      if (xh && do_e1) {
        hy_m = upd_h(L.hy_o, ch, L.exn_m - exk_m, ipz, ezk[0] - L.ez_mm, ipx_m);
        hz_m = upd_h(L.hz_o, ch, eyk[0] - L.ey_mm, ipx_m, L.ex_jm - exk_m, ipy);
      }
    }
    if constexpr (PF != 0) { if (k + 1 <= k1) pf_issue(k + 1); }      // (S1 has consumed this plane's slices)
    [[maybe_unused]] float ezn_pre[V];
    if constexpr (EZL) {                                                // E_z[k+1] of this row, for the next iteration (its own and the row below's)
      if (k + 1 <= k1) ldf<V, true>(ezn_pre, uni(a.ez + pb + g.sxy), ubc);
    }
    if constexpr (WHATIF != 6) {
      float4 t4;
      t4.x = hxn[0]; t4.y = hxn[1]; t4.z = hxn[2]; t4.w = hxn[3];
      xch[0 * slot + me] = t4;
      t4.x = hzn[0]; t4.y = hzn[1]; t4.z = hzn[2]; t4.w = hzn[3];
      xch[1 * slot + me] = t4;
    }
    if constexpr (WHATIF != 6 && WHATIF != 8) __syncthreads();
    // ---- S2: E1[k] ----
    float e1xn[V], e1yn[V], e1zn[V];
    unspecified<V>(e1xn); unspecified<V>(e1yn); unspecified<V>(e1zn);     // (row j0-2: nobody reads its E1)
    if (do_e1) {
      float hyx = lane_prev(hyn[V - 1]);
      float hzx = lane_prev(hzn[V - 1]);
      {
        if (txo == 0 || first_x) {
          if (xh) { hyx = hy_m; hzx = hz_m; }
          else if (pmc_x0) { hyx = -hyn[0]; hzx = -hzn[0]; }
          else { hyx = 0.f; hzx = 0.f; }
        }
        float hxj[V], hzj[V];
        if (WHATIF == 6) {                    // (no LDS exchange: the row's own values stand in for the row below)
#pragma unroll
          for (int e = 0; e < V; ++e) { hxj[e] = hxn[e]; hzj[e] = hzn[e]; }
        } else if (j > 0) {
          const float4 t0 = xch[0 * slot + me - 64];
          const float4 t1 = xch[1 * slot + me - 64];
          hxj[0] = t0.x; hxj[1] = t0.y; hxj[2] = t0.z; hxj[3] = t0.w;
          hzj[0] = t1.x; hzj[1] = t1.y; hzj[2] = t1.z; hzj[3] = t1.w;
        } else if (pmc_y0) {
#pragma unroll
          for (int e = 0; e < V; ++e) { hxj[e] = -hxn[e]; hzj[e] = -hzn[e]; }
        } else {
          zero<V>(hxj); zero<V>(hzj);
        }
        if (pmc_z0 && k == 0) {
#pragma unroll
          for (int e = 0; e < V; ++e) { h1x[e] = -hxn[e]; h1y[e] = -hyn[e]; }
        }
        const bool wall_z = (k == 0 && !pmc_z0) || (k == g.nz);
        // `coef(c, e)` yields (Ca, Cb) of component c of the lane's e-th cell (fused_step_kernel's e_phase)
        auto s2 = [&](auto coef) __attribute__((always_inline)) {
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const float hy_im = (e > 0) ? hyn[(e + V - 1) % V] : hyx;
            const float hz_im = (e > 0) ? hzn[(e + V - 1) % V] : hzx;
            float nex = upd_e(exk[e], coef(0, e).x, coef(0, e).y, hzn[e] - hzj[e], idy, hyn[e] - h1y[e], idz);
            float ney = upd_e(eyk[e], coef(1, e).x, coef(1, e).y, hxn[e] - h1x[e], idz, hzn[e] - hz_im, idx[e]);
            float nez = upd_e(ezk[e], coef(2, e).x, coef(2, e).y, hyn[e] - hy_im, idx[e], hxn[e] - hxj[e], idy);
            const bool wx = wall_x0 && (e == 0);
            if (wall_y || wall_z) nex = 0.f;
            if (wx || wall_z) ney = 0.f;
            if (wx || wall_y) nez = 0.f;
            e1xn[e] = nex; e1yn[e] = ney; e1zn[e] = nez;
          }
        };
        if constexpr (MAT) {
          if (rw != kMixedWord) {
            const float2 c0 = lut_s[rw & 1023u], c1 = lut_s[(rw >> 10) & 1023u], c2 = lut_s[(rw >> 20) & 1023u];
            s2([&](int c, int) { return c == 0 ? c0 : (c == 1 ? c1 : c2); });
          } else {
            uint32_t mw[V] = {kBgWord, kBgWord, kBgWord, kBgWord};
            if (act) ldm<V>(mw, at(uni(m.m4 + pb), ub));
            s2([&](int c, int e) { return lut_s[(mw[e] >> (10 * c)) & 1023u]; });
          }
        } else {
          const float2 c1 = make_float2(ca, cb);
          s2([&](int, int) { return c1; });
        }
        if constexpr (SRC) {                  // the E-side source terms of step n (behind the walls: a list kernel adds to what the sweep wrote)
          if (ss >= 0 && act) {
            const long long qs = ((long long)ss * 3) * 256 + tx * V;
            float tx_[V], ty_[V], tz_[V];
            ldv<V>(tx_, sr.t->e1 + qs); ldv<V>(ty_, sr.t->e1 + qs + 256); ldv<V>(tz_, sr.t->e1 + qs + 512);
#pragma unroll
            for (int e = 0; e < V; ++e) { e1xn[e] = e1xn[e] + tx_[e]; e1yn[e] = e1yn[e] + ty_[e]; e1zn[e] = e1zn[e] + tz_[e]; }
            if (sr.t->e1b) {
              ldv<V>(tx_, sr.t->e1b + qs); ldv<V>(ty_, sr.t->e1b + qs + 256); ldv<V>(tz_, sr.t->e1b + qs + 512);
#pragma unroll
              for (int e = 0; e < V; ++e) { e1xn[e] = e1xn[e] + tx_[e]; e1yn[e] = e1yn[e] + ty_[e]; e1zn[e] = e1zn[e] + tz_[e]; }
            }
          }
        }
        // The node table of the plane.  Codes 0 - 2: the E-side point sources of step n act on E^{n+1} before step n+1 reads
        // it.  Codes 8 - 13 (listed behind the sources of the plane; taken by the row's owner, once): what small time
        // monitors need of the middle step — E^{n+1} behind the sources, H^{n+1/2} — is copied out for pair_record_kernel.
        // (The rows of a plane are fetched 64 at a time, one per lane, and the lanes that hold a row of THIS wave's grid row are
        //  found by a vote: a plane with entries costs one vector load, not one scalar round trip per entry — a probe's 16
        //  entries per plane had added 30 us to every 200^3 step, profiles/r3x.)
        for (int qb = q0; qb < q1; qb += 64) {
          int4 ev = {0, -1, 0, 0};
          if (qb + tx < q1) ev = inj.ent[qb + tx];
          unsigned long long hit = __ballot(ev.y == j);
          while (hit) {
            const int l = __ffsll(hit) - 1;
            hit &= hit - 1;
            int4 en;
            en.x = lane_value(ev.x, l); en.z = lane_value(ev.z, l); en.w = lane_value(ev.w, l);
            const int d = en.x - i0;
            if (en.z < 3) {
              const float v = inj.val[en.w];
#pragma unroll
              for (int e = 0; e < V; ++e) {
                if (d == e) {
                  if (en.z == 0) e1xn[e] += v;
                  else if (en.z == 1) e1yn[e] += v;
                  else e1zn[e] += v;
                }
              }
            } else if (MON && en.z >= (DISP ? 11 : 8) && own && k >= k0 && k < k1) {     // (DISP: E^{n+1} is sampled behind the ADE update, below)
              const int c = en.z - 8;
#pragma unroll
              for (int e = 0; e < V; ++e) {
                if (d == e) {
                  float mid = c == 0 ? e1xn[e] : (c == 1 ? e1yn[e] : (c == 2 ? e1zn[e] : (c == 3 ? hxn[e] : (c == 4 ? hyn[e] : hzn[e]))));
                  if constexpr (DAMP) {       // E^{n+1} is recorded behind its damping (which follows the sources); H^{n+1/2} in front of its own
                    const float4 b4 = xdm[tx], c4 = xdm[64 + tx];
                    const float bx[V] = {b4.x, b4.y, b4.z, b4.w}, cx[V] = {c4.x, c4.y, c4.z, c4.w};
                    if (c == 0) mid *= cx[e] * byv * bzk; else if (c == 1) mid *= bx[e] * cyv * bzk; else if (c == 2) mid *= bx[e] * byv * czk;
                  }
                  inj.cap[en.w] = mid;
                }
              }
            }
          }
        }
        if constexpr (DAMP) {                 // E^{n+1}: behind the update and the sources of step n
          {
            const float4 c4 = xdm[64 + tx];
            const float cx[V] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int e = 0; e < V; ++e) e1xn[e] *= cx[e] * byv * bzk;
          }
          {
            const float4 b4 = xdm[tx];
            const float bx[V] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int e = 0; e < V; ++e) { e1yn[e] *= bx[e] * cyv * bzk; e1zn[e] *= bx[e] * byv * czk; }
          }
        }
        if constexpr (DISP) {
          if (ds >= 0 && act) {
            const long long qb = ((long long)ds * 3) * 256 + tx * V;
            float cx[V], cy[V], cz[V];
            ldv<V>(cx, dp.cs + qb); ldv<V>(cy, dp.cs + qb + 256); ldv<V>(cz, dp.cs + qb + 512);
#pragma unroll
            for (int e = 0; e < V; ++e) { e1xn[e] = e1xn[e] - cx[e]; e1yn[e] = e1yn[e] - cy[e]; e1zn[e] = e1zn[e] - cz[e]; }
            // (a clipped launch: the columns outside its box belong to the shell's boxes, which leave their own E^{n+1} there)
            if (own && k >= k0 && k < k1 && (!CLIP || (i0o >= clip.i0 && i0o < clip.i1))) {
              stv<V>(dp.e1 + qb, e1xn); stv<V>(dp.e1 + qb + 256, e1yn); stv<V>(dp.e1 + qb + 512, e1zn);
            }
          }
          if constexpr (MON) {                 // small time monitors: their E samples of the middle step, behind everything
            if (own && k >= k0 && k < k1) {
              for (int qb = q0; qb < q1; qb += 64) {
                int4 ev = {0, -1, 0, 0};
                if (qb + tx < q1) ev = inj.ent[qb + tx];
                unsigned long long hit = __ballot(ev.y == j && ev.z >= 8 && ev.z < 11);
                while (hit) {
                  const int l = __ffsll(hit) - 1;
                  hit &= hit - 1;
                  const int d = lane_value(ev.x, l) - i0, c = lane_value(ev.z, l) - 8, w = lane_value(ev.w, l);
#pragma unroll
                  for (int e = 0; e < V; ++e)
                    if (d == e) inj.cap[w] = c == 0 ? e1xn[e] : (c == 1 ? e1yn[e] : e1zn[e]);
                }
              }
            }
          }
        }
        // the middle step over the boxes of DFT monitors: H^{n+1/2} for records at step n, E^{n+1} (behind its sources and damping)
        // for records at step n+1 — what the record launches of two single steps would read
        if constexpr (MON) {
          if (own && k >= k0 && k < k1) {
            for (int q = d0; q < d1; ++q) {
              const DumpBox bx = inj.dboxes[inj.dlist[q]];
              const int ly = j - bx.lo1, lz = k - bx.lo2;
              if (ly >= 0 && ly < bx.ny) {
                const long long rowo = ((long long)lz * bx.ny + ly) * bx.nx;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                  const int lx = i0o + e - bx.lo0;
                  // (a clipped launch: the columns outside its box belong to the shell's launches — shell2 pairs let DFT monitors reach into the shell)
                  if (lx >= 0 && lx < bx.nx && act && (!CLIP || (i0o + e >= clip.i0 && i0o + e < clip.i1))) {
                    if (bx.off[0] >= 0) inj.dump[bx.off[0] + rowo + lx] = e1xn[e];
                    if (bx.off[1] >= 0) inj.dump[bx.off[1] + rowo + lx] = e1yn[e];
                    if (bx.off[2] >= 0) inj.dump[bx.off[2] + rowo + lx] = e1zn[e];
                    if (bx.off[3] >= 0) inj.dump[bx.off[3] + rowo + lx] = hxn[e];
                    if (bx.off[4] >= 0) inj.dump[bx.off[4] + rowo + lx] = hyn[e];
                    if (bx.off[5] >= 0) inj.dump[bx.off[5] + rowo + lx] = hzn[e];
                  }
                }
              }
            }
          }
        }
        // what the neighbouring x tile needs of this step: repaired on the seam by seam_kernel
        if ((CLIP || (own && k >= k0 && k < k1)) && act) {
          float* sp = seam + seam_row + (long long)k * g.ny;
          if (seam_l) {
            sp[0] = hyn[V - 1];
            sp[seam_arr] = hzn[V - 1];
            sp[2 * seam_arr] = e1xn[V - 1];
            sp[3 * seam_arr] = e1yn[V - 1];
            sp[4 * seam_arr] = e1zn[V - 1];
          }
          if (seam_r) {
            sp[5 * seam_arr + seam_back] = e1yn[0];
            sp[6 * seam_arr + seam_back] = e1zn[0];
          }
        }
      }
    }
    // ---- S3: H2[k-1] ----
    float h2x[V], h2y[V], h2z[V];
    unspecified<V>(h2x); unspecified<V>(h2y); unspecified<V>(h2z);       // (rows j0-2, j0+R and the first iteration: nobody reads their H2)
    if (do_h2 && k > kA) {
      if constexpr (DAMP) {                   // H^{n+1/2} of plane k-1 is damped before step n+1 advances it (h1 is not read again here)
        {
          const float4 b4 = xdm[tx];
          const float bx[V] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int e = 0; e < V; ++e) h1x[e] *= bx[e] * cyv * cz_m;
        }
        {
          const float4 c4 = xdm[64 + tx];
          const float cx[V] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
          for (int e = 0; e < V; ++e) { h1y[e] *= cx[e] * byv * cz_m; h1z[e] *= cx[e] * cyv * bz_m; }
        }
      }
      if constexpr (SRC) {                    // the H-side source terms of step n+1 (behind the damping of H^{n+1/2}, as launch_sources follows launch_damp)
        if (ss_m >= 0 && act && sr.t->use_h2) {
          const long long qs = ((long long)ss_m * 3) * 256 + tx * V;
          float tx_[V], ty_[V], tz_[V];
          ldv<V>(tx_, sr.t->h2 + qs); ldv<V>(ty_, sr.t->h2 + qs + 256); ldv<V>(tz_, sr.t->h2 + qs + 512);
#pragma unroll
          for (int e = 0; e < V; ++e) { h1x[e] = h1x[e] + tx_[e]; h1y[e] = h1y[e] + ty_[e]; h1z[e] = h1z[e] + tz_[e]; }
          if (sr.t->h2b) {
            ldv<V>(tx_, sr.t->h2b + qs); ldv<V>(ty_, sr.t->h2b + qs + 256); ldv<V>(tz_, sr.t->h2b + qs + 512);
#pragma unroll
            for (int e = 0; e < V; ++e) { h1x[e] = h1x[e] + tx_[e]; h1y[e] = h1y[e] + ty_[e]; h1z[e] = h1z[e] + tz_[e]; }
          }
        }
      }
      // H-side point sources of step n+1 (codes 3 - 5 of the node table of plane k-1) act on H^{n+1/2} before step n+1 advances it
      // — E^{n+1} above was formed from the value without them, as in two single steps; h1 is not read again in this iteration
      if constexpr (MON) {
        if (inj.val2) {
          for (int qb = qm0; qb < qm1; qb += 64) {
            int4 ev = {0, -1, 8, 0};
            if (qb + tx < qm1) ev = inj.ent[qb + tx];
            unsigned long long hit = __ballot(ev.y == j && ev.z >= 3 && ev.z < 8);
            while (hit) {
              const int l = __ffsll(hit) - 1;
              hit &= hit - 1;
              const int d = lane_value(ev.x, l) - i0o, code = lane_value(ev.z, l);
              const float v = inj.val2[lane_value(ev.w, l)];
              // (selects, not conditional stores: merged into h1x[d] += v they send the three arrays to scratch memory — every
              //  instantiation that carries the monitor table ran 49 % slower for it at 512^3, profiles/r6/r6e)
#pragma unroll
              for (int e = 0; e < V; ++e) {
                const bool mine = d == e;
                h1x[e] = (mine && code == 3) ? h1x[e] + v : h1x[e];
                h1y[e] = (mine && code == 4) ? h1y[e] + v : h1y[e];
                h1z[e] = (mine && code == 5) ? h1z[e] + v : h1z[e];
              }
            }
          }
        }
      }
      float eyx = lane_next(e1y[0]);
      float ezx = lane_next(e1z[0]);
      if (txo == 63 || last_x) { eyx = 0.f; ezx = 0.f; }      // the wall, or a seam (repaired by seam_kernel)
      float4 t0, t1;
      if constexpr (WHATIF == 6) {
        t0 = make_float4(e1x[0], e1x[1], e1x[2], e1x[3]); t1 = make_float4(e1z[0], e1z[1], e1z[2], e1z[3]);
      } else {
        t0 = xch[(4 + (E1S ? 0 : (cur ^ 1) * 2) + 0) * slot + me + 64];
        t1 = xch[(4 + (E1S ? 0 : (cur ^ 1) * 2) + 1) * slot + me + 64];
      }
      const float exj1[V] = {t0.x, t0.y, t0.z, t0.w}, ezj1[V] = {t1.x, t1.y, t1.z, t1.w};
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float ey_ip = (e + 1 < V) ? e1y[(e + 1) % V] : eyx;
        const float ez_ip = (e + 1 < V) ? e1z[(e + 1) % V] : ezx;
        h2x[e] = upd_h(h1x[e], ch, ezj1[e] - e1z[e], ipy, e1yn[e] - e1y[e], ipz_m);
        h2y[e] = upd_h(h1y[e], ch, e1xn[e] - e1x[e], ipz_m, ez_ip - e1z[e], ipx[e]);
        h2z[e] = upd_h(h1z[e], ch, ey_ip - e1y[e], ipx[e], exj1[e] - e1x[e], ipy);
      }
      if constexpr (CLIP) {                    // H2 next to the seams, of every row / plane computed here (S4 writes the owned ones otherwise)
        if (act) {
          float* sq = seam + seam_row + (long long)(k - 1) * g.ny;
          if (seam_l) {
            sq[7 * seam_arr] = h2x[V - 1];
            sq[8 * seam_arr] = h2y[V - 2];
            sq[9 * seam_arr] = h2z[V - 2];
          }
          if (seam_r) {
            sq[10 * seam_arr + seam_back] = h2x[0];
            sq[11 * seam_arr + seam_back] = h2y[0];
            sq[12 * seam_arr + seam_back] = h2z[0];
          }
        }
      }
    }
    if constexpr (WHATIF != 6) {
      float4 t4;
      if constexpr (!E1S) {
        t4.x = e1xn[0]; t4.y = e1xn[1]; t4.z = e1xn[2]; t4.w = e1xn[3];
        xch[(4 + cur * 2 + 0) * slot + me] = t4;
        t4.x = e1zn[0]; t4.y = e1zn[1]; t4.z = e1zn[2]; t4.w = e1zn[3];
        xch[(4 + cur * 2 + 1) * slot + me] = t4;
      }
      if constexpr (EXJ) {          // E_x[k+1] of this row for the wave below (its E_x of the row above in S1 of the next iteration)
        t4.x = exn[0]; t4.y = exn[1]; t4.z = exn[2]; t4.w = exn[3];
        xch[XE * slot + me] = t4;
      }
      if constexpr (EZL) {
        if (k + 1 <= k1) {
          t4.x = ezn_pre[0]; t4.y = ezn_pre[1]; t4.z = ezn_pre[2]; t4.w = ezn_pre[3];
          xch[XZ * slot + me] = t4;
        }
      }
      t4.x = h2x[0]; t4.y = h2x[1]; t4.z = h2x[2]; t4.w = h2x[3];
      xch[2 * slot + me] = t4;
      t4.x = h2z[0]; t4.y = h2z[1]; t4.z = h2z[2]; t4.w = h2z[3];
      xch[3 * slot + me] = t4;
    }
    if constexpr (WHATIF != 2 && WHATIF != 6 && WHATIF != 8) __syncthreads();      // (2: no second barrier — the upper bound of a one-barrier pipeline)
    if constexpr (E1S) {          // E1[k] for the row below: read in S3 of the NEXT iteration, behind its first barrier — one buffer
      float4 t4;
      t4.x = e1xn[0]; t4.y = e1xn[1]; t4.z = e1xn[2]; t4.w = e1xn[3];
      xch[4 * slot + me] = t4;
      t4.x = e1zn[0]; t4.y = e1zn[1]; t4.z = e1zn[2]; t4.w = e1zn[3];
      xch[5 * slot + me] = t4;
    }
    // ---- S4: E2[k-1] ----
    if (own && k > k0) {
      float hyx = lane_prev(h2y[V - 1]);
      float hzx = lane_prev(h2z[V - 1]);
      if (txo == 0 || first_x) {                              // a seam (repaired by seam_kernel), or the wall
        if (first_x && pmc_x0) { hyx = -h2y[0]; hzx = -h2z[0]; }
        else { hyx = 0.f; hzx = 0.f; }
      }
      float hxj[V], hzj[V];
      if (WHATIF == 6) {
#pragma unroll
        for (int e = 0; e < V; ++e) { hxj[e] = h2x[e]; hzj[e] = h2z[e]; }
      } else if (j > 0) {
        const float4 t0 = xch[2 * slot + me - 64];
        const float4 t1 = xch[3 * slot + me - 64];
        hxj[0] = t0.x; hxj[1] = t0.y; hxj[2] = t0.z; hxj[3] = t0.w;
        hzj[0] = t1.x; hzj[1] = t1.y; hzj[2] = t1.z; hzj[3] = t1.w;
      } else if (pmc_y0) {
#pragma unroll
        for (int e = 0; e < V; ++e) { hxj[e] = -h2x[e]; hzj[e] = -h2z[e]; }
      } else {
        zero<V>(hxj); zero<V>(hzj);
      }
      if (pmc_z0 && k - 1 == 0) {
#pragma unroll
        for (int e = 0; e < V; ++e) { h2xm[e] = -h2x[e]; h2ym[e] = -h2y[e]; }
      }
      const bool wall_z = (k - 1 == 0) && !pmc_z0;
      float ex[V], ey[V], ez[V];
      auto s4 = [&](auto coef) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float hy_im = (e > 0) ? h2y[(e + V - 1) % V] : hyx;
          const float hz_im = (e > 0) ? h2z[(e + V - 1) % V] : hzx;
          float nex = upd_e(e1x[e], coef(0, e).x, coef(0, e).y, h2z[e] - hzj[e], idy, h2y[e] - h2ym[e], idz_m);
          float ney = upd_e(e1y[e], coef(1, e).x, coef(1, e).y, h2x[e] - h2xm[e], idz_m, h2z[e] - hz_im, idx[e]);
          float nez = upd_e(e1z[e], coef(2, e).x, coef(2, e).y, h2y[e] - hy_im, idx[e], h2x[e] - hxj[e], idy);
          const bool wx = wall_x0 && (e == 0);
          if (wall_y || wall_z) nex = 0.f;
          if (wx || wall_z) ney = 0.f;
          if (wx || wall_y) nez = 0.f;
          ex[e] = nex; ey[e] = ney; ez[e] = nez;
        }
      };
      if constexpr (MAT) {
        if (rw_m != kMixedWord) {
          const float2 c0 = lut_s[rw_m & 1023u], c1 = lut_s[(rw_m >> 10) & 1023u], c2 = lut_s[(rw_m >> 20) & 1023u];
          s4([&](int c, int) { return c == 0 ? c0 : (c == 1 ? c1 : c2); });
        } else {
          // (the packed words of plane k-1 are read again rather than carried: four registers the instantiation does not have)
          uint32_t mw_m[V] = {kBgWord, kBgWord, kBgWord, kBgWord};
          if (act) ldm<V>(mw_m, at(uni(m.m4 + pb - g.sxy), ub));
          s4([&](int c, int e) { return lut_s[(mw_m[e] >> (10 * c)) & 1023u]; });
        }
      } else {
        const float2 c1 = make_float2(ca, cb);
        s4([&](int, int) { return c1; });
      }
      if constexpr (SRC) {                    // the E-side source terms of step n+1
        if (ss_m >= 0 && act && sr.t->use_e2) {
          const long long qs = ((long long)ss_m * 3) * 256 + tx * V;
          float tx_[V], ty_[V], tz_[V];
          ldv<V>(tx_, sr.t->e2 + qs); ldv<V>(ty_, sr.t->e2 + qs + 256); ldv<V>(tz_, sr.t->e2 + qs + 512);
#pragma unroll
          for (int e = 0; e < V; ++e) { ex[e] = ex[e] + tx_[e]; ey[e] = ey[e] + ty_[e]; ez[e] = ez[e] + tz_[e]; }
          if (sr.t->e2b) {
            ldv<V>(tx_, sr.t->e2b + qs); ldv<V>(ty_, sr.t->e2b + qs + 256); ldv<V>(tz_, sr.t->e2b + qs + 512);
#pragma unroll
            for (int e = 0; e < V; ++e) { ex[e] = ex[e] + tx_[e]; ey[e] = ey[e] + ty_[e]; ez[e] = ez[e] + tz_[e]; }
          }
        }
      }
      // the node table of plane k-1: the E-side sources of step n+1 (when the launch carries them) act on E^{n+2}
      if (inj.val2 && inj.e2_in_sweep) {
        for (int qb = qm0; qb < qm1; qb += 64) {
          int4 ev = {0, -1, 8, 0};
          if (qb + tx < qm1) ev = inj.ent[qb + tx];
          unsigned long long hit = __ballot(ev.y == j && ev.z < 3);
          while (hit) {
            const int l = __ffsll(hit) - 1;
            hit &= hit - 1;
            const int d = lane_value(ev.x, l) - i0, code = lane_value(ev.z, l);
            const float v = inj.val2[lane_value(ev.w, l)];
#pragma unroll
            for (int e = 0; e < V; ++e) {
              if (d == e) {
                if (code == 0) ex[e] += v;
                else if (code == 1) ey[e] += v;
                else ez[e] += v;
              }
            }
          }
        }
      }
      if constexpr (DAMP) {
        if (dmp.e2) {                         // E^{n+2}: behind the update and the sources of step n+1 (else the caller damps it behind those)
          {
            const float4 c4 = xdm[64 + tx];
            const float cx[V] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int e = 0; e < V; ++e) ex[e] *= cx[e] * byv * bz_m;
          }
          {
            const float4 b4 = xdm[tx];
            const float bx[V] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int e = 0; e < V; ++e) { ey[e] *= bx[e] * cyv * bz_m; ez[e] *= bx[e] * byv * cz_m; }
          }
        }
      }
      if (act && (!CLIP || (i0o >= clip.i0 && i0o < clip.i1))) {
        // H2 next to the seams, for seam_kernel (so that it reads nothing but the scratch array, row-contiguous)
        float* sq = seam + seam_row + (long long)(k - 1) * g.ny;
        if (!CLIP && txo == 63 && !last_x) {
          sq[7 * seam_arr] = h2x[V - 1];
          sq[8 * seam_arr] = h2y[V - 2];
          sq[9 * seam_arr] = h2z[V - 2];
        }
        if (!CLIP && txo == 0 && tile_x > 0) {
          sq[10 * seam_arr - kSeamArrays * seam_arr] = h2x[0];
          sq[11 * seam_arr - kSeamArrays * seam_arr] = h2y[0];
          sq[12 * seam_arr - kSeamArrays * seam_arr] = h2z[0];
        }
        const long long po = pb - g.sxy + i0;
        if (WHATIF != 5 || ex[0] == 3.0e38f) {          // (5: no field stores — the condition keeps what leads to them alive)
          stv_h<V, NT>(b.hx + po, h2x);
          stv_h<V, NT>(b.hy + po, h2y);
          stv_h<V, NT>(b.hz + po, h2z);
          stv_h<V, NT>(b.ex + po, ex);
          stv_h<V, NT>(b.ey + po, ey);
          stv_h<V, NT>(b.ez + po, ez);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      h2xm[e] = h2x[e]; h2ym[e] = h2y[e];
      h1x[e] = hxn[e]; h1y[e] = hyn[e]; h1z[e] = hzn[e];
      e1x[e] = e1xn[e]; e1y[e] = e1yn[e]; e1z[e] = e1zn[e];
      exk[e] = exn[e]; eyk[e] = eyn[e];
    }
    exk_m = L.exn_m;
    ipz_m = ipz; idz_m = idz;
    qm0 = q0; qm1 = q1;
    if constexpr (DAMP) { bz_m = bzk; cz_m = czk; }
    if constexpr (MAT) rw_m = rw;
    if constexpr (SRC) ss_m = ss;
    cur ^= 1;
  };
  // two planes per trip: the carried values alternate between two register sets instead of being copied
  if constexpr (PF != 0) pf_issue(kA);
  for (int k = kA; k <= k1; k += 2) {
    body(k, LA);
    if (k + 1 <= k1) body(k + 1, LB2);
  }
}

// Tile classes (round 5): the materials instantiation sits at 128 VGPRs with spilled registers and looks coefficients up per row
// segment — 12 % slower than the plain one (V1 174 against V0 197 Gcells/s) — although four tiles in five of a body in a box of
// background hold nothing but the background medium.  With tcl.cls set, the workgroup of a tile whose cells (halo rows and planes
// included) are all background runs the PLAIN sweep (the uniform coefficients are the table's entry 1: the same bits), inside the
// same launch: two launches, one per class, were measured first and lose to the tail of the short one — 272 material tiles on 256
// CUs are two rounds of workgroups (profiles/r5/tile_classes.md).
template <int LB, int OPT>
__device__ __forceinline__ void fused2_tile_by_class(const GridP& g, const FieldP& a, const FieldP& b, const StepP& s, const MatP& m,
                                                     int zchunk, int nbx, int nby, int nbz, const InjP& inj, float* __restrict__ seam,
                                                     const DampT& dmp, const ClipP& clip, int t, const DispP& dp, const SrcP& sr, const TileClassP& tcl) {
  if constexpr ((OPT & 2) != 0) {
    if (tcl.cls) {
      const int cl = tcl.cls[t] & 3;
      if (cl == 0) {
        fused2_step_tile<LB, (OPT & ~(2 | 32))>(g, a, b, s, m, zchunk, nbx, nby, nbz, inj, seam, dmp, clip, t, dp, sr);
        return;
      }
      if constexpr ((OPT & 32) != 0) {       // (a tile without dispersive cells in a launch that carries them: the materials sweep)
        if (cl == 1) {
          fused2_step_tile<LB, (OPT & ~32)>(g, a, b, s, m, zchunk, nbx, nby, nbz, inj, seam, dmp, clip, t, dp, sr);
          return;
        }
      }
    }
  }
  fused2_step_tile<LB, OPT>(g, a, b, s, m, zchunk, nbx, nby, nbz, inj, seam, dmp, clip, t, dp, sr);
}

template <int LB, int OPT>
__global__ __launch_bounds__(LB, (LB == 512 ? 4 : 1)) void fused2_step_kernel(GridP g, FieldP a, FieldP b, StepP s, MatP m,
                                                         int zchunk, int nbx, int nby, int nbz, int xcd_remap,
                                                         InjP inj, float* __restrict__ seam, DampT dmp, ClipP clip, TileClassP tcl, DispP dp, SrcP sr) {
  const int total = nbx * nby * nbz;
  int t = blockIdx.x;
  if (xcd_remap == 1) {
    const int per = (total + 7) >> 3;
    t = (t & 7) * per + (t >> 3);
    if (t >= total) return;
  } else if (xcd_remap > 1) {
    const int G = xcd_remap, full = total / (8 * G) * (8 * G);
    if (t < full) {
      const int x = t & 7, mloc = t >> 3;
      t = ((mloc / G) * 8 + x) * G + mloc % G;
    }
    if (t >= total) return;
  }
  // (bit 2 of a tile's class, launches that add paged source terms: a row segment of the tile holds a source node — the others run
  //  the instantiation without those lines: the mode plane of BASELINE config 3 crosses 1 tile row in 64, and the lines cost 8 %)
  // (every tile body of this kernel shares the LDS the launcher sized from the kernel's own OPT word: fused2_xch_arrays)
  constexpr int OPTK = OPT | (fused2_exj(LB, OPT) ? 0 : 4096);
  if constexpr ((OPT & 64) != 0) {
    if (tcl.cls && !(tcl.cls[t] & 4)) {
      fused2_tile_by_class<LB, (OPTK & ~64)>(g, a, b, s, m, zchunk, nbx, nby, nbz, inj, seam, dmp, clip, t, dp, sr, tcl);
      return;
    }
  }
  fused2_tile_by_class<LB, OPTK>(g, a, b, s, m, zchunk, nbx, nby, nbz, inj, seam, dmp, clip, t, dp, sr, tcl);
}

// ---- the seams between x tiles -------------------------------------------------------------------------------------
// One thread per (seam, j, k); c = first column of the right tile.  H2_{y,z}[c-1] needs E1_{y,z}[c] of the right tile, and
// E2_{x,y,z}[c-1], E2_{y,z}[c] differentiate H2_{y,z}[c-1]: recomputed here with the formulas of the sweep from what both
// tiles left in the scratch array [seam][13][nz + 2][ny] (read row-contiguously; plane nz and what lies beyond the walls
// stay zero).  H2_{y,z}[c-1] of the row below and of the plane below are recomputed rather than exchanged: one launch.
// Round 6: the E-side source terms of step n+1 on a seam column are added here when the sweep added the others (inj.e2_in_sweep):
// a node next to a seam no longer sends all of them behind the launch.
__global__ __launch_bounds__(256) void seam_kernel(GridP g, FieldP b, StepP s, MatP m,
                                                   const float* __restrict__ seam, int n_seams, DampT dmp, ClipP clip, InjP inj, SrcP sr) {
  // (clip: the box the sweep wrote — the whole grid, or the bulk of a grid whose shell takes single steps; rows and planes
  //  are those of the box, and a seam column that lies outside it is left alone)
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nj = clip.j1 - clip.j0;
  const long long per = (long long)nj * (clip.k1 - clip.k0);
  if (t >= per * n_seams) return;
  const int sm = (int)(t / per);
  const int k = clip.k0 + (int)((t % per) / nj), j = clip.j0 + (int)(t % nj);
  // (periodic x: the last seam is the wrap — its left column is the row's last, nx - 1, its right column the row's first)
  const bool wrap = g.bcx0 == BC_PERIODIC && sm == n_seams - 1;
  const int c = wrap ? g.nx : (sm + 1) * 256;
  const int cc = wrap ? 0 : c;                          // the right column's index in its row
  const bool wl = c - 1 >= clip.i0 && c - 1 < clip.i1, wr = cc >= clip.i0 && cc < clip.i1;
  if (!wl && !wr) return;
  const float ch = g.ch;
  const float ipx = s.ipx[c - 1];
  auto A = [&](int a, int jj, int kk) { return seam[seam_at(g, sm, a, jj, kk)]; };
  // H2_{y,z}[c-1] at (jj, kk)
  auto h2 = [&](int jj, int kk, float& hy, float& hz) {
    const float e1x = A(2, jj, kk), e1y = A(3, jj, kk), e1z = A(4, jj, kk);
    const float e1x_jp = (jj + 1 < g.ny) ? A(2, jj + 1, kk) : 0.f;
    float h1y = A(0, jj, kk), h1z = A(1, jj, kk);
    if (dmp.fb[0]) {                        // absorber layers: H^{n+1/2} is damped before step n+1 advances it (column c-1: x centre)
      const float cxv = dmp.fc[0][c - 1];
      h1y *= cxv * dmp.fb[1][jj] * dmp.fc[2][kk];
      h1z *= cxv * dmp.fc[1][jj] * dmp.fb[2][kk];
    }
    if (sr.sseg && sr.t->use_h2) {             // paged H-side source terms of step n+1 at column c-1
      const int blk = sr.sseg[((long long)kk * g.ny + jj) * ((g.nx + 255) >> 8) + ((c - 1) >> 8)];
      if (blk >= 0) {
        const long long qs = ((long long)blk * 3) * 256 + ((c - 1) & 255);
        h1y = h1y + sr.t->h2[qs + 256];
        h1z = h1z + sr.t->h2[qs + 512];
        if (sr.t->h2b) { h1y = h1y + sr.t->h2b[qs + 256]; h1z = h1z + sr.t->h2b[qs + 512]; }
      }
    }
    hy = upd_h(h1y, ch, A(2, jj, kk + 1) - e1x, s.ipz[kk], A(6, jj, kk) - e1z, ipx);
    hz = upd_h(h1z, ch, A(5, jj, kk) - e1y, ipx, e1x_jp - e1x, s.ipy[jj]);
  };
  float hy_m, hz_m;
  h2(j, k, hy_m, hz_m);
  // (row 0 / plane 0: a PEC wall, or a PMC one with H mirrored behind it)
  const bool pmc_y0 = g.bcy0 == BC_PMC, pmc_z0 = !g.pec_z0;
  const bool wall_y = (j == 0) && !pmc_y0, wall_z = (k == 0) && !pmc_z0;
  const bool mir_y = (j == 0) && pmc_y0, mir_z = (k == 0) && pmc_z0;
  const long long pr = (long long)k * g.sxy + (long long)j * g.nx;
  const long long p = pr + cc, pl = pr + c - 1;                              // the right column (c, or the row's first) / the left one
  if (wl) { b.hy[pl] = hy_m; b.hz[pl] = hz_m; }
  const float idy = s.idy[j], idz = s.idz[k], idx_m = s.idx[c - 1], idx_c = s.idx[cc];
  const float hx_m = A(7, j, k), hy_mm = A(8, j, k), hz_mm = A(9, j, k);
  const float hx_c = A(10, j, k), hy_c = A(11, j, k), hz_c = A(12, j, k);
  // (Ca, Cb) of component c at column c-1 (p - 1) / c (p): the table entry of the cell's medium word, or the uniform medium
  auto coef = [&](long long cell, int c) {
    return m.m4 ? m.lut[(m.m4[cell] >> (10 * c)) & 1023u] : make_float2(m.ca1, m.cb1);
  };
  float ex_m = 0.f, ey_m = 0.f, ez_m = 0.f, ey_c = 0.f, ez_c = 0.f;
  float hy_k = 0.f, hz_j = 0.f, dum;
  if (mir_z) hy_k = -hy_m; else if (!wall_z) h2(j, k - 1, hy_k, dum);
  if (mir_y) hz_j = -hz_m; else if (!wall_y) h2(j - 1, k, dum, hz_j);
  // H2_x of the plane / row below (columns c-1 and c)
  const float hxm_k = wall_z ? 0.f : (mir_z ? -hx_m : A(7, j, k - 1)), hxc_k = wall_z ? 0.f : (mir_z ? -hx_c : A(10, j, k - 1));
  const float hxm_j = wall_y ? 0.f : (mir_y ? -hx_m : A(7, j - 1, k)), hxc_j = wall_y ? 0.f : (mir_y ? -hx_c : A(10, j - 1, k));
  if (!wall_y && !wall_z) { const float2 q = coef(pl, 0); ex_m = upd_e(A(2, j, k), q.x, q.y, hz_m - hz_j, idy, hy_m - hy_k, idz); }
  if (!wall_z) {
    const float2 qm = coef(pl, 1), qc = coef(p, 1);
    ey_m = upd_e(A(3, j, k), qm.x, qm.y, hx_m - hxm_k, idz, hz_m - hz_mm, idx_m);
    ey_c = upd_e(A(5, j, k), qc.x, qc.y, hx_c - hxc_k, idz, hz_c - hz_m, idx_c);
  }
  if (!wall_y) {
    const float2 qm = coef(pl, 2), qc = coef(p, 2);
    ez_m = upd_e(A(4, j, k), qm.x, qm.y, hy_m - hy_mm, idx_m, hx_m - hxm_j, idy);
    ez_c = upd_e(A(6, j, k), qc.x, qc.y, hy_c - hy_m, idx_c, hx_c - hxc_j, idy);
  }
  if (sr.sseg && sr.t->use_e2) {                // paged E-side source terms of step n+1 on these two columns
    const int nbxg = (g.nx + 255) >> 8;
    const int bl = sr.sseg[((long long)k * g.ny + j) * nbxg + ((c - 1) >> 8)], br = sr.sseg[((long long)k * g.ny + j) * nbxg + (cc >> 8)];
    if (bl >= 0) {
      const long long qs = ((long long)bl * 3) * 256 + ((c - 1) & 255);
      ex_m = ex_m + sr.t->e2[qs]; ey_m = ey_m + sr.t->e2[qs + 256]; ez_m = ez_m + sr.t->e2[qs + 512];
      if (sr.t->e2b) { ex_m = ex_m + sr.t->e2b[qs]; ey_m = ey_m + sr.t->e2b[qs + 256]; ez_m = ez_m + sr.t->e2b[qs + 512]; }
    }
    if (br >= 0) {
      const long long qs = ((long long)br * 3) * 256 + (cc & 255);
      ey_c = ey_c + sr.t->e2[qs + 256]; ez_c = ez_c + sr.t->e2[qs + 512];
      if (sr.t->e2b) { ey_c = ey_c + sr.t->e2b[qs + 256]; ez_c = ez_c + sr.t->e2b[qs + 512]; }
    }
  }
  if (inj.val2 && inj.e2_in_sweep) {         // the E-side sources of step n+1 on these two columns (the plane's table rows in their order)
    const int q0 = inj.start[k], q1 = inj.start[k + 1];
    for (int q = q0; q < q1; ++q) {
      const int4 en = inj.ent[q];
      if (en.y != j || en.z >= 3) continue;
      const float v = inj.val2[en.w];
      if (en.x == c - 1) { if (en.z == 0) ex_m += v; else if (en.z == 1) ey_m += v; else ez_m += v; }
      else if (en.x == cc) { if (en.z == 1) ey_c += v; else if (en.z == 2) ez_c += v; }
    }
  }
  if (dmp.fb[0] && dmp.e2) {                 // E^{n+2} damped as in the sweep
    const float bxm = dmp.fb[0][c - 1], cxm = dmp.fc[0][c - 1], bxc = dmp.fb[0][cc];
    const float byv = dmp.fb[1][j], cyv = dmp.fc[1][j], bzv = dmp.fb[2][k], czv = dmp.fc[2][k];
    ex_m *= cxm * byv * bzv; ey_m *= bxm * cyv * bzv; ez_m *= bxm * byv * czv;
    ey_c *= bxc * cyv * bzv; ez_c *= bxc * byv * czv;
  }
  if (wl) { b.ex[pl] = ex_m; b.ey[pl] = ey_m; b.ez[pl] = ez_m; }
  if (wr) { b.ey[p] = ey_c; b.ez[p] = ez_c; }
}

}  // namespace fdtd
