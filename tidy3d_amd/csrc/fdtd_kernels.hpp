// Hand-written CDNA4 (gfx950) kernels of the FDTD hot path.  HIP only — no CUDA shims, no
// dual paths.  SURVEY.md section 8(a) rows K1-K7; the reference contains no counterpart (its
// solver is a closed service), conventions are pinned as cited in tidy3d_amd/coeffs.py.
//
// Memory layout: every field is [nz+2][ny][nx] fp32, x fastest, with one ghost xy-plane below
// (k = -1) and above (k = nz) the slab; kernels receive pointers to interior plane k = 0.  The
// ghost planes carry the z boundary condition (zeros for PEC, wrapped copy for periodic,
// negated copy for PMC) or the neighbouring rank's plane in a z-slab decomposition, so the hot
// kernels have no z special case.  A wavefront (64 lanes) always covers 64*V consecutive x
// cells of ONE row: global loads are fully coalesced 256 B / 1 KiB segments, x+-1 neighbours
// come from the adjacent lane by __shfl, z+-1 neighbours from registers carried along the
// z-march, y+-1 neighbours from the row loaded by the neighbouring wave (L1/L2 hit).
//
// Roofline: HBM-bound stencil; algorithmic traffic 36 B/cell per pass (3 reads of the curl
// source + 3 reads and 3 writes of the updated field), no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdtd {

enum { BC_PEC = 0, BC_PMC = 1, BC_PERIODIC = 2, BC_NEIGHBOR = 3 };

struct GridP {
  int nx, ny, nz;
  long long sxy;                 // nx * ny
  int bcx0, bcx1, bcy0, bcy1;    // boundary codes of the x / y faces
  int pec_z0;                    // tangential E on local plane k == 0 is a PEC wall
  float ch;                      // dt / mu0
  int psi_ghost;                 // plane -1 belongs to another rank: its H-side psi (x, y axes) sits in slot nz of the psi arrays
};

struct FieldP {
  float* ex; float* ey; float* ez;
  float* hx; float* hy; float* hz;
};

struct StepP {
  const float* ipx; const float* ipy; const float* ipz;   // 1 / primal steps
  const float* idx; const float* idy; const float* idz;   // 1 / dual steps
};

// Material words: one 32-bit word per cell, bits [10c, 10c+10) = medium index of E_c (<= 1023 media:
// enough for the 0.2 % permittivity steps sub-pixel averaging and CustomMedium quantise to, at the
// same 4 B/cell as 8-bit indices).  `roww` holds one word per row segment of 256 cells (one wavefront
// of the sweep): the word all its cells share, or kMixedWord — a scalar load that lets the sweep skip
// the packed words and the per-cell table look-ups wherever the medium is uniform along the segment.
constexpr int kMaxMedia = 1024;
constexpr uint32_t kBgWord = 1u | (1u << 10) | (1u << 20);
constexpr uint32_t kMixedWord = 0xFFFFFFFFu;
// More than 1022 media (ref scene.py:52 allows 65530; a CustomMedium with thousands of distinct (permittivity, conductivity)
// levels): the WIDE layout — 16-bit indices, two words per cell (m4: E_x | E_y << 16, m4b: E_z), coefficients read from the table in
// global memory (it does not fit LDS).  Two-pass kernels only (e_update_kernel, the slab-form CPML): the sweeps keep the packed
// 10-bit words and their LDS table; a run with a wide table takes FDTD_VARIANT_ZMARCH.
constexpr int kMaxMediaWide = 65531;
struct MatP {
  const uint32_t* m4;             // packed material indices
  const uint32_t* m4b;            // wide layout: the second word per cell (E_z index); nullptr = 10-bit packed words in m4
  const uint32_t* roww;           // [nz][ny][ceil(nx / 256)] row-segment words
  const float2* lut;              // (ca, cb) per medium
  int n_media;
  float ca1, cb1;                 // uniform medium (entry 1)
};

// ---- small vector helpers -------------------------------------------------------------------
template <int V>
__device__ __forceinline__ void ldv(float (&r)[V], const float* __restrict__ p) {
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) r[e] = p[e];
  }
}

template <int V>
__device__ __forceinline__ void stv(float* __restrict__ p, const float (&r)[V]) {
  if constexpr (V == 4) {
    float4 t; t.x = r[0]; t.y = r[1]; t.z = r[2]; t.w = r[3];
    *reinterpret_cast<float4*>(p) = t;
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) p[e] = r[e];
  }
}

// Streaming store: what the sweep writes is not read again before the next step, 6 GB of traffic later.  The
// non-temporal hint keeps those lines from displacing the rows that NEIGHBOURING workgroups are about to re-read
// (halo rows) from the 4 MiB L2 of the XCD.
template <int V, bool NT>
__device__ __forceinline__ void stv_h(float* __restrict__ p, const float (&r)[V]) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (NT && V == 4) {
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    v4f_ t; t.x = r[0]; t.y = r[1]; t.z = r[2]; t.w = r[3];
    __builtin_nontemporal_store(t, reinterpret_cast<v4f_*>(p));
    return;
  }
#endif
  stv<V>(p, r);
}

// packed material words of V consecutive cells: ONE 16-byte load per thread (V = 4) instead of
// three 4-byte ones
template <int V>
__device__ __forceinline__ void ldm(uint32_t (&m)[V], const uint32_t* __restrict__ p) {
  if constexpr (V == 4) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    m[0] = t.x; m[1] = t.y; m[2] = t.z; m[3] = t.w;
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = p[e];
  }
}

// The two update formulas, written once with explicit fused multiply-adds.  The library is built
// with -ffp-contract=off, so these are the ONLY fma's: every code path (vector body, tile-edge
// scalar column, chunk prologue, two-pass and fused kernels) rounds identically and results do
// not depend on the launch geometry or on the kernel variant.
//   H_new = h - ch * (a * wa - b * wb)
__device__ __forceinline__ float upd_h(float h, float ch, float a, float wa, float b, float wb) {
  return fmaf(-ch, fmaf(-b, wb, a * wa), h);
}
//   E_new = ca * e + cb * (a * wa - b * wb)
__device__ __forceinline__ float upd_e(float e, float ca, float cb, float a, float wa, float b, float wb) {
  return fmaf(cb, fmaf(-b, wb, a * wa), ca * e);
}

// A wave-uniform pointer, pinned to SGPRs and opaque to the optimiser.  Without it loop-invariant code
// motion splits  base + k * plane + lane offset  into a per-lane 64-bit (base + lane offset) that it
// keeps in a VGPR pair for the whole z-march — one pair per array, ~30 VGPRs of the sweep — instead
// of the scalar-base + 32-bit lane offset addressing the hardware offers.  Emits no instruction.
template <typename T>
__device__ __forceinline__ T* uni(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+s"(p));
#endif
  return p;
}

// uniform base pointer + per-lane 32-bit byte offset: the form the scalar-base global loads take
__device__ __forceinline__ const float* at(const float* base, unsigned byte_off) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float* at(float* base, unsigned byte_off) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off);
}
__device__ __forceinline__ const uint32_t* at(const uint32_t* base, unsigned byte_off) {
  return reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + byte_off);
}

template <int V>
__device__ __forceinline__ void zero(float (&r)[V]) {
#pragma unroll
  for (int e = 0; e < V; ++e) r[e] = 0.f;
}

// Registers of lanes that take no part (cells beyond the row end, rows beyond the grid): nothing they hold is
// ever stored or read by an active lane, so they need no defined value — and no v_mov per plane to give them
// one (the zero-fills were 32 of the ~200 vector instructions of a plane).
template <int V>
__device__ __forceinline__ void unspecified(float (&r)[V]) {
#if defined(__has_builtin)
#if __has_builtin(__builtin_nondeterministic_value)
#pragma unroll
  for (int e = 0; e < V; ++e) r[e] = __builtin_nondeterministic_value(r[e]);
  return;
#endif
#endif
  zero<V>(r);
}

// =============================================================================================
// K1  H-update:  H -= ch * curl_primal(E)      (one workgroup = ROWS rows x 64*V cells, marched
//                                               over `zchunk` planes with E_x,E_y carried in
//                                               registers between planes)
// =============================================================================================
template <int V>
__global__ __launch_bounds__(512) void h_update_kernel(GridP g, FieldP f, StepP s, int kbeg, int kend,
                                                        int zchunk) {
  const int tx = threadIdx.x;
  const int i0 = (blockIdx.x * 64 + tx) * V;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (j >= g.ny) return;                       // whole wavefront (one row) leaves together
  const int k0 = kbeg + blockIdx.z * zchunk;
  const int k1 = min(k0 + zchunk, kend);
  const bool act = i0 < g.nx;
  const bool last_x = (i0 + V >= g.nx);
  const bool per_x = g.bcx1 == BC_PERIODIC;
  const bool use_jp = (j + 1 < g.ny) || (g.bcy1 == BC_PERIODIC);
  const long long row = (long long)j * g.nx + i0;
  const long long rowp = (j + 1 < g.ny) ? row + g.nx : (long long)i0;    // periodic wrap -> row 0
  const float ch = g.ch;

  float ipx[V];
  zero<V>(ipx);
  if (act) ldv<V>(ipx, s.ipx + i0);
  const float ipy = s.ipy[j];

  float exk[V], eyk[V];
  zero<V>(exk); zero<V>(eyk);
  if (act) {
    ldv<V>(exk, f.ex + (long long)k0 * g.sxy + row);
    ldv<V>(eyk, f.ey + (long long)k0 * g.sxy + row);
  }
  for (int k = k0; k < k1; ++k) {
    const long long p = (long long)k * g.sxy + row;
    const long long pj = (long long)k * g.sxy + rowp;
    const float ipz = s.ipz[k];
    float exn[V], eyn[V], ezk[V], exj[V], ezj[V], hx[V], hy[V], hz[V];
    zero<V>(exn); zero<V>(eyn); zero<V>(ezk); zero<V>(exj); zero<V>(ezj);
    if (act) {
      ldv<V>(exn, f.ex + p + g.sxy);
      ldv<V>(eyn, f.ey + p + g.sxy);
      ldv<V>(ezk, f.ez + p);
      if (use_jp) {
        ldv<V>(exj, f.ex + pj);
        ldv<V>(ezj, f.ez + pj);
      }
      ldv<V>(hx, f.hx + p);
      ldv<V>(hy, f.hy + p);
      ldv<V>(hz, f.hz + p);
    }
    // x+1 neighbour of the last element: lane+1's first element
    float eyx = __shfl_down(eyk[0], 1);
    float ezx = __shfl_down(ezk[0], 1);
    if (act && (tx == 63 || last_x)) {
      if (!last_x) { eyx = f.ey[p + V]; ezx = f.ez[p + V]; }
      else if (per_x) { eyx = f.ey[p - i0]; ezx = f.ez[p - i0]; }
      else { eyx = 0.f; ezx = 0.f; }
    }
    if (act) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float ey_ip = (e + 1 < V) ? eyk[(e + 1) % V] : eyx;
        const float ez_ip = (e + 1 < V) ? ezk[(e + 1) % V] : ezx;
        hx[e] = upd_h(hx[e], ch, ezj[e] - ezk[e], ipy, eyn[e] - eyk[e], ipz);
        hy[e] = upd_h(hy[e], ch, exn[e] - exk[e], ipz, ez_ip - ezk[e], ipx[e]);
        hz[e] = upd_h(hz[e], ch, ey_ip - eyk[e], ipx[e], exj[e] - exk[e], ipy);
      }
      stv<V>(f.hx + p, hx);
      stv<V>(f.hy + p, hy);
      stv<V>(f.hz + p, hz);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { exk[e] = exn[e]; eyk[e] = eyn[e]; }
  }
}

// =============================================================================================
// K2  E-update:  E = Ca E + Cb * curl_dual(H), PEC walls on the min faces, optional uint8
//                material index -> (Ca, Cb) look-up table staged in LDS.
// =============================================================================================
template <int V, bool MAT>
__global__ __launch_bounds__(512) void e_update_kernel(GridP g, FieldP f, StepP s, MatP m, int kbeg,
                                                        int kend, int zchunk) {
  __shared__ float2 lut_s[MAT ? kMaxMedia : 1];
  if constexpr (MAT) {
    for (int t = threadIdx.y * blockDim.x + threadIdx.x; t < min(m.n_media, kMaxMedia); t += blockDim.x * blockDim.y)
      lut_s[t] = m.lut[t];
    __syncthreads();
  }
  const int tx = threadIdx.x;
  const int i0 = (blockIdx.x * 64 + tx) * V;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (j >= g.ny) return;
  const int k0 = kbeg + blockIdx.z * zchunk;
  const int k1 = min(k0 + zchunk, kend);
  const bool act = i0 < g.nx;
  const long long row = (long long)j * g.nx + i0;
  // row j-1 and how to use it
  const int bcy = g.bcy0, bcx = g.bcx0;
  const long long rowm = (j > 0) ? row - g.nx : row + (long long)(g.ny - 1) * g.nx;  // periodic wrap
  const int ymode = (j > 0) ? 0 : (bcy == BC_PERIODIC ? 0 : (bcy == BC_PMC ? 1 : 2));  // 0 load,1 negate,2 zero
  const bool first_x = (i0 == 0);
  const bool wall_y = (j == 0) && (bcy == BC_PEC);
  const bool wall_x0 = first_x && (bcx == BC_PEC);

  float idx[V];
  zero<V>(idx);
  if (act) ldv<V>(idx, s.idx + i0);
  const float idy = s.idy[j];

  float hxm[V], hym[V];
  zero<V>(hxm); zero<V>(hym);
  if (act) {
    ldv<V>(hxm, f.hx + (long long)(k0 - 1) * g.sxy + row);
    ldv<V>(hym, f.hy + (long long)(k0 - 1) * g.sxy + row);
  }
  for (int k = k0; k < k1; ++k) {
    const long long p = (long long)k * g.sxy + row;
    const long long pj = (long long)k * g.sxy + rowm;
    const float idz = s.idz[k];
    float hxk[V], hyk[V], hzk[V], hxj[V], hzj[V], ex[V], ey[V], ez[V];
    zero<V>(hxk); zero<V>(hyk); zero<V>(hzk); zero<V>(hxj); zero<V>(hzj);
    if (act) {
      ldv<V>(hxk, f.hx + p);
      ldv<V>(hyk, f.hy + p);
      ldv<V>(hzk, f.hz + p);
      if (ymode == 0) {
        ldv<V>(hxj, f.hx + pj);
        ldv<V>(hzj, f.hz + pj);
      } else if (ymode == 1) {
#pragma unroll
        for (int e = 0; e < V; ++e) { hxj[e] = -hxk[e]; hzj[e] = -hzk[e]; }
      }
      ldv<V>(ex, f.ex + p);
      ldv<V>(ey, f.ey + p);
      ldv<V>(ez, f.ez + p);
    }
    // x-1 neighbour of the first element: lane-1's last element
    float hyx = __shfl_up(hyk[V - 1], 1);
    float hzx = __shfl_up(hzk[V - 1], 1);
    if (act && (tx == 0 || first_x)) {
      if (!first_x) { hyx = f.hy[p - 1]; hzx = f.hz[p - 1]; }
      else if (bcx == BC_PERIODIC) { hyx = f.hy[p + g.nx - 1]; hzx = f.hz[p + g.nx - 1]; }
      else if (bcx == BC_PMC) { hyx = -hyk[0]; hzx = -hzk[0]; }
      else { hyx = 0.f; hzx = 0.f; }
    }
    if (act) {
      float cax[V], cbx[V], cay[V], cby[V], caz[V], cbz[V];
      if constexpr (MAT) {
        uint32_t mw[V];
        ldm<V>(mw, m.m4 + p);
        if (m.m4b) {                 // wide layout (wave-uniform): 16-bit indices, the table in global memory
          uint32_t mz[V];
          ldm<V>(mz, m.m4b + p);
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const float2 c0 = m.lut[mw[e] & 0xFFFFu], c1 = m.lut[mw[e] >> 16], c2 = m.lut[mz[e] & 0xFFFFu];
            cax[e] = c0.x; cbx[e] = c0.y; cay[e] = c1.x; cby[e] = c1.y; caz[e] = c2.x; cbz[e] = c2.y;
          }
        } else
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float2 c0 = lut_s[mw[e] & 1023u];
          const float2 c1 = lut_s[(mw[e] >> 10) & 1023u];
          const float2 c2 = lut_s[(mw[e] >> 20) & 1023u];
          cax[e] = c0.x; cbx[e] = c0.y; cay[e] = c1.x; cby[e] = c1.y; caz[e] = c2.x; cbz[e] = c2.y;
        }
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          cax[e] = cay[e] = caz[e] = m.ca1;
          cbx[e] = cby[e] = cbz[e] = m.cb1;
        }
      }
      const bool wall_z = (k == 0) && g.pec_z0;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float hy_im = (e > 0) ? hyk[(e + V - 1) % V] : hyx;
        const float hz_im = (e > 0) ? hzk[(e + V - 1) % V] : hzx;
        float nex = upd_e(ex[e], cax[e], cbx[e], hzk[e] - hzj[e], idy, hyk[e] - hym[e], idz);
        float ney = upd_e(ey[e], cay[e], cby[e], hxk[e] - hxm[e], idz, hzk[e] - hz_im, idx[e]);
        float nez = upd_e(ez[e], caz[e], cbz[e], hyk[e] - hy_im, idx[e], hxk[e] - hxj[e], idy);
        const bool wx = wall_x0 && (e == 0);
        if (wall_y || wall_z) nex = 0.f;
        if (wx || wall_z) ney = 0.f;
        if (wx || wall_y) nez = 0.f;
        ex[e] = nex; ey[e] = ney; ez[e] = nez;
      }
      stv<V>(f.ex + p, ex);
      stv<V>(f.ey + p, ey);
      stv<V>(f.ez + p, ez);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { hxm[e] = hxk[e]; hym[e] = hyk[e]; }
  }
}

// ---- CPML inside the fused sweep ------------------------------------------------------------
// Same recursion and the same operation order as the slab kernels pml_h_kernel / pml_e_kernel
// (K3 below), applied to registers: H-side corrections to H^{n-1/2} before upd_h, E-side ones to
// E^{n+1} after upd_e.  Halo rows / columns / prologue planes recompute the corrected value from
// psi without storing it, so psi is updated exactly once per cell.
//
// Slab membership is the same on the E and the H side: index ia of axis a is a member when
// ia < lo or ia >= hi0 (the tables are identity on the few extra cells this adds on the E side),
// slab index si = ia (low face) or lo + ia - hi0 (high face), psi extent ns = lo + n - hi0.
// Along x the ranges are rounded to multiples of 4 cells, so the float4 of a lane lies entirely
// inside or outside a slab and its psi values are ONE 16-byte load per array.
//
// The parameter block lives in device memory and is passed by pointer: its fields are fetched by
// scalar loads where they are used instead of occupying ~90 SGPRs for the whole kernel (the
// by-value form spilled SGPRs into VGPR lanes and cost an occupancy step).
// Pointers that come out of a parameter block in memory have no known address space: a plain access
// through them is a FLAT load with a 64-bit VGPR address.  These helpers say what they are — global
// memory for psi, constant (never written while a kernel runs) for the coefficient tables — so the
// accesses become scalar-base global loads / scalar loads.  (The host-side emulator build has one
// address space.)
#if defined(__HIPCC__)
#define FDTD_AS_GLOBAL __attribute__((address_space(1)))
#define FDTD_AS_CONST __attribute__((address_space(4)))
#else
#define FDTD_AS_GLOBAL
#define FDTD_AS_CONST
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
// (base = wave-uniform pointer, off = the lane's 32-bit byte offset)
__device__ __forceinline__ void ldg4(float (&r)[4], const float* base, unsigned off) {
  const v4f t = *(const FDTD_AS_GLOBAL v4f*)((const FDTD_AS_GLOBAL char*)base + off);
  r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
}
// Field row load: uniform base + the lane's byte offset.  GLOBAL = through an explicit global-address-space pointer
// (a pointer that went through uni()'s asm is generic to the compiler, which then emits flat_load): -0.15 ... -0.35 %
// measured inside engines on the instantiations without CPML (profiles/r03h), -0.3 ... -0.5 % on the CPML step (r04h).
template <int V, bool GLOBAL>
__device__ __forceinline__ void ldf(float (&r)[V], const float* base, unsigned off) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (GLOBAL && V == 4) {
    const v4f t = *(const FDTD_AS_GLOBAL v4f*)((const FDTD_AS_GLOBAL char*)base + off);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    return;
  }
#endif
  ldv<V>(r, at(base, off));
}
// Streaming load of data no other wave reads (E_y and H_y have no y-neighbour in the curl).
template <int V, bool NT>
__device__ __forceinline__ void ldv_h(float (&r)[V], const float* base, unsigned off) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (NT && V == 4) {
    const v4f t = __builtin_nontemporal_load((const FDTD_AS_GLOBAL v4f*)((const FDTD_AS_GLOBAL char*)base + off));
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    return;
  }
#endif
  ldv<V>(r, at(base, off));
}
__device__ __forceinline__ void stg4(float* base, unsigned off, const float (&r)[4]) {
  v4f t; t.x = r[0]; t.y = r[1]; t.z = r[2]; t.w = r[3];
  *(FDTD_AS_GLOBAL v4f*)((FDTD_AS_GLOBAL char*)base + off) = t;
}
__device__ __forceinline__ float ldg1(const float* p) { return *(const FDTD_AS_GLOBAL float*)p; }
__device__ __forceinline__ void ldc4(float (&r)[4], const float* base, unsigned off) {
  const v4f t = *(const FDTD_AS_CONST v4f*)((const FDTD_AS_CONST char*)base + off);
  r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
}
__device__ __forceinline__ float4 ldc_f4(const float4* p) {
  const v4f t = *(const FDTD_AS_CONST v4f*)p;
  float4 o; o.x = t.x; o.y = t.y; o.z = t.z; o.w = t.w;
  return o;
}

struct PmlAxisP {
  const float4* ce4;      // [n] {1/kappa_e - 1, b_e, c_e, 0}  (y / z: wave-uniform index -> scalar load)
  const float4* ch4;      // [n] {1/kappa_h - 1, b_h, c_h, 0}
  const float* kv_e; const float* b_e; const float* c_e;     // the same, one array each (x: float4 loads along the row)
  const float* kv_h; const float* b_h; const float* c_h;
  float* pe0; float* pe1;               // psi of E_{a+1}, E_{a+2}
  const float* ph0; const float* ph1;   // psi of H_{a+1}, H_{a+2}: READ set  (halo rows, the x-halo column and
  float* ph0n; float* ph1n;             // chunk prologues of OTHER workgroups re-read the old values, so the
                                        // H-side psi is ping-ponged like the fields) / WRITE set
  float* pe0n; float* pe1n;             // WRITE set of the E-side psi: pe0 / pe1 themselves (updated in place) for the single-step
                                        // kernels; the other set for shell2_step_kernel, whose halo rows re-read the old values
  int lo, hi0, ns, n;
};
struct PmlP { PmlAxisP ax[3]; };

__device__ __forceinline__ int pml_si(const PmlAxisP& A, int ia) {
  if (ia < A.lo) return ia;
  if (ia >= A.hi0) return A.lo + (ia - A.hi0);
  return -1;
}
__device__ __forceinline__ long long pml_q(const GridP& g, int a, int ns, int i, int j, int k, int si) {
  if (a == 0) return ((long long)k * g.ny + j) * ns + si;
  if (a == 1) return ((long long)k * ns + si) * g.nx + i;
  return ((long long)si * g.ny + j) * g.nx + i;
}
// one cell, one axis, read-only psi:  h1 += ch (kv d2 + p1),  h2 -= ch (kv d1 + p2)
__device__ __forceinline__ void pml_h_cell(float& h1, float& h2, float d1, float d2, const PmlAxisP& A,
                                           long long q, int ia, float ch) {
  const float4 cf = ldc_f4(A.ch4 + ia);
  const float p1 = cf.y * ldg1(A.ph0 + q) + cf.z * d2;
  const float p2 = cf.y * ldg1(A.ph1 + q) + cf.z * d1;
  h1 += ch * (cf.x * d2 + p1);
  h2 -= ch * (cf.x * d1 + p2);
}
// the recursion on psi values in registers (s1, s2 are replaced by their new values)
__device__ __forceinline__ void pml_h_apply(float& h1, float& h2, float d1, float d2, float& s1, float& s2,
                                            float kv, float b, float c, float ch) {
  const float p1 = b * s1 + c * d2;
  const float p2 = b * s2 + c * d1;
  s1 = p1; s2 = p2;
  h1 += ch * (kv * d2 + p1);
  h2 -= ch * (kv * d1 + p2);
}

// The sweep is compiled for 3 waves per SIMD (<= 168 VGPRs): the measured optimum of the plain
// sweep, and the budget the CPML-carrying instantiations are written to stay inside.
#if defined(__HIPCC__)
#define FDTD_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#else
#define FDTD_WAVES_PER_EU(lo, hi)
#endif

// =============================================================================================
// K1+K2 fused: one sweep advances H by curl E AND E by curl of the *new* H, reading set `a`
// (E^n, H^{n-1/2}) and writing set `b` (E^{n+1}, H^{n+1/2})  — 6 reads + 6 writes = 48 B per
// cell-step of HBM traffic instead of 72 B for the two-pass form.  Ping-pong buffers make the
// sweep race-free; every additive correction (CPML, sources, ADE) commutes with it: H-side
// corrections are pre-applied to H^{n-1/2} in `a`, E-side ones post-applied to E^{n+1} in `b`.
//
// Workgroup = (64 lanes x 4 cells) x (R rows + 1 halo row below); marches `zchunk` planes.
//   * H^{n+1/2}[k] is computed in registers by every wave (the halo wave recomputes H_x, H_z of
//     row j0-1);
//   * E^{n+1}[k] needs H^{n+1/2} at x-1 (lane-1 via __shfl_up; the lane at the tile edge
//     recomputes the x-halo column itself), at y-1 (previous wave via a double-buffered LDS
//     slot, one barrier per plane) and at z-1 (registers carried from the previous plane; the
//     first plane of a chunk recomputes H^{n+1/2}[k0-1] in a prologue).
// Everything that depends on the row only (threadIdx.y is wave-uniform) is kept in SGPRs.
// =============================================================================================
template <bool MAT, int LB, int PML, int HINT = 0>   // PML: bit a set = CPML of axis a runs inside the sweep; HINT: bit 0 = non-temporal field stores, bit 1 = non-temporal loads of E_y, H_y (measured: +0.7 %, not instantiated), bit 3 = H stores ahead of the row exchange
__global__ __launch_bounds__(LB, (LB == 256 ? (PML == 0 ? 4 : (PML == 1 ? 3 : 2)) : (LB == 512 ? (PML ? 2 : 4) : 4))) void fused_step_kernel(GridP g, FieldP a, FieldP b, StepP s, MatP m,
                                                          int kbeg, int kend, int zchunk, int pmc_z0,
                                                          int nbx, int nby, int nbz, int xcd_remap,
                                                          const PmlP* __restrict__ pmq,
                                                          int nbz1, int k2beg, int k2end, int ty_a, int ty_gap,
                                                          int ex_j0, int ex_j1) {
  constexpr int V = 4;
  // 1-D launch; logical tile (bx, by, bz) with by fastest.  Hardware block L runs on XCD L % 8 (observed dispatch
  // order, used for speed only); y-neighbouring tiles share their halo row, and meet in an XCD's L2 when they are
  // resident there at about the same time.  xcd_remap = 0: plain order (neighbours on different XCDs);
  // 1: XCD x gets the contiguous range [x * per, (x+1) * per) of tiles; G > 1: see below (the default, G = 8).
  const int total = nbx * nby * nbz;
  int t = blockIdx.x;
  if (xcd_remap == 1) {
    const int per = (total + 7) >> 3;
    t = (t & 7) * per + (t >> 3);
    if (t >= total) return;              // whole workgroup leaves before any barrier
  } else if (xcd_remap > 1) {
    // grouped order: runs of G = xcd_remap consecutive tiles (y-neighbours) go to one XCD, the runs round-robin over
    // the XCDs — the halo rows inside a run still meet in that XCD's L2, while all eight XCDs work in the same
    // few z-chunks (the contiguous split above has them stream eight distant regions of every array at once)
    const int G = xcd_remap, full = total / (8 * G) * (8 * G);
    if (t < full) {
      const int x = t & 7, mloc = t >> 3;
      t = ((mloc / G) * 8 + x) * G + mloc % G;
    }
    if (t >= total) return;
  }
  // tile rows of this launch: the first ty_a, then (after a gap of ty_gap) the rest — the launch that
  // carries the y / z recursions covers the bottom and top tile rows only, a leaner one the middle
  int tile_y = t % nby;
  if (tile_y >= ty_a) tile_y += ty_gap;
  const int tile_x = (t / nby) % nbx;
  const int tile_z = t / (nby * nbx);
  __shared__ float2 lut_s[MAT ? kMaxMedia : 1];
  HIP_DYNAMIC_SHARED(float4, xch)      // [2 buffers][2 comps][blockDim.y][64] (+ [6][64] x-CPML coefficients)
  if constexpr (MAT) {
    for (int q = threadIdx.y * blockDim.x + threadIdx.x; q < m.n_media; q += blockDim.x * blockDim.y)
      lut_s[q] = m.lut[q];
  }
  // x-CPML coefficients of the tile's 256 cells {kv_h, b_h, c_h, kv_e, b_e, c_e}: read from LDS where they
  // are used (24 VGPRs if held, a global round trip if fetched on the spot)
  [[maybe_unused]] float4* xco = xch + 2 * 2 * blockDim.y * 64;
  if constexpr ((PML & 1) != 0) {
    const PmlAxisP& A = pmq->ax[0];
    const unsigned o = (unsigned)((tile_x * 64 + threadIdx.x) * V) * 4u;
    for (int q = threadIdx.y; q < 6; q += blockDim.y) {
      const float* src = q == 0 ? A.kv_h : (q == 1 ? A.b_h : (q == 2 ? A.c_h : (q == 3 ? A.kv_e : (q == 4 ? A.b_e : A.c_e))));
      float r[V] = {0.f, 0.f, 0.f, 0.f};
      if ((int)(o / 4u) < g.nx && (A.lo > 0 || A.hi0 < A.n)) ldc4(r, src, o);   // (no tables on an axis without members)
      float4 t4; t4.x = r[0]; t4.y = r[1]; t4.z = r[2]; t4.w = r[3];
      xco[q * 64 + threadIdx.x] = t4;
    }
  }
  if constexpr (MAT || (PML & 1) != 0) __syncthreads();
  const int tx = threadIdx.x;
  const int ty = __builtin_amdgcn_readfirstlane((int)threadIdx.y);     // one row per wave
  const int R = blockDim.y - 1;
  const int i0 = (tile_x * 64 + tx) * V;
  const unsigned ux = (unsigned)i0;
  const unsigned ub = ux * 4u;            // lane's byte offset along the row: every row access is  uniform base + 32-bit lane offset
  const unsigned ubc = (i0 < g.nx) ? ub : 0u;   // the same, clamped into the row for loads of idle lanes
  const bool per_x = g.bcx0 == BC_PERIODIC, per_y = g.bcy0 == BC_PERIODIC;
  int j = tile_y * R + ty - 1;
  // rows [ex_j0, ex_j1) belong to another launch (the y slabs of a shell step leave the rows of the bulk alone): a wave on
  // such a row acts as a halo wave — it publishes H_x, H_z for the row above and stores nothing
  const bool halo = (ty == 0) || (j >= ex_j0 && j < ex_j1);
  bool row_ok = (j >= 0) && (j < g.ny);
  if (j < 0 && per_y) { j = g.ny - 1; row_ok = true; }
  if (!row_ok) j = 0;                    // keeps every address of an idle wave inside the arrays
  const bool act = row_ok && (i0 < g.nx);
  // z tiles [0, nbz1) march through [kbeg, kend), tiles [nbz1, nbz) through a second plane range
  // [k2beg, k2end): the bottom and top boundary chunks of a z-slab go out as ONE launch
  const bool second = tile_z >= nbz1;
  const int k0 = second ? k2beg + (tile_z - nbz1) * zchunk : kbeg + tile_z * zchunk;
  const int k1 = min(k0 + zchunk, second ? k2end : kend);
  const float ch = g.ch;
  const bool last_x = (i0 + V >= g.nx);
  const bool first_x = (i0 == 0);
  const bool use_jp = (j + 1 < g.ny) || (g.bcy1 == BC_PERIODIC);
  const long long rowb = (long long)j * g.nx;                               // scalar row bases
  const long long rowpb = (j + 1 < g.ny) ? rowb + g.nx : 0;
  // x-halo column handled by the first lane of the tile (or the wrapped column for periodic x)
  const bool xh = act && (tx == 0) && (!first_x || per_x);
  const int im = first_x ? g.nx - 1 : i0 - 1;
  const bool wall_y = (j == 0) && (g.bcy0 == BC_PEC);
  const bool wall_x0 = first_x && (g.bcx0 == BC_PEC);

  float ipx[V], idx[V];
  zero<V>(ipx); zero<V>(idx);
  float ipy = 0.f, idy = 0.f, ipx_m = 0.f;
  if (act) {
    ldv<V>(ipx, at(uni(s.ipx), ub));
    ldv<V>(idx, at(uni(s.idx), ub));
    ipx_m = s.ipx[im];
  }
  if (row_ok) { ipy = s.ipy[j]; idy = s.idy[j]; }
  // CPML membership that does not change along the march (x: per lane, y: per row)
  [[maybe_unused]] int sx = -1, sx_m = -1, sy = -1;
  [[maybe_unused]] unsigned sxb = 0;      // lane's byte offset into an x-slab psi row
  if constexpr ((PML & 1) != 0) {
    if (act) sx = pml_si(pmq->ax[0], i0);
    sxb = (unsigned)max(sx, 0) * 4u;
    if (xh) sx_m = pml_si(pmq->ax[0], im);
  }
  [[maybe_unused]] float4 cyh = {0.f, 0.f, 0.f, 0.f}, cye = {0.f, 0.f, 0.f, 0.f};
  if constexpr ((PML & 2) != 0) {
    if (row_ok) sy = pml_si(pmq->ax[1], j);
    if (sy >= 0) { cyh = ldc_f4(pmq->ax[1].ch4 + j); cye = ldc_f4(pmq->ax[1].ce4 + j); }
  }

  float exk[V], eyk[V], hxm[V], hym[V];
  zero<V>(hxm); zero<V>(hym);
  float exk_m = 0.f;
  {
    const long long p0 = (long long)k0 * g.sxy + rowb;
    ldf<V, true>(exk, uni(a.ex + p0), ubc);
    ldf<V, true>(eyk, uni(a.ey + p0), ubc);
    if (xh) exk_m = a.ex[p0 + im];
  }
  // ---- prologue: H^{n+1/2}[k0-1] (x, y components) of the own cells --------------------------
  if (!halo) {
    const bool skip = (pmc_z0 && k0 == 0);
    const long long pb = (long long)(k0 - 1) * g.sxy + rowb;
    float ezm[V], ezj[V], exm[V], eym[V], ho[V], hoy[V];
    zero<V>(ezm); zero<V>(ezj); zero<V>(exm); zero<V>(eym);
    if (act && !skip) {
      ldf<V, true>(ezm, uni(a.ez + pb), ub);
      ldf<V, true>(exm, uni(a.ex + pb), ub);
      ldf<V, true>(eym, uni(a.ey + pb), ub);
      if (use_jp) ldf<V, true>(ezj, uni(a.ez + (long long)(k0 - 1) * g.sxy + rowpb), ub);
    }
    float ezx = __shfl_down(ezm[0], 1);
    if (act && !skip) {
      if (tx == 63 || last_x) {
        if (!last_x) ezx = a.ez[pb + ux + V];
        else if (g.bcx1 == BC_PERIODIC) ezx = a.ez[pb];
        else ezx = 0.f;
      }
      const float ipz = s.ipz[k0 - 1];
      ldf<V, true>(ho, uni(a.hx + pb), ub);
      ldf<V, true>(hoy, uni(a.hy + pb), ub);
      if constexpr (PML != 0) {
        // corrected H^{n-1/2}_{x,y} of plane k0-1 (read-only psi; the plane's owner stores it)
        // periodic z on one GPU: the ghost plane is the top plane; on a z-slab: the lower neighbour's top plane, whose psi
        // arrives with the ghost fields in an extra plane (slot nz) of the x / y psi arrays
        const int kk = (k0 - 1 < 0) ? (g.psi_ghost ? g.nz : g.nz - 1) : k0 - 1;
        if (k0 > 0 || !g.pec_z0) {
          // axis x: Hy += ch (kv dEz/dx + p1)
          if constexpr ((PML & 1) != 0) {
            if (sx >= 0) {
              const PmlAxisP& A = pmq->ax[0];
              float s1[V];
              ldg4(s1, uni(A.ph0 + ((long long)kk * g.ny + j) * A.ns), sxb);
              const float4 kv4 = xco[0 * 64 + tx], bb4 = xco[1 * 64 + tx], cc4 = xco[2 * 64 + tx];
              const float kv[V] = {kv4.x, kv4.y, kv4.z, kv4.w}, bb[V] = {bb4.x, bb4.y, bb4.z, bb4.w},
                          cc[V] = {cc4.x, cc4.y, cc4.z, cc4.w};
#pragma unroll
              for (int e = 0; e < V; ++e) {
                const float ez_ip = (e + 1 < V) ? ezm[(e + 1) % V] : ezx;
                const float d2 = (ez_ip - ezm[e]) * ipx[e];
                const float p1 = bb[e] * s1[e] + cc[e] * d2;
                hoy[e] += ch * (kv[e] * d2 + p1);
              }
            }
          }
          // axis y: Hx -= ch (kv dEz/dy + p2)
          if constexpr ((PML & 2) != 0) {
            if (sy >= 0) {
              const PmlAxisP& A = pmq->ax[1];
              const float4 cf = cyh;
              float s2[V];
              ldg4(s2, uni(A.ph1 + ((long long)kk * A.ns + sy) * g.nx), ub);
#pragma unroll
              for (int e = 0; e < V; ++e) {
                const float d1 = (ezj[e] - ezm[e]) * ipy;
                const float p2 = cf.y * s2[e] + cf.z * d1;
                ho[e] -= ch * (cf.x * d1 + p2);
              }
            }
          }
          // axis z: Hx += ch (kv dEy/dz + p1), Hy -= ch (kv dEx/dz + p2)   (only inside this slab)
          if constexpr ((PML & 4) != 0) {
            const PmlAxisP& A = pmq->ax[2];
            const int sz = (k0 - 1 >= 0) ? pml_si(A, k0 - 1) : -1;
            if (sz >= 0) {
              const float4 cf = ldc_f4(A.ch4 + k0 - 1);
              float s1[V], s2[V];
              const long long q = ((long long)sz * g.ny + j) * g.nx;           // uniform part of the psi index
              ldg4(s1, uni(A.ph0 + q), ub);
              ldg4(s2, uni(A.ph1 + q), ub);
#pragma unroll
              for (int e = 0; e < V; ++e)
                pml_h_apply(ho[e], hoy[e], (exk[e] - exm[e]) * ipz, (eyk[e] - eym[e]) * ipz, s1[e], s2[e],
                            cf.x, cf.y, cf.z, ch);
            }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < V; ++e) hxm[e] = upd_h(ho[e], ch, ezj[e] - ezm[e], ipy, eyk[e] - eym[e], ipz);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float ez_ip = (e + 1 < V) ? ezm[(e + 1) % V] : ezx;
        hym[e] = upd_h(hoy[e], ch, exk[e] - exm[e], ipz, ez_ip - ezm[e], ipx[e]);
      }
    }
  }                                      // (the halo wave needs no H^{n+1/2}[k0-1])
  int cur = 0;
  const int slot = (int)blockDim.y * 64;           // float4 entries per component per buffer
  // HINT bit 8 (256): the E values of a plane are stored one H phase later, behind the loads of the next plane
  [[maybe_unused]] float pend_ex[V], pend_ey[V], pend_ez[V];
  [[maybe_unused]] long long pend_p = -1;
  for (int k = k0; k < k1; ++k) {
    const long long pb = (long long)k * g.sxy + rowb;      // scalar
    const long long pjb = (long long)k * g.sxy + rowpb;
    // Field loads of the plane: unconditional.  Lanes beyond the row end read the row's first cells instead
    // (ubc), waves beyond the grid read row 0 (j): same cache lines as their neighbours, no bandwidth, no
    // lane predicate, no per-plane zero-fill of 32 registers; nothing such a lane computes is ever stored.
    float exn[V], eyn[V], ezk[V], exj[V], ezj[V], hxn[V], hyn[V], hzn[V];
    const float ipz = s.ipz[k], idz = s.idz[k];
    ldf<V, true>(exn, uni(a.ex + pb + g.sxy), ubc);
    ldv_h<V, (HINT & 2) != 0>(eyn, uni(a.ey + pb + g.sxy), ubc);
    ldf<V, true>(ezk, uni(a.ez + pb), ubc);
    if (use_jp) {
      ldf<V, true>(exj, uni(a.ex + pjb), ubc);
      ldf<V, true>(ezj, uni(a.ez + pjb), ubc);
    } else {
      zero<V>(exj); zero<V>(ezj);                // row j+1 beyond a wall: E = 0 there
    }
    ldf<V, true>(hxn, uni(a.hx + pb), ubc);
    if (!halo) ldv_h<V, (HINT & 2) != 0>(hyn, uni(a.hy + pb), ubc);      // the halo wave only publishes H_x and H_z
    else zero<V>(hyn);
    ldf<V, true>(hzn, uni(a.hz + pb), ubc);
    // ---- CPML state of this plane: EVERY psi load is issued here, with the field loads, so that ONE memory
    // round trip per plane covers them.  Loaded where they are used they chain two more round trips per plane
    // (H side, then E side behind the barrier): +0.63 ms per 512^3 step, worse than the slab kernels
    // (profiles/r02a_probe_shapes_pml.jsonl).
    [[maybe_unused]] float xh1[V], xh2[V], xe1[V], xe2[V], yh1[V], yh2[V], ye1[V], ye2[V], zh1[V], zh2[V], ze1[V], ze2[V];
    // uniform parts of the psi indices (formed outside the lane predicate: they must stay wave-uniform)
    [[maybe_unused]] long long qx = 0, qy = 0, qz = 0;
    [[maybe_unused]] int sz = -1;
    [[maybe_unused]] float4 czh = {0.f, 0.f, 0.f, 0.f}, cze = {0.f, 0.f, 0.f, 0.f};
    const bool wall_z = (k == 0) && g.pec_z0;
    if constexpr ((PML & 4) != 0) {
      sz = pml_si(pmq->ax[2], k);                                 // outside the lane predicate: stays wave-uniform
      if (sz >= 0) { czh = ldc_f4(pmq->ax[2].ch4 + k); cze = ldc_f4(pmq->ax[2].ce4 + k); }
      qz = ((long long)max(sz, 0) * g.ny + j) * g.nx;
    }
    if constexpr ((PML & 1) != 0) qx = ((long long)k * g.ny + j) * pmq->ax[0].ns;
    if constexpr ((PML & 2) != 0) qy = ((long long)k * pmq->ax[1].ns + max(sy, 0)) * g.nx;
    if constexpr (PML != 0) {
      if (act) {
        if constexpr ((PML & 1) != 0) {
          if (sx >= 0) {
            const PmlAxisP& A = pmq->ax[0];
            ldg4(xh1, uni(A.ph0 + qx), sxb);
            ldg4(xh2, uni(A.ph1 + qx), sxb);
            if (!halo) { ldg4(xe1, uni(A.pe0 + qx), sxb); ldg4(xe2, uni(A.pe1 + qx), sxb); }
          }
        }
        if constexpr ((PML & 2) != 0) {
          if (sy >= 0) {
            const PmlAxisP& A = pmq->ax[1];
            ldg4(yh1, uni(A.ph0 + qy), ub);
            ldg4(yh2, uni(A.ph1 + qy), ub);
            if (!halo && !wall_y) { ldg4(ye1, uni(A.pe0 + qy), ub); ldg4(ye2, uni(A.pe1 + qy), ub); }
          }
        }
        if constexpr ((PML & 4) != 0) {
          if (sz >= 0) {
            const PmlAxisP& A = pmq->ax[2];
            ldg4(zh1, uni(A.ph0 + qz), ub);
            ldg4(zh2, uni(A.ph1 + qz), ub);
            if (!halo && !wall_z) { ldg4(ze1, uni(A.pe0 + qz), ub); ldg4(ze2, uni(A.pe1 + qz), ub); }
          }
        }
      }
    }
    // material row-segment word of this plane (scalar load) and, where the segment is mixed, the packed words
    [[maybe_unused]] uint32_t rw = kBgWord;
    [[maybe_unused]] uint32_t mw[V] = {kBgWord, kBgWord, kBgWord, kBgWord};
    if constexpr (MAT) {
      if (!halo && row_ok) {
        rw = m.roww[((long long)k * g.ny + j) * nbx + tile_x];
        if (rw == kMixedWord && act) ldm<V>(mw, at(uni(m.m4 + pb), ub));
      }
    }
    float eyx = __shfl_down(eyk[0], 1);
    float ezx = __shfl_down(ezk[0], 1);
    if (act && (tx == 63 || last_x)) {
      if (!last_x) { eyx = a.ey[pb + ux + V]; ezx = a.ez[pb + ux + V]; }
      else if (g.bcx1 == BC_PERIODIC) { eyx = a.ey[pb]; ezx = a.ez[pb]; }
      else { eyx = 0.f; ezx = 0.f; }
    }
    // ---- H-side CPML: pre-corrections of H^{n-1/2}, axes in the order x, y, z ----------------
    if constexpr (PML != 0) {
      if (act) {
        // axis x: Hy += ch (kv dEz/dx + p1), Hz -= ch (kv dEy/dx + p2)
        if constexpr ((PML & 1) != 0) {
          if (sx >= 0) {
            const PmlAxisP& A = pmq->ax[0];
            const float4 kv4 = xco[0 * 64 + tx], bb4 = xco[1 * 64 + tx], cc4 = xco[2 * 64 + tx];
            const float kv[V] = {kv4.x, kv4.y, kv4.z, kv4.w}, bb[V] = {bb4.x, bb4.y, bb4.z, bb4.w},
                        cc[V] = {cc4.x, cc4.y, cc4.z, cc4.w};
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const float ey_ip = (e + 1 < V) ? eyk[(e + 1) % V] : eyx;
              const float ez_ip = (e + 1 < V) ? ezk[(e + 1) % V] : ezx;
              pml_h_apply(hyn[e], hzn[e], (ey_ip - eyk[e]) * ipx[e], (ez_ip - ezk[e]) * ipx[e], xh1[e], xh2[e],
                          kv[e], bb[e], cc[e], ch);
            }
            if constexpr ((HINT & 128) == 0) { if (!halo) { stg4(uni(A.ph0n + qx), sxb, xh1); stg4(uni(A.ph1n + qx), sxb, xh2); } }
          }
        }
        // axis y: Hz += ch (kv dEx/dy + p1), Hx -= ch (kv dEz/dy + p2)
        if constexpr ((PML & 2) != 0) {
          if (sy >= 0) {
            const PmlAxisP& A = pmq->ax[1];
#pragma unroll
            for (int e = 0; e < V; ++e)
              pml_h_apply(hzn[e], hxn[e], (ezj[e] - ezk[e]) * ipy, (exj[e] - exk[e]) * ipy, yh1[e], yh2[e],
                          cyh.x, cyh.y, cyh.z, ch);
            if constexpr ((HINT & 128) == 0) { if (!halo) { stg4(uni(A.ph0n + qy), ub, yh1); stg4(uni(A.ph1n + qy), ub, yh2); } }
          }
        }
        // axis z: Hx += ch (kv dEy/dz + p1), Hy -= ch (kv dEx/dz + p2)
        if constexpr ((PML & 4) != 0) {
          if (sz >= 0) {
            const PmlAxisP& A = pmq->ax[2];
#pragma unroll
            for (int e = 0; e < V; ++e)
              pml_h_apply(hxn[e], hyn[e], (exn[e] - exk[e]) * ipz, (eyn[e] - eyk[e]) * ipz, zh1[e], zh2[e],
                          czh.x, czh.y, czh.z, ch);
            if constexpr ((HINT & 128) == 0) { if (!halo) { stg4(uni(A.ph0n + qz), ub, zh1); stg4(uni(A.ph1n + qz), ub, zh2); } }
          }
        }
      }
    }
    if (act) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float ey_ip = (e + 1 < V) ? eyk[(e + 1) % V] : eyx;
        const float ez_ip = (e + 1 < V) ? ezk[(e + 1) % V] : ezx;
        hxn[e] = upd_h(hxn[e], ch, ezj[e] - ezk[e], ipy, eyn[e] - eyk[e], ipz);
        if (!halo) hyn[e] = upd_h(hyn[e], ch, exn[e] - exk[e], ipz, ez_ip - ezk[e], ipx[e]);
        hzn[e] = upd_h(hzn[e], ch, ey_ip - eyk[e], ipx[e], exj[e] - exk[e], ipy);
      }
    }
    // x-halo column: H^{n+1/2}_{y,z} at i0-1 recomputed by the tile's first lane
    float hy_m = 0.f, hz_m = 0.f, exn_m = 0.f;
    if (xh && !halo) {
      const long long pm = pb + im;
      exn_m = a.ex[pm + g.sxy];
      const float ez_mm = a.ez[pm], ey_mm = a.ey[pm];
      const float ex_jm = use_jp ? a.ex[pjb + im] : 0.f;
      float hy_o = a.hy[pm], hz_o = a.hz[pm];
      if constexpr (PML != 0) {
        float dum = 0.f;
        if constexpr ((PML & 1) != 0) {                 // axis x: Hy += .., Hz -= ..
          if (sx_m >= 0)
            pml_h_cell(hy_o, hz_o, (eyk[0] - ey_mm) * ipx_m, (ezk[0] - ez_mm) * ipx_m, pmq->ax[0],
                       pml_q(g, 0, pmq->ax[0].ns, im, j, k, sx_m), im, ch); }
        if constexpr ((PML & 2) != 0) {                 // axis y: Hz += ch (kv dEx/dy + p1)
          if (sy >= 0)
            pml_h_cell(hz_o, dum, 0.f, (ex_jm - exk_m) * ipy, pmq->ax[1],
                       pml_q(g, 1, pmq->ax[1].ns, im, j, k, sy), j, ch); }
        if constexpr ((PML & 4) != 0) {                 // axis z: Hy -= ch (kv dEx/dz + p2)
          if (sz >= 0)
            pml_h_cell(dum, hy_o, (exn_m - exk_m) * ipz, 0.f, pmq->ax[2],
                       pml_q(g, 2, pmq->ax[2].ns, im, j, k, sz), k, ch); }
      }
      hy_m = upd_h(hy_o, ch, exn_m - exk_m, ipz, ezk[0] - ez_mm, ipx_m);
      hz_m = upd_h(hz_o, ch, eyk[0] - ey_mm, ipx_m, ex_jm - exk_m, ipy);
    }
    if constexpr ((HINT & 256) != 0) {
      if (act && !halo && pend_p >= 0) {
        stv_h<V, (HINT & 1) != 0>(b.ex + pend_p + i0, pend_ex);
        stv_h<V, (HINT & 1) != 0>(b.ey + pend_p + i0, pend_ey);
        stv_h<V, (HINT & 1) != 0>(b.ez + pend_p + i0, pend_ez);
      }
    }
    if constexpr ((HINT & 8) != 0) {      // H stores ahead of the exchange: their completion overlaps the E phase
      if (act && !halo) {
        stv_h<V, (HINT & 1) != 0>(b.hx + pb + i0, hxn);
        stv_h<V, (HINT & 1) != 0>(b.hy + pb + i0, hyn);
        stv_h<V, (HINT & 1) != 0>(b.hz + pb + i0, hzn);
      }
    }
    // publish H^{n+1/2}_{x,z} of this row for the row above
    {
      float4 t4;
      t4.x = hxn[0]; t4.y = hxn[1]; t4.z = hxn[2]; t4.w = hxn[3];
      xch[(cur * 2 + 0) * slot + ty * 64 + tx] = t4;
      t4.x = hzn[0]; t4.y = hzn[1]; t4.z = hzn[2]; t4.w = hzn[3];
      xch[(cur * 2 + 1) * slot + ty * 64 + tx] = t4;
    }
    __syncthreads();
    float hyx = __shfl_up(hyn[V - 1], 1);
    float hzx = __shfl_up(hzn[V - 1], 1);
    if (pmc_z0 && k == 0) {
#pragma unroll
      for (int e = 0; e < V; ++e) { hxm[e] = -hxn[e]; hym[e] = -hyn[e]; }
    }
    // E^{n+1} of the own rows.  `coef(c, e)` yields (Ca, Cb) of component c of the lane's e-th cell: scalars
    // for a uniform medium, per-cell look-ups in the LDS table otherwise, fetched where they are used so
    // that no coefficient stays live across the update (the packed words are all that is held).
    auto e_phase = [&](auto coef) {
      if (tx == 0 || first_x) {
        if (xh) { hyx = hy_m; hzx = hz_m; }
        else if (g.bcx0 == BC_PMC) { hyx = -hyn[0]; hzx = -hzn[0]; }
        else { hyx = 0.f; hzx = 0.f; }
      }
      float hxj[V], hzj[V];
      if (j > 0 || per_y) {
        const float4 t0 = xch[(cur * 2 + 0) * slot + (ty - 1) * 64 + tx];
        const float4 t1 = xch[(cur * 2 + 1) * slot + (ty - 1) * 64 + tx];
        hxj[0] = t0.x; hxj[1] = t0.y; hxj[2] = t0.z; hxj[3] = t0.w;
        hzj[0] = t1.x; hzj[1] = t1.y; hzj[2] = t1.z; hzj[3] = t1.w;
      } else if (g.bcy0 == BC_PMC) {
#pragma unroll
        for (int e = 0; e < V; ++e) { hxj[e] = -hxn[e]; hzj[e] = -hzn[e]; }
      } else {
        zero<V>(hxj); zero<V>(hzj);
      }
      float ex[V], ey[V], ez[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float hy_im = (e > 0) ? hyn[(e + V - 1) % V] : hyx;
        const float hz_im = (e > 0) ? hzn[(e + V - 1) % V] : hzx;
        float nex = upd_e(exk[e], coef(0, e).x, coef(0, e).y, hzn[e] - hzj[e], idy, hyn[e] - hym[e], idz);
        float ney = upd_e(eyk[e], coef(1, e).x, coef(1, e).y, hxn[e] - hxm[e], idz, hzn[e] - hz_im, idx[e]);
        float nez = upd_e(ezk[e], coef(2, e).x, coef(2, e).y, hyn[e] - hy_im, idx[e], hxn[e] - hxj[e], idy);
        const bool wx = wall_x0 && (e == 0);
        if (wall_y || wall_z) nex = 0.f;
        if (wx || wall_z) ney = 0.f;
        if (wx || wall_y) nez = 0.f;
        ex[e] = nex; ey[e] = ney; ez[e] = nez;
      }
      if constexpr (PML != 0) {
        // E-side CPML, axes in the order y, z, x (= launch_pml's E-side order):
        //   E_{a+1} -= cb (kv d2 + p1),  E_{a+2} += cb (kv d1 + p2),  d1 = d_a H_{a+1}, d2 = d_a H_{a+2}
        // A cell on the min wall of the PML axis itself keeps its psi untouched (both components
        // are wall-tangential there), exactly like the slab kernels.
        // axis y: E_z -= cb (kv dHx/dy + p1),  E_x += cb (kv dHz/dy + p2)
        if constexpr ((PML & 2) != 0) {
          if (sy >= 0 && !wall_y) {
            const PmlAxisP& A = pmq->ax[1];
            const float4 cf = cye;
            float (&s1)[V] = ye1, (&s2)[V] = ye2;
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const float d1 = (hzn[e] - hzj[e]) * idy;
              const float d2 = (hxn[e] - hxj[e]) * idy;
              const float p1 = cf.y * s1[e] + cf.z * d2;
              const float p2 = cf.y * s2[e] + cf.z * d1;
              s1[e] = p1; s2[e] = p2;
              const bool wx = wall_x0 && (e == 0);
              if (!wx) ez[e] -= coef(2, e).y * (cf.x * d2 + p1);       // E_z is tangential to the x wall
              if (!wall_z) ex[e] += coef(0, e).y * (cf.x * d1 + p2);   // E_x is tangential to the z wall
            }
            stg4(uni(A.pe0n + qy), ub, s1);       // (pe0n == pe0 for a step on its own: in place; the middle steps of a pair write another set)
            stg4(uni(A.pe1n + qy), ub, s2);
          }
        }
        // axis z: E_x -= cb (kv dHy/dz + p1),  E_y += cb (kv dHx/dz + p2)
        if constexpr ((PML & 4) != 0) {
          if (sz >= 0 && !wall_z) {
            const PmlAxisP& A = pmq->ax[2];
            const float4 cf = cze;
            float (&s1)[V] = ze1, (&s2)[V] = ze2;
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const float d1 = (hxn[e] - hxm[e]) * idz;
              const float d2 = (hyn[e] - hym[e]) * idz;
              const float p1 = cf.y * s1[e] + cf.z * d2;
              const float p2 = cf.y * s2[e] + cf.z * d1;
              s1[e] = p1; s2[e] = p2;
              const bool wx = wall_x0 && (e == 0);
              if (!wall_y) ex[e] -= coef(0, e).y * (cf.x * d2 + p1);   // E_x is tangential to the y wall
              if (!wx) ey[e] += coef(1, e).y * (cf.x * d1 + p2);       // E_y is tangential to the x wall
            }
            stg4(uni(A.pe0n + qz), ub, s1);
            stg4(uni(A.pe1n + qz), ub, s2);
          }
        }
        // axis x: E_y -= cb (kv dHz/dx + p1),  E_z += cb (kv dHy/dx + p2)
        if constexpr ((PML & 1) != 0) {
          if (sx >= 0) {
            const PmlAxisP& A = pmq->ax[0];
            float (&s1)[V] = xe1, (&s2)[V] = xe2;
            const float4 kv4 = xco[3 * 64 + tx], bb4 = xco[4 * 64 + tx], cc4 = xco[5 * 64 + tx];
            const float kv[V] = {kv4.x, kv4.y, kv4.z, kv4.w}, bb[V] = {bb4.x, bb4.y, bb4.z, bb4.w},
                        cc[V] = {cc4.x, cc4.y, cc4.z, cc4.w};
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const bool wx = wall_x0 && (e == 0);
              if (!wx) {
                const float hy_im = (e > 0) ? hyn[(e + V - 1) % V] : hyx;
                const float hz_im = (e > 0) ? hzn[(e + V - 1) % V] : hzx;
                const float d1 = (hyn[e] - hy_im) * idx[e];
                const float d2 = (hzn[e] - hz_im) * idx[e];
                const float p1 = bb[e] * s1[e] + cc[e] * d2;
                const float p2 = bb[e] * s2[e] + cc[e] * d1;
                s1[e] = p1; s2[e] = p2;
                if (!wall_z) ey[e] -= coef(1, e).y * (kv[e] * d2 + p1);     // E_y is tangential to the z wall
                if (!wall_y) ez[e] += coef(2, e).y * (kv[e] * d1 + p2);     // E_z is tangential to the y wall
              }
            }
            stg4(uni(A.pe0n + qx), sxb, s1);
            stg4(uni(A.pe1n + qx), sxb, s2);
          }
        }
      }
      // HINT bit 7: the H-side psi of this plane is stored HERE, behind the E update, not in the H phase in front of the
      // row exchange (the values wait in registers): a store issued before the barrier costs the CPML instantiations
      // 2.4 ... 3 % of the whole step (profiles/r04i), as the H field stores do (+12 %, r03k)
      if constexpr (PML != 0 && (HINT & 128) != 0) {
        if constexpr ((PML & 1) != 0) {
          if (sx >= 0) { const PmlAxisP& A = pmq->ax[0]; stg4(uni(A.ph0n + qx), sxb, xh1); stg4(uni(A.ph1n + qx), sxb, xh2); }
        }
        if constexpr ((PML & 2) != 0) {
          if (sy >= 0) { const PmlAxisP& A = pmq->ax[1]; stg4(uni(A.ph0n + qy), ub, yh1); stg4(uni(A.ph1n + qy), ub, yh2); }
        }
        if constexpr ((PML & 4) != 0) {
          if (sz >= 0) { const PmlAxisP& A = pmq->ax[2]; stg4(uni(A.ph0n + qz), ub, zh1); stg4(uni(A.ph1n + qz), ub, zh2); }
        }
      }
      if constexpr ((HINT & 8) == 0) {
        stv_h<V, (HINT & 1) != 0>(b.hx + pb + i0, hxn);
        stv_h<V, (HINT & 1) != 0>(b.hy + pb + i0, hyn);
        stv_h<V, (HINT & 1) != 0>(b.hz + pb + i0, hzn);
      }
      if constexpr ((HINT & 256) != 0) {
#pragma unroll
        for (int e = 0; e < V; ++e) { pend_ex[e] = ex[e]; pend_ey[e] = ey[e]; pend_ez[e] = ez[e]; }
        pend_p = pb;
      } else {
        stv_h<V, (HINT & 1) != 0>(b.ex + pb + i0, ex);
        stv_h<V, (HINT & 1) != 0>(b.ey + pb + i0, ey);
        stv_h<V, (HINT & 1) != 0>(b.ez + pb + i0, ez);
      }
    };
    if (act && !halo) {
      if constexpr (MAT) {
        // one word per row segment (256 cells of one row): the medium word of all its cells when they
        // agree — then the packed words are not read at all — or kMixedWord (both fetched at the top of the plane)
        if (rw != kMixedWord) {
          const float2 c0 = lut_s[rw & 1023u], c1 = lut_s[(rw >> 10) & 1023u], c2 = lut_s[(rw >> 20) & 1023u];
          e_phase([&](int c, int) { return c == 0 ? c0 : (c == 1 ? c1 : c2); });
        } else {
          e_phase([&](int c, int e) { return lut_s[(mw[e] >> (10 * c)) & 1023u]; });
        }
      } else {
        const float2 c1 = make_float2(m.ca1, m.cb1);
        e_phase([&](int, int) { return c1; });
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { hxm[e] = hxn[e]; hym[e] = hyn[e]; exk[e] = exn[e]; eyk[e] = eyn[e]; }
    exk_m = exn_m;
    cur ^= 1;
  }
  if constexpr ((HINT & 256) != 0) {
    if (act && !halo && pend_p >= 0) {
      stv_h<V, (HINT & 1) != 0>(b.ex + pend_p + i0, pend_ex);
      stv_h<V, (HINT & 1) != 0>(b.ey + pend_p + i0, pend_ey);
      stv_h<V, (HINT & 1) != 0>(b.ez + pend_p + i0, pend_ez);
    }
  }
}

// =============================================================================================
// K3  CPML slab corrections (post-correction form: the main kernels use the plain derivative
//     everywhere; inside a slab the difference  (1/kappa - 1) d + psi  is added afterwards,
//     which is exact because both updates are linear in the curl).
// =============================================================================================
// medium index of component c of a cell: from its packed word (10 bits per component), or — wide layout — 16 bits per component in
// two words per cell (mw: E_x | E_y << 16; m4b[cell]: E_z)
__device__ __forceinline__ uint32_t medium_of(uint32_t mw, const uint32_t* m4b, long long cell, int c) {
  if (!m4b) return (mw >> (10 * c)) & 1023u;
  return c == 0 ? (mw & 0xFFFFu) : (c == 1 ? (mw >> 16) : (m4b[cell] & 0xFFFFu));
}

struct SlabP {
  int a;              // PML axis
  int s_lo, s_n;      // slab index range [s_lo, s_lo + s_n) along a
  int psi_base;       // slab-local index of s_lo inside the psi array
  int psi_ns;         // extent of the psi array along a (n_lo + n_hi entries)
  int kbeg, kend;     // z-range processed by this launch (already intersected for a == 2)
  int kpsi0;          // local k that maps to psi index 0 along z (0 unless a == 2: then unused)
};

__device__ __forceinline__ long long psi_index(const GridP& g, const SlabP& sl, int i, int j, int k, int si) {
  if (sl.a == 0) return ((long long)k * g.ny + j) * sl.psi_ns + si;
  if (sl.a == 1) return ((long long)k * sl.psi_ns + si) * g.nx + i;
  return ((long long)si * g.ny + j) * g.nx + i;
}

// E-side:  E_{a+1} -= Cb * ((kinv-1) d(H_{a+2})/da + psi1),  E_{a+2} += Cb * ((kinv-1) d(H_{a+1})/da + psi2)
__global__ __launch_bounds__(256) void pml_e_kernel(GridP g, SlabP sl_lo, SlabP sl_hi, float* e1, float* e2, const float* h1,
                                                     const float* h2, float* psi1, float* psi2,
                                                     const float4* cf4,
                                                     const float* idl, const uint32_t* m4,
                                                     const float2* lut, float cb_uniform, const uint32_t* m4b) {
  const SlabP sl = blockIdx.y ? sl_hi : sl_lo;   // both faces of an axis in one launch (disjoint cells)
  const int bx = (sl.a == 0) ? sl.s_n : g.nx;
  const int by = (sl.a == 1) ? sl.s_n : g.ny;
  const int bz = sl.kend - sl.kbeg;
  const long long total = (long long)bx * by * bz;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int lx = (int)(t % bx);
  const int ly = (int)((t / bx) % by);
  const int lz = (int)(t / ((long long)bx * by));
  const int i = (sl.a == 0) ? sl.s_lo + lx : lx;
  const int j = (sl.a == 1) ? sl.s_lo + ly : ly;
  const int k = sl.kbeg + lz;
  const int ia = (sl.a == 0) ? i : (sl.a == 1 ? j : k);
  const int c1 = (sl.a + 1) % 3, c2 = (sl.a + 2) % 3;
  const int idx3[3] = {i, j, k};
  const int bc0[3] = {g.bcx0, g.bcy0, g.pec_z0 ? BC_PEC : BC_NEIGHBOR};
  const long long stride = (sl.a == 0) ? 1 : (sl.a == 1 ? (long long)g.nx : g.sxy);
  const long long p = (long long)k * g.sxy + (long long)j * g.nx + i;
  float d1, d2;   // d/da of H_{a+1}, H_{a+2}
  if (ia == 0 && sl.a != 2) {
    if (bc0[sl.a] == BC_PEC) return;                  // both components are wall-tangential: stay 0
    if (bc0[sl.a] == BC_PMC) { d1 = 2.f * h1[p] * idl[ia]; d2 = 2.f * h2[p] * idl[ia]; }
    else { const long long w = (long long)((sl.a == 0 ? g.nx : g.ny) - 1) * stride;
           d1 = (h1[p] - h1[p + w]) * idl[ia]; d2 = (h2[p] - h2[p + w]) * idl[ia]; }
  } else {
    if (ia == 0 && g.pec_z0) return;                  // z wall (ghost plane otherwise)
    d1 = (h1[p] - h1[p - stride]) * idl[ia];
    d2 = (h2[p] - h2[p - stride]) * idl[ia];
  }
  const int si = sl.psi_base + (ia - sl.s_lo);
  const long long q = psi_index(g, sl, i, j, k, si);
  const float4 cf = cf4[ia];
  const float kv = cf.x, b = cf.y, c = cf.z;
  const float p1 = b * psi1[q] + c * d2;     // psi of E_{a+1} follows d(H_{a+2})/da
  const float p2 = b * psi2[q] + c * d1;     // psi of E_{a+2} follows d(H_{a+1})/da
  psi1[q] = p1;
  psi2[q] = p2;
  const uint32_t mw = m4 ? m4[p] : 0u;
  const float cb1 = m4 ? lut[medium_of(mw, m4b, p, c1)].y : cb_uniform;
  const float cb2 = m4 ? lut[medium_of(mw, m4b, p, c2)].y : cb_uniform;
  // PEC walls of the other transverse axis
  const bool w1 = (idx3[c2] == 0) && (bc0[c2] == BC_PEC);   // E_{c1} is tangential to the c2-wall
  const bool w2 = (idx3[c1] == 0) && (bc0[c1] == BC_PEC);
  if (!w1) e1[p] -= cb1 * (kv * d2 + p1);
  if (!w2) e2[p] += cb2 * (kv * d1 + p2);
}

// H-side:  H_{a+1} += ch * ((kinv-1) d(E_{a+2})/da + psi1),  H_{a+2} -= ch * ((kinv-1) d(E_{a+1})/da + psi2)
__global__ __launch_bounds__(256) void pml_h_kernel(GridP g, SlabP sl_lo, SlabP sl_hi, float* h1, float* h2, const float* e1,
                                                     const float* e2, float* psi1, float* psi2,
                                                     const float4* cf4,
                                                     const float* ipl) {
  const SlabP sl = blockIdx.y ? sl_hi : sl_lo;   // both faces of an axis in one launch (disjoint cells)
  const int bx = (sl.a == 0) ? sl.s_n : g.nx;
  const int by = (sl.a == 1) ? sl.s_n : g.ny;
  const int bz = sl.kend - sl.kbeg;
  const long long total = (long long)bx * by * bz;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int lx = (int)(t % bx);
  const int ly = (int)((t / bx) % by);
  const int lz = (int)(t / ((long long)bx * by));
  const int i = (sl.a == 0) ? sl.s_lo + lx : lx;
  const int j = (sl.a == 1) ? sl.s_lo + ly : ly;
  const int k = sl.kbeg + lz;
  const int ia = (sl.a == 0) ? i : (sl.a == 1 ? j : k);
  const int na = (sl.a == 0) ? g.nx : (sl.a == 1 ? g.ny : g.nz);
  const int bc1 = (sl.a == 0) ? g.bcx1 : (sl.a == 1 ? g.bcy1 : BC_NEIGHBOR);
  const long long stride = (sl.a == 0) ? 1 : (sl.a == 1 ? (long long)g.nx : g.sxy);
  const long long p = (long long)k * g.sxy + (long long)j * g.nx + i;
  float n1, n2;   // E_{a+1}, E_{a+2} at ia + 1
  if (ia == na - 1 && sl.a != 2) {
    if (bc1 == BC_PERIODIC) { n1 = e1[p - (long long)(na - 1) * stride]; n2 = e2[p - (long long)(na - 1) * stride]; }
    else { n1 = 0.f; n2 = 0.f; }
  } else {
    n1 = e1[p + stride];      // for a == 2 the ghost plane holds the boundary value
    n2 = e2[p + stride];
  }
  const float d1 = (n1 - e1[p]) * ipl[ia];
  const float d2 = (n2 - e2[p]) * ipl[ia];
  const int si = sl.psi_base + (ia - sl.s_lo);
  const long long q = psi_index(g, sl, i, j, k, si);
  const float4 cf = cf4[ia];
  const float kv = cf.x, b = cf.y, c = cf.z;
  const float p1 = b * psi1[q] + c * d2;
  const float p2 = b * psi2[q] + c * d1;
  psi1[q] = p1;
  psi2[q] = p2;
  h1[p] += g.ch * (kv * d2 + p1);
  h2[p] -= g.ch * (kv * d1 + p2);
}

// float4 versions for the y and z slabs (rows are contiguous along x): one thread = 4 cells, same
// per-element arithmetic as the scalar kernels above.
__global__ __launch_bounds__(256) void pml_e4_kernel(GridP g, SlabP sl_lo, SlabP sl_hi, float* e1, float* e2, const float* h1,
                                                      const float* h2, float* psi1, float* psi2,
                                                      const float4* cf4,
                                                      const float* idl, const uint32_t* m4,
                                                      const float2* lut, float cb_uniform, const uint32_t* m4b) {
  const SlabP sl = blockIdx.y ? sl_hi : sl_lo;   // both faces of an axis in one launch (disjoint cells)
  constexpr int V = 4;
  const int bx = g.nx / V;
  const int by = (sl.a == 1) ? sl.s_n : g.ny;
  const int bz = sl.kend - sl.kbeg;
  const long long total = (long long)bx * by * bz;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int i0 = (int)(t % bx) * V;
  const int ly = (int)((t / bx) % by);
  const int lz = (int)(t / ((long long)bx * by));
  const int j = (sl.a == 1) ? sl.s_lo + ly : ly;
  const int k = sl.kbeg + lz;
  const int ia = (sl.a == 1) ? j : k;
  const int c1 = (sl.a + 1) % 3, c2 = (sl.a + 2) % 3;
  const int bcz0 = g.pec_z0 ? BC_PEC : BC_NEIGHBOR;
  const long long stride = (sl.a == 1) ? (long long)g.nx : g.sxy;
  const long long p = (long long)k * g.sxy + (long long)j * g.nx + i0;
  float a1[V], a2[V], b1[V], b2[V];
  ldv<V>(a1, h1 + p);
  ldv<V>(a2, h2 + p);
  float s1 = 1.f;          // derivative = (a - s * b) * idl  with b the lower neighbour
  if (ia == 0 && sl.a == 1) {
    if (g.bcy0 == BC_PEC) return;
    if (g.bcy0 == BC_PMC) { for (int e = 0; e < V; ++e) { b1[e] = a1[e]; b2[e] = a2[e]; } s1 = -1.f; }
    else { ldv<V>(b1, h1 + p + (long long)(g.ny - 1) * stride); ldv<V>(b2, h2 + p + (long long)(g.ny - 1) * stride); }
  } else {
    if (ia == 0 && g.pec_z0) return;
    ldv<V>(b1, h1 + p - stride);
    ldv<V>(b2, h2 + p - stride);
  }
  const int si = sl.psi_base + (ia - sl.s_lo);
  const long long q = psi_index(g, sl, i0, j, k, si);
  const float4 cf = cf4[ia];
  const float kv = cf.x, b = cf.y, c = cf.z, w = idl[ia];
  float q1[V], q2[V], x1[V], x2[V];
  ldv<V>(q1, psi1 + q);
  ldv<V>(q2, psi2 + q);
  ldv<V>(x1, e1 + p);
  ldv<V>(x2, e2 + p);
  uint32_t mw[V] = {0u, 0u, 0u, 0u};
  if (m4) ldm<V>(mw, m4 + p);
  // E_{c1} is tangential to the c2-wall, E_{c2} to the c1-wall; along x only element i == 0 can sit on it
  const int jk[3] = {0, j, k};
  const int bc0[3] = {g.bcx0, g.bcy0, bcz0};
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float d1 = (s1 > 0.f) ? (a1[e] - b1[e]) * w : 2.f * a1[e] * w;
    const float d2 = (s1 > 0.f) ? (a2[e] - b2[e]) * w : 2.f * a2[e] * w;
    const float p1 = b * q1[e] + c * d2;
    const float p2 = b * q2[e] + c * d1;
    q1[e] = p1;
    q2[e] = p2;
    const float cb1 = m4 ? lut[medium_of(mw[e], m4b, p + e, c1)].y : cb_uniform;
    const float cb2 = m4 ? lut[medium_of(mw[e], m4b, p + e, c2)].y : cb_uniform;
    const int i3c2 = (c2 == 0) ? i0 + e : jk[c2];
    const int i3c1 = (c1 == 0) ? i0 + e : jk[c1];
    const bool w1 = (i3c2 == 0) && (bc0[c2] == BC_PEC);
    const bool w2 = (i3c1 == 0) && (bc0[c1] == BC_PEC);
    if (!w1) x1[e] -= cb1 * (kv * d2 + p1);
    if (!w2) x2[e] += cb2 * (kv * d1 + p2);
  }
  stv<V>(psi1 + q, q1);
  stv<V>(psi2 + q, q2);
  stv<V>(e1 + p, x1);
  stv<V>(e2 + p, x2);
}

__global__ __launch_bounds__(256) void pml_h4_kernel(GridP g, SlabP sl_lo, SlabP sl_hi, float* h1, float* h2, const float* e1,
                                                      const float* e2, float* psi1, float* psi2,
                                                      const float4* cf4,
                                                      const float* ipl) {
  const SlabP sl = blockIdx.y ? sl_hi : sl_lo;   // both faces of an axis in one launch (disjoint cells)
  constexpr int V = 4;
  const int bx = g.nx / V;
  const int by = (sl.a == 1) ? sl.s_n : g.ny;
  const int bz = sl.kend - sl.kbeg;
  const long long total = (long long)bx * by * bz;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int i0 = (int)(t % bx) * V;
  const int ly = (int)((t / bx) % by);
  const int lz = (int)(t / ((long long)bx * by));
  const int j = (sl.a == 1) ? sl.s_lo + ly : ly;
  const int k = sl.kbeg + lz;
  const int ia = (sl.a == 1) ? j : k;
  const long long stride = (sl.a == 1) ? (long long)g.nx : g.sxy;
  const long long p = (long long)k * g.sxy + (long long)j * g.nx + i0;
  float n1[V], n2[V], c1v[V], c2v[V];
  if (ia == g.ny - 1 && sl.a == 1) {
    if (g.bcy1 == BC_PERIODIC) {
      ldv<V>(n1, e1 + p - (long long)(g.ny - 1) * stride);
      ldv<V>(n2, e2 + p - (long long)(g.ny - 1) * stride);
    } else { zero<V>(n1); zero<V>(n2); }
  } else {
    ldv<V>(n1, e1 + p + stride);
    ldv<V>(n2, e2 + p + stride);
  }
  ldv<V>(c1v, e1 + p);
  ldv<V>(c2v, e2 + p);
  const int si = sl.psi_base + (ia - sl.s_lo);
  const long long q = psi_index(g, sl, i0, j, k, si);
  const float4 cf = cf4[ia];
  const float kv = cf.x, b = cf.y, c = cf.z, w = ipl[ia];
  float q1[V], q2[V], x1[V], x2[V];
  ldv<V>(q1, psi1 + q);
  ldv<V>(q2, psi2 + q);
  ldv<V>(x1, h1 + p);
  ldv<V>(x2, h2 + p);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float d1 = (n1[e] - c1v[e]) * w;
    const float d2 = (n2[e] - c2v[e]) * w;
    const float p1 = b * q1[e] + c * d2;
    const float p2 = b * q2[e] + c * d1;
    q1[e] = p1;
    q2[e] = p2;
    x1[e] += g.ch * (kv * d2 + p1);
    x2[e] -= g.ch * (kv * d1 + p2);
  }
  stv<V>(psi1 + q, q1);
  stv<V>(psi2 + q, q2);
  stv<V>(h1 + p, x1);
  stv<V>(h2 + p, x2);
}

// =============================================================================================
// K4  ADE (pole-residue) post-update on the compact list of dispersive cells of one component
//     and one medium:  E <- E* - cc * S(Q),  Q <- kap Q + bet (E_new + E_old)
// =============================================================================================
constexpr int kMaxPoles = 8;
struct AdeP {
  int n_poles;
  float cc;
  float2 kap[kMaxPoles];
  float2 bet[kMaxPoles];
};

// (the list is sorted by cell index: a launch over a plane range covers the entries [t0, t0 + n) of those planes only —
//  `cell`, `e_old`, `q` arrive offset by t0, `qs` is the stride between the poles of one entry = the length of the list)
// (`cs` != nullptr — round 6, runs whose step pairs advance the dispersive cells, fdtd_fused2.hpp DispP: the memory term of the
//  NEXT step, cc S(Q^{n+1}), goes to the paged array the two-step sweep subtracts from E^{n+2}; qoff[t] = the entry's slot in it)
__global__ __launch_bounds__(256) void ade_kernel(float* e, const uint32_t* cell, float* e_old, float2* q,
                                                   long long n, long long qs, long long zlo, long long zhi, AdeP a,
                                                   float* cs, const uint32_t* qoff) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long p = cell[t];
  if (p < zlo || p >= zhi) return;        // z-range filter (unsorted lists)
  const float es = e[p];
  const float eo = e_old[t];
  float S = 0.f;
  float2 qq[kMaxPoles];
#pragma unroll
  for (int k = 0; k < kMaxPoles; ++k) {
    if (k < a.n_poles) {
      qq[k] = q[(long long)k * qs + t];
      // 2 Re[(kap - 1) Q]
      S += 2.f * ((a.kap[k].x - 1.f) * qq[k].x - a.kap[k].y * qq[k].y);
    }
  }
  const float en = es - a.cc * S;
  e[p] = en;
  e_old[t] = en;
  const float se = en + eo;
  float S1 = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxPoles; ++k) {
    if (k < a.n_poles) {
      float2 r;
      r.x = a.kap[k].x * qq[k].x - a.kap[k].y * qq[k].y + a.bet[k].x * se;
      r.y = a.kap[k].x * qq[k].y + a.kap[k].y * qq[k].x + a.bet[k].y * se;
      q[(long long)k * qs + t] = r;
      S1 += 2.f * ((a.kap[k].x - 1.f) * r.x - a.kap[k].y * r.y);
    }
  }
  if (cs) cs[qoff[t]] = a.cc * S1;
}

// K4 twice, behind a two-step sweep that subtracted the memory term of step n itself (fused2_step_kernel, OPT bit 5) and left
// E^{n+1} of the dispersive cells in the paged array `e1`: per entry  Q^{n+1} = kap Q^n + bet (E^{n+1} + E^n),
// E^{n+2} <- E^{n+2} - cc S(Q^{n+1}),  Q^{n+2} = kap Q^{n+1} + bet (E^{n+2} + E^{n+1}),  and the memory term of the step after,
// cc S(Q^{n+2}), into `cs` — the operations of two ade_kernel launches in their order, the pole states read and written once.
// e = E^{n+2} as the sweep (and the sources / damping of step n+1) left it; e_old[t] = E^n on entry, E^{n+2} on exit.
__global__ __launch_bounds__(256) void ade2_kernel(float* e, const uint32_t* cell, float* e_old, float2* q, long long n, AdeP a,
                                                    const float* e1, float* cs, const uint32_t* qoff) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long p = cell[t];
  const uint32_t o = qoff[t];
  const float em = e1[o];
  const float se1 = em + e_old[t];
  float S = 0.f;
  float2 qq[kMaxPoles];
#pragma unroll
  for (int k = 0; k < kMaxPoles; ++k) {
    if (k < a.n_poles) {
      const float2 q0 = q[(long long)k * n + t];
      qq[k].x = a.kap[k].x * q0.x - a.kap[k].y * q0.y + a.bet[k].x * se1;
      qq[k].y = a.kap[k].x * q0.y + a.kap[k].y * q0.x + a.bet[k].y * se1;
      S += 2.f * ((a.kap[k].x - 1.f) * qq[k].x - a.kap[k].y * qq[k].y);
    }
  }
  const float en = e[p] - a.cc * S;
  e[p] = en;
  e_old[t] = en;
  const float se = en + em;
  float S1 = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxPoles; ++k) {
    if (k < a.n_poles) {
      float2 r;
      r.x = a.kap[k].x * qq[k].x - a.kap[k].y * qq[k].y + a.bet[k].x * se;
      r.y = a.kap[k].x * qq[k].y + a.kap[k].y * qq[k].x + a.bet[k].y * se;
      q[(long long)k * n + t] = r;
      S1 += 2.f * ((a.kap[k].x - 1.f) * r.x - a.kap[k].y * r.y);
    }
  }
  cs[o] = a.cc * S1;
}

// ---- paged memory terms (round 6): set-up kernels, each run once per engine ----------------------------------------------------
// mark the row segments (256 cells of a row) that hold an entry of the list
__global__ __launch_bounds__(256) void disp_mark_kernel(const uint32_t* cell, long long n, int nx, int nbx, int* flag) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long p = cell[t];
  flag[(p / nx) * nbx + (int)(p % nx) / 256] = 1;
}
// qoff[t]: the slot of entry t of a list of component `comp` in the paged arrays, and the memory term of the state as it is,
// cc S(Q), into it; box[0 .. 5]: the bounding box of the dispersive cells
__global__ __launch_bounds__(256) void disp_qoff_kernel(const uint32_t* cell, long long n, int nx, int ny, int nbx, const int* dseg,
                                                         int comp, const float2* q, AdeP a, uint32_t* qoff, float* cs, int* box) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long p = cell[t];
  const int i = (int)(p % nx);
  const long long blk = dseg[(p / nx) * nbx + i / 256];
  const uint32_t o = (uint32_t)((blk * 3 + comp) * 256 + (i & 255));
  qoff[t] = o;
  float S = 0.f;
  for (int k = 0; k < a.n_poles; ++k) {
    const float2 qk = q[(long long)k * n + t];
    S += 2.f * ((a.kap[k].x - 1.f) * qk.x - a.kap[k].y * qk.y);
  }
  cs[o] = a.cc * S;
  const int j = (int)((p / nx) % ny), k = (int)(p / ((long long)nx * ny));
  // (one atomic per lane on six words: a one-off of a few ms for 10^7 entries)
  atomicMin(&box[0], i); atomicMax(&box[1], i); atomicMin(&box[2], j); atomicMax(&box[3], j); atomicMin(&box[4], k); atomicMax(&box[5], k);
}

// =============================================================================================
// K5  sources
// =============================================================================================
// F[comp][cell] += w_re * Re(wave[n]) - w_im * Im(wave[n])
// (time step n = step, or *step_dev + step when the launch is a node of a captured graph: a graph bakes its kernel
//  arguments, so the step counter of a replayed step pair lives in device memory, fdtd_capi.hip graph_pair)
__global__ __launch_bounds__(256) void point_source_kernel(float* f0, float* f1, float* f2, const int32_t* comp,
                                                            const uint32_t* cell, const float* w_re,
                                                            const float* w_im, const float2* wave, long long step,
                                                            long long n, long long zlo, long long zhi,
                                                            const long long* step_dev) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long p = cell[t];
  if (p < zlo || p >= zhi) return;
  if (step_dev) step += *step_dev;
  const float2 a = wave[step];
  const int c = comp[t] % 3;
  float* f = (c == 0) ? f0 : (c == 1 ? f1 : f2);
  f[p] += w_re[t] * a.x - w_im[t] * a.y;
}

// TFSF surface correction:  F[comp][cell] += sum_e w[e] * aux[aux_index[e]]
// One thread per TARGET node (the host groups the entries by (component, cell), entries of a node in
// their given order: a node on a box edge or corner receives two or three): no atomics, the sum is
// formed in a fixed order and the result is bitwise repeatable.
__global__ __launch_bounds__(256) void tfsf_corr_kernel(float* f0, float* f1, float* f2, const int32_t* comp,
                                                         const uint32_t* cell, const int32_t* start, const float* w,
                                                         const int32_t* ai, const float* aux, long long n_targets,
                                                         long long zlo, long long zhi) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_targets) return;
  const long long p = cell[t];
  if (p < zlo || p >= zhi) return;
  const int c = comp[t] % 3;
  float* f = (c == 0) ? f0 : (c == 1 ? f1 : f2);
  float acc = 0.f;
  for (int e = start[t]; e < start[t + 1]; ++e) acc += w[e] * aux[ai[e]];
  f[p] += acc;
}

// 1-D auxiliary grid of the incident plane wave
__global__ void tfsf_aux_h_kernel(float* h1, const float* e1, const float* ah, const float* bh, int n_aux) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_aux; i += gridDim.x * blockDim.x)
    h1[i] = ah[i] * h1[i] - bh[i] * (e1[i + 1] - e1[i]);
}

// single workgroup: interior update (the grid ends in matched lossy pads; end nodes stay 0), soft source
__global__ void tfsf_aux_e_kernel(float* e1, const float* h1, const float* ae, const float* be, int n_aux,
                                  int src_cell, const float* wave, long long step, const long long* step_dev) {
  for (int i = 1 + threadIdx.x; i < n_aux; i += blockDim.x)
    e1[i] = ae[i] * e1[i] - be[i] * (h1[i] - h1[i - 1]);
  __syncthreads();
  if (threadIdx.x == 0) e1[src_cell] += wave[step_dev ? *step_dev + step : step];
}

// ---- paged source terms of a step pair (round 6, fdtd_fused2.hpp SrcP) -------------------------------------------------------------
// soff[t]: the slot of node t of a list in the paged arrays; count[slot] += 1 (two nodes of one side on one slot: the pair cannot
// form E + term exactly — the host then keeps single steps while the lists inject); box: bounding box of the nodes
__global__ __launch_bounds__(256) void src_soff_kernel(const uint32_t* cell, const int32_t* comp, long long n, int nx, int ny, int nbx,
                                                        const int* sseg, uint32_t* soff, int* box) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long p = cell[t];
  const int i = (int)(p % nx);
  const long long blk = sseg[(p / nx) * nbx + i / 256];
  const uint32_t o = (uint32_t)((blk * 3 + comp[t] % 3) * 256 + (i & 255));
  soff[t] = o;
  const int j = (int)((p / nx) % ny), k = (int)(p / ((long long)nx * ny));
  atomicMin(&box[0], i); atomicMax(&box[1], i); atomicMin(&box[2], j); atomicMax(&box[3], j); atomicMin(&box[4], k); atomicMax(&box[5], k);
}
// mark != 0: the list takes its slots in this layer's occupancy map (hit[1] counts slots taken twice: a node listed twice);
// mark == 0: hit[0] counts the list's slots an earlier list holds in this layer
__global__ __launch_bounds__(256) void src_layer_kernel(const uint32_t* soff, long long n, int* occ, int* hit, int mark) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  if (mark) { if (atomicAdd(&occ[soff[t]], 1) > 0) atomicAdd(&hit[1], 1); }
  else if (occ[soff[t]] > 0) atomicAdd(&hit[0], 1);
}
// the term point_source_kernel would add at `step`, formed by its operations, into the node's slot (zero != 0: the list is spent)
__global__ __launch_bounds__(256) void src_fill_points_kernel(float* val, const uint32_t* soff, const float* w_re, const float* w_im,
                                                               const float2* wave, long long step, long long n, int zero) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  if (zero) { val[soff[t]] = 0.f; return; }
  const float2 a = wave[step];
  val[soff[t]] = w_re[t] * a.x - w_im[t] * a.y;
}
// the term tfsf_corr_kernel would add from the incident grid as it stands, formed by its operations
__global__ __launch_bounds__(256) void src_fill_tfsf_kernel(float* val, const uint32_t* soff, const int32_t* start, const float* w,
                                                             const int32_t* ai, const float* aux, long long n_targets, int zero) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_targets) return;
  float acc = 0.f;
  if (!zero) for (int e = start[t]; e < start[t + 1]; ++e) acc += w[e] * aux[ai[e]];
  val[soff[t]] = acc;
}

// the device-side step counter of captured step pairs: set (when the host stepped outside a graph) / advanced (last node)
__global__ void step_counter_kernel(long long* step_dev, long long value, int add) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *step_dev = add ? *step_dev + value : value;
}

// =============================================================================================
// K6  monitors
// =============================================================================================
struct BoxP { int lo0, lo1, lo2; int nx, ny, nz; };   // box origin (i,j,k) and extents

// All components of one monitor in ONE launch (blockIdx.y = entry): small grids are launch-bound —
// at 200^3 nine single-component record launches of 4.7 us each were 31 % of a step
// (profiles/r01h_small_grid.txt).
struct RecP {
  const float* f[6];    // field array of each entry
  int slot[6];          // component slot of the monitor buffer
  float scale[6];       // time monitors: 1 for E, 0.5 for each of the two H half-samples
  int acc[6];           // time monitors: accumulate (H) or overwrite (E)
  int n;
};

// out[slot][cell] (=|+=) scale * F[box cell]
__global__ __launch_bounds__(256) void time_record_multi_kernel(RecP r, GridP g, BoxP b, float* out, long long cells) {
  const int q = blockIdx.y;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cells) return;
  const int lx = (int)(t % b.nx), ly = (int)((t / b.nx) % b.ny), lz = (int)(t / ((long long)b.nx * b.ny));
  const float v = r.scale[q] * r.f[q][(long long)(b.lo2 + lz) * g.sxy + (long long)(b.lo1 + ly) * g.nx + b.lo0 + lx];
  float* o = out + (long long)r.slot[q] * cells;
  o[t] = r.acc[q] ? o[t] + v : v;
}

// acc[f][slot][cell] += F[box cell] * phase[f]
__global__ __launch_bounds__(256) void dft_record_multi_kernel(RecP r, GridP g, BoxP b, float2* acc, long long cells,
                                                                long long fstride, const float2* phase, int nf) {
  const int q = blockIdx.y;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cells) return;
  const int lx = (int)(t % b.nx), ly = (int)((t / b.nx) % b.ny), lz = (int)(t / ((long long)b.nx * b.ny));
  const float v = r.f[q][(long long)(b.lo2 + lz) * g.sxy + (long long)(b.lo1 + ly) * g.nx + b.lo0 + lx];
  float2* a0 = acc + (long long)r.slot[q] * cells;
  for (int k = 0; k < nf; ++k) {
    const float2 ph = phase[k];
    float2 a = a0[(long long)k * fstride + t];
    a.x += v * ph.x;
    a.y += v * ph.y;
    a0[(long long)k * fstride + t] = a;
  }
}

// =============================================================================================
// K7  field-energy reduction  W = sum |E|^2 + eta0^2 sum |H|^2  over the slab (shutoff / divergence
//     detection; eta0^2 = mu0 / eps0 puts the two terms on the same scale, so a standing wave whose
//     electric energy passes through zero twice per period does not look decayed).  Two passes with a
//     FIXED launch geometry and fixed summation trees — no atomics: the value, and with it the step a run
//     shuts off at, is bitwise repeatable.
// =============================================================================================
constexpr double kEta0Sq = 376.730313668 * 376.730313668;     // (mu0 c0)^2, ref constants.py:32
constexpr int kEnergyBlocks = 1024;

__device__ __forceinline__ double block_sum_256(double acc, double* part) {
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) part[wv] = acc;
  __syncthreads();
  return (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void energy_partial_kernel(const float* ex, const float* ey, const float* ez,
                                                              const float* hx, const float* hy, const float* hz,
                                                              long long n, double* partial) {
  __shared__ double part[4];
  double ae = 0.0, ah = 0.0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (long long)gridDim.x * blockDim.x) {
    const float a = ex[t], b = ey[t], c = ez[t];
    const float u = hx[t], v = hy[t], w = hz[t];
    // squares in double: a field that has grown past 1.8e19 must not overflow the sum before it is itself Inf
    ae += (double)a * a + (double)b * b + (double)c * c;
    ah += (double)u * u + (double)v * v + (double)w * w;
  }
  const double tot = block_sum_256(ae + kEta0Sq * ah, part);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// one workgroup: partial[0 .. n_part) -> out[0]
__global__ __launch_bounds__(256) void energy_final_kernel(const double* partial, int n_part, double* out) {
  __shared__ double part[4];
  double acc = 0.0;
  for (int t = threadIdx.x; t < n_part; t += 256) acc += partial[t];
  const double tot = block_sum_256(acc, part);
  if (threadIdx.x == 0) out[0] = tot;
}

// e_old[t] = E[cell[t]]   (after fdtd_set_field: the ADE recursion needs E^n of its cells)
__global__ __launch_bounds__(256) void ade_gather_kernel(const float* e, const uint32_t* cell, float* e_old, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) e_old[t] = e[cell[t]];
}

// ---- absorber layers (K3b) -------------------------------------------------------------------
// Matched-conductivity damping of the Absorber boundary (ref boundary.py:427-476; formulas in
// tidy3d_amd/coeffs.py damping_tables): inside the layers every component is multiplied once per
// step by  f = f_x[i] f_y[j] f_z[k],  each axis factor taken at the component's Yee location (fc on
// the axes where it sits on cell centres, fb elsewhere).  One launch covers the slab [s_lo, s_lo+s_n)
// of axis a over the planes [kbeg, kend); cells that also lie in a slab of a lower axis belong to
// that axis' launch.  Pure streaming: 3 loads + 3 stores per cell, x fastest -> coalesced rows.
struct DampP {
  const float* fb[3];
  const float* fc[3];
  int lo[3], hi[3];        // layers of axis b: index < lo[b] or index >= hi[b]
};

__global__ __launch_bounds__(256) void damp_kernel(GridP g, DampP d, float* f0, float* f1, float* f2, int is_h,
                                                   int a, int s_lo, int s_n, int kbeg, int kend) {
  const long long bx = (a == 0) ? s_n : g.nx, by = (a == 1) ? s_n : g.ny;
  const long long bz = (a == 2) ? s_n : (kend - kbeg);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bx * by * bz) return;
  const int i = (int)(t % bx) + (a == 0 ? s_lo : 0);
  const int j = (int)((t / bx) % by) + (a == 1 ? s_lo : 0);
  const int k = (int)(t / (bx * by)) + (a == 2 ? s_lo : kbeg);
  if (a >= 1 && (i < d.lo[0] || i >= d.hi[0])) return;
  if (a == 2 && (j < d.lo[1] || j >= d.hi[1])) return;
  const float bxv = d.fb[0][i], cxv = d.fc[0][i];
  const float byv = d.fb[1][j], cyv = d.fc[1][j];
  const float bzv = d.fb[2][k], czv = d.fc[2][k];
  float w0, w1, w2;
  if (is_h) {           // H_c: boundary along c, centres along the other two axes
    w0 = bxv * cyv * czv; w1 = cxv * byv * czv; w2 = cxv * cyv * bzv;
  } else {              // E_c: centre along c, boundaries along the other two
    w0 = cxv * byv * bzv; w1 = bxv * cyv * bzv; w2 = bxv * byv * czv;
  }
  const long long idx = (long long)k * g.sxy + (long long)j * g.nx + i;
  f0[idx] *= w0;
  f1[idx] *= w1;
  f2[idx] *= w2;
}

// y / z layers on float4-aligned rows: four x cells per thread (one 16-byte load and store per component)
__global__ __launch_bounds__(256) void damp4_kernel(GridP g, DampP d, float* f0, float* f1, float* f2, int is_h,
                                                    int a, int s_lo, int s_n, int kbeg, int kend) {
  const long long bx = g.nx / 4, by = (a == 1) ? s_n : g.ny;
  const long long bz = (a == 2) ? s_n : (kend - kbeg);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bx * by * bz) return;
  const int i = 4 * (int)(t % bx);
  const int j = (int)((t / bx) % by) + (a == 1 ? s_lo : 0);
  const int k = (int)(t / (bx * by)) + (a == 2 ? s_lo : kbeg);
  if (a == 2 && (j < d.lo[1] || j >= d.hi[1])) return;
  const float4 bx4 = *reinterpret_cast<const float4*>(d.fb[0] + i);
  const float4 cx4 = *reinterpret_cast<const float4*>(d.fc[0] + i);
  const float byv = d.fb[1][j], cyv = d.fc[1][j];
  const float bzv = d.fb[2][k], czv = d.fc[2][k];
  const float bxs[4] = {bx4.x, bx4.y, bx4.z, bx4.w}, cxs[4] = {cx4.x, cx4.y, cx4.z, cx4.w};
  const long long idx = (long long)k * g.sxy + (long long)j * g.nx + i;
  float4 v0 = *reinterpret_cast<float4*>(f0 + idx);
  float4 v1 = *reinterpret_cast<float4*>(f1 + idx);
  float4 v2 = *reinterpret_cast<float4*>(f2 + idx);
  float* p0 = reinterpret_cast<float*>(&v0);
  float* p1 = reinterpret_cast<float*>(&v1);
  float* p2 = reinterpret_cast<float*>(&v2);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // cells of this row that lie in an x layer belong to the x launch (scalar kernel)
    if (i + q < d.lo[0] || i + q >= d.hi[0]) continue;
    float w0, w1, w2;            // same products, in the same order, as damp_kernel
    if (is_h) { w0 = bxs[q] * cyv * czv; w1 = cxs[q] * byv * czv; w2 = cxs[q] * cyv * bzv; }
    else      { w0 = cxs[q] * byv * bzv; w1 = bxs[q] * cyv * bzv; w2 = bxs[q] * byv * czv; }
    p0[q] *= w0; p1[q] *= w1; p2[q] *= w2;
  }
  *reinterpret_cast<float4*>(f0 + idx) = v0;
  *reinterpret_cast<float4*>(f1 + idx) = v1;
  *reinterpret_cast<float4*>(f2 + idx) = v2;
}

// ---- Bloch boundaries (complex fields as a (Re, Im) pair of real field sets) ------------------
// F(r + L_a) = exp(i phi_a) F(r).  A Bloch axis carries one ghost cell at each end of the device grid
// (index 0 and n_real + 1; the real cells are 1 .. n_real; the kernels see PEC walls there and never
// wrap), refilled once per step, before the main kernels, with the rotated copy of the cell one period
// away:   F[0] = exp(-i phi) F[n_real],   F[n_real + 1] = exp(+i phi) F[1]   (all six components).
// The update kernels — the fused sweep included — then run unchanged: every real cell finds valid
// neighbours, what they write INTO the ghost cells is overwritten by the next fill, and the z direction
// uses the ghost planes it always had, filled with the rotated copy (bloch_plane_kernel).
struct CplxP {
  float* re; float* im;
};

struct Cplx6P {
  CplxP f[6];
};

// axis a in {0, 1}; threads over the planes perpendicular to a (the other in-plane axis with ITS ghost
// cells — corners — and all nz planes)
__global__ __launch_bounds__(256) void bloch_ghost_fill_kernel(GridP g, int a, int n_real, Cplx6P F, float cphi,
                                                               float sphi, int nz) {
  const int nb = (a == 0) ? g.ny : g.nx;
  const long long total = (long long)nb * nz;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int o = (int)(t % nb), k = (int)(t / nb);
  const long long stride = (a == 0) ? 1 : (long long)g.nx;
  const long long base = (long long)k * g.sxy + ((a == 0) ? (long long)o * g.nx : (long long)o);
  const long long p_lo = base, p_first = base + stride, p_last = base + (long long)n_real * stride,
                  p_hi = base + (long long)(n_real + 1) * stride;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const float lr = F.f[c].re[p_last], li = F.f[c].im[p_last];
    const float fr = F.f[c].re[p_first], fi = F.f[c].im[p_first];
    F.f[c].re[p_lo] = cphi * lr + sphi * li;          // exp(-i phi) (lr + i li)
    F.f[c].im[p_lo] = cphi * li - sphi * lr;
    F.f[c].re[p_hi] = cphi * fr - sphi * fi;          // exp(+i phi) (fr + i fi)
    F.f[c].im[p_hi] = cphi * fi + sphi * fr;
  }
}

// dst = exp(i phi) src on one xy-plane (z ghost planes; pass -sin(phi) for exp(-i phi))
__global__ __launch_bounds__(256) void bloch_plane_kernel(float* dre, float* dim, const float* sre, const float* sim,
                                                          float cphi, float sphi, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float a = sre[t], b = sim[t];
  dre[t] = cphi * a - sphi * b;
  dim[t] = cphi * b + sphi * a;
}

// =============================================================================================
// K9  far-field integration (near -> far projection): for every observation direction d the four surface integrals
//         N_c(d) = sum_{iu,iv} w_u[iu] w_v[iv] F_c[iu,iv] exp(-i k r_hat(d) . r'[iu,iv]),   F = (J_u, J_v, M_u, M_v)
//     over one surface of a projection monitor (ref components/field_projection.py:360-368 `integrate_2d`, :370
//     `_far_fields_for_surface`: the phase is exp(-i k (u r_u + v r_v + w r_w)), trapezoid weights in u and v).
//     One workgroup per direction; phases and sums in fp64 (k r' reaches 1e3 rad); fixed reduction tree -> repeatable.
// =============================================================================================
struct FarP {
  const double* u; const double* v; const double* wu; const double* wv;
  const double2* cur;                    // [4][n_u * n_v]: J_u, J_v, M_u, M_v on the (u, v) lattice, v fastest
  int n_u, n_v;
  double w0, k_re, k_im;
  const double* r_u; const double* r_v; const double* r_w;      // direction cosines along u, v and the surface normal
  double* out;                           // [n_dir][4][2]
};

__global__ __launch_bounds__(256) void far_field_kernel(FarP p) {
  __shared__ double red[256][8];
  const int d = blockIdx.x;
  const double ru = p.r_u[d], rv = p.r_v[d], rw = p.r_w[d];
  double acc[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  const int n = p.n_u * p.n_v;
  for (int q = threadIdx.x; q < n; q += 256) {
    const int iu = q / p.n_v, iv = q - iu * p.n_v;
    const double x = p.u[iu] * ru + p.v[iv] * rv + p.w0 * rw;
    // exp(-i k x), k = k_re + i k_im:  exp(k_im x) (cos(k_re x) - i sin(k_re x))
    const double ang = p.k_re * x;
    const double a = (p.k_im != 0.0 ? exp(p.k_im * x) : 1.0) * p.wu[iu] * p.wv[iv];
    const double pr = a * cos(ang), pi = -a * sin(ang);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double2 f = p.cur[(long long)m * n + q];
      acc[2 * m] += f.x * pr - f.y * pi;
      acc[2 * m + 1] += f.x * pi + f.y * pr;
    }
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) red[threadIdx.x][m] = acc[m];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int m = 0; m < 8; ++m) red[threadIdx.x][m] += red[threadIdx.x + s][m];
    }
    __syncthreads();
  }
  if (threadIdx.x < 8) p.out[(long long)d * 8 + threadIdx.x] = red[0][threadIdx.x];
}

// Occupies one wavefront for `ticks` of the constant-rate wall clock: the host times two of these on the engine's two
// streams to see whether the streams really run concurrently (fdtd_capi.hip probe_stream_overlap).
__global__ void spin_kernel(long long ticks) {
#if defined(__HIP_DEVICE_COMPILE__)
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
#else
  (void)ticks;
#endif
}

// ---- PMC on a PLUS face ------------------------------------------------------------------------------------------
// The wall is the cell boundary of index N along axis a; the grid carries two ghost cells beyond it.  The wall nodes
// (tangential E, normal H: located on boundaries along a) are unknowns like any other; everything beyond is the mirror
// image of the inside, refreshed at the start of every step — E_tan, H_norm even, E_norm, H_tan odd:
//   components on cell boundaries along a:  F[N + 1] = +F[N - 1]
//   components on cell centres along a:     F[N] = -F[N - 1],  F[N + 1] = -F[N - 2]
// The update equations preserve the mirror symmetry, so the sweep itself produces the right H^{n+1/2} in the ghost cell N
// (which the E update of the wall nodes differentiates); only what the truncation at N + 2 spoils is refreshed here.
// One thread per line along a.
// (x / y walls: the planes [k0, k0 + nk) — a z-slab rank refreshes its boundary planes and its interior on different streams)
__global__ __launch_bounds__(256) void mirror_fill_kernel(GridP g, FieldP f, int a, int N, int k0, int nk) {
  const int n1 = (a == 0) ? g.ny : g.nx;                // fastest transverse extent
  const int n2 = (a == 2) ? g.ny : nk;                  // slowest transverse extent
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n1 * n2) return;
  const int u = (int)(t % n1), w = (int)(t / n1) + (a == 2 ? 0 : k0);
  const long long stride = (a == 0) ? 1 : (a == 1 ? (long long)g.nx : g.sxy);
  long long base;
  if (a == 0) base = (long long)w * g.sxy + (long long)u * g.nx;          // u = j, w = k
  else if (a == 1) base = (long long)w * g.sxy + u;                       // u = i, w = k
  else base = (long long)w * g.nx + u;                                    // u = i, w = j
  float* F[6] = {f.ex, f.ey, f.ez, f.hx, f.hy, f.hz};
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const bool on_center = ((c % 3) == a) != (c >= 3);
    float* p = F[c] + base;
    if (on_center) {
      p[(long long)N * stride] = -p[(long long)(N - 1) * stride];
      p[(long long)(N + 1) * stride] = -p[(long long)(N - 2) * stride];
    } else {
      p[(long long)(N + 1) * stride] = p[(long long)(N - 1) * stride];
    }
  }
}

// plane copy as a kernel (ghost planes of a periodic z inside a captured graph: memcpy nodes did not capture, profiles/r3g)
__global__ __launch_bounds__(256) void copy_kernel(float* dst, const float* src, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = src[t];
}

// ghost-plane helpers (single-GPU z boundary conditions)
__global__ __launch_bounds__(256) void negate_copy_kernel(float* dst, const float* src, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = -src[t];
}

}  // namespace fdtd
