// x strips of a shell step (fdtd_capi.hip, shell pairs): ONE time step of the fused sweep (fused_step_kernel: H^{n+1/2} from
// set `a`, then E^{n+1}, both written to set `b`) over a box that is only a few cells wide along x — the x-CPML slab and
// its collar beside the bulk that the two-step sweep advances.  fused_step_kernel lays a wavefront along 256 cells of ONE
// row: on a 16-cell strip 60 of its 64 lanes would idle through ~450 instructions per plane.  Here a wavefront covers
// Q lanes x 4 cells of 64 / Q consecutive ROWS (Q = the strip's width in lanes: 4 for the 16 columns of a 12-layer slab and
// its collar, 5 for the 20 columns of step one, which reaches one lane further; lanes beyond Q * (64 / Q) idle); every quantity
// that fused_step_kernel keeps per row in SGPRs is per lane.  A strip is a few per cent of the cells, so this kernel is written for being right, not
// for the last register: all lanes load (idle ones from clamped addresses) and compute, only the stores are predicated —
// which also keeps every shuffle and barrier in uniform control flow.
//
// Same formulas in the same order as fused_step_kernel<MAT, ., PML = 1> (upd_h / upd_e, pml_h_apply, the E-side x recursion,
// the wall rules, the x-halo column with read-only psi): the same bits (tests/test_emu_shell.py).  The box's rows and planes
// lie clear of the y / z CPML slabs (those corners belong to the y / z slab launches of the shell), so x is the only axis
// whose recursions run here.  No periodic faces (a shell pair is not taken then).
#pragma once
#include "fdtd_kernels.hpp"

namespace fdtd {

constexpr int kStripWaves = 4;
constexpr int kStripMaxQ = 16;                            // lanes per row at most (64 columns per x tile)

struct StripP {
  int q;               // lanes per row (1 ... kStripMaxQ): 64 / q rows per wavefront, 4 * (64 / q) row slots per workgroup (slot 0: halo row)
  int xorg;            // first column of x tile 0 (a multiple of 4); tile t covers columns xorg + 4 q t ...
  int ci0, ci1;        // columns written: [ci0, ci1), multiples of 4
  int j0, j1;          // rows written: [j0, j1)
  int kbeg, kend;      // planes written
  int zchunk;          // planes marched per workgroup
  int nbx, nby, nbz;   // tiles
};

template <bool MAT, int OCC>      // OCC: workgroups per CU the register budget is cut for (3: <= 168 VGPRs, 4: <= 128)
__global__ __launch_bounds__(256, OCC) void strip_step_kernel(GridP g, FieldP a, FieldP b, StepP s, MatP m,
                                                         const PmlP* __restrict__ pmq, StripP sp, int pmc_z0) {
  constexpr int V = 4;
  const int Q = sp.q, rows_w = 64 / Q, n_slots = rows_w * kStripWaves;
  const int t = blockIdx.x;
  const int tile_y = t % sp.nby;
  const int tile_x = (t / sp.nby) % sp.nbx;
  const int tile_z = t / (sp.nby * sp.nbx);
  __shared__ float4 xch[2 * 2 * 64 * kStripWaves];         // [2 buffers][H_x, H_z][row slot][lane of the row]
  __shared__ float4 xco[8 * kStripMaxQ];                   // per lane of a row: {kv_h, b_h, c_h, kv_e, b_e, c_e, 1 / primal step, 1 / dual step} of its 4 cells
  __shared__ float2 lut_s[MAT ? kMaxMedia : 1];
  const int tx = threadIdx.x, ty = threadIdx.y;
  if constexpr (MAT) {
    for (int q = ty * 64 + tx; q < m.n_media; q += 256) lut_s[q] = m.lut[q];
  }
  const bool lane_on = tx < Q * rows_w;                    // (64 is not a multiple of 5: the last lanes of a wavefront idle)
  const int q = lane_on ? tx % Q : 0, r = lane_on ? tx / Q : 0;
  const int slot_i = ty * rows_w + r;
  const int me = ty * 64 + tx;
  const int mb = slot_i > 0 ? (r > 0 ? me - Q : (ty - 1) * 64 + (rows_w - 1) * Q + q) : me;     // the row below (slot 0 stores nothing: what it reads there is unused)
  constexpr int slot = 64 * kStripWaves;
  int j = sp.j0 + tile_y * (n_slots - 1) + slot_i - 1;
  const bool halo = slot_i == 0 || j >= sp.j1;
  const bool row_ok = lane_on && j >= 0 && j < g.ny;
  if (!row_ok) j = 0;                                      // keeps every address inside the arrays
  const int i0r = sp.xorg + (tile_x * Q + q) * V;
  const bool in_x = i0r < g.nx;
  const int i0 = in_x ? i0r : 0;
  const bool act = row_ok && in_x;
  const bool st_ok = act && !halo && i0 >= sp.ci0 && i0 < sp.ci1;
  const int k0 = sp.kbeg + tile_z * sp.zchunk;
  const int k1 = min(k0 + sp.zchunk, sp.kend);
  const float ch = g.ch;
  const bool last_x = (i0 + V >= g.nx);
  const bool first_x = (i0 == 0);
  const bool use_jp = (j + 1 < g.ny);
  const long long rowo = (long long)j * g.nx + i0;         // this lane's first cell in a plane
  const bool xh = act && q == 0 && !first_x;               // the tile's first lane recomputes H_{y,z} of column i0 - 1
  const int im = first_x ? 0 : i0 - 1;
  const long long rowm = (long long)j * g.nx + im;
  const bool wall_y = (j == 0) && (g.bcy0 == BC_PEC);
  const bool wall_x0 = first_x && (g.bcx0 == BC_PEC);
  const bool take_next = (q == Q - 1) || last_x;           // E of column i0 + 4: not in the next lane
  const bool take_prev = (q == 0) || first_x;              // H of column i0 - 1: not in the previous lane

  // x-CPML membership of this lane's four cells (the x ranges are multiples of 4 cells); the coefficients and steps of the
  // tile's columns are the same for every row: staged in LDS once, read where they are used
  int sx = -1, sx_m = -1;
  if (pmq) {
    const PmlAxisP& A = pmq->ax[0];
    if (act) sx = pml_si(A, i0);
    if (xh) sx_m = pml_si(A, im);
  }
  if (ty == 0 && tx < Q) {
    const int ic = sp.xorg + (tile_x * Q + tx) * V;
    const int icc = ic < g.nx ? ic : 0;
    float4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bool mem = pmq && pml_si(pmq->ax[0], icc) >= 0;
    xco[0 * kStripMaxQ + tx] = mem ? *reinterpret_cast<const float4*>(pmq->ax[0].kv_h + icc) : z4;
    xco[1 * kStripMaxQ + tx] = mem ? *reinterpret_cast<const float4*>(pmq->ax[0].b_h + icc) : z4;
    xco[2 * kStripMaxQ + tx] = mem ? *reinterpret_cast<const float4*>(pmq->ax[0].c_h + icc) : z4;
    xco[3 * kStripMaxQ + tx] = mem ? *reinterpret_cast<const float4*>(pmq->ax[0].kv_e + icc) : z4;
    xco[4 * kStripMaxQ + tx] = mem ? *reinterpret_cast<const float4*>(pmq->ax[0].b_e + icc) : z4;
    xco[5 * kStripMaxQ + tx] = mem ? *reinterpret_cast<const float4*>(pmq->ax[0].c_e + icc) : z4;
    xco[6 * kStripMaxQ + tx] = *reinterpret_cast<const float4*>(s.ipx + icc);
    xco[7 * kStripMaxQ + tx] = *reinterpret_cast<const float4*>(s.idx + icc);
  }
  __syncthreads();
  auto co = [&](int w, float (&o)[V]) { const float4 c4 = xco[w * kStripMaxQ + q]; o[0] = c4.x; o[1] = c4.y; o[2] = c4.z; o[3] = c4.w; };
  const float ipy = s.ipy[j], idy = s.idy[j], ipx_m = s.ipx[im];

  float exk[V], eyk[V], hxm[V], hym[V];
  zero<V>(hxm); zero<V>(hym);
  float exk_m = 0.f;
  {
    const long long p0 = (long long)k0 * g.sxy;
    ldv<V>(exk, a.ex + p0 + rowo);
    ldv<V>(eyk, a.ey + p0 + rowo);
    exk_m = a.ex[p0 + rowm];
  }
  // ---- prologue: H^{n+1/2}_{x,y}[k0-1] of the own cells ----
  {
    const bool skip = (pmc_z0 && k0 == 0);
    const long long pk = (long long)(k0 - 1) * g.sxy;
    float ezm[V], ezj[V], exm[V], eym[V], ho[V], hoy[V];
    zero<V>(ezj);
    ldv<V>(ezm, a.ez + pk + rowo);
    ldv<V>(exm, a.ex + pk + rowo);
    ldv<V>(eym, a.ey + pk + rowo);
    if (use_jp) ldv<V>(ezj, a.ez + pk + rowo + g.nx);
    float ezx = __shfl_down(ezm[0], 1);
    if (take_next) ezx = last_x ? 0.f : a.ez[pk + rowo + V];
    const float ipz = s.ipz[k0 - 1];
    ldv<V>(ho, a.hx + pk + rowo);
    ldv<V>(hoy, a.hy + pk + rowo);
    float ipx[V];
    co(6, ipx);
    if (!skip) {
      // corrected H^{n-1/2}_y of plane k0-1 (read-only psi; the plane's owner stores it): Hy += ch (kv dEz/dx + p1)
      if (sx >= 0 && (k0 > 0 || !g.pec_z0)) {
        const PmlAxisP& A = pmq->ax[0];
        const int kk = (k0 - 1 < 0) ? (g.psi_ghost ? g.nz : g.nz - 1) : k0 - 1;
        float s1[V], kvh[V], bh[V], chc[V];
        co(0, kvh); co(1, bh); co(2, chc);
        ldv<V>(s1, A.ph0 + ((long long)kk * g.ny + j) * A.ns + sx);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float ez_ip = (e + 1 < V) ? ezm[(e + 1) % V] : ezx;
          const float d2 = (ez_ip - ezm[e]) * ipx[e];
          const float p1 = bh[e] * s1[e] + chc[e] * d2;
          hoy[e] += ch * (kvh[e] * d2 + p1);
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
  }
  int cur = 0;
  for (int k = k0; k < k1; ++k) {
    const long long pk = (long long)k * g.sxy;
    const long long pb = pk + rowo;
    float exn[V], eyn[V], ezk[V], exj[V], ezj[V], hxn[V], hyn[V], hzn[V];
    const float ipz = s.ipz[k], idz = s.idz[k];
    ldv<V>(exn, a.ex + pb + g.sxy);
    ldv<V>(eyn, a.ey + pb + g.sxy);
    ldv<V>(ezk, a.ez + pb);
    if (use_jp) {
      ldv<V>(exj, a.ex + pb + g.nx);
      ldv<V>(ezj, a.ez + pb + g.nx);
    } else {
      zero<V>(exj); zero<V>(ezj);
    }
    ldv<V>(hxn, a.hx + pb);
    ldv<V>(hyn, a.hy + pb);
    ldv<V>(hzn, a.hz + pb);
    // column i0 - 1 (what the tile's first lane recomputes H_{y,z} of): loaded by every lane, with the plane's other loads —
    // the same cache lines, no second round trip for one lane in Q
    const long long pm = pk + rowm;
    const float exn_m = a.ex[pm + g.sxy];
    const float ez_mm = a.ez[pm], ey_mm = a.ey[pm];
    const float ex_jm = use_jp ? a.ex[pm + g.nx] : 0.f;
    float hy_o = a.hy[pm], hz_o = a.hz[pm];
    const bool wall_z = (k == 0) && g.pec_z0;
    // x-CPML state of the lane's cells in this plane
    float xh1[V], xh2[V], xe1[V], xe2[V];
    zero<V>(xh1); zero<V>(xh2); zero<V>(xe1); zero<V>(xe2);
    long long qx = 0;
    if (sx >= 0) {
      const PmlAxisP& A = pmq->ax[0];
      qx = ((long long)k * g.ny + j) * A.ns + sx;
      ldv<V>(xh1, A.ph0 + qx);
      ldv<V>(xh2, A.ph1 + qx);
      ldv<V>(xe1, A.pe0 + qx);
      ldv<V>(xe2, A.pe1 + qx);
    }
    [[maybe_unused]] uint32_t mw[V] = {kBgWord, kBgWord, kBgWord, kBgWord};
    if constexpr (MAT) ldm<V>(mw, m.m4 + pb);
    float eyx = __shfl_down(eyk[0], 1);
    float ezx = __shfl_down(ezk[0], 1);
    if (take_next) {
      if (!last_x) { eyx = a.ey[pb + V]; ezx = a.ez[pb + V]; }
      else { eyx = 0.f; ezx = 0.f; }
    }
    // ---- H-side x recursion: Hy += ch (kv dEz/dx + p1), Hz -= ch (kv dEy/dx + p2) ----
    float ipx[V];
    co(6, ipx);
    if (sx >= 0) {
      float kvh[V], bh[V], chc[V];
      co(0, kvh); co(1, bh); co(2, chc);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float ey_ip = (e + 1 < V) ? eyk[(e + 1) % V] : eyx;
        const float ez_ip = (e + 1 < V) ? ezk[(e + 1) % V] : ezx;
        pml_h_apply(hyn[e], hzn[e], (ey_ip - eyk[e]) * ipx[e], (ez_ip - ezk[e]) * ipx[e], xh1[e], xh2[e],
                    kvh[e], bh[e], chc[e], ch);
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float ey_ip = (e + 1 < V) ? eyk[(e + 1) % V] : eyx;
      const float ez_ip = (e + 1 < V) ? ezk[(e + 1) % V] : ezx;
      hxn[e] = upd_h(hxn[e], ch, ezj[e] - ezk[e], ipy, eyn[e] - eyk[e], ipz);
      hyn[e] = upd_h(hyn[e], ch, exn[e] - exk[e], ipz, ez_ip - ezk[e], ipx[e]);
      hzn[e] = upd_h(hzn[e], ch, ey_ip - eyk[e], ipx[e], exj[e] - exk[e], ipy);
    }
    // x-halo column: H^{n+1/2}_{y,z} at i0-1 recomputed by the tile's first lane
    float hy_m = 0.f, hz_m = 0.f;
    if (xh) {
      if (sx_m >= 0)
        pml_h_cell(hy_o, hz_o, (eyk[0] - ey_mm) * ipx_m, (ezk[0] - ez_mm) * ipx_m, pmq->ax[0],
                   pml_q(g, 0, pmq->ax[0].ns, im, j, k, sx_m), im, ch);
      hy_m = upd_h(hy_o, ch, exn_m - exk_m, ipz, ezk[0] - ez_mm, ipx_m);
      hz_m = upd_h(hz_o, ch, eyk[0] - ey_mm, ipx_m, ex_jm - exk_m, ipy);
    }
    {
      float4 t4;
      t4.x = hxn[0]; t4.y = hxn[1]; t4.z = hxn[2]; t4.w = hxn[3];
      xch[(cur * 2 + 0) * slot + me] = t4;
      t4.x = hzn[0]; t4.y = hzn[1]; t4.z = hzn[2]; t4.w = hzn[3];
      xch[(cur * 2 + 1) * slot + me] = t4;
    }
    __syncthreads();
    float hyx = __shfl_up(hyn[V - 1], 1);
    float hzx = __shfl_up(hzn[V - 1], 1);
    if (pmc_z0 && k == 0) {
#pragma unroll
      for (int e = 0; e < V; ++e) { hxm[e] = -hxn[e]; hym[e] = -hyn[e]; }
    }
    if (take_prev) {
      if (xh) { hyx = hy_m; hzx = hz_m; }
      else if (g.bcx0 == BC_PMC) { hyx = -hyn[0]; hzx = -hzn[0]; }
      else { hyx = 0.f; hzx = 0.f; }
    }
    float hxj[V], hzj[V];
    if (j > 0) {
      const float4 t0 = xch[(cur * 2 + 0) * slot + mb];
      const float4 t1 = xch[(cur * 2 + 1) * slot + mb];
      hxj[0] = t0.x; hxj[1] = t0.y; hxj[2] = t0.z; hxj[3] = t0.w;
      hzj[0] = t1.x; hzj[1] = t1.y; hzj[2] = t1.z; hzj[3] = t1.w;
    } else if (g.bcy0 == BC_PMC) {
#pragma unroll
      for (int e = 0; e < V; ++e) { hxj[e] = -hxn[e]; hzj[e] = -hzn[e]; }
    } else {
      zero<V>(hxj); zero<V>(hzj);
    }
    auto coef = [&](int c, int e) {
      if constexpr (MAT) return lut_s[(mw[e] >> (10 * c)) & 1023u];
      else return make_float2(m.ca1, m.cb1);
    };
    float ex[V], ey[V], ez[V], idx[V];
    co(7, idx);
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
    // ---- E-side x recursion: E_y -= cb (kv dHz/dx + p1), E_z += cb (kv dHy/dx + p2) ----
    if (sx >= 0) {
      float kve[V], be[V], cec[V];
      co(3, kve); co(4, be); co(5, cec);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const bool wx = wall_x0 && (e == 0);
        if (!wx) {
          const float hy_im = (e > 0) ? hyn[(e + V - 1) % V] : hyx;
          const float hz_im = (e > 0) ? hzn[(e + V - 1) % V] : hzx;
          const float d1 = (hyn[e] - hy_im) * idx[e];
          const float d2 = (hzn[e] - hz_im) * idx[e];
          const float p1 = be[e] * xe1[e] + cec[e] * d2;
          const float p2 = be[e] * xe2[e] + cec[e] * d1;
          xe1[e] = p1; xe2[e] = p2;
          if (!wall_z) ey[e] -= coef(1, e).y * (kve[e] * d2 + p1);     // E_y is tangential to the z wall
          if (!wall_y) ez[e] += coef(2, e).y * (kve[e] * d1 + p2);     // E_z is tangential to the y wall
        }
      }
    }
    if (st_ok) {
      if (sx >= 0) {
        const PmlAxisP& A = pmq->ax[0];
        stv<V>(A.pe0n + qx, xe1);
        stv<V>(A.pe1n + qx, xe2);
        stv<V>(A.ph0n + qx, xh1);
        stv<V>(A.ph1n + qx, xh2);
      }
      stv<V>(b.hx + pb, hxn);
      stv<V>(b.hy + pb, hyn);
      stv<V>(b.hz + pb, hzn);
      stv<V>(b.ex + pb, ex);
      stv<V>(b.ey + pb, ey);
      stv<V>(b.ez + pb, ez);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { hxm[e] = hxn[e]; hym[e] = hyn[e]; exk[e] = exn[e]; eyk[e] = eyn[e]; }
    exk_m = exn_m;
    cur ^= 1;
  }
}

}  // namespace fdtd
