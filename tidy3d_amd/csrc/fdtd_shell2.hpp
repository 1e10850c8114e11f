// Two time steps per sweep WITH the CPML recursions carried through both of them (round 5), gfx950.
//
// fused2_step_kernel (fdtd_kernels2.hpp) advances the bulk of a CPML-walled grid by two steps per pass; until round 4 the shell
// around it — the CPML slabs and their two-cell collar — took two single steps of the production kernels through a third field
// set, which moved its fields and psi twice per pair (12.7 GB per 512^3 pair against 7.4 GB for the bulk alone).
// shell2_step_kernel advances a BOX of the shell by both steps in one pass: it reads set `a` (E^n, H^{n-1/2}) and the psi READ
// sets (psi_H^{n-1/2}, psi_E^n), keeps E^{n+1}, H^{n+1/2}, psi_H^{n+1/2}, psi_E^{n+1} in registers and LDS, and writes set `b`
// (E^{n+2}, H^{n+3/2}) and the psi WRITE sets (both sides are ping-ponged: halo rows / planes / lanes of OTHER workgroups
// recompute what this one owns from the old values).  Same pipeline as fused2_step_kernel:
//   iteration k:  S1  H1[k]   = H^{n+1/2}[k]    (H-side recursions of step n,   axes x, y, z, on H^{n-1/2})       publish H1_{x,z}
//                 ---- barrier ----
//                 S2  E1[k]   = E^{n+1}[k]      (E-side recursions of step n,   axes y, z, x)                      publish E1_{x,z}
//                 S3  H2[k-1] = H^{n+3/2}[k-1]  (H-side recursions of step n+1 on H1[k-1], from E1[k-1], E1[k])    publish H2_{x,z}
//                 ---- barrier ----
//                 S4  E2[k-1] = E^{n+2}[k-1]    (E-side recursions of step n+1)            store E2, H2, psi_H^{n+3/2}, psi_E^{n+2}
// with the update formulas, the recursion arithmetic, the axis order and the wall rules of fused_step_kernel<MAT, ., PML = 7>
// (fdtd_kernels.hpp) for each of the two steps: the same bits as two single steps (tests/test_emu_shell2.py on the emulator,
// tests/test_gpu_shell_pairs.py on the device).
//
// Lane layout (as strip_step_kernel): a wavefront covers Q lanes x 4 cells of 64 / Q consecutive ROWS, so that one kernel serves
// the wide boxes (z slabs, y slabs: Q = 16 ... 64) and the x strips (16 columns + a halo lane: Q = 5).  Every quantity
// fused2_step_kernel keeps per row in SGPRs is per lane here; all lanes load (idle ones from clamped addresses) and compute,
// only the stores are predicated — every lane shift and barrier stays in uniform control flow.
//   rows:    row slots 0 .. S-1 of a workgroup (S = (64 / Q) * waves) are rows j0 - 2 .. j0 + S - 3 of its tile; slots 2 .. S-2
//            are written (two halo rows below, one above, recomputed — as in fused2_step_kernel).  A box on the y-min wall that fits one
//            tile row starts with slot 0 on row 0 (Shell2P::jlo = 0): nothing lies below the wall.
//   columns: x tiles OVERLAP by two lanes: lane 0 and lane Q-1 of a tile are halo lanes (what they hold after two steps is wrong
//            two / one cells deep) unless they sit on an x wall — no seam scratch, no seam kernel, no edge-column loads.
//   planes:  a chunk [k0, k1) runs iterations k0-1 .. k1; a prologue supplies H1_{x,y}[k0-2] from read-only psi.
// Scope (fdtd_capi.hip checks it): PEC walls behind the layers (PMC allowed on min faces); a periodic x wraps through halo lanes, a
// periodic y keeps the boxes two rows clear of the wrap (those rows take single steps beside them, as z holes do); non-dispersive
// media in the boxes (uniform or packed medium words), no sources / monitors inside the boxes while a pair is taken.
#pragma once
#include "fdtd_kernels2.hpp"
#include "fdtd_shell2_host.hpp"

namespace fdtd {

// AXES: the axes whose recursions the boxes of this launch can meet (bit a = axis a): a box off the slabs of an axis — its halo
// rows / lanes / planes then lie off them too (the collar) — runs an instantiation without that axis' 32 psi registers
template <bool MAT, int AXES>
__global__ __launch_bounds__(512, (AXES == 7 ? 2 : 3)) void shell2_step_kernel(GridP g, FieldP a, FieldP b, StepP s, MatP m,
                                                          const PmlP* __restrict__ pmq, Shell2M boxes, Shell2Dump dmp, DispP dp, SrcP sr) {
  constexpr int V = 4;
  // ONE launch covers all boxes of the shell (the workgroups of box q are [first[q], first[q + 1])): six small launches one
  // behind the other on a stream left the machine half empty between them (profiles/r5)
  int bq = 0;
#pragma unroll
  for (int q = 1; q < kShell2Boxes; ++q) if (q < boxes.n && (int)blockIdx.x >= boxes.first[q]) bq = q;
  const Shell2P& sp = boxes.box[bq];
  const int Q = sp.q, RW = 64 / Q;
  const int W = blockDim.y;
  const int S = RW * W;                                  // row slots of the workgroup
  const int SL = W * 64;                                 // float4 entries per exchange array
  const int t = (int)blockIdx.x - boxes.first[bq];
  const int tile_y = t % sp.nby;
  const int tile_x = (t / sp.nby) % sp.nbx;
  const int tile_z = t / (sp.nby * sp.nbx);
  HIP_DYNAMIC_SHARED(float4, lds)
  float4* xch = lds;                                     // [8][SL]: H1_x H1_z | H2_x H2_z | E1_x E1_z (buffer 0) | E1_x E1_z (buffer 1)
  float4* xco = lds + 8 * SL;                            // [8][64]: per lane of a row {kv_h, b_h, c_h, kv_e, b_e, c_e, 1 / primal step, 1 / dual step} of its 4 cells
  float4* yco = xco + 8 * kShell2MaxQ;                   // [2][32 W]: per row slot {1/kappa - 1, b, c, 0} of the H side / the E side (zero: not a member)
  __shared__ float2 lut_s[MAT ? kMaxMedia : 1];
  const int tx = threadIdx.x, ty = threadIdx.y;
  if constexpr (MAT) {
    for (int q = ty * 64 + tx; q < m.n_media; q += SL) lut_s[q] = m.lut[q];
  }
  const bool lane_on = tx < Q * RW;                      // (64 is not a multiple of every Q: the last lanes of a wavefront idle)
  const int q = lane_on ? tx % Q : 0, r = lane_on ? tx / Q : 0;
  const int slot_i = ty * RW + r;
  const int me = lane_on ? slot_i * Q + q : 0;             // compact index: the row below is me - Q, the row above me + Q (idle lanes publish nothing)
  const int mb = slot_i > 0 ? me - Q : me;               // (slot 0 / the top slot read their own entry: what they form from it is never used)
  const int ma = slot_i < S - 1 ? me + Q : me;
  const int R = S - 1 - sp.jlo;
  const int jr = sp.j0 + tile_y * R + slot_i - sp.jlo;        // the lane's row
  const bool row_ok = lane_on && jr >= 0 && jr < g.ny;
  const int j = row_ok ? jr : 0;                         // (keeps every address inside the arrays)
  const int i0r = sp.xorg + (tile_x * (Q - 2) + q) * V;
  // periodic x (a box that spans the whole row): the row has no wall — the tiles start one lane left of column 0 and end one lane
  // right of column nx - 1, and those two lanes are halo lanes like any other tile edge, holding the wrapped columns
  const bool per_x = g.bcx0 == BC_PERIODIC;
  const bool in_x = per_x ? (i0r <= g.nx) : (i0r < g.nx);
  const int i0 = in_x ? (per_x ? (i0r < 0 ? i0r + g.nx : (i0r >= g.nx ? i0r - g.nx : i0r)) : i0r) : 0;
  const bool act = row_ok && in_x;
  const bool last_x = !per_x && in_x && (i0 + V >= g.nx);
  const bool first_x = !per_x && in_x && (i0 == 0);
  const bool own_row = slot_i >= sp.jlo && slot_i <= S - 2 && row_ok && jr < sp.j1;
  // a lane on a tile edge holds wrong values two / one cells deep after two steps (its neighbour lane belongs to the next tile) — unless it sits on a wall
  const bool own_col = act && i0r >= sp.ci0 && i0r < sp.ci1 && (q >= 1 || first_x) && (q <= Q - 2 || last_x);
  const bool st_lane = own_row && own_col;
  const bool take_next = (q == Q - 1) || last_x || !lane_on;      // E of column i0 + 4 is not in the next lane
  const bool take_prev = (q == 0) || first_x || !lane_on;         // H of column i0 - 1 is not in the previous lane
  const int kc0 = sp.k0 + tile_z * sp.zchunk;
  const int kc1 = min(kc0 + sp.zchunk, sp.k1);
  const int kA = kc0 > 0 ? kc0 - 1 : 0;
  const float ch = g.ch;
  const bool use_jp = (j + 1 < g.ny);
  const unsigned ob = (unsigned)(j * g.nx + i0) * 4u;    // the lane's byte offset inside a plane
  const bool pmc_x0 = g.bcx0 == BC_PMC, pmc_y0 = g.bcy0 == BC_PMC, pmc_z0 = !g.pec_z0;
  const bool wall_y = (j == 0) && !pmc_y0;
  const bool wall_x0 = first_x && !pmc_x0;
  const float ipy = s.ipy[j], idy = s.idy[j];
  // CPML membership: x per lane (the x ranges are multiples of 4 cells), y per row, z per plane
  const PmlAxisP& AX = pmq->ax[0];
  const PmlAxisP& AY = pmq->ax[1];
  const PmlAxisP& AZ = pmq->ax[2];
  const int sx = ((AXES & 1) && act) ? pml_si(AX, i0) : -1;
  const int sy = ((AXES & 2) && row_ok) ? pml_si(AY, j) : -1;
  const unsigned oxb = (unsigned)(j * AX.ns + max(sx, 0)) * 4u;            // the lane's byte offset inside a plane of the x psi arrays
  const unsigned oyb = (unsigned)(max(sy, 0) * g.nx + i0) * 4u;            //                                      ... of the y psi arrays
  const long long xpl = (long long)g.ny * AX.ns, ypl = (long long)AY.ns * g.nx;   // entries per plane of them
  if (ty == 0 && tx < Q) {
    const int ic = sp.xorg + (tile_x * (Q - 2) + tx) * V;
    const int icc = per_x ? (ic < 0 ? ic + g.nx : (ic == g.nx ? 0 : (ic < g.nx ? ic : 0))) : (ic < g.nx ? ic : 0);
    const float4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bool mem = pml_si(AX, icc) >= 0;
    xco[0 * kShell2MaxQ + tx] = mem ? *reinterpret_cast<const float4*>(AX.kv_h + icc) : z4;
    xco[1 * kShell2MaxQ + tx] = mem ? *reinterpret_cast<const float4*>(AX.b_h + icc) : z4;
    xco[2 * kShell2MaxQ + tx] = mem ? *reinterpret_cast<const float4*>(AX.c_h + icc) : z4;
    xco[3 * kShell2MaxQ + tx] = mem ? *reinterpret_cast<const float4*>(AX.kv_e + icc) : z4;
    xco[4 * kShell2MaxQ + tx] = mem ? *reinterpret_cast<const float4*>(AX.b_e + icc) : z4;
    xco[5 * kShell2MaxQ + tx] = mem ? *reinterpret_cast<const float4*>(AX.c_e + icc) : z4;
    xco[6 * kShell2MaxQ + tx] = *reinterpret_cast<const float4*>(s.ipx + icc);
    xco[7 * kShell2MaxQ + tx] = *reinterpret_cast<const float4*>(s.idx + icc);
  }
  if (lane_on && q == 0) {
    const float4 z4 = {0.f, 0.f, 0.f, 0.f};
    yco[slot_i] = sy >= 0 ? ldc_f4(AY.ch4 + j) : z4;
    yco[32 * W + slot_i] = sy >= 0 ? ldc_f4(AY.ce4 + j) : z4;
  }
  {
    const float4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) xch[w * SL + ty * 64 + tx] = z4;
  }
  __syncthreads();
  auto co = [&](int w, float (&o)[V]) __attribute__((always_inline)) {
    const float4 c4 = xco[w * kShell2MaxQ + q]; o[0] = c4.x; o[1] = c4.y; o[2] = c4.z; o[3] = c4.w;
  };
  auto put = [&](int w, const float (&v)[V]) __attribute__((always_inline)) {
    float4 t4; t4.x = v[0]; t4.y = v[1]; t4.z = v[2]; t4.w = v[3];
    if (lane_on) xch[w * SL + me] = t4;
  };
  auto get = [&](int w, int at_i, float (&o)[V]) __attribute__((always_inline)) {
    const float4 t4 = xch[w * SL + at_i]; o[0] = t4.x; o[1] = t4.y; o[2] = t4.z; o[3] = t4.w;
  };
  const float ca = m.ca1, cb = m.cb1;

  // carried along the march
  float exk[V], eyk[V];                    // E^n_{x,y}[k]
  float h1x[V], h1y[V], h1z[V];            // H1[k-1]
  float e1x[V], e1y[V], e1z[V];            // E1[k-1]
  float h2xm[V], h2ym[V];                  // H2_{x,y}[k-2]
  float pxh1[V], pxh2[V], pyh1[V], pyh2[V], pzh1[V], pzh2[V];     // psi_H^{n+1/2}[k-1] (x: H_y H_z, y: H_z H_x, z: H_x H_y)
  float pxe1[V], pxe2[V], pye1[V], pye2[V], pze1[V], pze2[V];     // psi_E^{n+1}[k-1]   (x: E_y E_z, y: E_z E_x, z: E_x E_y)
  zero<V>(h1x); zero<V>(h1y); zero<V>(h1z); zero<V>(e1x); zero<V>(e1y); zero<V>(e1z); zero<V>(h2xm); zero<V>(h2ym);
  zero<V>(pxh1); zero<V>(pxh2); zero<V>(pyh1); zero<V>(pyh2); zero<V>(pzh1); zero<V>(pzh2);
  zero<V>(pxe1); zero<V>(pxe2); zero<V>(pye1); zero<V>(pye2); zero<V>(pze1); zero<V>(pze2);
  float ipz_m = 0.f, idz_m = 0.f;          // 1 / steps of plane k-1
  [[maybe_unused]] uint32_t mwm[V] = {kBgWord, kBgWord, kBgWord, kBgWord};       // packed medium words of plane k-1 (S4)
  int ss_m = -1;                           // block of the lane's row segment in the paged source terms, plane k-1 (S3, S4)
  const bool box_src = sr.sseg && (sp.paged & 1), box_disp = dp.dseg && (sp.paged & 2);
  int sz_m = -1;                           // z membership of plane k-1 and its coefficients
  float4 czh_m = {0.f, 0.f, 0.f, 0.f}, cze_m = {0.f, 0.f, 0.f, 0.f};
  {
    const long long p0 = (long long)kA * g.sxy;
    ldf<V, true>(exk, uni(a.ex + p0), ob);
    ldf<V, true>(eyk, uni(a.ey + p0), ob);
  }
  // ---- prologue: H1_{x,y}[kA-1], with the H-side recursions of step n from read-only psi (the plane's owner stores them) -----
  if (kA > 0) {
    const int kk = kA - 1;
    const long long pk = (long long)kk * g.sxy;
    float ezm[V], ezj[V], exm[V], eym[V], ho[V], hoy[V], ipx[V];
    zero<V>(ezj);
    ldf<V, true>(ezm, uni(a.ez + pk), ob);
    ldf<V, true>(exm, uni(a.ex + pk), ob);
    ldf<V, true>(eym, uni(a.ey + pk), ob);
    if (use_jp) ldf<V, true>(ezj, uni(a.ez + pk + g.nx), ob);
    float ezx = lane_next(ezm[0]);
    if (take_next) ezx = 0.f;
    const float ipz = s.ipz[kk];
    ldf<V, true>(ho, uni(a.hx + pk), ob);
    ldf<V, true>(hoy, uni(a.hy + pk), ob);
    co(6, ipx);
    // axis x: Hy += ch (kv dEz/dx + p1)
    if (sx >= 0) {
      float s1[V], kv[V], bb[V], cc[V];
      ldg4(s1, uni(AX.ph0 + (long long)kk * xpl), oxb);
      co(0, kv); co(1, bb); co(2, cc);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float ez_ip = (e + 1 < V) ? ezm[(e + 1) % V] : ezx;
        const float d2 = (ez_ip - ezm[e]) * ipx[e];
        const float p1 = bb[e] * s1[e] + cc[e] * d2;
        hoy[e] += ch * (kv[e] * d2 + p1);
      }
    }
    // axis y: Hx -= ch (kv dEz/dy + p2)
    if (sy >= 0) {
      const float4 cf = yco[slot_i];
      float s2[V];
      ldg4(s2, uni(AY.ph1 + (long long)kk * ypl), oyb);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float d1 = (ezj[e] - ezm[e]) * ipy;
        const float p2 = cf.y * s2[e] + cf.z * d1;
        ho[e] -= ch * (cf.x * d1 + p2);
      }
    }
    // axis z: Hx += ch (kv dEy/dz + p1), Hy -= ch (kv dEx/dz + p2)
    {
      const int sz = (AXES & 4) ? pml_si(AZ, kk) : -1;
      if (sz >= 0) {
        const float4 cf = ldc_f4(AZ.ch4 + kk);
        float s1[V], s2[V];
        ldg4(s1, uni(AZ.ph0 + (long long)sz * g.sxy), ob);
        ldg4(s2, uni(AZ.ph1 + (long long)sz * g.sxy), ob);
#pragma unroll
        for (int e = 0; e < V; ++e)
          pml_h_apply(ho[e], hoy[e], (exk[e] - exm[e]) * ipz, (eyk[e] - eym[e]) * ipz, s1[e], s2[e], cf.x, cf.y, cf.z, ch);
      }
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
  for (int k = kA; k <= kc1; ++k) {
    // (iteration k = nz, last chunk only: plane nz is the z-max wall, E1_{x,y}[nz] = 0 is all it contributes; its loads read the
    //  ghost plane, its other results are never used; no psi there)
    const bool in_z = k < g.nz;
    const long long pk = (long long)k * g.sxy;
    const long long up = in_z ? g.sxy : 0;
    const float ipz = s.ipz[k], idz = s.idz[k];          // (the step arrays carry one ghost entry at each end)
    const int sz = ((AXES & 4) && in_z) ? pml_si(AZ, k) : -1;
    float4 czh = {0.f, 0.f, 0.f, 0.f}, cze = {0.f, 0.f, 0.f, 0.f};
    if (sz >= 0) { czh = ldc_f4(AZ.ch4 + k); cze = ldc_f4(AZ.ce4 + k); }
    const bool mem_x = sx >= 0 && in_z, mem_y = sy >= 0 && in_z;
    int ss = -1;                             // block of the lane's row segment in the paged source terms, this plane
    float exn[V], eyn[V], ezk[V], exj[V], ezj[V], hxn[V], hyn[V], hzn[V];
    ldf<V, true>(exn, uni(a.ex + pk + up), ob);
    ldf<V, true>(eyn, uni(a.ey + pk + up), ob);
    ldf<V, true>(ezk, uni(a.ez + pk), ob);
    if (use_jp) {
      ldf<V, true>(exj, uni(a.ex + pk + g.nx), ob);
      ldf<V, true>(ezj, uni(a.ez + pk + g.nx), ob);
    } else {
      zero<V>(exj); zero<V>(ezj);
    }
    ldf<V, true>(hxn, uni(a.hx + pk), ob);
    ldf<V, true>(hyn, uni(a.hy + pk), ob);
    ldf<V, true>(hzn, uni(a.hz + pk), ob);
    // psi of this plane (read sets): every load goes out here, with the field loads — one memory round trip per plane
    float xh1[V], xh2[V], yh1[V], yh2[V], zh1[V], zh2[V], xe1[V], xe2[V], ye1[V], ye2[V], ze1[V], ze2[V];
    zero<V>(xh1); zero<V>(xh2); zero<V>(yh1); zero<V>(yh2); zero<V>(zh1); zero<V>(zh2);
    zero<V>(xe1); zero<V>(xe2); zero<V>(ye1); zero<V>(ye2); zero<V>(ze1); zero<V>(ze2);
    if (mem_x) {
      ldg4(xh1, uni(AX.ph0 + (long long)k * xpl), oxb); ldg4(xh2, uni(AX.ph1 + (long long)k * xpl), oxb);
      ldg4(xe1, uni(AX.pe0 + (long long)k * xpl), oxb); ldg4(xe2, uni(AX.pe1 + (long long)k * xpl), oxb);
    }
    if (mem_y) {
      ldg4(yh1, uni(AY.ph0 + (long long)k * ypl), oyb); ldg4(yh2, uni(AY.ph1 + (long long)k * ypl), oyb);
      ldg4(ye1, uni(AY.pe0 + (long long)k * ypl), oyb); ldg4(ye2, uni(AY.pe1 + (long long)k * ypl), oyb);
    }
    if (sz >= 0) {
      ldg4(zh1, uni(AZ.ph0 + (long long)sz * g.sxy), ob); ldg4(zh2, uni(AZ.ph1 + (long long)sz * g.sxy), ob);
      ldg4(ze1, uni(AZ.pe0 + (long long)sz * g.sxy), ob); ldg4(ze2, uni(AZ.pe1 + (long long)sz * g.sxy), ob);
    }
    [[maybe_unused]] uint32_t mw[V] = {kBgWord, kBgWord, kBgWord, kBgWord};
    if constexpr (MAT) ldm<V>(mw, at(uni(m.m4 + pk), ob));
    float ipx[V], idx[V];
    co(6, ipx); co(7, idx);
    // ---- S1: H1[k] = H^{n+1/2}[k]; H-side recursions of step n in the order x, y, z --------------------------------------------
    {
      float eyx = lane_next(eyk[0]);
      float ezx = lane_next(ezk[0]);
      if (take_next) { eyx = 0.f; ezx = 0.f; }           // the x-max wall, or a halo lane (whatever it forms is never stored)
      if (mem_x) {
        float kv[V], bb[V], cc[V];
        co(0, kv); co(1, bb); co(2, cc);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float ey_ip = (e + 1 < V) ? eyk[(e + 1) % V] : eyx;
          const float ez_ip = (e + 1 < V) ? ezk[(e + 1) % V] : ezx;
          pml_h_apply(hyn[e], hzn[e], (ey_ip - eyk[e]) * ipx[e], (ez_ip - ezk[e]) * ipx[e], xh1[e], xh2[e], kv[e], bb[e], cc[e], ch);
        }
      }
      if (mem_y) {
        const float4 cf = yco[slot_i];
#pragma unroll
        for (int e = 0; e < V; ++e)
          pml_h_apply(hzn[e], hxn[e], (ezj[e] - ezk[e]) * ipy, (exj[e] - exk[e]) * ipy, yh1[e], yh2[e], cf.x, cf.y, cf.z, ch);
      }
      if (sz >= 0) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          pml_h_apply(hxn[e], hyn[e], (exn[e] - exk[e]) * ipz, (eyn[e] - eyk[e]) * ipz, zh1[e], zh2[e], czh.x, czh.y, czh.z, ch);
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float ey_ip = (e + 1 < V) ? eyk[(e + 1) % V] : eyx;
        const float ez_ip = (e + 1 < V) ? ezk[(e + 1) % V] : ezx;
        hxn[e] = upd_h(hxn[e], ch, ezj[e] - ezk[e], ipy, eyn[e] - eyk[e], ipz);
        hyn[e] = upd_h(hyn[e], ch, exn[e] - exk[e], ipz, ez_ip - ezk[e], ipx[e]);
        hzn[e] = upd_h(hzn[e], ch, ey_ip - eyk[e], ipx[e], exj[e] - exk[e], ipy);
      }
    }
    put(0, hxn); put(1, hzn);
    __syncthreads();
    // ---- S2: E1[k] = E^{n+1}[k]; E-side recursions of step n in the order y, z, x ----------------------------------------------
    float e1xn[V], e1yn[V], e1zn[V];
    auto coef = [&](const uint32_t (&w)[V], int c, int e) __attribute__((always_inline)) {
      if constexpr (MAT) return lut_s[(w[e] >> (10 * c)) & 1023u];
      else return make_float2(ca, cb);
    };
    {
      float hyx = lane_prev(hyn[V - 1]);
      float hzx = lane_prev(hzn[V - 1]);
      if (take_prev) {
        if (first_x && pmc_x0) { hyx = -hyn[0]; hzx = -hzn[0]; }
        else { hyx = 0.f; hzx = 0.f; }                    // the x-min wall, or a halo lane
      }
      float hxj[V], hzj[V];
      if (j > 0) { get(0, mb, hxj); get(1, mb, hzj); }
      else if (pmc_y0) {
#pragma unroll
        for (int e = 0; e < V; ++e) { hxj[e] = -hxn[e]; hzj[e] = -hzn[e]; }
      } else { zero<V>(hxj); zero<V>(hzj); }
      if (pmc_z0 && k == 0) {
#pragma unroll
        for (int e = 0; e < V; ++e) { h1x[e] = -hxn[e]; h1y[e] = -hyn[e]; }
      }
      const bool wall_z = (k == 0 && !pmc_z0) || !in_z;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float hy_im = (e > 0) ? hyn[(e + V - 1) % V] : hyx;
        const float hz_im = (e > 0) ? hzn[(e + V - 1) % V] : hzx;
        float nex = upd_e(exk[e], coef(mw, 0, e).x, coef(mw, 0, e).y, hzn[e] - hzj[e], idy, hyn[e] - h1y[e], idz);
        float ney = upd_e(eyk[e], coef(mw, 1, e).x, coef(mw, 1, e).y, hxn[e] - h1x[e], idz, hzn[e] - hz_im, idx[e]);
        float nez = upd_e(ezk[e], coef(mw, 2, e).x, coef(mw, 2, e).y, hyn[e] - hy_im, idx[e], hxn[e] - hxj[e], idy);
        const bool wx = wall_x0 && (e == 0);
        if (wall_y || wall_z) nex = 0.f;
        if (wx || wall_z) ney = 0.f;
        if (wx || wall_y) nez = 0.f;
        e1xn[e] = nex; e1yn[e] = ney; e1zn[e] = nez;
      }
      // axis y: E_z -= cb (kv dHx/dy + p1),  E_x += cb (kv dHz/dy + p2)
      if (mem_y && !wall_y) {
        const float4 cf = yco[32 * W + slot_i];
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float d1 = (hzn[e] - hzj[e]) * idy;
          const float d2 = (hxn[e] - hxj[e]) * idy;
          const float p1 = cf.y * ye1[e] + cf.z * d2;
          const float p2 = cf.y * ye2[e] + cf.z * d1;
          ye1[e] = p1; ye2[e] = p2;
          const bool wx = wall_x0 && (e == 0);
          if (!wx) e1zn[e] -= coef(mw, 2, e).y * (cf.x * d2 + p1);
          if (!wall_z) e1xn[e] += coef(mw, 0, e).y * (cf.x * d1 + p2);
        }
      }
      // axis z: E_x -= cb (kv dHy/dz + p1),  E_y += cb (kv dHx/dz + p2)
      if (sz >= 0 && !wall_z) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float d1 = (hxn[e] - h1x[e]) * idz;
          const float d2 = (hyn[e] - h1y[e]) * idz;
          const float p1 = cze.y * ze1[e] + cze.z * d2;
          const float p2 = cze.y * ze2[e] + cze.z * d1;
          ze1[e] = p1; ze2[e] = p2;
          const bool wx = wall_x0 && (e == 0);
          if (!wall_y) e1xn[e] -= coef(mw, 0, e).y * (cze.x * d2 + p1);
          if (!wx) e1yn[e] += coef(mw, 1, e).y * (cze.x * d1 + p2);
        }
      }
      // axis x: E_y -= cb (kv dHz/dx + p1),  E_z += cb (kv dHy/dx + p2)
      if (mem_x) {
        float kv[V], bb[V], cc[V];
        co(3, kv); co(4, bb); co(5, cc);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const bool wx = wall_x0 && (e == 0);
          if (!wx) {
            const float hy_im = (e > 0) ? hyn[(e + V - 1) % V] : hyx;
            const float hz_im = (e > 0) ? hzn[(e + V - 1) % V] : hzx;
            const float d1 = (hyn[e] - hy_im) * idx[e];
            const float d2 = (hzn[e] - hz_im) * idx[e];
            const float p1 = bb[e] * xe1[e] + cc[e] * d2;
            const float p2 = bb[e] * xe2[e] + cc[e] * d1;
            xe1[e] = p1; xe2[e] = p2;
            if (!wall_z) e1yn[e] -= coef(mw, 1, e).y * (kv[e] * d2 + p1);
            if (!wall_y) e1zn[e] += coef(mw, 2, e).y * (kv[e] * d1 + p2);
          }
        }
      }
      if (!row_ok) { zero<V>(e1xn); zero<V>(e1zn); }      // rows beyond the grid publish E = 0 (the wall)
      // paged source terms (round 6, fdtd_fused2.hpp SrcP: a mode plane / current sheet that runs through the layers while it injects):
      // the E-side terms of step n, behind the E-side recursions as launch_sources follows the sweep
      if (box_src && act && in_z) {
        ss = sr.sseg[((long long)k * g.ny + j) * ((g.nx + 255) >> 8) + (i0 >> 8)];
        if (ss >= 0) {
          const long long qs = ((long long)ss * 3) * 256 + (i0 & 255);
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
      // dispersive cells inside the shell (round 6, as fused2_step_kernel's OPT bit 5): E^{n+1} <- E^{n+1} - cc S(Q^n) from the paged
      // memory terms, last of all; the lane that owns the cells leaves E^{n+1} for ade2_kernel
      if (box_disp && act && in_z) {
        const int ds = dp.dseg[((long long)k * g.ny + j) * ((g.nx + 255) >> 8) + (i0 >> 8)];
        if (ds >= 0) {
          const long long qb = ((long long)ds * 3) * 256 + (i0 & 255);
          float cx[V], cy[V], cz[V];
          ldv<V>(cx, dp.cs + qb); ldv<V>(cy, dp.cs + qb + 256); ldv<V>(cz, dp.cs + qb + 512);
#pragma unroll
          for (int e = 0; e < V; ++e) { e1xn[e] = e1xn[e] - cx[e]; e1yn[e] = e1yn[e] - cy[e]; e1zn[e] = e1zn[e] - cz[e]; }
          if (st_lane && k >= kc0 && k < kc1) { stv<V>(dp.e1 + qb, e1xn); stv<V>(dp.e1 + qb + 256, e1yn); stv<V>(dp.e1 + qb + 512, e1zn); }
        }
      }
      // the middle step over the boxes of DFT monitors that reach into the shell (fused2_step_kernel does the same over the bulk):
      // H^{n+1/2} for records at step n, E^{n+1} for records at step n + 1 — of the cells this box owns
      if (dmp.dstart && st_lane && k >= kc0 && k < kc1) {
        for (int qd = dmp.dstart[k]; qd < dmp.dstart[k + 1]; ++qd) {
          const DumpBox bx = dmp.dboxes[dmp.dlist[qd]];
          const int ly = j - bx.lo1, lz = k - bx.lo2;
          if (ly >= 0 && ly < bx.ny) {
            const long long rowd = ((long long)lz * bx.ny + ly) * bx.nx;
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const int lx = i0 + e - bx.lo0;
              if (lx >= 0 && lx < bx.nx) {
                if (bx.off[0] >= 0) dmp.dump[bx.off[0] + rowd + lx] = e1xn[e];
                if (bx.off[1] >= 0) dmp.dump[bx.off[1] + rowd + lx] = e1yn[e];
                if (bx.off[2] >= 0) dmp.dump[bx.off[2] + rowd + lx] = e1zn[e];
                if (bx.off[3] >= 0) dmp.dump[bx.off[3] + rowd + lx] = hxn[e];
                if (bx.off[4] >= 0) dmp.dump[bx.off[4] + rowd + lx] = hyn[e];
                if (bx.off[5] >= 0) dmp.dump[bx.off[5] + rowd + lx] = hzn[e];
              }
            }
          }
        }
      }
    }
    // ---- S3: H2[k-1] = H^{n+3/2}[k-1]; H-side recursions of step n+1 on H1[k-1] -----------------------------------------------
    float h2x[V], h2y[V], h2z[V];
    {
      float eyx = lane_next(e1y[0]);
      float ezx = lane_next(e1z[0]);
      if (take_next) { eyx = 0.f; ezx = 0.f; }
      float exj1[V], ezj1[V];
      get(4 + (cur ^ 1) * 2 + 0, ma, exj1);
      get(4 + (cur ^ 1) * 2 + 1, ma, ezj1);
      // paged H-side source terms of step n+1 on H^{n+1/2}[k-1], in front of the H-side recursions (launch_sources precedes the sweep)
      if (box_src && k > kA) {
        if (ss_m >= 0 && sr.t->use_h2) {
          const long long qs = ((long long)ss_m * 3) * 256 + (i0 & 255);
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
      if (sx >= 0) {
        float kv[V], bb[V], cc[V];
        co(0, kv); co(1, bb); co(2, cc);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float ey_ip = (e + 1 < V) ? e1y[(e + 1) % V] : eyx;
          const float ez_ip = (e + 1 < V) ? e1z[(e + 1) % V] : ezx;
          pml_h_apply(h1y[e], h1z[e], (ey_ip - e1y[e]) * ipx[e], (ez_ip - e1z[e]) * ipx[e], pxh1[e], pxh2[e], kv[e], bb[e], cc[e], ch);
        }
      }
      if (sy >= 0) {
        const float4 cf = yco[slot_i];
#pragma unroll
        for (int e = 0; e < V; ++e)
          pml_h_apply(h1z[e], h1x[e], (ezj1[e] - e1z[e]) * ipy, (exj1[e] - e1x[e]) * ipy, pyh1[e], pyh2[e], cf.x, cf.y, cf.z, ch);
      }
      if (sz_m >= 0) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          pml_h_apply(h1x[e], h1y[e], (e1xn[e] - e1x[e]) * ipz_m, (e1yn[e] - e1y[e]) * ipz_m, pzh1[e], pzh2[e], czh_m.x, czh_m.y, czh_m.z, ch);
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float ey_ip = (e + 1 < V) ? e1y[(e + 1) % V] : eyx;
        const float ez_ip = (e + 1 < V) ? e1z[(e + 1) % V] : ezx;
        h2x[e] = upd_h(h1x[e], ch, ezj1[e] - e1z[e], ipy, e1yn[e] - e1y[e], ipz_m);
        h2y[e] = upd_h(h1y[e], ch, e1xn[e] - e1x[e], ipz_m, ez_ip - e1z[e], ipx[e]);
        h2z[e] = upd_h(h1z[e], ch, ey_ip - e1y[e], ipx[e], exj1[e] - e1x[e], ipy);
      }
    }
    put(4 + cur * 2 + 0, e1xn); put(4 + cur * 2 + 1, e1zn);
    put(2, h2x); put(3, h2z);
    __syncthreads();
    // ---- S4: E2[k-1] = E^{n+2}[k-1]; E-side recursions of step n+1; stores -------------------------------------------------------
    if (k > kA) {
      float hyx = lane_prev(h2y[V - 1]);
      float hzx = lane_prev(h2z[V - 1]);
      if (take_prev) {
        if (first_x && pmc_x0) { hyx = -h2y[0]; hzx = -h2z[0]; }
        else { hyx = 0.f; hzx = 0.f; }
      }
      float hxj[V], hzj[V];
      if (j > 0) { get(2, mb, hxj); get(3, mb, hzj); }
      else if (pmc_y0) {
#pragma unroll
        for (int e = 0; e < V; ++e) { hxj[e] = -h2x[e]; hzj[e] = -h2z[e]; }
      } else { zero<V>(hxj); zero<V>(hzj); }
      if (pmc_z0 && k - 1 == 0) {
#pragma unroll
        for (int e = 0; e < V; ++e) { h2xm[e] = -h2x[e]; h2ym[e] = -h2y[e]; }
      }
      const bool wall_z = (k - 1 == 0) && !pmc_z0;
      float ex[V], ey[V], ez[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float hy_im = (e > 0) ? h2y[(e + V - 1) % V] : hyx;
        const float hz_im = (e > 0) ? h2z[(e + V - 1) % V] : hzx;
        float nex = upd_e(e1x[e], coef(mwm, 0, e).x, coef(mwm, 0, e).y, h2z[e] - hzj[e], idy, h2y[e] - h2ym[e], idz_m);
        float ney = upd_e(e1y[e], coef(mwm, 1, e).x, coef(mwm, 1, e).y, h2x[e] - h2xm[e], idz_m, h2z[e] - hz_im, idx[e]);
        float nez = upd_e(e1z[e], coef(mwm, 2, e).x, coef(mwm, 2, e).y, h2y[e] - hy_im, idx[e], h2x[e] - hxj[e], idy);
        const bool wx = wall_x0 && (e == 0);
        if (wall_y || wall_z) nex = 0.f;
        if (wx || wall_z) ney = 0.f;
        if (wx || wall_y) nez = 0.f;
        ex[e] = nex; ey[e] = ney; ez[e] = nez;
      }
      if (sy >= 0 && !wall_y) {
        const float4 cf = yco[32 * W + slot_i];
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float d1 = (h2z[e] - hzj[e]) * idy;
          const float d2 = (h2x[e] - hxj[e]) * idy;
          const float p1 = cf.y * pye1[e] + cf.z * d2;
          const float p2 = cf.y * pye2[e] + cf.z * d1;
          pye1[e] = p1; pye2[e] = p2;
          const bool wx = wall_x0 && (e == 0);
          if (!wx) ez[e] -= coef(mwm, 2, e).y * (cf.x * d2 + p1);
          if (!wall_z) ex[e] += coef(mwm, 0, e).y * (cf.x * d1 + p2);
        }
      }
      if (sz_m >= 0 && !wall_z) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float d1 = (h2x[e] - h2xm[e]) * idz_m;
          const float d2 = (h2y[e] - h2ym[e]) * idz_m;
          const float p1 = cze_m.y * pze1[e] + cze_m.z * d2;
          const float p2 = cze_m.y * pze2[e] + cze_m.z * d1;
          pze1[e] = p1; pze2[e] = p2;
          const bool wx = wall_x0 && (e == 0);
          if (!wall_y) ex[e] -= coef(mwm, 0, e).y * (cze_m.x * d2 + p1);
          if (!wx) ey[e] += coef(mwm, 1, e).y * (cze_m.x * d1 + p2);
        }
      }
      if (sx >= 0) {
        float kv[V], bb[V], cc[V];
        co(3, kv); co(4, bb); co(5, cc);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const bool wx = wall_x0 && (e == 0);
          if (!wx) {
            const float hy_im = (e > 0) ? h2y[(e + V - 1) % V] : hyx;
            const float hz_im = (e > 0) ? h2z[(e + V - 1) % V] : hzx;
            const float d1 = (h2y[e] - hy_im) * idx[e];
            const float d2 = (h2z[e] - hz_im) * idx[e];
            const float p1 = bb[e] * pxe1[e] + cc[e] * d2;
            const float p2 = bb[e] * pxe2[e] + cc[e] * d1;
            pxe1[e] = p1; pxe2[e] = p2;
            if (!wall_z) ey[e] -= coef(mwm, 1, e).y * (kv[e] * d2 + p1);
            if (!wall_y) ez[e] += coef(mwm, 2, e).y * (kv[e] * d1 + p2);
          }
        }
      }
      if (box_src) {                            // paged E-side source terms of step n+1 on E^{n+2}[k-1]
        if (ss_m >= 0 && sr.t->use_e2) {
          const long long qs = ((long long)ss_m * 3) * 256 + (i0 & 255);
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
      if (st_lane && k > kc0) {
        const long long po = pk - g.sxy;
        stg4(uni(b.hx + po), ob, h2x);
        stg4(uni(b.hy + po), ob, h2y);
        stg4(uni(b.hz + po), ob, h2z);
        stg4(uni(b.ex + po), ob, ex);
        stg4(uni(b.ey + po), ob, ey);
        stg4(uni(b.ez + po), ob, ez);
        if (sx >= 0) {
          const long long qo = (long long)(k - 1) * xpl;
          stg4(uni(AX.ph0n + qo), oxb, pxh1); stg4(uni(AX.ph1n + qo), oxb, pxh2);
          stg4(uni(AX.pe0n + qo), oxb, pxe1); stg4(uni(AX.pe1n + qo), oxb, pxe2);
        }
        if (sy >= 0) {
          const long long qo = (long long)(k - 1) * ypl;
          stg4(uni(AY.ph0n + qo), oyb, pyh1); stg4(uni(AY.ph1n + qo), oyb, pyh2);
          stg4(uni(AY.pe0n + qo), oyb, pye1); stg4(uni(AY.pe1n + qo), oyb, pye2);
        }
        if (sz_m >= 0) {
          const long long qo = (long long)sz_m * g.sxy;
          stg4(uni(AZ.ph0n + qo), ob, pzh1); stg4(uni(AZ.ph1n + qo), ob, pzh2);
          stg4(uni(AZ.pe0n + qo), ob, pze1); stg4(uni(AZ.pe1n + qo), ob, pze2);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      h2xm[e] = h2x[e]; h2ym[e] = h2y[e];
      h1x[e] = hxn[e]; h1y[e] = hyn[e]; h1z[e] = hzn[e];
      e1x[e] = e1xn[e]; e1y[e] = e1yn[e]; e1z[e] = e1zn[e];
      exk[e] = exn[e]; eyk[e] = eyn[e];
      pxh1[e] = xh1[e]; pxh2[e] = xh2[e]; pyh1[e] = yh1[e]; pyh2[e] = yh2[e]; pzh1[e] = zh1[e]; pzh2[e] = zh2[e];
      pxe1[e] = xe1[e]; pxe2[e] = xe2[e]; pye1[e] = ye1[e]; pye2[e] = ye2[e]; pze1[e] = ze1[e]; pze2[e] = ze2[e];
    }
    ipz_m = ipz; idz_m = idz;
    sz_m = sz; czh_m = czh; cze_m = cze;
    ss_m = ss;
    if constexpr (MAT) {
#pragma unroll
      for (int e = 0; e < V; ++e) mwm[e] = mw[e];
    }
    cur ^= 1;
  }
}

}  // namespace fdtd
