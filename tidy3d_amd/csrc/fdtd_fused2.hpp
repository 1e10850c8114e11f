// Two time steps per sweep: what fdtd_capi.hip and fdtd_fused2.hip share (the kernels live in fdtd_kernels2.hpp and are
// compiled in their own translation unit, fdtd_fused2.hip, with the SLP vectorizer off: packed-fp32 code for this kernel
// spends more on moves into aligned register pairs than it saves — 0.810 -> 0.701 ms per step inside one engine,
// profiles/r3q_*).
#pragma once
#include "fdtd_kernels.hpp"

namespace fdtd {

constexpr int kMaxInj = 256;
struct InjP {
  int n;                                   // nodes that receive a source term between the two steps (0: none alive)
  const int* start;                        // [nz + 2] entries of plane k: [start[k], start[k + 1])
  const int4* ent;                         // (i, j, component, index into val), sorted by plane, list order kept within a plane
  const float* val;                        // the terms (inject_values_kernel)
};
constexpr int kSeamArrays = 13;  // of step one: H1_y, H1_z, E1_x, E1_y, E1_z [c-1], E1_y, E1_z [c]; of step two: H2_x [c-1], H2_y, H2_z [c-2], H2_x, H2_y, H2_z [c]
                                 // (c = first column of the right tile)

// host-side launchers (fdtd_fused2.hip)
void launch_inject_values(hipStream_t st, float* val, const float* w_re, const float* w_im, const float2* wave,
                          long long step, int n);
// waves = rows per workgroup (W - 3 of them written); opt: bit 0 non-temporal stores, bit 1 prefetch
void launch_fused2_step(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                        const FieldP& b, const StepP& s, float ca, float cb, int zchunk, int nbx, int nby, int nbz,
                        int xcd_remap, const InjP& inj, float* seam);
void launch_seams(hipStream_t st, const GridP& g, const FieldP& b, const StepP& s, float ca, float cb, const float* seam,
                  int n_seams);

}  // namespace fdtd
