// Translation unit of the two-steps-per-sweep kernels (fdtd_kernels2.hpp) and their host-side launchers.
// Built with -fno-slp-vectorize (tidy3d_amd/build.py): see fdtd_fused2.hpp.
#include <hip/hip_runtime.h>
// fdtd_kernels.hpp defines its kernels in the header (it was written for one translation unit).  This second unit only
// needs its types and device helpers: here every kernel gets internal linkage, and the ones not launched from this
// file are dropped.
#undef __global__
#if defined(__HIPCC__)
#define __global__ static __attribute__((global))
#else
#define __global__ static
#endif
#include "fdtd_kernels2.hpp"

namespace fdtd {

void launch_inject_values(hipStream_t st, float* val, const float* w_re, const float* w_im, const float2* wave,
                          long long step, int n) {
  hipLaunchKernelGGL(inject_values_kernel, dim3(1), dim3(256), 0, st, val, w_re, w_im, wave, step, n);
}

void launch_fused2_step(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                        const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                        int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip, const TileClassP& tcl, const DispP& dp, const SrcP& sr) {
  if (opt & 64) {
    launch_fused2_step_src(st, waves, opt, grid_blocks, g, a, b, s, m, zchunk, nbx, nby, nbz, xcd_remap, inj, seam, dmp, clip, tcl, dp, sr);
    return;
  }
  if (opt >> 8) {
    launch_fused2_step_whatif(st, waves, opt, grid_blocks, g, a, b, s, m, zchunk, nbx, nby, nbz, xcd_remap, inj, seam, dmp, clip);
    return;
  }
  if (opt & 32) {
    launch_fused2_step_disp(st, waves, opt, grid_blocks, g, a, b, s, m, zchunk, nbx, nby, nbz, xcd_remap, inj, seam, dmp, clip, tcl, dp);
    return;
  }
  if (opt & 16) {
    launch_fused2_step_clip(st, waves, opt, grid_blocks, g, a, b, s, m, zchunk, nbx, nby, nbz, xcd_remap, inj, seam, dmp, clip, tcl);
    return;
  }
  const dim3 grid(grid_blocks, 1, 1), block(64, waves, 1);
#define FDTD_F2_O(LBV, OV)                                                                                             \
  hipLaunchKernelGGL((fused2_step_kernel<LBV, OV>), grid, block, fused2_lds_bytes(LBV, OV, waves), st, g, a, b, s, m, zchunk, nbx, nby, nbz,     \
                     xcd_remap, inj, seam, dmp, clip, tcl, dp, sr)
#define FDTD_F2(LBV)                                                                                                   \
  do {                                                                                                                 \
    switch (opt & 15) {                                                                                                \
      case 0: FDTD_F2_O(LBV, 0); break; case 1: FDTD_F2_O(LBV, 1); break; case 2: FDTD_F2_O(LBV, 2); break;            \
      case 3: FDTD_F2_O(LBV, 3); break; case 4: FDTD_F2_O(LBV, 4); break; case 5: FDTD_F2_O(LBV, 5); break;            \
      case 6: FDTD_F2_O(LBV, 6); break; case 7: FDTD_F2_O(LBV, 7); break; case 8: FDTD_F2_O(LBV, 8); break;            \
      case 9: FDTD_F2_O(LBV, 9); break; case 10: FDTD_F2_O(LBV, 10); break; case 11: FDTD_F2_O(LBV, 11); break;        \
      case 12: FDTD_F2_O(LBV, 12); break; case 13: FDTD_F2_O(LBV, 13); break; case 14: FDTD_F2_O(LBV, 14); break;      \
      default: FDTD_F2_O(LBV, 15); break;                                                                              \
    }                                                                                                                  \
  } while (0)
  if (waves <= 8) FDTD_F2(512);
  else if (waves <= 12) FDTD_F2(768);
  else FDTD_F2(1024);
#undef FDTD_F2
#undef FDTD_F2_O
}

void launch_inject_table(hipStream_t st, float* tab, long long stride, long long off, const float* w_re, const float* w_im,
                         const float2* wave, long long n_steps, int n) {
  const unsigned blocks = (unsigned)((n_steps * n + 255) / 256);
  hipLaunchKernelGGL(inject_table_kernel, dim3(blocks), dim3(256), 0, st, tab, stride, off, w_re, w_im, wave, n_steps, n);
}

void launch_pair_record(hipStream_t st, const PairRecP& r, long long max_cells, const GridP& g, const FieldP& a, const FieldP& b,
                        const float* cap) {
  const dim3 grid((unsigned)((max_cells + 255) / 256), (unsigned)(r.n_mon * 6));
  hipLaunchKernelGGL(pair_record_kernel, grid, dim3(256), 0, st, r, g, a, b, cap);
}

void launch_dft_record_dump(hipStream_t st, const DftDumpP& r, const float* dump, float2* acc, long long cells, long long fstride,
                            const float2* phase, int nf) {
  const dim3 grid((unsigned)((cells + 255) / 256), (unsigned)r.n);
  hipLaunchKernelGGL(dft_record_dump_kernel, grid, dim3(256), 0, st, r, dump, acc, cells, fstride, phase, nf);
}

void launch_seams(hipStream_t st, const GridP& g, const FieldP& b, const StepP& s, const MatP& m, const float* seam,
                  int n_seams, const DampT& dmp, const ClipP& clip, const InjP& inj, const SrcP& sr) {
  const long long nt = (long long)n_seams * (clip.j1 - clip.j0) * (clip.k1 - clip.k0);
  if (nt <= 0) return;
  const unsigned blocks = (unsigned)((nt + 255) / 256);
  hipLaunchKernelGGL(seam_kernel, dim3(blocks), dim3(256), 0, st, g, b, s, m, seam, n_seams, dmp, clip, inj, sr);
}

}  // namespace fdtd
