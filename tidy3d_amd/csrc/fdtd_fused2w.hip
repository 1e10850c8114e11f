// Translation unit of the WHAT-IF and PREFETCH instantiations of the two-steps-per-sweep kernel (fdtd_kernels2.hpp, OPT bits 8 - 11;
// FDTD_OPT_WHATIF): measuring aids that skip part of the sweep's work — wrong results, meaningful times — switched inside one
// engine by scripts/probe_whatif.py.  Vacuum instantiation, 16-wave workgroups only.  Own unit: compiles beside the others.
#include <hip/hip_runtime.h>
#undef __global__
#if defined(__HIPCC__)
#define __global__ static __attribute__((global))
#else
#define __global__ static
#endif
#include "fdtd_kernels2.hpp"

namespace fdtd {

void launch_fused2_step_whatif(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                               const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                               int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip) {
  const dim3 grid(grid_blocks, 1, 1), block(64, waves, 1);
  // (prefetch instantiations 10 - 12: six exchange arrays + the arrays of the next plane that travel through LDS)
  const int wv = opt >> 8;
  const size_t shmem = ((size_t)(wv == 13 ? 9 : ((wv >= 10 && wv <= 12) ? 6 + (wv == 11 ? 2 : 3) : 8)) * waves * 64) * sizeof(float4);
  const TileClassP tcl{nullptr};
  const DispP dp{nullptr, nullptr, nullptr};
#define FDTD_F2_W(WV)                                                                                                  \
  case WV: hipLaunchKernelGGL((fused2_step_kernel<1024, 1 | (WV << 8)>), grid, block, shmem, st, g, a, b, s, m, zchunk, nbx, nby,  \
                              nbz, xcd_remap, inj, seam, dmp, clip, tcl, dp, SrcP{}); break
  switch (opt >> 8) {
    FDTD_F2_W(1); FDTD_F2_W(2); FDTD_F2_W(3); FDTD_F2_W(4); FDTD_F2_W(5); FDTD_F2_W(6); FDTD_F2_W(7); FDTD_F2_W(8);
    FDTD_F2_W(10); FDTD_F2_W(11); FDTD_F2_W(12); FDTD_F2_W(13); FDTD_F2_W(14); FDTD_F2_W(15);
    default: break;
  }
#undef FDTD_F2_W
}

}  // namespace fdtd
