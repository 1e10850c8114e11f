// Translation unit of the instantiations of the two-steps-per-sweep kernel that advance DISPERSIVE cells inside the sweep
// (fdtd_kernels2.hpp, OPT bit 5; round 6): materials + ADE, with / without the monitor table, whole grid (with / without absorber
// layers) or clipped to the bulk of a shell pair.  Always non-temporal stores; workgroups of up to 8 waves run under
// __launch_bounds__(512), larger ones under 1024.  Own unit so that it compiles beside the others; same flags (-fno-slp-vectorize).
#include <hip/hip_runtime.h>
#undef __global__
#if defined(__HIPCC__)
#define __global__ static __attribute__((global))
#else
#define __global__ static
#endif
#include "fdtd_kernels2.hpp"

namespace fdtd {

void launch_fused2_step_disp(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                             const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                             int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip, const TileClassP& tcl, const DispP& dp) {
  const dim3 grid(grid_blocks, 1, 1), block(64, waves, 1);
#define FDTD_F2_O(LBV, OV)                                                                                             \
  hipLaunchKernelGGL((fused2_step_kernel<LBV, OV>), grid, block, fused2_lds_bytes(LBV, OV, waves), st, g, a, b, s, m, zchunk, nbx, nby, nbz,     \
                     xcd_remap, inj, seam, dmp, clip, tcl, dp, SrcP{})
  // 32 + 2 + 1 = 35: materials + ADE, non-temporal stores; + 4: monitor table; + 8: absorber layers; + 16: clipped
#define FDTD_F2(LBV)                                                                                                   \
  do {                                                                                                                 \
    switch (opt & (4 | 8 | 16)) {                                                                                      \
      case 0: FDTD_F2_O(LBV, 35); break; case 4: FDTD_F2_O(LBV, 39); break;                                            \
      case 8: FDTD_F2_O(LBV, 43); break; case 12: FDTD_F2_O(LBV, 47); break;                                           \
      case 16: FDTD_F2_O(LBV, 51); break; default: FDTD_F2_O(LBV, 55); break;                                          \
    }                                                                                                                  \
  } while (0)
  if (waves <= 8) FDTD_F2(512);
  else FDTD_F2(1024);
#undef FDTD_F2
#undef FDTD_F2_O
}

}  // namespace fdtd
