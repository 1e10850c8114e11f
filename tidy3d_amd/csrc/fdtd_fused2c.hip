// Translation unit of the CLIPPED instantiations of the two-steps-per-sweep kernel (fdtd_kernels2.hpp, OPT bit 4): the bulk
// launch of a step pair whose shell — CPML slabs and their collar, or the boundary planes of a z-slab rank — is advanced by
// single steps beside it (fdtd_capi.hip).  Own unit so that it compiles beside fdtd_fused2.hip; same flags (-fno-slp-vectorize).
#include <hip/hip_runtime.h>
#undef __global__
#if defined(__HIPCC__)
#define __global__ static __attribute__((global))
#else
#define __global__ static
#endif
#include "fdtd_kernels2.hpp"

namespace fdtd {

void launch_fused2_step_clip(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                             const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                             int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip, const TileClassP& tcl) {
  const dim3 grid(grid_blocks, 1, 1), block(64, waves, 1);
#define FDTD_F2_O(LBV, OV)                                                                                             \
  hipLaunchKernelGGL((fused2_step_kernel<LBV, OV>), grid, block, fused2_lds_bytes(LBV, OV, waves), st, g, a, b, s, m, zchunk, nbx, nby, nbz,     \
                     xcd_remap, inj, seam, dmp, clip, tcl, DispP{nullptr, nullptr, nullptr}, SrcP{})
#define FDTD_F2(LBV)                                                                                                   \
  do {                                                                                                                 \
    switch (opt & 7) {                                                                                                 \
      case 0: FDTD_F2_O(LBV, 16); break; case 1: FDTD_F2_O(LBV, 17); break; case 2: FDTD_F2_O(LBV, 18); break;         \
      case 3: FDTD_F2_O(LBV, 19); break; case 4: FDTD_F2_O(LBV, 20); break; case 5: FDTD_F2_O(LBV, 21); break;         \
      case 6: FDTD_F2_O(LBV, 22); break; default: FDTD_F2_O(LBV, 23); break;                                           \
    }                                                                                                                  \
  } while (0)
  if (waves <= 8) FDTD_F2(512);
  else if (waves <= 12) FDTD_F2(768);
  else FDTD_F2(1024);
#undef FDTD_F2
#undef FDTD_F2_O
}

}  // namespace fdtd
