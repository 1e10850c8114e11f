// Translation unit of the instantiations of the two-steps-per-sweep kernel that add PAGED SOURCE TERMS (fdtd_kernels2.hpp, OPT bit 6;
// round 6): step pairs while a TFSF box, a mode plane, a current sheet or any list of more than kMaxInj nodes injects.  Always with
// non-temporal stores and the monitor table (bits 0 and 2); with / without materials; whole grid (with / without absorber layers)
// or clipped to the bulk of a shell pair; the materials ones also with the dispersive cells' memory terms (bit 5).
// Workgroups of up to 8 waves run under __launch_bounds__(512), larger ones under 1024.  Own unit: compiles beside the others.
#include <hip/hip_runtime.h>
#undef __global__
#if defined(__HIPCC__)
#define __global__ static __attribute__((global))
#else
#define __global__ static
#endif
#include "fdtd_kernels2.hpp"

namespace fdtd {

void launch_fused2_step_src(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                            const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                            int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip, const TileClassP& tcl,
                            const DispP& dp, const SrcP& sr) {
  const dim3 grid(grid_blocks, 1, 1), block(64, waves, 1);
#define FDTD_F2_O(LBV, OV)                                                                                             \
  hipLaunchKernelGGL((fused2_step_kernel<LBV, OV>), grid, block, fused2_lds_bytes(LBV, OV, waves), st, g, a, b, s, m, zchunk, nbx, nby, nbz,     \
                     xcd_remap, inj, seam, dmp, clip, tcl, dp, sr)
  // 64 + 4 + 1 = 69: source terms, monitor table, non-temporal stores; + 2: materials; + 8: absorber layers; + 16: clipped; + 32: memory terms
#define FDTD_F2(LBV)                                                                                                   \
  do {                                                                                                                 \
    switch (opt & (2 | 8 | 16 | 32)) {                                                                                 \
      case 0: FDTD_F2_O(LBV, 69); break; case 2: FDTD_F2_O(LBV, 71); break;                                            \
      case 8: FDTD_F2_O(LBV, 77); break; case 10: FDTD_F2_O(LBV, 79); break;                                           \
      case 16: FDTD_F2_O(LBV, 85); break; case 18: FDTD_F2_O(LBV, 87); break;                                          \
      case 2 | 16 | 32: FDTD_F2_O(LBV, 119); break;                                                                    \
      case 2 | 32: FDTD_F2_O(LBV, 103); break; case 2 | 8 | 32: FDTD_F2_O(LBV, 111); break;                            \
      default: break;       /* (fdtd_capi.hip asks for nothing else) */                                                \
    }                                                                                                                  \
  } while (0)
  if (waves <= 8) FDTD_F2(512);
  else FDTD_F2(1024);
#undef FDTD_F2
#undef FDTD_F2_O
}

}  // namespace fdtd
