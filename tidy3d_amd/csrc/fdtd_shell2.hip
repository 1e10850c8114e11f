// Translation unit of shell2_step_kernel (fdtd_shell2.hpp): two time steps per sweep with the CPML recursions carried through
// both — the shell launches of a step pair on a CPML-walled grid (fdtd_capi.hip).  Own unit, built like fdtd_fused2.hip
// (-fno-slp-vectorize): it compiles beside the others.
#include <hip/hip_runtime.h>
#undef __global__
#if defined(__HIPCC__)
#define __global__ static __attribute__((global))
#else
#define __global__ static
#endif
#include "fdtd_shell2.hpp"

namespace fdtd {

void launch_shell2_step(hipStream_t st, int waves, bool mat, int axes, const GridP& g, const FieldP& a, const FieldP& b, const StepP& s,
                        const MatP& m, const PmlP* pm, const Shell2M& boxes, const Shell2Dump& dmp, const DispP& dp, const SrcP& sr) {
  const long long total = boxes.first[boxes.n];
  if (total <= 0) return;
  const dim3 grid((unsigned)total, 1, 1), block(64, waves, 1);
  const size_t shmem = ((size_t)8 * waves * 64 + 8 * kShell2MaxQ + 2 * 32 * waves) * sizeof(float4);
#define FDTD_S2(MATV, AXV) hipLaunchKernelGGL((shell2_step_kernel<MATV, AXV>), grid, block, shmem, st, g, a, b, s, m, pm, boxes, dmp, dp, sr)
  // one axis (the middle of an x strip, a y slab, the middle rows of a z slab: 94 % of a 512^3 shell's cells) or all of them (edges and corners)
  if (mat) { if (axes == 1) FDTD_S2(true, 1); else if (axes == 2) FDTD_S2(true, 2); else if (axes == 4) FDTD_S2(true, 4); else FDTD_S2(true, 7); }
  else { if (axes == 1) FDTD_S2(false, 1); else if (axes == 2) FDTD_S2(false, 2); else if (axes == 4) FDTD_S2(false, 4); else FDTD_S2(false, 7); }
#undef FDTD_S2
}

}  // namespace fdtd
