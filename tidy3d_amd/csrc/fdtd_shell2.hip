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

void launch_shell2_step(hipStream_t st, int waves, bool mat, const GridP& g, const FieldP& a, const FieldP& b, const StepP& s,
                        const MatP& m, const PmlP* pm, const Shell2P& sp) {
  const long long total = (long long)sp.nbx * sp.nby * sp.nbz;
  if (total <= 0) return;
  const dim3 grid((unsigned)total, 1, 1), block(64, waves, 1);
  const size_t shmem = ((size_t)8 * waves * 64 + 8 * kShell2MaxQ + 2 * 32 * waves) * sizeof(float4);
  if (mat) hipLaunchKernelGGL((shell2_step_kernel<true>), grid, block, shmem, st, g, a, b, s, m, pm, sp);
  else hipLaunchKernelGGL((shell2_step_kernel<false>), grid, block, shmem, st, g, a, b, s, m, pm, sp);
}

}  // namespace fdtd
