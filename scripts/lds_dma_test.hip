// LDS-DMA (global_load_lds_dwordx4 through inline asm: absent from hipcc's s_waitcnt bookkeeping, so nothing drains it at a barrier)
// — the addressing the two-step sweep's prefetch relies on, checked on the device: a wave's 64 lanes land at base + lane * 16.
// build: hipcc --offload-arch=gfx950 -O3 scripts/lds_dma_test.hip -o variants/lds_dma_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma16(const float* base, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(base) : "memory");
}
__global__ __launch_bounds__(1024) void k(const float* src, float* dst, int n4) {
  extern __shared__ float4 lds[];
  const int tx = threadIdx.x, ty = __builtin_amdgcn_readfirstlane((int)threadIdx.y), W = blockDim.y;
  const int me = ty * 64 + tx;
  // slot far into the 160 KB: array 9 of [10][W * 64]
  float4* slot = lds + 9 * W * 64;
  const unsigned lds_addr = (unsigned)(size_t)(slot + ty * 64);
  const unsigned voff = (unsigned)((blockIdx.x * W * 64 + me) * 16);
  dma16(src, voff, __builtin_amdgcn_readfirstlane(lds_addr));
  __syncthreads();                       // (nothing here waits for the DMA)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float4 v = slot[me];
  reinterpret_cast<float4*>(dst)[blockIdx.x * W * 64 + me] = v;
}
int main() {
  const int W = 16, blocks = 512, n4 = blocks * W * 64;
  std::vector<float> h(n4 * 4), o(n4 * 4);
  for (int i = 0; i < n4 * 4; ++i) h[i] = (float)i;
  float *s, *d;
  hipMalloc(&s, n4 * 16); hipMalloc(&d, n4 * 16);
  hipMemcpy(s, h.data(), n4 * 16, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64, W), (size_t)10 * W * 64 * 16, 0, s, d, n4);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(o.data(), d, n4 * 16, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n4 * 4; ++i) bad += o[i] != h[i];
  printf("lds_dma_test: %s, %ld mismatches of %d\n", hipGetErrorString(e), bad, n4 * 4);
  return bad != 0;
}
