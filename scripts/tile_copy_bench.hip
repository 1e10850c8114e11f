// Memory-system floor of the sweep's access pattern (round 6): copy kernels that move what fused2_step_kernel moves
// — 6 arrays in, 6 arrays out, [nz][ny][nx] fp32 — under different traversals, without any arithmetic.  The what-if instantiation 7
// of the real kernel (loads + stores only) runs as long as the kernel itself (profiles/r6/r6h): the sweep sits on the floor of ITS
// traversal.  Which traversal has a lower floor?
//   linear     grid-stride float4 copy, one array after the other                      (the achievable roof on this box)
//   linear6    the same with the six arrays interleaved per element                    (12 streams per wave, contiguous)
//   zmarch     workgroup = W waves = W rows x 256 cells, marching `zc` planes, tiles y-fastest in runs of `run` per XCD — the sweep's
//              traversal; `halo` re-reads two arrays of the row above (as the sweep does), nt = non-temporal stores
//   ymarch     workgroup = W waves = W PLANES x 256 cells, marching `zc` rows (stride one row): the same tile with y and z swapped
//   zmarch512  a wave covers a whole 512-cell row (two float4 per lane)
// build: hipcc --offload-arch=gfx950 -O3 scripts/tile_copy_bench.hip -o variants/tile_copy_bench ;  run: variants/tile_copy_bench [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <array>

struct Arr { const float* s[6]; float* d[6]; };
typedef float v4f_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(float4 v, float* p) {
  v4f_ t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<v4f_*>(p));
}

__global__ __launch_bounds__(256) void linear_kernel(const float4* __restrict__ s, float4* __restrict__ d, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ __launch_bounds__(256) void linear6_kernel(Arr a, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = reinterpret_cast<const float4*>(a.s[c])[i];
#pragma unroll
    for (int c = 0; c < 6; ++c) nt_store(v[c], a.d[c] + 4 * i);
  }
}

__device__ __forceinline__ int tile_of(int t, int total, int run) {
  if (run > 1) {
    const int full = total / (8 * run) * (8 * run);
    if (t < full) { const int x = t & 7, m = t >> 3; t = ((m / run) * 8 + x) * run + m % run; }
  }
  return t;
}

// mode 0: z-march (rows = waves), 1: y-march (planes = waves)
template <int MODE, int XW, bool NT, bool HALO>
__global__ __launch_bounds__(1024) void march_kernel(Arr a, int nx, int ny, int nz, int zc, int nbx, int nb1, int nb2, int run) {
  const int W = blockDim.y;
  const int total = nbx * nb1 * nb2;
  int t = tile_of(blockIdx.x, total, run);
  if (t >= total) return;
  const int t1 = t % nb1, tx = (t / nb1) % nbx, t2 = t / (nb1 * nbx);
  const int w = threadIdx.y, lane = threadIdx.x;
  const size_t sxy = (size_t)nx * ny;
  // MODE 0: waves span rows (t1 over y), march over planes (t2 over z);  MODE 1: waves span planes (t1 over z), march over rows (t2 over y)
  const int fixed = t1 * W + w;
  const int lim_fixed = MODE == 0 ? ny : nz, lim_march = MODE == 0 ? nz : ny;
  if (fixed >= lim_fixed) return;
  const int m0 = t2 * zc, m1 = min(m0 + zc, lim_march);
  for (int m = m0; m < m1; ++m) {
    const int j = MODE == 0 ? fixed : m, k = MODE == 0 ? m : fixed;
#pragma unroll
    for (int h = 0; h < XW; ++h) {
      const int i = (tx * 64 * XW + h * 64 + lane) * 4;
      if (i >= nx) continue;
      const size_t p = (size_t)k * sxy + (size_t)j * nx + i;
      float4 v[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) v[c] = *reinterpret_cast<const float4*>(a.s[c] + p);
      if (HALO && j + 1 < ny) {
        const float4 h0 = *reinterpret_cast<const float4*>(a.s[0] + p + nx), h2 = *reinterpret_cast<const float4*>(a.s[2] + p + nx);
        v[0].x += h0.x; v[2].x += h2.x;
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (NT) nt_store(v[c], a.d[c] + p);
        else *reinterpret_cast<float4*>(a.d[c] + p) = v[c];
      }
    }
  }
}

// the sweep's own tile: W waves = rows j0-2 .. j0+W-3 of a tile whose rows start every RS rows; rows j0 .. j0+R-1 are stored, the other
// waves (two below, one above) load everything and store nothing; a chunk of zc planes runs zc + 2 iterations; every wave reads its
// six arrays and two arrays of the row above
__global__ __launch_bounds__(1024) void sweep_like_kernel(Arr a, int nx, int ny, int nz, int zc, int nbx, int nby, int nbz, int run, int R, int RS, int extra) {
  const int W = blockDim.y;
  const int total = nbx * nby * nbz;
  int t = tile_of(blockIdx.x, total, run);
  if (t >= total) return;
  const int ty = t % nby, tx = (t / nby) % nbx, tz = t / (nby * nbx);
  const int w = threadIdx.y, lane = threadIdx.x;
  const size_t sxy = (size_t)nx * ny;
  const int j = ty * RS + w - 2;
  if (j < 0 || j >= ny) return;
  const bool own = w >= 2 && w < 2 + R;
  const int i = (tx * 64 + lane) * 4;
  if (i >= nx) return;
  const int k0 = tz * zc, k1 = min(k0 + zc, nz);
  for (int k = max(k0 - extra, 0); k < k1; ++k) {
    const size_t p = (size_t)k * sxy + (size_t)j * nx + i;
    float4 v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = *reinterpret_cast<const float4*>(a.s[c] + p);
    if (j + 1 < ny) {
      const float4 h0 = *reinterpret_cast<const float4*>(a.s[0] + p + nx), h2 = *reinterpret_cast<const float4*>(a.s[2] + p + nx);
      v[0].x += h0.x; v[2].x += h2.x;
    }
    if (own && k >= k0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) nt_store(v[c], a.d[c] + p);
    } else {
      if (v[0].x == 3.0e38f) a.d[0][p] = v[1].x + v[2].x + v[3].x + v[4].x + v[5].x;      // (keeps the loads alive)
    }
  }
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int nx = n, ny = n, nz = n;
  const size_t cells = (size_t)nx * ny * nz;
  Arr a;
  for (int c = 0; c < 6; ++c) {
    float *s, *d;
    CHK(hipMalloc(&s, cells * 4)); CHK(hipMalloc(&d, cells * 4));
    CHK(hipMemset(s, 0, cells * 4)); CHK(hipMemset(d, 0, cells * 4));
    a.s[c] = s; a.d[c] = d;
  }
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const double bytes = 12.0 * cells * 4;
  auto timeit = [&](const char* name, auto launch) {
    std::vector<float> ms;
    for (int r = 0; r < 9; ++r) {
      CHK(hipEventRecord(e0));
      launch();
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float t; CHK(hipEventElapsedTime(&t, e0, e1));
      if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const float med = ms[ms.size() / 2];
    printf("{\"n\": %d, \"pattern\": \"%s\", \"ms\": %.4f, \"min_ms\": %.4f, \"max_ms\": %.4f, \"TB_per_s\": %.3f}\n", n, name, med, ms.front(), ms.back(), bytes / med * 1e-9);
    fflush(stdout);
  };
  timeit("linear (6 copies, 16384 blocks)", [&] {
    for (int c = 0; c < 6; ++c) hipLaunchKernelGGL(linear_kernel, dim3(16384), dim3(256), 0, 0, (const float4*)a.s[c], (float4*)a.d[c], cells / 4);
  });
  timeit("linear6 (six arrays per element, nt)", [&] { hipLaunchKernelGGL(linear6_kernel, dim3(8192), dim3(256), 0, 0, a, cells / 4); });
  size_t lds_bytes = 0;        // dynamic LDS per workgroup: 128 KB leaves ONE 16-wave workgroup per CU, as the sweep's exchange arrays do
  auto zm = [&](auto kern, int W, int zc, int run, int mode, int xw) {
    const int nbx = (nx + 256 * xw - 1) / (256 * xw);
    const int nb1 = ((mode == 0 ? ny : nz) + W - 1) / W, nb2 = ((mode == 0 ? nz : ny) + zc - 1) / zc;
    const int total = nbx * nb1 * nb2, blocks = (total + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64, W), lds_bytes, 0, a, nx, ny, nz, zc, nbx, nb1, nb2, run);
  };
  char name[160];
  for (int W : {16, 8}) for (int zc : {32, 8}) for (int run : {32, 1}) {
    snprintf(name, sizeof name, "zmarch W=%d zc=%d run=%d nt halo", W, zc, run);
    timeit(name, [&] { zm(march_kernel<0, 1, true, true>, W, zc, run, 0, 1); });
  }
  for (size_t lb : {(size_t)128 << 10, (size_t)64 << 10, (size_t)40 << 10}) {       // 1 / 2 / 3 sixteen-wave workgroups per CU
    lds_bytes = lb;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&march_kernel<0, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
    snprintf(name, sizeof name, "zmarch W=16 zc=32 run=32 nt halo, %zu KB of LDS per workgroup (%d workgroups per CU)", lb >> 10, (int)(160 / (lb >> 10)) > 2 ? 2 : (int)(160 / (lb >> 10)));
    timeit(name, [&] { zm(march_kernel<0, 1, true, true>, 16, 32, 32, 0, 1); });
    snprintf(name, sizeof name, "zmarch W=8 zc=32 run=32 nt halo, %zu KB of LDS per workgroup", lb >> 11);
    lds_bytes = lb / 2;
    timeit(name, [&] { zm(march_kernel<0, 1, true, true>, 8, 32, 32, 0, 1); });
  }
  lds_bytes = 0;
  timeit("zmarch W=16 zc=32 run=32 nt no-halo", [&] { zm(march_kernel<0, 1, true, false>, 16, 32, 32, 0, 1); });
  timeit("zmarch W=16 zc=32 run=32 plain-stores halo", [&] { zm(march_kernel<0, 1, false, true>, 16, 32, 32, 0, 1); });
  timeit("zmarch W=16 zc=512 run=32 nt halo (whole column)", [&] { zm(march_kernel<0, 1, true, true>, 16, 512, 32, 0, 1); });
  timeit("zmarch512 W=16 zc=32 run=32 nt halo (whole rows)", [&] { zm(march_kernel<0, 2, true, true>, 16, 32, 32, 0, 2); });
  timeit("zmarch512 W=8 zc=32 run=32 nt halo (whole rows)", [&] { zm(march_kernel<0, 2, true, true>, 8, 32, 32, 0, 2); });
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_like_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
  for (int lds_kb : {128, 0}) for (auto cfg : {std::array<int, 4>{16, 13, 13, 2}, std::array<int, 4>{16, 13, 16, 2}, std::array<int, 4>{16, 13, 13, 0},
                                               std::array<int, 4>{16, 16, 16, 2}, std::array<int, 4>{19, 16, 16, 2}, std::array<int, 4>{8, 5, 5, 2}}) {
    const int W = cfg[0], R = cfg[1], RS = cfg[2], extra = cfg[3], zc = 32, run = 32;
    if (W > 16 && lds_kb) continue;
    const int nbx = (nx + 255) / 256, nby = (ny + RS - 1) / RS, nbz = (nz + zc - 1) / zc;
    const int total = nbx * nby * nbz, blocks = (total + 7) / 8 * 8;
    snprintf(name, sizeof name, "sweep-like W=%d stored rows=%d tile stride=%d planes per chunk=%d+%d, %d KB LDS", W, R, RS, zc, extra, lds_kb * W / 16);
    timeit(name, [&] { hipLaunchKernelGGL(sweep_like_kernel, dim3(blocks), dim3(64, W), (size_t)lds_kb * 1024 * W / 16, 0, a, nx, ny, nz, zc, nbx, nby, nbz, run, R, RS, extra); });
  }
  for (int W : {16, 8}) for (int zc : {32, 512}) {
    snprintf(name, sizeof name, "ymarch W=%d rows=%d run=32 nt", W, zc);
    timeit(name, [&] { zm(march_kernel<1, 1, true, false>, W, zc, 32, 1, 1); });
  }
  timeit("ymarch512 W=16 rows=512 run=32 nt (whole planes per wave)", [&] { zm(march_kernel<1, 2, true, false>, 16, 512, 32, 1, 2); });
  timeit("ymarch512 W=4 rows=512 run=1 nt", [&] { zm(march_kernel<1, 2, true, false>, 4, 512, 1, 1, 2); });
  return 0;
}
