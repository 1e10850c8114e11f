// TEST INFRASTRUCTURE — a minimal single-threaded emulation of the HIP programming model.
//
// Purpose: the GPU box is a scarce resource (minutes per call), so the *logic* of the
// hand-written gfx950 kernels (indexing, boundary rules, halo handling, wave shuffles, LDS
// tiles) is first validated on the CPU by compiling the UNMODIFIED kernel sources
// (tidy3d_amd/csrc/*.hip) with a host C++ compiler against this header instead of ROCm's
// <hip/hip_runtime.h>.  Every GPU thread of a block becomes a fiber (own stack, hand-written context switch); __syncthreads
// and the wave-64 shuffles are fiber rendezvous points.  Nothing here is shipped or used by
// the product path: tests/hipemu/build_emu.py builds `libfdtd_emu.so`, which only
// tests/test_emu_*.py load.  The product library is built by hipcc from the same sources and
// fails loudly when absent.
#pragma once
#if !defined(__x86_64__)
#include <ucontext.h>
#endif
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <vector>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict
#define __constant__ static

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
struct double2 { double x, y; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return {x, y, z, w}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorNotSupported = 801 };
typedef struct hipemuStream* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                     hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventDefault = 0 };

namespace hipemu {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

// A fiber's saved context.  x86-64: the stack pointer of a frame that holds the callee-saved registers and the FP control words —
// switched by twenty instructions (hip_emu.cpp hipemu_switch) instead of swapcontext's two signal-mask system calls per switch,
// which were a third of the emulator's run time.  Elsewhere: ucontext.
#if defined(__x86_64__)
struct Ctx { void* sp = nullptr; };
#else
struct Ctx { ucontext_t uc; };
#endif

struct Fiber {
  Ctx ctx;
  char* stack = nullptr;
  int state = 0;   // 0 not started, 1 runnable, 2 at block barrier, 3 at wave rendezvous, 4 done
  dim3 tid;
  int lin = 0;     // linear thread index in block
};

struct State {
  Ctx sched;
  Fiber* cur = nullptr;
  dim3 bidx, bdim, gdim;
  std::vector<Fiber> fibers;
  std::function<void()> body;
  // wave rendezvous buffers (values exchanged by shuffles), double-buffered by phase parity
  std::vector<uint64_t> xbuf;
  std::vector<int> wave_phase;
  char* dyn_smem = nullptr;
  std::vector<char*> stack_pool;
};
State& S();
void yield_to_sched();
void block_barrier();
void wave_rendezvous();
void run_grid(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);

struct EventRec { std::chrono::steady_clock::time_point t; };

}  // namespace hipemu

#define threadIdx (hipemu::S().cur->tid)
#define blockIdx (hipemu::S().bidx)
#define blockDim (hipemu::S().bdim)
#define gridDim (hipemu::S().gdim)
#define warpSize 64
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::S().dyn_smem);

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

namespace hipemu {
template <typename T>
static inline T shfl_generic(T v, int src_lane_of_me /* absolute lane within wave, or -1 keep */) {
  State& s = S();
  int lin = s.cur->lin;
  int wave = lin / kWave, lane = lin % kWave;
  int ph = s.wave_phase[wave] & 1;
  uint64_t bits = 0;
  static_assert(sizeof(T) <= 8, "shuffle of >8 bytes");
  std::memcpy(&bits, &v, sizeof(T));
  s.xbuf[(size_t)(ph * (s.fibers.size() + kWave)) + wave * kWave + lane] = bits;
  wave_rendezvous();
  T out = v;
  int nthreads = (int)s.fibers.size();
  if (src_lane_of_me >= 0 && src_lane_of_me < kWave && wave * kWave + src_lane_of_me < nthreads) {
    uint64_t b = s.xbuf[(size_t)(ph * (s.fibers.size() + kWave)) + wave * kWave + src_lane_of_me];
    std::memcpy(&out, &b, sizeof(T));
  }
  return out;
}
}  // namespace hipemu

template <typename T> static inline T __shfl(T v, int src, int width = 64) {
  int lane = hipemu::S().cur->lin % 64;
  int base = lane / width * width;
  return hipemu::shfl_generic(v, base + ((src % width) + width) % width);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = hipemu::S().cur->lin % 64;
  int src = lane + (int)d;
  if (src / width != lane / width) src = -1;
  return hipemu::shfl_generic(v, src);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int lane = hipemu::S().cur->lin % 64;
  int src = lane - (int)d;
  if (src < 0 || src / width != lane / width) src = -1;
  return hipemu::shfl_generic(v, src);
}
template <typename T> static inline T __shfl_xor(T v, int m, int width = 64) {
  int lane = hipemu::S().cur->lin % 64;
  int src = lane ^ m;
  if (src / width != lane / width) src = -1;
  return hipemu::shfl_generic(v, src);
}

// wave-wide vote: bit l = predicate of lane l (lanes that have left the kernel vote 0); one rendezvous
static inline unsigned long long __ballot(int pred) {
  hipemu::State& s = hipemu::S();
  const int lin = s.cur->lin, wave = lin / hipemu::kWave, lane = lin % hipemu::kWave;
  const int ph = s.wave_phase[wave] & 1;
  const size_t base = (size_t)(ph * (s.fibers.size() + hipemu::kWave)) + (size_t)wave * hipemu::kWave;
  s.xbuf[base + lane] = pred ? 1 : 0;
  hipemu::wave_rendezvous();
  unsigned long long out = 0;
  for (int l = 0; l < hipemu::kWave; ++l) {
    const size_t gl = (size_t)wave * hipemu::kWave + l;
    if (gl < s.fibers.size() && s.fibers[gl].state != 4 && s.xbuf[base + l]) out |= 1ull << l;
  }
  return out;
}
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }

// atomics (single OS thread: plain RMW is atomic w.r.t. fibers because fibers only switch at
// explicit rendezvous points)
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }

static inline float __ldg(const float* p) { return *p; }
template <typename T> static inline T __ldg(const T* p) { return *p; }
static inline float __fmaf_rn(float a, float b, float c) { return std::fmaf(a, b, c); }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
using std::min;
using std::max;

// ---- host runtime ---------------------------------------------------------------------
hipError_t hipMalloc(void** p, size_t bytes);
template <typename T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc((void**)p, bytes); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags = 0);
template <typename T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags = 0) { return hipHostMalloc((void**)p, bytes, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipDeviceSynchronize();
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int prio);
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
// graphs: not emulated — stream capture reports "not supported" and the library keeps launching directly
typedef struct hipemuGraph* hipGraph_t;
typedef struct hipemuGraphExec* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

template <typename... KArgs, typename... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem,
                                      hipStream_t, Args... args) {
  std::function<void()> body = [=]() { kernel(static_cast<KArgs>(args)...); };
  hipemu::run_grid(grid, block, shmem, body);
}
