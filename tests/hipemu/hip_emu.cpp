// TEST INFRASTRUCTURE — runtime of the HIP-on-CPU emulator (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace hipemu {

State& S() {
  static State s;
  return s;
}

#if defined(__x86_64__)
// save the callee-saved registers and the FP control words of the running context on its stack, store its stack pointer in
// *save_sp, continue on the stack load_sp (a frame of the same layout: another fiber, the scheduler, or a fresh fiber's first frame)
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch,.-hipemu_switch
)");
static inline void switch_ctx(Ctx& from, Ctx& to) { hipemu_switch(&from.sp, to.sp); }
static void trampoline();
// the first frame of a fresh fiber: what hipemu_switch pops before it "returns" into the trampoline
static void init_ctx(Ctx& c, char* stack, size_t size, State&) {
  uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
  uint64_t* sp = (uint64_t*)(top - 8);          // (the slot a caller's return address would take: the trampoline never returns)
  *sp = 0;
  *--sp = (uint64_t)(uintptr_t)&trampoline;     // ret -> trampoline, with rsp = top - 8 (the alignment of a function entry)
  for (int q = 0; q < 6; ++q) *--sp = 0;        // rbp, rbx, r12 ... r15
  --sp;
  uint32_t words[2] = {0x1F80u, 0x037Fu};       // MXCSR and the x87 control word: those of the launching thread (as getcontext took them)
  uint16_t cw = 0x037F;
  asm volatile("stmxcsr %0" : "=m"(words[0]));
  asm volatile("fnstcw %0" : "=m"(cw));
  words[0] &= ~0x3Fu;                           // (no pending exception flags)
  words[1] = cw;
  std::memcpy(sp, words, 8);
  c.sp = sp;
}
#else
static inline void switch_ctx(Ctx& from, Ctx& to) { swapcontext(&from.uc, &to.uc); }
static void trampoline();
static void init_ctx(Ctx& c, char* stack, size_t size, State& s) {
  getcontext(&c.uc);
  c.uc.uc_stack.ss_sp = stack;
  c.uc.uc_stack.ss_size = size;
  c.uc.uc_link = &s.sched.uc;
  makecontext(&c.uc, (void (*)())trampoline, 0);
}
#endif

static void trampoline() {
  State& s = S();
  s.body();
  s.cur->state = 4;
  switch_ctx(s.cur->ctx, s.sched);
  std::abort();                                 // (a finished fiber is never resumed)
}

void yield_to_sched() {
  State& s = S();
  switch_ctx(s.cur->ctx, s.sched);
}

void block_barrier() {
  State& s = S();
  s.cur->state = 2;
  yield_to_sched();
}

void wave_rendezvous() {
  State& s = S();
  s.cur->state = 3;
  yield_to_sched();
}

static char* get_stack(State& s) {
  if (!s.stack_pool.empty()) {
    char* p = s.stack_pool.back();
    s.stack_pool.pop_back();
    return p;
  }
  return (char*)std::malloc(kStack);
}

void run_grid(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
  State& s = S();
  s.gdim = grid;
  s.bdim = block;
  s.body = body;
  const int nthreads = (int)(block.x * block.y * block.z);
  const int nwaves = (nthreads + kWave - 1) / kWave;
  std::vector<char> smem(shmem + 64);
  s.dyn_smem = smem.data();
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.bidx = dim3(bx, by, bz);
        s.fibers.assign(nthreads, Fiber());
        s.xbuf.assign(2 * (size_t)(nthreads + kWave), 0);
        s.wave_phase.assign(nwaves, 0);
        int lin = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++lin) {
              s.fibers[lin].tid = dim3(tx, ty, tz);
              s.fibers[lin].lin = lin;
            }
        int done = 0;
        while (done < nthreads) {
          bool progress = false;
          for (int i = 0; i < nthreads; ++i) {
            Fiber& f = s.fibers[i];
            if (f.state == 0) {
              f.stack = get_stack(s);
              init_ctx(f.ctx, f.stack, kStack, s);
              f.state = 1;
            }
            if (f.state == 1) {
              s.cur = &f;
              switch_ctx(s.sched, f.ctx);
              progress = true;
              if (f.state == 4) {
                ++done;
                s.stack_pool.push_back(f.stack);
                f.stack = nullptr;
              }
            }
          }
          // release barriers
          int at_bar = 0, live = 0;
          for (auto& f : s.fibers) {
            if (f.state != 4) ++live;
            if (f.state == 2) ++at_bar;
          }
          if (live > 0 && at_bar == live) {
            for (auto& f : s.fibers)
              if (f.state == 2) f.state = 1;
            progress = true;
          }
          for (int w = 0; w < nwaves; ++w) {
            int wl = 0, wr = 0;
            for (int l = w * kWave; l < std::min(nthreads, (w + 1) * kWave); ++l) {
              if (s.fibers[l].state != 4) ++wl;
              if (s.fibers[l].state == 3) ++wr;
            }
            if (wl > 0 && wr == wl) {
              for (int l = w * kWave; l < std::min(nthreads, (w + 1) * kWave); ++l)
                if (s.fibers[l].state == 3) s.fibers[l].state = 1;
              s.wave_phase[w]++;
              progress = true;
            }
          }
          if (!progress) {
            std::fprintf(stderr, "hipemu: deadlock (divergent barrier/shuffle) in block %u,%u,%u\n", bx, by, bz);
            std::abort();
          }
        }
      }
  s.cur = nullptr;
}

}  // namespace hipemu

struct hipemuStream { int id; };
struct hipemuEvent { std::chrono::steady_clock::time_point t; };

// $HIPEMU_POISON: device memory is handed out filled with that byte (0xFF: NaN floats, -1 indices) — a kernel that reads what nobody wrote
// shows at once (on the device a fresh allocation is often zero, and a block recycled from an earlier engine of the process is not)
hipError_t hipMalloc(void** p, size_t bytes) {
  *p = std::malloc(bytes ? bytes : 1);
  if (*p) {
    static const char* poison = std::getenv("HIPEMU_POISON");
    if (poison) std::memset(*p, (int)std::strtol(poison, nullptr, 0), bytes ? bytes : 1);
  }
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) std::memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipemuStream{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)16 << 30; return hipSuccess; }

// ---- RCCL shim: point-to-point only, executed through a callback installed by the test ----
static hipemu_exchange_fn g_exchange = nullptr;
static void* g_exchange_user = nullptr;
struct ncclCommEmu { int rank, nranks; };
static std::vector<hipemu_p2p_op> g_ops;
static int g_group_depth = 0;

extern "C" void hipemu_set_exchange(hipemu_exchange_fn fn, void* user) { g_exchange = fn; g_exchange_user = user; }

static ncclResult_t flush_ops() {
  if (g_ops.empty()) return ncclSuccess;
  if (!g_exchange) {
    // no callback installed: a 1-rank communicator exchanging with itself — RCCL pairs the k-th
    // send to a peer with the k-th receive from it
    std::vector<hipemu_p2p_op> sends, recvs;
    for (const hipemu_p2p_op& o : g_ops) (o.kind == 1 ? sends : recvs).push_back(o);
    bool ok = sends.size() == recvs.size();
    for (size_t k = 0; ok && k < sends.size(); ++k) ok = sends[k].bytes == recvs[k].bytes && recvs[k].kind == 0;
    if (!ok) { std::fprintf(stderr, "hipemu: unmatched self Send/Recv and no exchange callback\n"); g_ops.clear(); return ncclSystemError; }
    // receives may alias later sends' sources only through distinct planes: stage through copies
    std::vector<std::vector<char>> tmp(sends.size());
    for (size_t k = 0; k < sends.size(); ++k) tmp[k].assign((char*)sends[k].buf, (char*)sends[k].buf + sends[k].bytes);
    for (size_t k = 0; k < sends.size(); ++k) std::memcpy(recvs[k].buf, tmp[k].data(), tmp[k].size());
    g_ops.clear();
    return ncclSuccess;
  }
  int rc = g_exchange(g_ops.data(), (int)g_ops.size(), g_exchange_user);
  g_ops.clear();
  return rc == 0 ? ncclSuccess : ncclSystemError;
}
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id, 0x5a, sizeof(*id)); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* c, int n, ncclUniqueId, int r) { *c = new ncclCommEmu{r, n}; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }
ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (--g_group_depth == 0) return flush_ops(); return ncclSuccess; }
static size_t dsize(ncclDataType_t t) { return t == ncclFloat ? 4 : (t == ncclDouble ? 8 : 1); }
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t, hipStream_t) {
  g_ops.push_back({1, peer, const_cast<void*>(buf), count * dsize(t)});
  return g_group_depth ? ncclSuccess : flush_ops();
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t, hipStream_t) {
  g_ops.push_back({0, peer, buf, count * dsize(t)});
  return g_group_depth ? ncclSuccess : flush_ops();
}
ncclResult_t ncclAllReduce(const void* s, void* r, size_t count, ncclDataType_t t, ncclRedOp_t, ncclComm_t c, hipStream_t) {
  // emulated as an exchange with op kind 2 (sum all-reduce over `bytes`, in place in recvbuf)
  if (s != r) std::memmove(r, s, count * dsize(t));
  if (c->nranks == 1) return ncclSuccess;
  g_ops.push_back({t == ncclDouble ? 3 : 2, -1, r, count * dsize(t)});
  return g_group_depth ? ncclSuccess : flush_ops();
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ncclSuccess" : "nccl emu error"; }
