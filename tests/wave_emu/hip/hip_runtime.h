/*
 * TEST INFRASTRUCTURE ONLY -- a lock-step 64-lane wavefront emulator for debugging the device source of
 * petlion.jl_amd/csrc/*.hip on a machine without a GPU (this container).  It is NOT a CPU fallback of the
 * product: the package never loads it, it is built only by tests/wave_emu/build_emu.py, into
 * tests/wave_emu/libpetlion_emu.so, and only the `-m "not gpu"` tests (and developers) use it.
 *
 * Model: one workgroup = one or two waves of 64 lanes; every lane is a ucontext fiber with its own stack; the scheduler runs
 * the lanes round-robin from yield point to yield point (wave_emu::yield = the intra-wave phase separator PL_SYNC, and the two
 * steps of every __shfl_*), which reproduces SIMT semantics for code that is convergent within a wave.  __syncthreads() is a real
 * workgroup barrier (a lane spins -- yielding -- until every live lane of the block has arrived), so the two waves of a cell may
 * run different code between barriers.  Blocks run one after another.  __shared__ -> static, so exactly one block is live.
 */
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define PL_WAVE_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct emu_dim3 { unsigned x, y, z; emu_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef emu_dim3 dim3;

namespace wave_emu {
constexpr int LANES = 64;          // lanes per wave
constexpr int MAXT = 128;          // threads per block (two waves)
struct State {
  ucontext_t main_ctx;
  ucontext_t lane_ctx[MAXT];
  bool done[MAXT];
  bool waiting[MAXT];               // spinning in __syncthreads
  int cur;
  int nthreads;
  unsigned block;
  unsigned grid;
  double xbuf[MAXT];
  int bar_arrived; unsigned bar_gen;
  std::function<void()> body;
  std::vector<char*> stacks;
};
inline State& st() { static State s; return s; }
inline void yield() { State& s = st(); swapcontext(&s.lane_ctx[s.cur], &s.main_ctx); }
inline void lane_entry() { State& s = st(); s.body(); s.done[s.cur] = true; swapcontext(&s.lane_ctx[s.cur], &s.main_ctx); }
inline void run_block(unsigned b, unsigned grid, unsigned nthreads, const std::function<void()>& body) {
  State& s = st();
  const size_t STK = 1 << 20;
  while ((int)s.stacks.size() < MAXT) s.stacks.push_back((char*)malloc(STK));
  s.body = body; s.block = b; s.grid = grid; s.nthreads = (int)nthreads; s.bar_arrived = 0; s.bar_gen = 0;
  const int LANES = (int)nthreads;                                    // (threads of this block; shadows the per-wave constant in this function)
  static const bool poison = getenv("PL_EMU_POISON") != nullptr;     // uninitialised locals read garbage instead of a recycled stack
  for (int l = 0; l < LANES; l++) {
    if (poison) memset(s.stacks[l], 0x7f, STK);
    s.done[l] = false; s.waiting[l] = false; getcontext(&s.lane_ctx[l]);
    s.lane_ctx[l].uc_stack.ss_sp = s.stacks[l]; s.lane_ctx[l].uc_stack.ss_size = STK; s.lane_ctx[l].uc_link = &s.main_ctx;
    makecontext(&s.lane_ctx[l], (void (*)())lane_entry, 0);
  }
  for (;;) {
    int alive = 0;
    // PL_EMU_ORDER=reverse runs the lanes 63..0 between sync points: a cross-lane LDS hand-over that lacks a sync point (= a compiler
    // barrier on the GPU, where nothing else stops the compiler from moving the load above the store) then reads stale data in one of the
    // two orders, so the tests are run in both
    static const bool rev = getenv("PL_EMU_ORDER") && !strcmp(getenv("PL_EMU_ORDER"), "reverse");
    // PL_EMU_WAVE=0|1 (two-wave blocks): the preferred wave runs alone for as long as any of its lanes can make progress (is neither done
    // nor waiting at the workgroup barrier), so it gets as far ahead of the other wave as the barriers allow -- a cross-wave LDS hand-over
    // that lacks a barrier reads stale data in one of the two preferences
    static const int pref = getenv("PL_EMU_WAVE") ? atoi(getenv("PL_EMU_WAVE")) : -1;
    int only = -1;
    if (pref >= 0 && LANES > wave_emu::LANES) {
      bool can = false;
      for (int l = pref * wave_emu::LANES; l < (pref + 1) * wave_emu::LANES; l++) can = can || (!s.done[l] && !s.waiting[l]);
      only = can ? pref : 1 - pref;
    }
    for (int q = 0; q < LANES; q++) {
      const int l = rev ? LANES - 1 - q : q;
      if (only >= 0 && l / wave_emu::LANES != only) { alive += !s.done[l]; continue; }
      if (!s.done[l]) { alive++; s.cur = l; swapcontext(&s.main_ctx, &s.lane_ctx[l]); }
    }
    if (!alive) break;
  }
}
struct TidProxy { struct X { operator unsigned() const { return (unsigned)st().cur; } } x; };
struct BidProxy { struct X { operator unsigned() const { return st().block; } } x; };
struct GdimProxy { struct X { operator unsigned() const { return st().grid; } } x; };
}  // namespace wave_emu

static wave_emu::TidProxy threadIdx;
static wave_emu::BidProxy blockIdx;
static wave_emu::GdimProxy gridDim;

// workgroup barrier: every live lane of the block must arrive (lanes that returned from the kernel do not count)
inline void __syncthreads() {
  auto& s = wave_emu::st();
  const unsigned gen = s.bar_gen;
  ++s.bar_arrived;
  for (;;) {
    if (s.bar_gen != gen) { s.waiting[s.cur] = false; return; }
    int live = 0; for (int l = 0; l < s.nthreads; l++) live += !s.done[l];       // (a lane that has returned since no longer counts)
    if (s.bar_arrived >= live) { s.bar_arrived = 0; s.bar_gen++; for (int l = 0; l < s.nthreads; l++) s.waiting[l] = false; wave_emu::yield(); return; }
    s.waiting[s.cur] = true; wave_emu::yield();
  }
}
// shuffles are per wave: `src` is a lane index within the caller's wave
inline double __shfl(double v, int src) {
  auto& s = wave_emu::st(); int me = s.cur; const int base = me & ~(wave_emu::LANES - 1); s.xbuf[me] = v; wave_emu::yield();
  double r = (src >= 0 && src < wave_emu::LANES) ? s.xbuf[base + src] : v; wave_emu::yield(); return r;
}
inline double __shfl_down(double v, int d) { return __shfl(v, (wave_emu::st().cur & (wave_emu::LANES - 1)) + d); }
inline double __shfl_up(double v, int d) { return __shfl(v, (wave_emu::st().cur & (wave_emu::LANES - 1)) - d); }
inline double __shfl_xor(double v, int m) { return __shfl(v, (wave_emu::st().cur & (wave_emu::LANES - 1)) ^ m); }
inline int __shfl(int v, int src) { return (int)__shfl((double)v, src); }

/* ---- minimal runtime API used by the C ABI layer ---- */
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct emu_event_s { double t; }* hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
#define hipHostMallocDefault 0
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, int, hipStream_t) { for (size_t r = 0; r < h; r++) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, w); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_s{0}; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = -1.f; return 0; }

#define PL_LAUNCH(kernel, grid, block, stream, ...)                                      \
  do { unsigned g__ = (grid);                                                            \
       for (unsigned b__ = 0; b__ < g__; b__++) wave_emu::run_block(b__, g__, (unsigned)(block), [&]() { kernel(__VA_ARGS__); }); } while (0)
