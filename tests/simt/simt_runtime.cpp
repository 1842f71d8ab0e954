// Fiber scheduler of the SIMT emulation shim (test infrastructure only; see hip/hip_runtime.h).
//
// Context switches: on x86-64 a twelve-instruction stack switch (callee-saved registers + stack pointer) instead of swapcontext, which
// saves and restores the signal mask with two system calls per switch -- an emulated MFMA is ~250 switches per wave, and the CPU suite
// spent most of its time in rt_sigprocmask.  Other architectures keep ucontext.
#include <hip/hip_runtime.h>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace simt {
namespace {
constexpr size_t STACK = 256 * 1024;
#if defined(__x86_64__)
extern "C" void simt_switch(void** save_sp, void* load_sp);
asm(R"(
  .text
  .globl simt_switch
  .type simt_switch,@function
simt_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
  .size simt_switch,.-simt_switch
)");
struct Ctx { void* sp = nullptr; };
#else
struct Ctx { ucontext_t uc; };
#endif
struct Fiber {
  Ctx ctx;
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
  int linear = 0;
};
std::vector<Fiber> fibers;
Ctx sched_ctx;
void fiber_main();
#if defined(__x86_64__)
inline void ctx_switch(Ctx& from, Ctx& to) { simt_switch(&from.sp, to.sp); }
void ctx_make(Fiber& f) {
  // initial frame: six callee-saved registers, then the "return address" of the first switch = fiber_main, entered with the stack
  // pointer where a call would have left it (8 mod 16)
  uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
  uint64_t* a = (uint64_t*)(top - 16);
  a[0] = (uint64_t)(uintptr_t)&fiber_main;
  a[1] = 0;
  uint64_t* sp = a - 6;
  for (int k = 0; k < 6; ++k) sp[k] = 0;
  f.ctx.sp = sp;
}
#else
inline void ctx_switch(Ctx& from, Ctx& to) { swapcontext(&from.uc, &to.uc); }
void ctx_make(Fiber& f) {
  getcontext(&f.ctx.uc);
  f.ctx.uc.uc_stack.ss_sp = f.stack;
  f.ctx.uc.uc_stack.ss_size = STACK;
  f.ctx.uc.uc_link = nullptr;
  makecontext(&f.ctx.uc, fiber_main, 0);
}
#endif
int cur = -1;
int n_threads = 0;
const std::function<void()>* g_body = nullptr;
// block barrier
int live = 0, bar_arrived = 0;
unsigned bar_gen = 0;
// wave state
struct Wave {
  int live = 0, arrived = 0;
  unsigned gen = 0;
  char a[64][16], b[64][16];
};
std::vector<Wave> waves;

void yield() {
  ctx_switch(fibers[cur].ctx, sched_ctx);
}
void fiber_main() {
  (*g_body)();
  Fiber& f = fibers[cur];
  f.done = true;
  --live;
  --waves[f.linear >> 6].live;
  // a finished thread may complete a barrier others are waiting on
  if (live > 0 && bar_arrived == live && bar_arrived > 0) { bar_arrived = 0; ++bar_gen; }
  Wave& w = waves[f.linear >> 6];
  if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; ++w.gen; }
  for (;;) ctx_switch(f.ctx, sched_ctx);        // (never resumed: the scheduler skips finished fibers)
}
void wave_sync() {
  Wave& w = waves[fibers[cur].linear >> 6];
  const unsigned g = w.gen;
  if (++w.arrived == w.live) { w.arrived = 0; ++w.gen; return; }
  while (w.gen == g) yield();
}
}  // namespace

int lane_id() { return fibers[cur].linear & 63; }

char* dyn_smem(size_t ensure_bytes) {
  static std::vector<char> buf;
  if (ensure_bytes > buf.size()) buf.resize(ensure_bytes);
  // poison on (re)size requests so that uninitialised LDS reads show up as NaN
  if (ensure_bytes) memset(buf.data(), 0xFF, buf.size());
  return buf.data();
}

void sync_block() {
  const unsigned g = bar_gen;
  if (++bar_arrived == live) { bar_arrived = 0; ++bar_gen; return; }
  while (bar_gen == g) yield();
}

void wave_exchange(const void* mine, size_t n, int src_lane, void* out) {
  Wave& w = waves[fibers[cur].linear >> 6];
  memcpy(w.a[lane_id()], mine, n);
  wave_sync();
  memcpy(out, w.a[src_lane], n);
  wave_sync();
}
void wave_gather_begin(const void* a, const void* b, size_t n) {
  Wave& w = waves[fibers[cur].linear >> 6];
  memcpy(w.a[lane_id()], a, n);
  memcpy(w.b[lane_id()], b, n);
  wave_sync();
}
const char* wave_slot_a(int lane) { return waves[fibers[cur].linear >> 6].a[lane]; }
const char* wave_slot_b(int lane) { return waves[fibers[cur].linear >> 6].b[lane]; }
void wave_gather_end() { wave_sync(); }

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
  n_threads = (int)(block.x * block.y * block.z);
  if ((int)fibers.size() < n_threads) {
    size_t old = fibers.size();
    fibers.resize(n_threads);
    for (size_t i = old; i < fibers.size(); ++i) fibers[i].stack = (char*)malloc(STACK);
  }
  waves.assign((n_threads + 63) / 64, Wave());
  g_body = &body;
  blockDim = block;
  gridDim = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        live = n_threads;
        bar_arrived = 0;
        for (auto& w : waves) { w.live = 0; w.arrived = 0; }
        for (int t = 0; t < n_threads; ++t) {
          Fiber& f = fibers[t];
          f.done = false;
          f.linear = t;
          f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          ++waves[t >> 6].live;
          ctx_make(f);
        }
        int remaining = n_threads;
        while (remaining > 0) {
          for (int t = 0; t < n_threads; ++t) {
            Fiber& f = fibers[t];
            if (f.done) continue;
            cur = t;
            threadIdx = f.tid;
            blockIdx = dim3(bx, by, bz);
            ctx_switch(sched_ctx, f.ctx);
            if (f.done) --remaining;
          }
        }
      }
  cur = -1;
}
}  // namespace simt
