// Fiber scheduler of the SIMT emulation shim (test infrastructure only; see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace simt {
namespace {
constexpr size_t STACK = 256 * 1024;
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
  int linear = 0;
};
std::vector<Fiber> fibers;
ucontext_t sched_ctx;
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
  swapcontext(&fibers[cur].ctx, &sched_ctx);
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
  swapcontext(&f.ctx, &sched_ctx);
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
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = STACK;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, fiber_main, 0);
        }
        int remaining = n_threads;
        while (remaining > 0) {
          for (int t = 0; t < n_threads; ++t) {
            Fiber& f = fibers[t];
            if (f.done) continue;
            cur = t;
            threadIdx = f.tid;
            blockIdx = dim3(bx, by, bz);
            swapcontext(&sched_ctx, &f.ctx);
            if (f.done) --remaining;
          }
        }
      }
  cur = -1;
}
}  // namespace simt
