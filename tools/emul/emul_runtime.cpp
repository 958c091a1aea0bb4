// Fiber runtime + C-ABI remainder of the emulated library (see hip/hip_runtime.h).  One OS thread; a workgroup = blockDim.x ucontext
// fibers; a fiber runs until it reaches a rendezvous (workgroup barrier, wave exchange) that is not complete yet, then the scheduler
// resumes the next runnable one.  A rendezvous nobody can complete is a deadlock and reported as such (on the GPU: a hang).
#include <ucontext.h>
#include <sys/mman.h>

#include <cstdarg>
#include <ctime>
#include <functional>
#include <mutex>
#include <deque>
#include <type_traits>
#include <vector>

#include "hip/hip_runtime.h"
#include "imagen_hip.h"

namespace emul {
namespace {
constexpr size_t kStack = 1 << 20;
struct Dma { const void* src; char* dst; int bytes; };
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  emul_uint3 tid{0, 0, 0};
  bool done = false;
  std::deque<Dma> dma;            // this lane's direct-to-LDS copies in flight, oldest first
};
struct Group {
  int size = 0, count = 0;
  std::vector<int> waiters;
};
std::vector<Fiber> fibers;
std::deque<int> runq;
ucontext_t sched_ctx;
int cur = -1;
emul_uint3 g_block{0, 0, 0}, g_bdim{1, 1, 1}, g_gdim{1, 1, 1};
Group wg_group;
struct Wave {
  Group arrive, release;
  unsigned long long alive = 0;    // lanes that exist and have not returned
  alignas(64) char buf[64 * 64];   // 64 lanes x up to 64 bytes
};
std::vector<Wave> waves;
void (*g_body)(void*) = nullptr;
long g_divergent_wave_ops = 0;   // wave operations completed with part of the wave (see launch())
void* g_arg = nullptr;

void block_here() { swapcontext(&fibers[cur].ctx, &sched_ctx); }
void rendezvous(Group& g) {
  if (++g.count == g.size) {
    g.count = 0;
    for (int w : g.waiters) runq.push_back(w);
    g.waiters.clear();
  } else {
    g.waiters.push_back(cur);
    block_here();
  }
}
// a thread that has returned no longer takes part in barriers or wave operations (the hardware counts live waves / executes with the
// remaining lanes): shrink the groups and let a rendezvous go that was only waiting for this thread
void leave(Group& g) {
  --g.size;
  if (g.size > 0 && g.count == g.size) {
    g.count = 0;
    for (int w : g.waiters) runq.push_back(w);
    g.waiters.clear();
  }
}
void fiber_main() {
  g_body(g_arg);
  fibers[cur].done = true;
  leave(wg_group);
  leave(waves[fibers[cur].tid.x >> 6].arrive);
  leave(waves[fibers[cur].tid.x >> 6].release);
  waves[fibers[cur].tid.x >> 6].alive &= ~(1ull << (fibers[cur].tid.x & 63));
  swapcontext(&fibers[cur].ctx, &sched_ctx);
}
}  // namespace

const emul_uint3& thread_idx() { return fibers[cur].tid; }
const emul_uint3& block_idx() { return g_block; }
const emul_uint3& block_dim() { return g_bdim; }
const emul_uint3& grid_dim() { return g_gdim; }
void workgroup_barrier() { rendezvous(wg_group); }
void wave_exchange(const void* mine, void* all, size_t bytes) {
  if (bytes > 64) { fprintf(stderr, "emul: wave_exchange of %zu bytes\n", bytes); abort(); }
  Wave& w = waves[fibers[cur].tid.x >> 6];
  memcpy(w.buf + (fibers[cur].tid.x & 63) * 64, mine, bytes);
  rendezvous(w.arrive);
  for (int l = 0; l < 64; ++l) memcpy(static_cast<char*>(all) + l * bytes, w.buf + l * 64, bytes);
}
void wave_release() { rendezvous(waves[fibers[cur].tid.x >> 6].release); }
bool wave_any(bool pred) {
  int mine = pred, all[64];
  wave_exchange(&mine, all, sizeof(int));
  const unsigned long long alive = waves[fibers[cur].tid.x >> 6].alive;
  bool r = false;
  for (int l = 0; l < 64; ++l) r |= ((alive >> l) & 1) && all[l] != 0;
  wave_release();
  return r;
}

f16v mfma_f32_32x32x16_f16(h8 a, h8 b, f16v c) {
  struct AB { h8 a, b; };
  AB mine{a, b}, all[64];
  wave_exchange(&mine, all, sizeof(AB));
  const int l = fibers[cur].tid.x & 63, col = l & 31, hi = l >> 5;
  f16v d = c;
  for (int r = 0; r < 16; ++r) {
    const int row = 8 * (r >> 2) + 4 * hi + (r & 3);
    float acc = 0.f;
    for (int k = 0; k < 16; ++k) acc += (float)all[row + 32 * (k >> 3)].a[k & 7] * (float)all[col + 32 * (k >> 3)].b[k & 7];
    d[r] += acc;
  }
  wave_release();
  return d;
}
u32x2 permlane32_swap(unsigned old_v, unsigned src_v) {
  struct P { unsigned o, s; };
  P mine{old_v, src_v}, all[64];
  wave_exchange(&mine, all, sizeof(P));
  const int l = fibers[cur].tid.x & 63;
  u32x2 r;
  r[0] = l < 32 ? all[l].o : all[l - 32].s;      // new first operand: its upper half-wave now holds the second operand's lower half-wave
  r[1] = l < 32 ? all[l + 32].o : all[l].s;      // new second operand: its lower half-wave now holds the first operand's upper half-wave
  wave_release();
  return r;
}

int update_dpp(int old_v, int src_v, int ctrl, int row_mask, int, bool) {
  int all[64];
  wave_exchange(&src_v, all, sizeof(int));
  const int l = fibers[cur].tid.x & 63, row = l >> 4, i = l & 15;
  int r = old_v;
  if ((row_mask >> row) & 1) {
    if (ctrl >= 0x111 && ctrl <= 0x11f) {                 // row_shr:n — lane i of a 16-lane row reads lane i - n of the same row
      const int n = ctrl - 0x110;
      if (i >= n) r = all[l - n];
    } else if (ctrl == 0x142) {                            // row_bcast:15 — lane 15 of the previous row to every lane of this row
      if (row > 0) r = all[16 * row - 1];
    } else {
      fprintf(stderr, "emul: DPP control %#x is not emulated\n", ctrl);
      abort();
    }
  }
  wave_release();
  return r;
}
int readlane(int v, int lane) {
  int all[64];
  wave_exchange(&v, all, sizeof(int));
  const int r = all[lane & 63];
  wave_release();
  return r;
}
void dma16(const void* gsrc, unsigned lds_dst, char* lds) {
  Fiber& f = fibers[cur];
  f.dma.push_back(Dma{gsrc, lds + lds_dst + 16 * (f.tid.x & 63), 16});
}
void dma4(const void* gsrc, unsigned lds_dst, char* lds) {   // global_load_lds_dword: lane l -> LDS bytes [dst + 4 l, + 4)
  Fiber& f = fibers[cur];
  f.dma.push_back(Dma{gsrc, lds + lds_dst + 4 * (f.tid.x & 63), 4});
}
void wait_vm(int n) {
  Fiber& f = fibers[cur];
  while ((int)f.dma.size() > n) {
    memcpy(f.dma.front().dst, f.dma.front().src, f.dma.front().bytes);
    f.dma.pop_front();
  }
}
void s_waitcnt(int imm) {
  const int vm = (imm & 15) | (((imm >> 14) & 3) << 4);
  if (vm < 63) wait_vm(vm);
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, void (*body)(void*), void* arg, char* lds, size_t lds_cap) {
  static std::mutex mu;                       // one launch at a time (sampling lanes call in from their own threads); fibers run on the caller's thread
  std::lock_guard<std::mutex> lock(mu);
  if (lds_bytes > lds_cap) { fprintf(stderr, "emul: %zu bytes of LDS requested, %zu available\n", lds_bytes, lds_cap); abort(); }
  const int n = (int)block.x;
  if (block.y != 1 || block.z != 1) { fprintf(stderr, "emul: 1-D workgroups only\n"); abort(); }
  g_body = body;
  g_arg = arg;
  g_bdim = emul_uint3{block.x, 1, 1};
  g_gdim = emul_uint3{grid.x, grid.y, grid.z};
  if ((int)fibers.size() < n) {
    const size_t old = fibers.size();
    fibers.resize(n);
    for (size_t i = old; i < fibers.size(); ++i) {
      fibers[i].stack = static_cast<char*>(mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
      if (fibers[i].stack == MAP_FAILED) { perror("emul: mmap"); abort(); }
    }
  }
  const unsigned nblocks = grid.x * grid.y * grid.z;
  for (unsigned b = 0; b < nblocks; ++b) {
    g_block = emul_uint3{b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y)};
    memset(lds, 0xCD, std::min(lds_cap, lds_bytes + 4096));   // LDS content is undefined at workgroup start: fp16 0xCDCD = -23.2, fp32 -4.3e8 (loud, finite)
    wg_group = Group{n, 0, {}};
    waves.assign((n + 63) / 64, Wave{});
    for (size_t w = 0; w < waves.size(); ++w) {   // (a partial last wave)
      const int lanes = std::min(64, n - 64 * (int)w);
      waves[w].arrive.size = waves[w].release.size = lanes;
      waves[w].alive = lanes == 64 ? ~0ull : (1ull << lanes) - 1;
    }
    runq.clear();
    for (int i = 0; i < n; ++i) {
      Fiber& f = fibers[i];
      f.tid = emul_uint3{(unsigned)i, 0, 0};
      f.done = false;
      f.dma.clear();
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack;
      f.ctx.uc_stack.ss_size = kStack;
      f.ctx.uc_link = &sched_ctx;
      makecontext(&f.ctx, fiber_main, 0);
      runq.push_back(i);
    }
    int done = 0;
    while (done < n) {
      if (runq.empty()) {
        // a wave operation inside a divergent branch (a sub-group of the wave reduces with __shfl_xor while the other lanes have moved on
        // to the workgroup barrier): the hardware executes it with the lanes present.  When nothing else can run, let the oldest such
        // partial exchange go with the lanes it has (absent lanes' deposits are stale — a kernel that needed them shows it in its output)
        bool forced = false;
        for (auto& w : waves) {
          for (Group* g : {&w.arrive, &w.release}) {
            if (!forced && g->count > 0 && !g->waiters.empty()) {
              g->count = 0;
              for (int f : g->waiters) runq.push_back(f);
              g->waiters.clear();
              forced = true;
              ++g_divergent_wave_ops;
            }
          }
        }
        if (forced) continue;
        fprintf(stderr, "emul: DEADLOCK in workgroup %u: %d of %d threads finished, the rest wait at a rendezvous the others never reach "
                        "(workgroup barrier %d/%d", b, done, n, wg_group.count, wg_group.size);
        for (size_t w = 0; w < waves.size(); ++w)
          if (waves[w].arrive.count || waves[w].release.count) fprintf(stderr, ", wave %zu exchange %d/64 release %d/64", w, waves[w].arrive.count, waves[w].release.count);
        fprintf(stderr, ")\n");
        abort();
      }
      cur = runq.front();
      runq.pop_front();
      swapcontext(&sched_ctx, &fibers[cur].ctx);
      if (fibers[cur].done) {
        ++done;
        if (!fibers[cur].dma.empty()) {
          fprintf(stderr, "emul: thread %u of workgroup %u ended with %zu direct-to-LDS copies in flight (they would land in another "
                          "workgroup's LDS)\n", fibers[cur].tid.x, b, fibers[cur].dma.size());
          abort();
        }
      }
    }
    cur = -1;
  }
}
}  // namespace emul

hipError_t hipDeviceGetAttribute(int* v, int attr, int) {
  const char* e = getenv("IMAGEN_EMUL_CUS");
  *v = attr == hipDeviceAttributeMultiprocessorCount ? (e ? atoi(e) : 2) : 0;
  return hipSuccess;
}

void imagen_set_error(const char* fmt, ...);   // capi.hip
// ---- what the emulated library does not compile from csrc/: conv_lds.hip (superseded family, hand-scheduled asm throughout)
#define WEAK __attribute__((weak))   // the other families' translation units override these when they are part of the emulated library
int imagen_conv_lds_num_configs() { return 0; }
int imagen_conv_lds_config_info(int, int*, int*, int*) { return -1; }
int imagen_conv_lds_stage_slots(int, int, int) { return -1; }
long imagen_conv_lds_lds_bytes(int, int, int, int, int) { return -1; }
int launch_conv_lds(const ImagenIgemmParams*, int, hipStream_t) { imagen_set_error("emulated library: conv_lds is not emulated"); return -1; }
WEAK int imagen_conv_dma_num_configs() { return 0; }
WEAK int imagen_conv_dma_config_info(int, int*, int*, int*) { return -1; }
WEAK long imagen_conv_dma_lds_bytes(int, int, int, int, int) { return -1; }
WEAK int imagen_conv_dma_ring(int) { return 0; }
WEAK int launch_conv_dma(const ImagenIgemmParams*, int, hipStream_t) { return -1; }
WEAK int imagen_conv_stream_num_configs() { return 0; }
WEAK int imagen_conv_stream_config_info(int, int*, int*, int*) { return -1; }
WEAK long imagen_conv_stream_lds_bytes(int, int, int, int, int) { return -1; }
WEAK int launch_conv_stream(const ImagenIgemmParams*, int, hipStream_t) { return -1; }

// ---- graphs and events for capi.hip: a graph is the list of recorded launches (each with its by-value kernel arguments)
struct EmulGraph { std::vector<std::function<void()>> nodes; };
struct EmulEvent { double t = 0; };
namespace { thread_local EmulGraph* g_capture = nullptr; }   // (hipStreamCaptureModeThreadLocal: sampling lanes capture from their own threads)
namespace emul {
bool capturing() { return g_capture != nullptr; }
void record(std::function<void()> node) { g_capture->nodes.push_back(std::move(node)); }
}
hipError_t hipStreamBeginCapture(hipStream_t, int) {
  if (g_capture) return 1;
  g_capture = new EmulGraph();
  return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
  if (!g_capture) return 1;
  *g = g_capture;
  g_capture = nullptr;
  return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) { *e = new EmulGraph(*g); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
  for (auto& n : e->nodes) n();
  return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
static double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmulEvent(); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
