// Fiber runtime + C-ABI remainder of the emulated library (see hip/hip_runtime.h).  One OS thread; a workgroup = blockDim.x ucontext
// fibers; a fiber runs until it reaches a rendezvous (workgroup barrier, wave exchange) that is not complete yet, then the scheduler
// resumes the next runnable one.  A rendezvous nobody can complete is a deadlock and reported as such (on the GPU: a hang).
#include <ucontext.h>
#include <sys/mman.h>

#include <cstdarg>
#include <deque>
#include <type_traits>
#include <vector>

#include "hip/hip_runtime.h"
#include "imagen_hip.h"

namespace emul {
namespace {
constexpr size_t kStack = 1 << 20;
struct Dma { const void* src; char* dst; };
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
  alignas(64) char buf[64 * 64];   // 64 lanes x up to 64 bytes
};
std::vector<Wave> waves;
void (*g_body)(void*) = nullptr;
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
void fiber_main() {
  g_body(g_arg);
  fibers[cur].done = true;
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
  f.dma.push_back(Dma{gsrc, lds + lds_dst + 16 * (f.tid.x & 63)});
}
void wait_vm(int n) {
  Fiber& f = fibers[cur];
  while ((int)f.dma.size() > n) {
    memcpy(f.dma.front().dst, f.dma.front().src, 16);
    f.dma.pop_front();
  }
}
void s_waitcnt(int imm) {
  const int vm = (imm & 15) | (((imm >> 14) & 3) << 4);
  if (vm < 63) wait_vm(vm);
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, void (*body)(void*), void* arg, char* lds, size_t lds_cap) {
  if (lds_bytes > lds_cap) { fprintf(stderr, "emul: %zu bytes of LDS requested, %zu available\n", lds_bytes, lds_cap); abort(); }
  const int n = (int)block.x;
  if (n % 64 != 0 || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) { fprintf(stderr, "emul: 1-D launches of whole waves only\n"); abort(); }
  g_body = body;
  g_arg = arg;
  g_bdim = emul_uint3{block.x, 1, 1};
  g_gdim = emul_uint3{grid.x, 1, 1};
  if ((int)fibers.size() < n) {
    const size_t old = fibers.size();
    fibers.resize(n);
    for (size_t i = old; i < fibers.size(); ++i) {
      fibers[i].stack = static_cast<char*>(mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
      if (fibers[i].stack == MAP_FAILED) { perror("emul: mmap"); abort(); }
    }
  }
  for (unsigned b = 0; b < grid.x; ++b) {
    g_block = emul_uint3{b, 0, 0};
    memset(lds, 0xCD, lds_cap);                 // LDS content is undefined at workgroup start: fp16 0xCDCD = -23.2, fp32 -4.3e8 (loud, finite)
    wg_group = Group{n, 0, {}};
    waves.assign(n / 64, Wave{});
    for (auto& w : waves) { w.arrive.size = 64; w.release.size = 64; }
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

// ---- the rest of the C ABI the host side needs to drive IGEMM launches (capi.hip / codesize.hip / the other kernel families, reduced)
static thread_local char g_err[512] = "";
void imagen_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
unsigned imagen_kernel_code_bytes(const char*) { return 0; }   // no instruction warm-up in the emulation
int launch_igemm(const ImagenIgemmParams* p, hipStream_t s);
#define WEAK __attribute__((weak))   // the other families' translation units override these when they are part of the emulated library
int imagen_conv_lds_num_configs() { return 0; }                // (conv_lds.hip is not emulated)
int imagen_conv_lds_config_info(int, int*, int*, int*) { return -1; }
int imagen_conv_lds_stage_slots(int, int, int) { return -1; }
long imagen_conv_lds_lds_bytes(int, int, int, int, int) { return -1; }
int launch_conv_lds(const ImagenIgemmParams*, int, hipStream_t) { return -1; }
WEAK int imagen_conv_dma_num_configs() { return 0; }
WEAK int imagen_conv_dma_config_info(int, int*, int*, int*) { return -1; }
WEAK long imagen_conv_dma_lds_bytes(int, int, int, int, int) { return -1; }
WEAK int imagen_conv_dma_ring(int) { return 0; }
WEAK int launch_conv_dma(const ImagenIgemmParams*, int, hipStream_t) { return -1; }
WEAK int imagen_conv_stream_num_configs() { return 0; }
WEAK int imagen_conv_stream_config_info(int, int*, int*, int*) { return -1; }
WEAK long imagen_conv_stream_lds_bytes(int, int, int, int, int) { return -1; }
WEAK int launch_conv_stream(const ImagenIgemmParams*, int, hipStream_t) { return -1; }

extern "C" int imagen_abi_version(void) { return IMAGEN_ABI_VERSION; }
extern "C" const char* imagen_last_error(void) { return g_err; }
extern "C" size_t imagen_sizeof(int kind) {
  switch (kind) {
    case IMAGEN_OP_IGEMM: return sizeof(ImagenIgemmParams);
    case IMAGEN_OP_ROWSTAT: return sizeof(ImagenRowstatParams);
    case IMAGEN_OP_ATTENTION: return sizeof(ImagenAttentionParams);
    case IMAGEN_OP_KV_PREP: return sizeof(ImagenKvPrepParams);
    case IMAGEN_OP_QNORM: return sizeof(ImagenQnormParams);
    case IMAGEN_OP_GCA_PARTIAL: return sizeof(ImagenGcaPartialParams);
    case IMAGEN_OP_GCA_FINAL: return sizeof(ImagenGcaFinalParams);
    case IMAGEN_OP_GATE_RESIDUAL: return sizeof(ImagenGateResidualParams);
    case IMAGEN_OP_LN_RESIDUAL: return sizeof(ImagenLnResidualParams);
    case IMAGEN_OP_TIME_EMBED: return sizeof(ImagenTimeEmbedParams);
    case IMAGEN_OP_SCALE_SHIFT: return sizeof(ImagenScaleShiftParams);
    case IMAGEN_OP_PACK_IMAGE: return sizeof(ImagenPackImageParams);
    case IMAGEN_OP_CFG_X0: return sizeof(ImagenCfgX0Params);
    case IMAGEN_OP_QUANTILE: return sizeof(ImagenQuantileParams);
    case IMAGEN_OP_DDPM_UPDATE: return sizeof(ImagenDdpmUpdateParams);
    case IMAGEN_OP_ROWS_COPY: return sizeof(ImagenRowsCopyParams);
    case IMAGEN_OP_MEMSET32: return sizeof(ImagenMemset32Params);
    case IMAGEN_OP_SELECT_ROWS: return sizeof(ImagenSelectRowsParams);
    case IMAGEN_OP_MEAN_ROWS: return sizeof(ImagenMeanRowsParams);
    case IMAGEN_OP_RANDN: return sizeof(ImagenRandnParams);
    case IMAGEN_OP_LOWRES_PREP: return sizeof(ImagenLowresPrepParams);
    case IMAGEN_OP_LINCOMB: return sizeof(ImagenLincombParams);
    case IMAGEN_OP_KV_PREP_MULTI: return sizeof(ImagenKvPrepMultiParams);
    case IMAGEN_OP_TEMPORAL_PEG: return sizeof(ImagenTemporalPegParams);
    case IMAGEN_OP_TEMPORAL_ATTENTION: return sizeof(ImagenTemporalAttentionParams);
    case IMAGEN_OP_ACT_PREP: return sizeof(ImagenActPrepParams);
    default: return 0;
  }
}
extern "C" int imagen_launch(int kind, const void* params, imagen_stream_t stream) {
  if (kind != IMAGEN_OP_IGEMM) { imagen_set_error("emulated library: op kind %d is not emulated (IGEMM only)", kind); return -1; }
  return launch_igemm(static_cast<const ImagenIgemmParams*>(params), static_cast<hipStream_t>(stream));
}
extern "C" int imagen_plan_run(const ImagenOpRef* ops, int n, imagen_stream_t stream) {
  for (int i = 0; i < n; ++i) {
    const int rc = imagen_launch(ops[i].kind, ops[i].params, stream);
    if (rc != 0) return rc;
  }
  return 0;
}
extern "C" int imagen_graph_begin(imagen_stream_t) { imagen_set_error("emulated library: no graphs"); return -1; }
extern "C" int imagen_graph_end(imagen_stream_t, void**) { return -1; }
extern "C" int imagen_graph_launch(void*, imagen_stream_t) { return -1; }
extern "C" int imagen_graph_destroy(void*) { return -1; }
extern "C" int imagen_event_create(void**) { return -1; }
extern "C" int imagen_event_record(void*, imagen_stream_t) { return -1; }
extern "C" int imagen_event_elapsed_ms(void*, void*, float*) { return -1; }
extern "C" int imagen_event_destroy(void*) { return -1; }
