// CPU functional emulation of the HIP device environment for csrc/igemm.hip (TEST INFRASTRUCTURE: tools/emul/README.md).
// Included INSTEAD of <hip/hip_runtime.h> (-Itools/emul ahead of the ROCm include path, -DIMAGEN_EMUL): the kernel source is
// compiled as plain host C++ by clang, every workgroup runs as 512 cooperative fibers of one OS thread (emul_runtime.cpp), and the
// device builtins the kernel uses are restated here from the ISA documentation:
//   v_mfma_f32_32x32x16_f16   A: lane l holds row l%32, k = 8*(l/32) .. +7;  B: lane l holds column l%32, k = 8*(l/32) .. +7;
//                             D: register r of lane l is element (row 8*(r/4) + 4*(l/32) + r%4, column l%32)
//   v_permlane32_swap         the upper half-wave of the first operand is exchanged with the lower half-wave of the second
//   ds_bpermute-style shuffles, s_barrier, readfirstlane (identity: only ever applied to wave-uniform values here)
// Nothing here is tuned and nothing of it ships: it exists so that a kernel change can be executed and compared WITHOUT a GPU.
// The restatement is validated by the product build itself: the kernels are known-good on MI355X (tests/test_igemm_cfgs_gpu.py), so
// the emulated product build has to reproduce fp32 torch on the same cases (tests/test_igemm_emulated.py) before anything else is
// concluded from it.
#pragma once
#ifndef IMAGEN_EMUL
#error "tools/emul/hip/hip_runtime.h is the emulation shim: compile with -DIMAGEN_EMUL"
#endif
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <functional>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// one OS thread runs every fiber, so thread-local storage IS workgroup-shared storage: a kernel's `__shared__ T x[N]` becomes a
// block-scope thread_local (implicitly static) array, `extern __shared__ char smem[]` an extern thread_local one defined below
#define __shared__ thread_local

struct emul_uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
using std::max;
using std::min;

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum { hipDeviceAttributeMultiprocessorCount = 1, hipFuncAttributeMaxDynamicSharedMemorySize = 2 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, int attr, int dev);   // emul_runtime.cpp: IMAGEN_EMUL_CUS compute units (default 2)
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* nb, K, int, size_t) { *nb = 2; return hipSuccess; }

namespace emul {
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
const emul_uint3& thread_idx();
const emul_uint3& block_idx();
const emul_uint3& block_dim();
const emul_uint3& grid_dim();
void workgroup_barrier();                       // s_barrier / __syncthreads
void wave_exchange(const void* mine, void* all, size_t bytes);   // every lane of the wave deposits `bytes`, then sees all 64 deposits
void wave_release();                            // second half of the rendezvous: nobody overwrites the deposits before all have read
f16v mfma_f32_32x32x16_f16(h8 a, h8 b, f16v c);
u32x2 permlane32_swap(unsigned old_v, unsigned src_v);
int update_dpp(int old_v, int src_v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl);   // row_shr:1..15 and row_bcast:15 only
int readlane(int v, int lane);
// direct-to-LDS copies (global_load_lds_dwordx4: lane l writes 16 bytes at lds_dst + 16 l) with the hardware's ordering: a copy lands
// when a covering s_waitcnt vmcnt(n) of ITS wave-lane retires it — in issue order, the n youngest may stay in flight.  A kernel that
// reads LDS before the covering wait sees the poison (or the previous tile), exactly as it would see stale bytes on the GPU.
void dma16(const void* gsrc, unsigned lds_dst, char* lds);
void dma4(const void* gsrc, unsigned lds_dst, char* lds);
void wait_vm(int n);
void s_waitcnt(int imm);                        // gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4; the other counters need no emulation
void launch(dim3 grid, dim3 block, size_t lds_bytes, void (*body)(void*), void* arg, char* lds, size_t lds_cap);
template <class Tup> static void call_tuple(void* p) {
  std::apply([](auto k, auto... a) { k(a...); }, *static_cast<Tup*>(p));
}
// stream capture (capi.hip's imagen_graph_*): while a capture is open a launch is RECORDED with its by-value arguments, exactly what a
// hipGraph kernel node holds, and runs when the graph is launched
bool capturing();
void record(std::function<void()> node);
template <class K, class... A> static inline void launch_k(dim3 grid, dim3 block, size_t lds_bytes, char* lds, size_t cap, K kern, A... args) {
  auto tup = std::make_tuple(kern, args...);
  if (capturing()) {
    record([=]() mutable { launch(grid, block, lds_bytes, &call_tuple<decltype(tup)>, &tup, lds, cap); });
    return;
  }
  launch(grid, block, lds_bytes, &call_tuple<decltype(tup)>, &tup, lds, cap);
}
bool wave_any(bool pred);
template <class T> static inline T shfl_up(T v, int delta) {
  T all[64];
  wave_exchange(&v, all, sizeof(T));
  const int l = thread_idx().x & 63;
  const T r = l >= delta ? all[l - delta] : v;
  wave_release();
  return r;
}

template <class T> static inline T shfl_xor(T v, int mask) {
  T all[64];
  wave_exchange(&v, all, sizeof(T));
  const T r = all[(thread_idx().x & 63) ^ mask];
  wave_release();
  return r;
}
template <class T> static inline T shfl(T v, int src) {
  T all[64];
  wave_exchange(&v, all, sizeof(T));
  const T r = all[src & 63];
  wave_release();
  return r;
}
}  // namespace emul

// the dynamic LDS images of a translation unit, one per name the kernels declare (`extern __shared__ ... smem[] | smem_dyn[] | sm[] | lds[]`):
// a block-scope extern declaration in this TU's unnamed namespace binds to the variable of that name
namespace {
alignas(16) thread_local char smem[192 * 1024];
alignas(16) thread_local char smem_dyn[192 * 1024];
alignas(16) thread_local float sm[48 * 1024];
alignas(16) thread_local float lds[48 * 1024];
}
// single-threaded execution: atomics are plain read-modify-writes, fences are nothing
template <class T, class U> static inline T atomicAdd(T* p, U v) { const T o = *p; *p = o + (T)v; return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __shfl_up(v, d) emul::shfl_up((v), (d))
#define __any(p) emul::wave_any((p) != 0)
static inline float rsqrtf(float v) { return 1.0f / sqrtf(v); }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
// graphs and events (capi.hip): emul_runtime.cpp
typedef struct EmulGraph* hipGraph_t;
typedef struct EmulGraph* hipGraphExec_t;
typedef struct EmulEvent* hipEvent_t;
enum { hipStreamCaptureModeThreadLocal = 1 };
hipError_t hipStreamBeginCapture(hipStream_t, int);
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*);
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t);
hipError_t hipGraphDestroy(hipGraph_t);
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t);
hipError_t hipGraphExecDestroy(hipGraphExec_t);
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t*);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t);
hipError_t hipEventDestroy(hipEvent_t);

#define threadIdx (emul::thread_idx())
#define blockIdx (emul::block_idx())
#define blockDim (emul::block_dim())
#define gridDim (emul::grid_dim())
#define __logf(v) logf(v)
#define __sincosf(a, s, c) sincosf((a), (s), (c))
#define __expf(v) expf(v)   // HIP device intrinsics the kernels call by name (glibc declares, but does not export, these names)
#define __logf(v) logf(v)
#define __syncthreads() emul::workgroup_barrier()
#define __shfl_xor(v, m) emul::shfl_xor((v), (m))
#define __shfl(v, s) emul::shfl((v), (s))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emul::mfma_f32_32x32x16_f16((a), (b), (c))
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) emul::permlane32_swap((a), (b))
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_wave_barrier() do { int wb_ = 0; (void)emul::shfl_xor(wb_, 0); } while (0)   // a wave rendezvous: every lane's earlier stores precede every lane's later reads
#define __builtin_amdgcn_sched_group_barrier(m, n, id) ((void)0)
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_s_waitcnt(imm) emul::s_waitcnt(imm)
#define __builtin_amdgcn_update_dpp(o, s, ctrl, rm, bm, bc) emul::update_dpp((o), (s), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_readlane(v, l) emul::readlane((v), (l))
#define __builtin_amdgcn_s_getpc() (0ull)
#define __builtin_amdgcn_rcpf(v) (1.0f / (v))
#define __builtin_amdgcn_rsqf(v) (1.0f / sqrtf(v))
#define __builtin_amdgcn_exp2f(v) exp2f(v)

#define hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, ...) emul::launch_k(grid, block, lds_bytes, smem, sizeof(smem), kern, __VA_ARGS__)
