// probe.hip — calibration probes behind bench.py's `calibration` record: a fixed device-to-device copy and a fixed dense-MFMA loop, timed
// with HIP events on the caller's stream.  Boxes of the pool differ by up to +-20 % on the sampling workload at equal reported clocks;
// the two numbers, measured in the same process right beside the headline, let a reader normalise one round's line against another's.
// Not on the sampling path.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void probe_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// every wave: `iters` rounds of 4 independent v_mfma_f32_32x32x16_f16 (4 accumulators: the matrix pipe is never waiting on a dependency)
__global__ __launch_bounds__(256) void probe_mfma_kernel(float* sink, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
  f16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = (f16)(0.001f * (float)((threadIdx.x + j) & 7));
    b[j] = (f16)(0.002f * (float)((threadIdx.x * 3 + j) & 7));
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 12345.678f) sink[0] = s;   // (keeps the loop alive; never true)
}

// one lane follows a chain of 128-byte nodes (word 0 of node i = the next node): every load depends on the one before it, so the time per
// hop is the load-to-use latency of whichever level of the hierarchy the working set lives in
__global__ __launch_bounds__(64) void probe_chase_kernel(const unsigned* __restrict__ nodes, int hops, unsigned* out) {
  if (threadIdx.x != 0) return;
  unsigned i = out[0];   // continue where the previous launch stopped: the timed launch walks nodes the warm-up launch has NOT touched
  for (int h = 0; h < hops; ++h) {
    IMAGEN_OPAQUE(i);    // (a vector register: the hop is a global_load like the kernels' own, not a scalar-cache load)
    i = nodes[(size_t)i * 32];
  }
  out[0] = i;
}

// three distinct trivial kernels for the dependent-launch chain (a denoiser step never runs the same kernel twice in a row)
__global__ __launch_bounds__(64) void probe_tick_a(unsigned* c) { if (threadIdx.x == 0) c[0] += 1; }
__global__ __launch_bounds__(64) void probe_tick_b(unsigned* c) { if (threadIdx.x == 0) c[0] += 2; }
__global__ __launch_bounds__(64) void probe_tick_c(unsigned* c) { if (threadIdx.x == 0) c[0] += 3; }

struct Timer {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool ok = false;
  Timer() { ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess; }
  ~Timer() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
};

}  // namespace

extern "C" int imagen_probe_copy(void* dst, const void* src, size_t bytes, int reps, imagen_stream_t stream, float* gbs_out) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  IMAGEN_CHECK(dst && src && gbs_out && bytes >= 16 && bytes % 16 == 0 && reps >= 1, "probe_copy: bad arguments");
  Timer t;
  IMAGEN_CHECK(t.ok, "probe_copy: hipEventCreate failed");
  const size_t n = bytes / 16;
  const int grid = 256 * 8;
  hipLaunchKernelGGL(probe_copy_kernel, dim3(grid), dim3(256), 0, s, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n);   // warm
  (void)hipEventRecord(t.e0, s);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe_copy_kernel, dim3(grid), dim3(256), 0, s, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n);
  (void)hipEventRecord(t.e1, s);
  if (hipEventSynchronize(t.e1) != hipSuccess) return imagen_hip_status("probe_copy");
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, t.e0, t.e1);
  *gbs_out = ms > 0.f ? (float)(2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9) : 0.f;   // read + write bytes
  return imagen_hip_status("probe_copy");
}

extern "C" int imagen_probe_mfma(int iters, int reps, float* sink, imagen_stream_t stream, float* tflops_out) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  IMAGEN_CHECK(sink && tflops_out && iters >= 1 && reps >= 1, "probe_mfma: bad arguments");
  Timer t;
  IMAGEN_CHECK(t.ok, "probe_mfma: hipEventCreate failed");
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = cus * 2;   // 2 workgroups x 4 waves per CU: two waves per SIMD
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(grid), dim3(256), 0, s, sink, iters);   // warm
  (void)hipEventRecord(t.e0, s);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe_mfma_kernel, dim3(grid), dim3(256), 0, s, sink, iters);
  (void)hipEventRecord(t.e1, s);
  if (hipEventSynchronize(t.e1) != hipSuccess) return imagen_hip_status("probe_mfma");
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, t.e0, t.e1);
  const double flops = (double)grid * 4 /*waves*/ * (double)iters * 4 /*MFMAs*/ * 32768.0 * reps;   // 32x32x16: 2 * 32 * 32 * 16
  *tflops_out = ms > 0.f ? (float)(flops / (ms * 1e-3) / 1e12) : 0.f;
  return imagen_hip_status("probe_mfma");
}

// Dependent-load latency: `nodes` = a device array of n 128-byte nodes whose first words form ONE cycle through all of them (built by the
// caller: a random permutation), `hops` dependent loads by one lane, HIP-event timed -> ns per hop.  The working set (n * 128 bytes) picks
// the level: <= 2 MiB the XCD's L2, ~64-128 MiB the Infinity Cache, >= 1 GiB HBM.  The sampling path's small launches are chains of such
// round trips: this is the box-to-box difference the copy / MFMA probes do not see.
extern "C" int imagen_probe_latency(const void* nodes, int hops, void* out_word, imagen_stream_t stream, float* ns_per_hop) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  IMAGEN_CHECK(nodes && out_word && ns_per_hop && hops >= 1, "probe_latency: bad arguments");
  Timer t;
  IMAGEN_CHECK(t.ok, "probe_latency: hipEventCreate failed");
  (void)hipMemsetAsync(out_word, 0, sizeof(unsigned), s);
  hipLaunchKernelGGL(probe_chase_kernel, dim3(1), dim3(64), 0, s, static_cast<const unsigned*>(nodes), hops, static_cast<unsigned*>(out_word));   // warm: code, TLB reach, and — where the set fits a cache — the set
  (void)hipEventRecord(t.e0, s);
  hipLaunchKernelGGL(probe_chase_kernel, dim3(1), dim3(64), 0, s, static_cast<const unsigned*>(nodes), hops, static_cast<unsigned*>(out_word));
  (void)hipEventRecord(t.e1, s);
  if (hipEventSynchronize(t.e1) != hipSuccess) return imagen_hip_status("probe_latency");
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, t.e0, t.e1);
  *ns_per_hop = (float)((double)ms * 1e6 / hops);
  return imagen_hip_status("probe_latency");
}

// Dependent-launch cost inside a hipGraph: `n` one-wave kernels (three distinct symbols in rotation, each a read-modify-write of the same
// word, so every launch depends on the one before it) captured once and replayed `reps` times -> us per launch.  The floor under every
// small launch of the captured denoiser step.
extern "C" int imagen_probe_launch_chain(int n, int reps, void* counter_word, imagen_stream_t stream, float* us_per_launch) {
  IMAGEN_CHECK(counter_word && us_per_launch && n >= 1 && reps >= 1, "probe_launch_chain: bad arguments");
  (void)hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));
  hipStream_t s = nullptr;   // a private stream: the caller's may be the legacy default stream, which cannot be captured
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return imagen_hip_status("probe_launch_chain: hipStreamCreate");
  imagen_stream_t ps = reinterpret_cast<imagen_stream_t>(s);
  unsigned* c = static_cast<unsigned*>(counter_word);
  void* exec = nullptr;
  for (int k = 0; k < 3; ++k) {   // outside the capture first (module load)
    hipLaunchKernelGGL(probe_tick_a, dim3(1), dim3(64), 0, s, c);
    hipLaunchKernelGGL(probe_tick_b, dim3(1), dim3(64), 0, s, c);
    hipLaunchKernelGGL(probe_tick_c, dim3(1), dim3(64), 0, s, c);
  }
  int rc = imagen_graph_begin(ps);
  if (rc == 0) {
    for (int i = 0; i < n; ++i) {
      if (i % 3 == 0) hipLaunchKernelGGL(probe_tick_a, dim3(1), dim3(64), 0, s, c);
      else if (i % 3 == 1) hipLaunchKernelGGL(probe_tick_b, dim3(1), dim3(64), 0, s, c);
      else hipLaunchKernelGGL(probe_tick_c, dim3(1), dim3(64), 0, s, c);
    }
    rc = imagen_graph_end(ps, &exec);
  }
  Timer t;
  if (rc == 0 && !t.ok) rc = -1;
  if (rc == 0) rc = imagen_graph_launch(exec, ps);   // warm replay
  if (rc == 0) {
    (void)hipEventRecord(t.e0, s);
    for (int r = 0; r < reps && rc == 0; ++r) rc = imagen_graph_launch(exec, ps);
    (void)hipEventRecord(t.e1, s);
    if (hipEventSynchronize(t.e1) != hipSuccess) rc = imagen_hip_status("probe_launch_chain");
  }
  if (rc == 0) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t.e0, t.e1);
    *us_per_launch = (float)((double)ms * 1e3 / ((double)n * reps));
  }
  if (exec) (void)imagen_graph_destroy(exec);
  (void)hipStreamSynchronize(s);
  (void)hipStreamDestroy(s);
  return rc;
}
