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
