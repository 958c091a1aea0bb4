// pipe_probe.hip — issue-rate probe for gfx950 (a measurement tool, not on the sampling path): how many cycles do v_exp_f32, the softmax's
// VALU mix and v_mfma_f32_32x32x16_f16 take alone, and how much of the VALU work hides behind the matrix pipe when one wave interleaves
// them?  Everything is reported in units of the MFMA-only loop (one MFMA = 32 cycles when the matrix pipe is never starved), so the clock
// cancels.  Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/pipe_probe tools/probe/pipe_probe.hip && /tmp/pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// per iteration: NM MFMAs (two accumulators) and NE exponentials (+ the sums / conversions when MIX) on independent registers
template <int NM, int NE, bool MIX, bool PIN>
__global__ __launch_bounds__(512) void pipe_kernel(float* sink, int iters) {
  f32x16 acc[2];
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (f16)(0.001f * (float)((threadIdx.x + j) & 7));
    b[j] = (f16)(0.002f * (float)((threadIdx.x * 3 + j) & 7));
  }
  float x[32], s0 = 0.f, s1 = 0.f;
  for (int r = 0; r < 32; ++r) x[r] = 0.001f * (float)(threadIdx.x + r);
  unsigned pk = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 32; ++r) asm volatile("" : "+v"(x[r]));
    float e[32];
#pragma unroll
    for (int r = 0; r < NE; ++r) e[r] = __builtin_amdgcn_exp2f(x[r]);
    if constexpr (MIX) {
#pragma unroll
      for (int r = 0; r < NE; r += 2) {
        s0 += e[r];
        s1 += e[r + 1];
        typedef f16 f16x2 __attribute__((ext_vector_type(2)));
        f16x2 h = {(f16)e[r], (f16)e[r + 1]};
        pk ^= __builtin_bit_cast(unsigned, h);
      }
    } else {
#pragma unroll
      for (int r = 0; r < NE; ++r) asm volatile("" ::"v"(e[r]));
    }
#pragma unroll
    for (int k = 0; k < NM; ++k) acc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 1], 0, 0, 0);
    if constexpr (PIN && NM > 0 && NE > 0) {
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x400, NE / NM, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, MIX ? (3 * NE / 2) / NM / 2 + 1 : 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = s0 + s1 + (float)pk;
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 12345.678f) sink[0] = s;
}

template <class K>
static float run(K kern, int threads, int iters, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, sink, iters);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, sink, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms * 1e6f / 5.0f / (float)iters;   // ns per iteration
}

int main() {
  float* sink;
  hipMalloc(&sink, 256);
  const int iters = 20000;
  printf("{\"iters\": %d", iters);
  for (int threads : {256, 512}) {   // one / two waves per SIMD (one workgroup per CU)
    const float m8 = run(pipe_kernel<8, 0, false, false>, threads, iters, sink);
    const float cyc = m8 / (8.0f * 32.0f) / (threads / 256);   // ns per cycle, calibrated on the MFMA-only loop (waves share the matrix pipe)
    const float e32 = run(pipe_kernel<0, 32, false, false>, threads, iters, sink);
    const float x32 = run(pipe_kernel<0, 32, true, false>, threads, iters, sink);
    const float b_free = run(pipe_kernel<8, 32, true, false>, threads, iters, sink);
    const float b_pin = run(pipe_kernel<8, 32, true, true>, threads, iters, sink);
    const float e_pin = run(pipe_kernel<8, 32, false, true>, threads, iters, sink);
    const float b16 = run(pipe_kernel<8, 16, true, true>, threads, iters, sink);
    printf(", \"waves_per_simd_%d\": {\"ns_per_cycle\": %.4f, \"mfma8_cycles\": %.0f, \"exp32_cycles\": %.0f, \"softmax_mix32_cycles\": %.0f, "
           "\"mfma8_plus_mix32_compiler_order_cycles\": %.0f, \"mfma8_plus_mix32_interleaved_cycles\": %.0f, \"mfma8_plus_exp32_interleaved_cycles\": %.0f, "
           "\"mfma8_plus_mix16_interleaved_cycles\": %.0f}",
           threads / 256, cyc, m8 / cyc, e32 / cyc, x32 / cyc, b_free / cyc, b_pin / cyc, e_pin / cyc, b16 / cyc);
  }
  printf("}\n");
  return 0;
}
