// Internal helpers shared by the gfx950 kernels.  Not part of the C ABI (see include/imagen_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "imagen_hip.h"
#include "lds_dma.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define IMAGEN_WAVE 64

void imagen_set_error(const char* fmt, ...);

#define IMAGEN_CHECK(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      imagen_set_error(__VA_ARGS__);       \
      return -1;                           \
    }                                      \
  } while (0)

static inline int imagen_hip_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    imagen_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: the staging prologue of the C=32 layers is VALU-bound
// (round-2 probe igemm_probe.py ablations), and SiLU runs once per staged element.  exp2 with the log2(e) fold saves the v_mul of __expf.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
// exact-erf GELU (nn.GELU default, ip.py:413) with erfc by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far inside the fp16 output
// rounding): libm's erff is ~60 instructions per element and was measured at 18k of the 27k cycles of a FeedForward GEMM's tile
// (round-2 probe insitu_trace.py, ff.lin1); this form is one v_rcp, one v_exp and seven multiply-adds
__device__ __forceinline__ float gelu_f(float v) {
  const float x = fabsf(v) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float ec = poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);   // erfc(|x|)
  return 0.5f * v * (v >= 0.f ? 2.0f - ec : ec);
}
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }

// device code bytes of a kernel of this library by mangled symbol name, 0 if unknown (codesize.hip)
unsigned imagen_kernel_code_bytes(const char* mangled);

// Instruction warm-up, first statement of a kernel: read the kernel's own code range as data (one dword per 128-byte line, all
// threads of the workgroup at once).  Consecutive launches of the denoiser step run different kernels, so a kernel otherwise
// starts with a serial chain of instruction-cache misses to HBM; after this single parallel round trip they are L2 hits.  `code_bytes`
// comes from the launcher (0 = skip); the last KiB is left out so the range never leaves the function by more than it started
// behind its entry.  Returns a value that the caller must "use" (imagen_code_warm_sink) once its own first loads have landed.
// (The loop compiles to a load + s_waitcnt vmcnt(0) per round — the first two waves of a workgroup block on it.  Round 5, call J, replaced it by
// four non-blocking global loads sunk later: unet1 2.170 ms per step against 2.151 with the blocking loop — the waves that run ahead only meet
// the instruction misses the wait would have covered.  The loop stays.)
__device__ __forceinline__ unsigned imagen_code_warm(unsigned code_bytes, int tid, int nthreads) {
  unsigned v = 0;
  if (code_bytes > 1024u) {
    const char* pc = reinterpret_cast<const char*>(__builtin_amdgcn_s_getpc());
    for (unsigned off = (unsigned)tid * 128u; off + 1024u < code_bytes; off += (unsigned)nthreads * 128u)
      v ^= *reinterpret_cast<const volatile unsigned*>(pc + off);
  }
  return v;
}
__device__ __forceinline__ void imagen_code_warm_sink(unsigned v) { IMAGEN_SINK(v); }

// 16-byte output pieces from the MFMA accumulator layout.  A lane of a 32x32 fragment holds channel quads 8q + 4*half + {0..3} of
// its pixel (lanes l and l + 32 share the pixel), i.e. 8-byte pieces at 16-byte stride.  Quads q and q + 2 are exchanged between the
// half-waves (v_permlane32_swap, one per dword): the lower half-wave then owns channels 8q .. 8q+7, the upper one 16+8q .. 16+8q+7 — one
// 16-byte store per lane instead of two 8-byte ones (half the store instructions, twice the bytes per memory request; measured on
// the streaming conv: 36.7 -> 32.0 us for 32->32 @256^2, round-2 probe stream_probe.py).  Must be executed by ALL lanes of the wave.
typedef unsigned imagen_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ imagen_u32x4 imagen_pair_quads(const f16x4& q_lo, const f16x4& q_hi) {
  const uint2 lo = __builtin_bit_cast(uint2, q_lo), hi = __builtin_bit_cast(uint2, q_hi);
  const auto r0 = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
  return imagen_u32x4{r0[0], r1[0], r0[1], r1[1]};
}
// The inverse: a 16-byte piece loaded at the paired position (couts 8q + 16 * half .. + 7) back into the lane's own quads q and q + 2
// (the swap is an involution).  Must be executed by ALL lanes of the wave.
__device__ __forceinline__ void imagen_unpair_quads(const imagen_u32x4& v, f16x4& q_lo, f16x4& q_hi) {
  const auto r0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
  q_lo = __builtin_bit_cast(f16x4, uint2{r0[0], r1[0]});
  q_hi = __builtin_bit_cast(f16x4, uint2{r0[1], r1[1]});
}

// op launchers (one per translation unit)
int launch_igemm(const ImagenIgemmParams* p, hipStream_t s);
int launch_rowstat(const ImagenRowstatParams* p, hipStream_t s);
int launch_attention(const ImagenAttentionParams* p, hipStream_t s);
int launch_kv_prep(const ImagenKvPrepParams* p, hipStream_t s);
int launch_qnorm(const ImagenQnormParams* p, hipStream_t s);
int launch_gca_partial(const ImagenGcaPartialParams* p, hipStream_t s);
int launch_gca_final(const ImagenGcaFinalParams* p, hipStream_t s);
int launch_gate_residual(const ImagenGateResidualParams* p, hipStream_t s);
int launch_ln_residual(const ImagenLnResidualParams* p, hipStream_t s);
int launch_time_embed(const ImagenTimeEmbedParams* p, hipStream_t s);
int launch_scale_shift(const ImagenScaleShiftParams* p, hipStream_t s);
int launch_linear_f32(const ImagenLinearF32Params* p, hipStream_t s);
int launch_pack_image(const ImagenPackImageParams* p, hipStream_t s);
int launch_cfg_x0(const ImagenCfgX0Params* p, hipStream_t s);
int launch_quantile(const ImagenQuantileParams* p, hipStream_t s);
int launch_ddpm_update(const ImagenDdpmUpdateParams* p, hipStream_t s);
int launch_rows_copy(const ImagenRowsCopyParams* p, hipStream_t s);
int launch_memset32(const ImagenMemset32Params* p, hipStream_t s);
int launch_select_rows(const ImagenSelectRowsParams* p, hipStream_t s);
int launch_mean_rows(const ImagenMeanRowsParams* p, hipStream_t s);
int launch_randn(const ImagenRandnParams* p, hipStream_t s);
int launch_lowres_prep(const ImagenLowresPrepParams* p, hipStream_t s);
int launch_lincomb(const ImagenLincombParams* p, hipStream_t s);
int launch_kv_prep_multi(const ImagenKvPrepMultiParams* p, hipStream_t s);
int launch_temporal_peg(const ImagenTemporalPegParams* p, hipStream_t s);
int launch_temporal_attention(const ImagenTemporalAttentionParams* p, hipStream_t s);
int launch_act_prep(const ImagenActPrepParams* p, hipStream_t s);
int launch_gca_tail(const ImagenGcaTailParams* p, hipStream_t s);
int launch_step_slice(const ImagenStepSliceParams* p, hipStream_t s);
int launch_rowchain(const ImagenRowchainParams* p, hipStream_t s);
