// Internal helpers shared by the gfx950 kernels.  Not part of the C ABI (see include/imagen_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "imagen_hip.h"

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
// (tools/igemm_probe.py ablations), and SiLU runs once per staged element.  exp2 with the log2(e) fold saves the v_mul of __expf.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }

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
