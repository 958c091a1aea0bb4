// temporal.hip — the two kernels the Imagen-Video denoiser (Unet3D) adds to the image path: the depthwise temporal PEG and the
// per-pixel attention over the frame axis.  Both are small HBM-/latency-bound vector kernels (F <= 32 frames): no MFMA.
// STATUS: compiled for gfx950 and specified by include/imagen_hip.h + tests/plan_interp.py; not yet run on a GPU.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ temporal PEG
// one lane = 8 channels of one (b, f, p) position: three 16-byte loads (frames f-2..f or f-1..f+1), one store
__global__ __launch_bounds__(256) void temporal_peg_kernel(const ImagenTemporalPegParams p) {
  const int groups = p.C >> 3;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)p.B * p.F * p.P * groups;
  if (i >= n) return;
  const int g = (int)(i % groups);
  const size_t pos = i / groups;                    // (b*F + f)*P + px
  const int f = (int)((pos / p.P) % p.F);
  const size_t frame = (size_t)p.P * p.C;           // elements per frame
  const f16* x = reinterpret_cast<const f16*>(p.x) + pos * p.C + g * 8;
  const int first = p.causal ? -2 : -1;
  float acc[8];
  const f16x8 centre = *reinterpret_cast<const f16x8*>(x);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = (float)centre[j] + p.bias[g * 8 + j];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int ff = f + first + k;
    if (ff < 0 || ff >= p.F) continue;
    const f16x8 v = *reinterpret_cast<const f16x8*>(x + (ptrdiff_t)(first + k) * (ptrdiff_t)frame);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += p.w[(g * 8 + j) * 3 + k] * (float)v[j];
  }
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (f16)acc[j];
  *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.out) + pos * p.C + g * 8) = o;
}

// ------------------------------------------------------------------------------------------------ temporal attention
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int kMaxFrames = 32;
constexpr int kKvRow = 64;   // floats per key / value row in LDS (lane d reads column d: conflict-free)

// One wave per (clip b, pixel px); lane d owns dimension d of the 64-wide head.  The F keys / values of the pixel (shared by all
// heads) and the null key / value are normalised once into LDS; then for every head and query frame the F+1 similarities are
// wave reductions, the (online) softmax is computed redundantly by every lane, and lane d accumulates output dimension d.
__global__ __launch_bounds__(256) void temporal_attention_kernel(const ImagenTemporalAttentionParams p) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t item = (size_t)blockIdx.x * 4 + wave;       // (b, px)
  if (item >= (size_t)p.B * p.P) return;                    // whole wave exits together
  const int b = (int)(item / p.P), px = (int)(item - (size_t)b * p.P);
  const int F = p.F, J = F + 1;
  float* kh = lds + (size_t)wave * 2 * (kMaxFrames + 1) * kKvRow;
  float* vv = kh + (kMaxFrames + 1) * kKvRow;
  const f16* base = reinterpret_cast<const f16*>(p.qkv) + ((size_t)b * F * p.P + px) * p.ld;
  const size_t fstride = (size_t)p.P * p.ld;                // elements between consecutive frames of one pixel
  const int inner = p.heads * 64;
  const float ks = p.k_scale[lane], qs = p.q_scale[lane] * p.scale;
  {  // null key / value (row 0), then the F frames
    const float nk = p.null_kv[lane], nv = p.null_kv[64 + lane];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(nk * nk)), 1e-12f);
    kh[lane] = nk * inv * ks;
    vv[lane] = nv;
  }
  for (int j = 0; j < F; ++j) {
    const f16* row = base + (size_t)j * fstride + inner;
    const float k = (float)row[lane], v = (float)row[64 + lane];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(k * k)), 1e-12f);
    kh[(1 + j) * kKvRow + lane] = k * inv * ks;
    vv[(1 + j) * kKvRow + lane] = v;
  }
  // (each lane only ever reads back column `lane`, which it wrote itself: no barrier needed)
  f16* obase = reinterpret_cast<f16*>(p.o) + ((size_t)b * F * p.P + px) * p.ld_o;
  const size_t ostride = (size_t)p.P * p.ld_o;
  for (int h = 0; h < p.heads; ++h) {
    const float* bias_h = p.bias + (size_t)h * F * J;
    for (int i = 0; i < F; ++i) {
      const float q = (float)base[(size_t)i * fstride + h * 64 + lane];
      const float qn = q * (1.0f / fmaxf(sqrtf(wave_sum(q * q)), 1e-12f)) * qs;
      const int last = p.causal ? i + 1 : F;                // keys 0 (null) .. last are visible
      float mx = -3.0e38f, den = 0.f, acc = 0.f;             // online softmax: no per-key array (it would live in scratch)
      for (int j = 0; j <= last; ++j) {
        const float s = wave_sum(qn * kh[j * kKvRow + lane]) + bias_h[i * J + j];
        const float mn = fmaxf(mx, s);
        const float c = __expf(mx - mn), e = __expf(s - mn);
        den = den * c + e;
        acc = acc * c + e * vv[j * kKvRow + lane];
        mx = mn;
      }
      obase[(size_t)i * ostride + h * 64 + lane] = (f16)(acc / den);
    }
  }
}

}  // namespace

int launch_temporal_peg(const ImagenTemporalPegParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->x && p->w && p->bias && p->out, "temporal_peg: null pointer");
  IMAGEN_CHECK(p->C % 8 == 0 && p->B > 0 && p->F > 0 && p->P > 0, "temporal_peg: bad shape");
  const size_t n = (size_t)p->B * p->F * p->P * (p->C / 8);
  hipLaunchKernelGGL(temporal_peg_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *p);
  return imagen_hip_status("temporal_peg");
}

int launch_temporal_attention(const ImagenTemporalAttentionParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->qkv && p->null_kv && p->q_scale && p->k_scale && p->bias && p->o, "temporal_attention: null pointer");
  IMAGEN_CHECK(p->F > 0 && p->F <= kMaxFrames, "temporal_attention: 1 <= F <= 32");
  IMAGEN_CHECK(p->heads > 0 && p->B > 0 && p->P > 0, "temporal_attention: bad shape");
  const size_t items = (size_t)p->B * p->P;
  const size_t lds_bytes = (size_t)4 * 2 * (kMaxFrames + 1) * kKvRow * sizeof(float);
  static bool attr_set[16] = {};   // per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(temporal_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(temporal_attention_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), lds_bytes, s, *p);
  return imagen_hip_status("temporal_attention");
}
