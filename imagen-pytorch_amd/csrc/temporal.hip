// temporal.hip — the two kernels the Imagen-Video denoiser (Unet3D) adds to the image path: the depthwise temporal PEG and the
// per-pixel attention over the frame axis (a vector kernel for any F <= 32, and — round 4 — the MFMA kernel that takes every F <= 31).
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

// ---- the same attention on the matrix pipe (F <= 31: the F + 1 keys fit one 32-key tile).  Round-4 kernel table of BASELINE C5
// (profiles/r04_c5_kernel_stats.csv): the kernel above is 57 % of the C5 step — 757 us per launch for 235 MB of traffic, every one of the
// heads x F x (F + 1) similarities of a pixel a six-step wave reduction.  Here a wave still owns one (clip, pixel), but its heads x F query
// rows are MFMA rows: per block of 32 rows  S^T[key][row] = K^ . Q^T (v_mfma_f32_32x32x16_f16, the attention.hip layouts: lane = row, 16
// keys per lane), bias + causal mask + softmax in registers (one cross-half shuffle), O^T += V^T . P.  K^ and Q^ enter the MFMA as
// fp16 hi + lo pairs (three products per K step: hi.hi + lo.hi + hi.lo), so the logits keep the fp32 accuracy of the kernel above — they
// reach 18 with the scale vectors the reference trains, where a bare fp16 operand would cost 5e-3 in the softmax weights; P is fp16 as in
// attention.hip.  V^T (dims x 32 keys, fp16) and the bias table go through LDS; per pixel 64 MFMAs instead of ~1200 wave reductions.
constexpr int kTaVtRow = 72;   // LDS bytes per V^T row (32 keys x 2 B + 8: conflict-free ds_read_b64, attention.hip's VSTR)

__device__ __forceinline__ void ta_split(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hi[j] = (f16)x[j];
    lo[j] = (f16)(x[j] - (float)hi[j]);
  }
}

__global__ __launch_bounds__(256) void temporal_attention_mfma_kernel(const ImagenTemporalAttentionParams p) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int F = p.F, J = F + 1;
  // ---- the bias table of all heads -> LDS (shared by the four waves)
  float* s_bias = lds;                                                   // [heads][F][J]
  const int nb = p.heads * F * J;
  for (int i = threadIdx.x; i < nb; i += 256) s_bias[i] = p.bias[i];
  char* vt = reinterpret_cast<char*>(lds + ((nb + 3) & ~3)) + (size_t)wave * 64 * kTaVtRow;   // this wave's V^T: [64 dims][32 keys] fp16
  __syncthreads();
  const size_t item = (size_t)blockIdx.x * 4 + wave;       // (b, px)
  if (item >= (size_t)p.B * p.P) return;                    // (behind the only workgroup barrier)
  const int b = (int)(item / p.P), px = (int)(item - (size_t)b * p.P);
  const f16* base = reinterpret_cast<const f16*>(p.qkv) + ((size_t)b * F * p.P + px) * p.ld;
  const size_t fstride = (size_t)p.P * p.ld;
  const int inner = p.heads * 64;
  const int rows = p.heads * F;
  f16x8 qraw[4];   // the query rows of a 32-row block as loaded (row r0 + l31 = (head, frame), dims 16 s + 8 half ..)
  auto q_fetch = [&](int r0) __attribute__((always_inline)) {
    const int r = r0 + l31;
    const int h = r < rows ? r / F : 0, i = r < rows ? r - h * F : 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) qraw[s] = *reinterpret_cast<const f16x8*>(base + (size_t)i * fstride + h * 64 + 16 * s + 8 * half);
  };
  q_fetch(0);      // (in flight behind the V^T gather and the K^ rows)

  // ---- V^T: lane d gathers column d of the J value rows (null value first); keys J..31 are zero
  {
    f16* col = reinterpret_cast<f16*>(vt + lane * kTaVtRow);
    col[0] = (f16)p.null_kv[64 + lane];
    for (int j = 0; j < F; ++j) col[1 + j] = base[(size_t)j * fstride + inner + 64 + lane];
    for (int j = J; j < 32; ++j) col[j] = (f16)0.f;
  }
  // the tile is wave-private, but other LANES of the wave read what this lane stored: order the stores before the PV fragment reads
  // (hardware issues a wave's LDS operations in order; the fence keeps the compiler — and the CPU emulation's fibers — to that order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- K^ fragments (A operand: lane = key l31, dims 16 s + 8 half ..): l2norm * k_scale, as fp16 hi + lo; key 0 = the null key
  f16x8 kh[4], kl[4];
  {
    float kx[4][8];
    float ssq = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int d0 = 16 * s + 8 * half;
      if (l31 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) kx[s][j] = p.null_kv[d0 + j];
      } else if (l31 < J) {
        const f16x8 r = *reinterpret_cast<const f16x8*>(base + (size_t)(l31 - 1) * fstride + inner + d0);
#pragma unroll
        for (int j = 0; j < 8; ++j) kx[s][j] = (float)r[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) kx[s][j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ssq += kx[s][j] * kx[s][j];
    }
    ssq += __shfl_xor(ssq, 32);
    const float inv = 1.0f / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int d0 = 16 * s + 8 * half;
#pragma unroll
      for (int j = 0; j < 8; ++j) kx[s][j] *= inv * (p.k_scale[d0 + j] * p.q_scale[d0 + j] * p.scale);   // (q_scale * scale ride on K^: Q^ is the unit row)
      ta_split(kx[s], kh[s], kl[s]);
    }
  }
  // The null value is an fp32 parameter and the same vector for every pixel: its fp16 part rides the MFMA as V^T column 0, the remainder
  // nv - fp16(nv) is added behind it on the VALU with the null key's fp32 weight (32 FMAs per row block).  Dropped, it is a coherent
  // bias of the whole map — a first frame under the causal mask gives the null key about half its weight — and the C5 denoiser's
  // distance to the oracle moved 1.00e-3 -> 1.05e-3 when this kernel replaced the fp32 vector kernel in round 4 (plan interpreter with
  // the null value rounded to fp16: 1.008e-3 -> 1.044e-3, round-5 session 2).
  float nvlo[2][16];   // [db][4 qd + e]: dim 32 db + 8 qd + 4 half + e, the accumulator layout of O^T below
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float nv = p.null_kv[64 + 32 * db + 8 * (e >> 2) + 4 * half + (e & 3)];
      nvlo[db][e] = nv - (float)(f16)nv;
    }
  f16* obase = reinterpret_cast<f16*>(p.o) + ((size_t)b * F * p.P + px) * p.ld_o;
  const size_t ostride = (size_t)p.P * p.ld_o;
  for (int r0 = 0; r0 < rows; r0 += 32) {
    // ---- Q^ fragments of row r0 + l31 = (head, frame) (B operand: lane = row, the same dims); the rows of the NEXT block are requested
    //      as soon as this block's are in fp32 registers: a wave is one of two per SIMD (244 registers), the loads of one block at a time
    //      left the kernel at 2 TB/s (profiles/r06_k_c5_kernel_stats.csv)
    const int r = r0 + l31;
    const bool rok = r < rows;
    const int h = rok ? r / F : 0, i = rok ? r - h * F : 0;
    f16x8 qh[4], ql[4];
    {
      float qx[4][8];
      float ssq = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          qx[s][j] = rok ? (float)qraw[s][j] : 0.f;
          ssq += qx[s][j] * qx[s][j];
        }
      }
      if (r0 + 32 < rows) q_fetch(r0 + 32);
      ssq += __shfl_xor(ssq, 32);
      const float inv = 1.0f / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qx[s][j] *= inv;
        ta_split(qx[s], qh[s], ql[s]);
      }
    }
    // ---- S^T[key][row]: register e of this lane = key (e & 3) + 8 (e >> 2) + 4 half of row l31
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[s], qh[s], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[s], ql[s], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[s], qh[s], sacc, 0, 0, 0);
    }
    // ---- bias, causal mask (keys 0 (null) .. last are visible), softmax over the 32 key slots (this lane's 16 + lane ^ 32's)
    const int last = p.causal ? i + 1 : F;
    const float* brow = s_bias + ((size_t)h * F + i) * J;
    float mx = -3.0e38f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (e & 3) + 8 * (e >> 2) + 4 * half;
      const bool vis = key <= last && key < J;
      sacc[e] = vis ? sacc[e] + brow[vis ? key : 0] : -3.0e38f;
      mx = fmaxf(mx, sacc[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float den = 0.f;
    f16x8 pf[2], pl[2];   // the softmax weights as fp16 hi + lo pairs too (round 5): a bare fp16 P cost the C5 denoiser 6 % of its 1e-3 budget
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float w = sacc[e] > -1.0e38f ? __expf(sacc[e] - mx) : 0.f;
      den += w;
      const f16 wh = (f16)w;
      pf[e >> 3][e & 7] = wh;
      pl[e >> 3][e & 7] = (f16)(w - (float)wh);
    }
    den += __shfl_xor(den, 32);
    // the null key is key 0 = register 0 of the half-0 lane of the row
    const float w_null = __shfl(sacc[0] > -1.0e38f ? __expf(sacc[0] - mx) : 0.f, l31);
    // ---- O^T[d][row] += V^T . P   (k-step s covers the keys of accumulator registers 8 s .. 8 s + 7)
    f32x16 oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[db][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const char* vrow = vt + (32 * db + l31) * kTaVtRow + (16 * s + 4 * half) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16);
        uint4 packed = make_uint4(lo.x, lo.y, hi.x, hi.y);
        const f16x8 vf = *reinterpret_cast<const f16x8*>(&packed);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pl[s], oacc[db], 0, 0, 0);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], oacc[db], 0, 0, 0);
      }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[db][e] += w_null * nvlo[db][e];
    if (rok) {
      const float inv = 1.0f / den;
      f16* o = obase + (size_t)i * ostride + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(oacc[db][4 * qd + e] * inv);
          *reinterpret_cast<f16x4*>(o + 32 * db + 8 * qd + 4 * half) = v;
        }
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
  const size_t nb = (size_t)p->heads * p->F * (p->F + 1);
  const size_t lds = ((nb + 3) & ~(size_t)3) * sizeof(float) + (size_t)4 * 64 * kTaVtRow;
  // the MFMA kernel: F + 1 keys in one 32-key tile, aligned rows, the bias table of all heads in 64 KB of LDS (12+ heads at F = 31 do not
  // fit: those shapes keep the vector kernel below, as every shape did before round 4)
  if (p->F <= 31 && p->ld % 8 == 0 && p->ld_o % 4 == 0 && ((size_t)p->qkv & 15) == 0 && ((size_t)p->o & 7) == 0 && lds <= 64 * 1024) {
    hipLaunchKernelGGL(temporal_attention_mfma_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), lds, s, *p);
    return imagen_hip_status("temporal_attention");
  }
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
