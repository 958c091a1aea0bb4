// elementwise.hip — the HBM-bound glue of the Imagen denoiser (statistics, gates, residuals, token
// assembly, embeddings).  All kernels move 16 B per lane (8 fp16) with consecutive lanes on consecutive
// addresses; reductions over a row use a power-of-two sub-group of the wave64 and __shfl_xor.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "gca_device.h"

namespace {

__device__ __forceinline__ int lanes_per_row(int groups) {  // smallest power of two >= groups, capped at 64
  int l = 1;
  while (l < groups && l < 64) l <<= 1;
  return l;
}

__device__ __forceinline__ float group_sum(float v, int lpr) {
  for (int off = lpr >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------------------------------------------ rowstat
__global__ __launch_bounds__(256) void rowstat_kernel(const ImagenRowstatParams p, int lpr) {
  const int rows_per_block = 256 / lpr;
  const int sub = threadIdx.x / lpr, li = threadIdx.x % lpr;
  const int r = blockIdx.x * rows_per_block + sub;
  if (r >= p.rows) return;  // whole sub-group exits together
  const int b = r / p.rows_per_batch, rr = r - b * p.rows_per_batch;
  const f16* x1 = reinterpret_cast<const f16*>(p.x1) + (size_t)b * p.bs1 + (size_t)rr * p.ld1;
  const f16* x2 = p.x2 ? reinterpret_cast<const f16*>(p.x2) + (size_t)b * p.bs2 + (size_t)rr * p.ld2 : nullptr;
  const int g1 = p.C1 >> 3, g2 = p.x2 ? (p.C2 >> 3) : 0;
  if (p.mode == 0 || p.mode == 2) {
    float s1 = 0.f, s2 = 0.f;
    for (int g = li; g < g1; g += lpr) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(x1 + g * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s1 += (float)v[j] * (float)v[j];
    }
    for (int g = li; g < g2; g += lpr) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(x2 + g * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s2 += (float)v[j] * (float)v[j];
    }
    const float tot = group_sum(s1 + p.w2 * s2, lpr);
    if (li == 0) p.rs[r] = p.mode == 2 ? tot : 1.0f / fmaxf(sqrtf(tot), 1e-12f);
  } else {
    const int C = p.C1 + (p.x2 ? p.C2 : 0);
    float s = 0.f;
    for (int g = li; g < g1; g += lpr) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(x1 + g * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[j];
    }
    for (int g = li; g < g2; g += lpr) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(x2 + g * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[j];
    }
    const float mean = group_sum(s, lpr) / (float)C;
    float q = 0.f;
    for (int g = li; g < g1; g += lpr) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(x1 + g * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[j] - mean; q += d * d; }
    }
    for (int g = li; g < g2; g += lpr) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(x2 + g * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[j] - mean; q += d * d; }
    }
    const float var = group_sum(q, lpr) / (float)C;
    if (li == 0) {
      p.mu[r] = mean;
      p.rs[r] = rsqrtf(var + p.eps);
    }
  }
}

// ------------------------------------------------------------------------------------------------ gate_residual
__global__ __launch_bounds__(256) void gate_residual_kernel(const ImagenGateResidualParams p, int lpr) {
  const int rows_per_block = 256 / lpr;
  const int sub = threadIdx.x / lpr, li = threadIdx.x % lpr;
  const int r = blockIdx.x * rows_per_block + sub;
  if (r >= p.rows) return;
  const int b = r / p.rows_per_batch;
  const f16* h = reinterpret_cast<const f16*>(p.h) + (size_t)r * p.ld_h;
  const f16* res = reinterpret_cast<const f16*>(p.res) + (size_t)r * p.ld_res;
  f16* out = reinterpret_cast<f16*>(p.out) + (size_t)r * p.ld_out;
  const int groups = p.C >> 3;
  float ssq = 0.f;
  for (int g = li; g < groups; g += lpr) {
    const f16x8 hv = *reinterpret_cast<const f16x8*>(h + g * 8);
    const f16x8 rv = *reinterpret_cast<const f16x8*>(res + g * 8);
    float gt[8];
    if (p.gate) {
      const float4* gp = reinterpret_cast<const float4*>(p.gate + (size_t)b * p.C + g * 8);
      const float4 g0 = gp[0], g1 = gp[1];
      gt[0] = g0.x; gt[1] = g0.y; gt[2] = g0.z; gt[3] = g0.w; gt[4] = g1.x; gt[5] = g1.y; gt[6] = g1.z; gt[7] = g1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) gt[j] = 1.0f;
    }
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (float)hv[j] * gt[j] + (float)rv[j];
      o[j] = (f16)v;
      const float vr = (float)o[j];  // statistics of the value the consumer will actually read
      ssq += vr * vr;
    }
    *reinterpret_cast<f16x8*>(out + g * 8) = o;
  }
  if (p.rs_out) {
    const float tot = group_sum(ssq, lpr);
    if (li == 0) p.rs_out[r] = p.raw_ssq ? tot : 1.0f / fmaxf(sqrtf(tot), 1e-12f);
  }
}

// ------------------------------------------------------------------------------------------------ ln_residual
__global__ __launch_bounds__(256) void ln_residual_kernel(const ImagenLnResidualParams p, int lpr) {
  const int rows_per_block = 256 / lpr;
  const int sub = threadIdx.x / lpr, li = threadIdx.x % lpr;
  const int r = blockIdx.x * rows_per_block + sub;
  if (r >= p.rows) return;
  const int bb = r / p.rows_per_batch, rr = r - bb * p.rows_per_batch;
  const f16* y = reinterpret_cast<const f16*>(p.y) + (size_t)bb * p.bs_y + (size_t)rr * p.ld_y;
  const f16* res = p.res ? reinterpret_cast<const f16*>(p.res) + (size_t)bb * p.bs_res + (size_t)rr * p.ld_res : nullptr;
  f16* out = reinterpret_cast<f16*>(p.out) + (size_t)bb * p.bs_out + (size_t)rr * p.ld_out;
  const int groups = p.C >> 3;
  float s = 0.f;
  for (int g = li; g < groups; g += lpr) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(y + g * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)v[j];
  }
  const float mean = group_sum(s, lpr) / (float)p.C;
  float q = 0.f;
  for (int g = li; g < groups; g += lpr) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(y + g * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = (float)v[j] - mean; q += d * d; }
  }
  const float rstd = rsqrtf(group_sum(q, lpr) / (float)p.C + p.eps);
  float ssq = 0.f, so = 0.f;
  for (int g = li; g < groups; g += lpr) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(y + g * 8);
    f16x8 rv;
    if (res) rv = *reinterpret_cast<const f16x8*>(res + g * 8);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = ((float)v[j] - mean) * rstd * p.g[g * 8 + j];
      if (p.beta) t += p.beta[g * 8 + j];
      if (res) t += (float)rv[j];
      o[j] = (f16)t;
      const float tr = (float)o[j];
      ssq += tr * tr;
      so += tr;
    }
    *reinterpret_cast<f16x8*>(out + g * 8) = o;
  }
  if (p.ssq_out) {
    const float tot = group_sum(ssq, lpr);
    if (li == 0) p.ssq_out[r] = tot;
  }
  if (p.mu_out) {   // LayerNorm statistics of the stored row (two-pass: a lane re-reads the groups it has just written)
    const float mo = group_sum(so, lpr) / (float)p.C;
    float qo = 0.f;
    for (int g = li; g < groups; g += lpr) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(out + g * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[j] - mo; qo += d * d; }
    }
    const float ro = rsqrtf(group_sum(qo, lpr) / (float)p.C + p.eps_out);
    if (li == 0) {
      p.mu_out[r] = mo;
      p.rs_out[r] = ro;
    }
  }
}

// ------------------------------------------------------------------------------------------------ q / kv preparation
// head_dim / 8 lanes (8 | 4) per head row; each lane owns 8 consecutive dims.
__device__ __forceinline__ int head_dim_of(int v) { return v == 32 ? 32 : 64; }

__global__ __launch_bounds__(256) void qnorm_kernel(const ImagenQnormParams p) {
  const int D = head_dim_of(p.head_dim), lpr = D >> 3, sh = D == 64 ? 3 : 2;
  const int item = blockIdx.x * (256 >> sh) + (threadIdx.x >> sh);  // (row, head)
  const int dg = threadIdx.x & (lpr - 1);
  if (item >= p.rows * p.heads) return;
  const int row = item / p.heads, hd = item - row * p.heads;
  f16* q = reinterpret_cast<f16*>(p.q) + (size_t)row * p.ld + hd * D + dg * 8;
  const f16x8 v = *reinterpret_cast<const f16x8*>(q);
  float ssq = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) ssq += (float)v[j] * (float)v[j];
  ssq = group_sum(ssq, lpr);
  const float inv = p.mult / fmaxf(sqrtf(ssq), 1e-12f);
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (f16)((float)v[j] * inv * p.q_scale[dg * 8 + j]);
  *reinterpret_cast<f16x8*>(q) = o;
}

__device__ __forceinline__ void kv_prep_body(const ImagenKvPrepParams& p, int bx, int bh) {
  if (bh >= p.B * p.heads) return;
  const int b = bh / p.heads, hd = bh - b * p.heads;
  const int D = head_dim_of(p.head_dim), lpr = D >> 3, sh = D == 64 ? 3 : 2;
  const int row = bx * (256 >> sh) + (threadIdx.x >> sh);
  const int dg = threadIdx.x & (lpr - 1);
  if (row >= p.rows) return;
  const size_t soff = (size_t)b * p.src_bs + (size_t)row * p.src_rs + (size_t)hd * p.src_hs + dg * 8;
  float kv[8], vv[8];
  if (p.src_is_f32) {
    const float* ks = reinterpret_cast<const float*>(p.k_src) + soff;
    const float* vs = reinterpret_cast<const float*>(p.v_src) + soff;
#pragma unroll
    for (int j = 0; j < 8; ++j) { kv[j] = ks[j]; vv[j] = vs[j]; }
  } else {
    const f16x8 k8 = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p.k_src) + soff);
    const f16x8 v8 = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p.v_src) + soff);
#pragma unroll
    for (int j = 0; j < 8; ++j) { kv[j] = (float)k8[j]; vv[j] = (float)v8[j]; }
  }
  float ssq = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) ssq += kv[j] * kv[j];
  ssq = group_sum(ssq, lpr);
  const float inv = 1.0f / fmaxf(sqrtf(ssq), 1e-12f);
  f16x8 ko;
#pragma unroll
  for (int j = 0; j < 8; ++j) ko[j] = (f16)(kv[j] * inv * p.k_scale[dg * 8 + j]);
  f16* kh = reinterpret_cast<f16*>(p.khat) + (size_t)b * p.k_bs + (size_t)hd * p.k_hs + (size_t)(p.r0 + row) * p.k_rs + dg * 8;
  *reinterpret_cast<f16x8*>(kh) = ko;
  f16* vt = reinterpret_cast<f16*>(p.vt) + (size_t)b * p.vt_bs + (size_t)hd * p.vt_hs + (p.r0 + row);
#pragma unroll
  for (int j = 0; j < 8; ++j) vt[(size_t)(dg * 8 + j) * p.vt_ds] = (f16)vv[j];
}

__global__ __launch_bounds__(256) void kv_prep_kernel(const ImagenKvPrepParams p) { kv_prep_body(p, blockIdx.x, blockIdx.y); }

// grid.z = job: the params of the job are read from device memory (uniform address: scalar loads)
__global__ __launch_bounds__(256) void kv_prep_multi_kernel(const ImagenKvPrepMultiParams m) {
  const ImagenKvPrepParams p = m.jobs[blockIdx.z];
  kv_prep_body(p, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------ global context
// Block = one chunk of pixels of one image.  Phase 1: logits -> LDS, block max.  Phase 2: thread (pixel-lane,
// 8-channel group) accumulates exp(logit - max) * h over its pixels; cross-lane reduction through LDS.
constexpr int kGcaMaxChunk = 1024;

__global__ __launch_bounds__(256) void gca_partial_kernel(const ImagenGcaPartialParams p, int chunk_px) {
  __shared__ float s_logit[kGcaMaxChunk];
  __shared__ float s_red[256];
  __shared__ float s_acc[256 * 8];
  __shared__ float s_fin[2048];   // chunks == 1: ctx [C] | hid [hidden] | kGcaScratchFloats (C + hidden <= 1024 checked by the launcher)
  const int b = blockIdx.y, ch = blockIdx.x;
  const int p0 = ch * chunk_px;
  const int npx = min(chunk_px, p.HW - p0);
  const f16* h = reinterpret_cast<const f16*>(p.h) + ((size_t)b * p.HW + p0) * p.ld;
  const int groups = p.C >> 3;
  const int lpr = lanes_per_row(groups);
  const int ppb = 256 / lpr;
  const int sub = threadIdx.x / lpr, li = threadIdx.x % lpr;
  // phase 1
  float lmax = -3.0e38f;
  for (int px = sub; px < npx; px += ppb) {
    float d = 0.f;
    for (int g = li; g < groups; g += lpr) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(h + (size_t)px * p.ld + g * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) d += (float)v[j] * p.wk[g * 8 + j];
    }
    d = group_sum(d, lpr) + p.bk;
    if (li == 0) s_logit[px] = d;
    lmax = fmaxf(lmax, d);
  }
  s_red[threadIdx.x] = lmax;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) s_red[threadIdx.x] = fmaxf(s_red[threadIdx.x], s_red[threadIdx.x + off]);
    __syncthreads();
  }
  const float m = s_red[0];
  __syncthreads();
  // phase 2: thread -> (pixel lane pl, channel group cg)
  const int npl = 256 / groups;  // pixel lanes (threads beyond npl*groups idle)
  const int pl = threadIdx.x / groups, cg = threadIdx.x - pl * groups;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float se = 0.f;
  if (pl < npl) {
    for (int px = pl; px < npx; px += npl) {
      const float e = __expf(s_logit[px] - m);
      const f16x8 v = *reinterpret_cast<const f16x8*>(h + (size_t)px * p.ld + cg * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += e * (float)v[j];
      if (cg == 0) se += e;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s_acc[threadIdx.x * 8 + j] = acc[j];
  s_red[threadIdx.x] = se;
  __syncthreads();
  float* out = p.part + ((size_t)b * p.chunks + ch) * (p.C + 2);
  if (threadIdx.x < groups) {  // reduce over pixel lanes for channel group threadIdx.x
    float tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < npl; ++q) {
#pragma unroll
      for (int j = 0; j < 8; ++j) tot[j] += s_acc[(q * groups + threadIdx.x) * 8 + j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) out[2 + threadIdx.x * 8 + j] = tot[j];
  }
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int q = 0; q < npl; ++q) tot += s_red[q * groups];
    out[0] = m;
    out[1] = tot;
  }
}

// Single-pass variant (power-of-two C/8): thread = (pixel lane, 8-channel group); the `groups` consecutive lanes of a pixel
// reduce the logit with __shfl_xor, then every thread folds exp(logit - m) * h into its 8 accumulators with an online
// (running-max) rescale, so h is read ONCE.  Pixel lanes are merged through LDS at the end.
__global__ __launch_bounds__(256) void gca_partial_online_kernel(const ImagenGcaPartialParams p, int chunk_px) {
  __shared__ float s_m[256];
  __shared__ float s_se[256];
  __shared__ float s_acc[256 * 8];
  __shared__ float s_fin[2048];   // chunks == 1: ctx [C] | hid [hidden] | kGcaScratchFloats (C + hidden <= 1024 checked by the launcher)
  const int b = blockIdx.y, ch = blockIdx.x;
  const int p0 = ch * chunk_px;
  const int npx = min(chunk_px, p.HW - p0);
  const f16* h = reinterpret_cast<const f16*>(p.h) + ((size_t)b * p.HW + p0) * p.ld;
  const int groups = p.C >> 3;           // power of two <= 64
  const int npl = 256 / groups;          // pixel lanes
  const int pl = threadIdx.x / groups, cg = threadIdx.x & (groups - 1);
  float wk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) wk[j] = p.wk[cg * 8 + j];
  float m = -3.0e38f, se = 0.f;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int U = 8;   // pixels in flight per lane (the loop is a load-latency chain otherwise)
  for (int px0 = pl; px0 < npx; px0 += npl * U) {   // uniform trip count within each `groups`-lane team
    f16x8 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int px = px0 + u * npl;
      if (px < npx) v[u] = *reinterpret_cast<const f16x8*>(h + (size_t)px * p.ld + cg * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (px0 + u * npl < npx) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d += (float)v[u][j] * wk[j];
        d = group_sum(d, groups) + p.bk;
        const float mn = fmaxf(m, d);
        const float sc = __expf(m - mn), e = __expf(d - mn);
        m = mn;
        se = se * sc + e;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc[j] * sc + e * (float)v[u][j];
      }
    }
  }
  // merge the pixel lanes: inside a wave by butterfly shuffles over the lanes that share a channel group (lane distance
  // groups, 2*groups, ... 32), then the 4 waves through LDS
  for (int off = groups; off < 64; off <<= 1) {
    const float m2 = __shfl_xor(m, off), se2 = __shfl_xor(se, off);
    const float mn = fmaxf(m, m2);
    const float w1 = __expf(m - mn), w2 = __expf(m2 - mn);   // lanes that saw no pixel carry m = -3e38, se = acc = 0
    se = se * w1 + se2 * w2;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] * w1 + __shfl_xor(acc[j], off) * w2;
    m = mn;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = groups < 64 ? groups : 64;   // channel groups per wave
  if (lane < gw) {
    const int slot = wave * gw + lane;        // groups == 64: the wave's lanes ARE the groups; else every wave holds all groups
    s_m[slot] = m;
    s_se[slot] = se;
#pragma unroll
    for (int j = 0; j < 8; ++j) s_acc[slot * 8 + j] = acc[j];
  }
  __syncthreads();
  float* out = p.part + ((size_t)b * p.chunks + ch) * (p.C + 2);
  if (threadIdx.x < groups) {  // merge the 4 waves' entries of channel group threadIdx.x
    const int g = threadIdx.x & (gw - 1);
    float M = -3.0e38f;
    for (int q = 0; q < 4; ++q) M = fmaxf(M, s_m[q * gw + g]);
    float tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float S = 0.f;
    for (int q = 0; q < 4; ++q) {
      const float w = __expf(s_m[q * gw + g] - M);
      S += s_se[q * gw + g] * w;
#pragma unroll
      for (int j = 0; j < 8; ++j) tot[j] += s_acc[(q * gw + g) * 8 + j] * w;
    }
    if (p.chunks == 1 && p.w1t != nullptr) {
      // the whole image was this workgroup's (small feature maps): the pooled context goes straight to LDS (no partials in
      // global memory, no merge over chunks) and the squeeze MLP runs here
      const float inv = 1.0f / S;
#pragma unroll
      for (int j = 0; j < 8; ++j) s_fin[threadIdx.x * 8 + j] = tot[j] * inv;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) out[2 + threadIdx.x * 8 + j] = tot[j];
      if (threadIdx.x == 0) {
        out[0] = M;
        out[1] = S;
      }
    }
  }
  if (p.w1t == nullptr) return;   // partials only: a GCA_FINAL launch follows
  // one chunk per image (launcher-checked): the whole image was this workgroup's, the squeeze MLP runs here
  __syncthreads();
  gca_mlp(s_fin, s_fin + p.C, s_fin + p.C + p.hidden, p.C, p.hidden, p.w1t, p.b1, p.w2t, p.b2, p.gate + (size_t)b * p.C);
}

// One workgroup per image: merge the chunk partials and run the squeeze MLP (gca_device.h, shared with the fused igemm epilogue).
__global__ __launch_bounds__(256) void gca_final_kernel(const ImagenGcaFinalParams p) {
  extern __shared__ float sm[];  // C + hidden + chunks + kGcaScratchFloats floats
  const int b = blockIdx.x;
  gca_finalize(p.part + (size_t)b * p.chunks * (p.C + 2), p.chunks, p.C, p.hidden, p.w1t, p.b1, p.w2t, p.b2, p.gate + (size_t)b * p.C, sm);
}

// The same job as ONE global round trip.  gca_final_kernel is a chain of ~12 dependent round trips (chunk statistics, partial
// rows, then the two weight matrices in batches of 8 loads per thread), each 1-2 us once the weights have fallen out of L2 since
// the previous denoiser step: 11-20 us per launch, x 67 gates per step pair of the bench.  Here 1024 threads request EVERYTHING the
// gate depends on — their slice of both weight matrices (<= 8 float4 each), the chunk statistics and their partial rows — before
// the first wait, and all arithmetic runs out of registers and LDS.  Needs power-of-two C and hidden (launcher-checked).
constexpr int kGcaFastThreads = 1024;
// -DGCA_TRACE (tools/gca_bench.py --trace, a throw-away variant library: never in the product build): thread 0 of every workgroup stamps s_memtime
// at the phase boundaries of the gate derivation into a buffer handed over by imagen_debug_gca_trace()
#ifdef GCA_TRACE
__device__ unsigned long long* g_gca_trace = nullptr;
#define GCA_STAMP(i)                                                                                                                              \
  do {                                                                                                                                            \
    if (threadIdx.x == 0 && g_gca_trace) g_gca_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime();    \
  } while (0)
#else
#define GCA_STAMP(i) ((void)0)
#endif
constexpr int kGcaFastW = 8;    // prefetched float4 per thread and weight matrix (larger matrices: a second, loop-carried pass); the fused tail kernel takes 6
constexpr int kGcaFastP = 8;    // prefetched partial-row elements per thread

// out[o] = emit(o, sum_i wt[i][o] * in[i]); wt: [n_in][n_out] fp32, n_out a power of two in [4, 4096]; w: this thread's prefetched rows
template <int W, class Emit>
__device__ __forceinline__ void gca_fast_matvec(const float4 (&w)[W], const float* wt, int n_in, int n_out, const float* in, float4* red,
                                                Emit emit) {
  const int t = threadIdx.x;
  const int nvec = n_out >> 2, rpp = kGcaFastThreads / nvec;
  const int cg = t & (nvec - 1), r = t / nvec;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int row = r + k * rpp;
    const float x = row < n_in ? in[row] : 0.f;
    a.x += w[k].x * x; a.y += w[k].y * x; a.z += w[k].z * x; a.w += w[k].w * x;
  }
  for (int row = r + W * rpp; row < n_in; row += rpp) {
    const float4 q = *reinterpret_cast<const float4*>(wt + (size_t)row * n_out + cg * 4);
    const float x = in[row];
    a.x += q.x * x; a.y += q.y * x; a.z += q.z * x; a.w += q.w * x;
  }
  // (round 6, calls F / G: one thread per output column adds up its rpp = 1024 / nvec partials in a serial walk over LDS — 3.5k of the 20k cycles of a
  // derivation at C = 128, 9.3k at C = 32.  Per-wave shuffle sums + an unrolled 16-wave final sum cut the derivation by 15-25 % launched alone
  // (profiles/r06_g_gca_phase_timeline.json) and the sampling step by nothing (7.749 / 7.732 vs 7.749 / 7.750 ms): not kept.)
  red[t] = a;
  __syncthreads();
  for (int o = t; o < n_out; o += kGcaFastThreads) {
    const float* col = reinterpret_cast<const float*>(red + (o >> 2)) + (o & 3);
    float sum = 0.f;
    for (int rr = 0; rr < rpp; ++rr) sum += col[(size_t)rr * nvec * 4];
    emit(o, sum);
  }
  __syncthreads();
}

template <int W>
__device__ __forceinline__ void gca_fast_prefetch(float4 (&w)[W], const float* wt, int n_in, int n_out) {
  const int t = threadIdx.x;
  const int nvec = n_out >> 2, rpp = kGcaFastThreads / nvec;
  const int cg = t & (nvec - 1), r = t / nvec;
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int row = r + k * rpp;
    w[k] = row < n_in ? *reinterpret_cast<const float4*>(wt + (size_t)row * n_out + cg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// The finalisation of image b by the kGcaFastThreads threads of one workgroup: gate -> gate_a[C] and (optional) gate_b[C] (either may
// be LDS or global memory; visible to the workgroup after the function's final barrier).
template <int W>
__device__ __forceinline__ void gca_final_fast_body(const float* part_base, const float* w1t, const float* b1, const float* w2t, const float* b2,
                                                    int b, int C, int hidden, int chunks, float* gate_a, float* gate_b) {
  __shared__ float4 s_red[kGcaFastThreads];
  __shared__ float s_ctx[1024], s_hid[1024], s_wgt[1024], s_b1[1024], s_b2[1024], s_sc[40];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int stride = C + 2;
  const float* part = part_base + (size_t)b * chunks * stride;
  GCA_STAMP(1);
  // ---- every global load the gate depends on, before the first wait — in the order of use (vmcnt retires in issue order)
  float2 ms = make_float2(-3.0e38f, 0.f);
  if (t < chunks) ms = *reinterpret_cast<const float2*>(part + (size_t)t * stride);
  const int c = t & (C - 1), sl = t / C, nsl = kGcaFastThreads / C;   // C <= 1024
  float pv[kGcaFastP];
#pragma unroll
  for (int j = 0; j < kGcaFastP; ++j) {
    const int i = sl + j * nsl;
    pv[j] = i < chunks ? part[(size_t)i * stride + 2 + c] : 0.f;
  }
  const float bias1 = t < hidden ? b1[t] : 0.f, bias2 = t < C ? b2[t] : 0.f;
  float4 w1[W], w2[W];
  gca_fast_prefetch<W>(w1, w1t, C, hidden);
  gca_fast_prefetch<W>(w2, w2t, hidden, C);
  GCA_STAMP(2);
  // ---- softmax merge weights of the chunks
  float m = ms.x;
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (lane == 0) s_sc[wave] = m;
  __syncthreads();
  GCA_STAMP(3);
  float M = s_sc[0];
#pragma unroll
  for (int w = 1; w < kGcaFastThreads / 64; ++w) M = fmaxf(M, s_sc[w]);
  const float wg = t < chunks ? __expf(ms.x - M) : 0.f;
  if (t < chunks) s_wgt[t] = wg;
  float ssum = ms.y * wg;
  for (int off = 32; off > 0; off >>= 1) ssum += __shfl_xor(ssum, off);
  if (lane == 0) s_sc[16 + wave] = ssum;
  __syncthreads();
  float S = 0.f;
#pragma unroll
  for (int w = 0; w < kGcaFastThreads / 64; ++w) S += s_sc[16 + w];
  const float inv_S = 1.0f / S;
  GCA_STAMP(4);
  s_b1[t] = bias1;   // (parked in LDS: the weight registers leave no room to carry them)
  s_b2[t] = bias2;
  // ---- ctx[c] = sum_i part[i][2 + c] * wgt[i] / S
  float a = 0.f;
#pragma unroll
  for (int j = 0; j < kGcaFastP; ++j) {
    const int i = sl + j * nsl;
    a += pv[j] * (i < chunks ? s_wgt[i] : 0.f);
  }
  for (int i = sl + kGcaFastP * nsl; i < chunks; i += nsl) a += part[(size_t)i * stride + 2 + c] * s_wgt[i];
  float* red_f = reinterpret_cast<float*>(s_red);
  red_f[t] = a;
  __syncthreads();
  if (t < C) {
    float v = 0.f;
    for (int q = 0; q < nsl; ++q) v += red_f[q * C + t];
    s_ctx[t] = v * inv_S;
  }
  __syncthreads();
  GCA_STAMP(5);
  // ---- squeeze MLP out of the prefetched registers
  gca_fast_matvec<W>(w1, w1t, C, hidden, s_ctx, s_red, [&](int o, float v) __attribute__((always_inline)) { s_hid[o] = silu_f(v + s_b1[o]); });
  GCA_STAMP(6);
  gca_fast_matvec<W>(w2, w2t, hidden, C, s_hid, s_red, [&](int o, float v) __attribute__((always_inline)) {
    const float g = sigmoid_f(v + s_b2[o]);
    gate_a[o] = g;
    if (gate_b) gate_b[o] = g;
  });
  GCA_STAMP(7);
}

__global__ __launch_bounds__(kGcaFastThreads) void gca_final_fast_kernel(const ImagenGcaFinalParams p) {
  GCA_STAMP(0);
  gca_final_fast_body<kGcaFastW>(p.part, p.w1t, p.b1, p.w2t, p.b2, blockIdx.x, p.C, p.hidden, p.chunks, p.gate + (size_t)blockIdx.x * p.C, nullptr);
}

// The finalisation of a WIDE block (C * hidden >= 128 Ki elements: C2's 512- and 1024-channel levels, 1-4 MB of squeeze-MLP weights) as two
// launches over many workgroups instead of one workgroup per image streaming the weights alone (39.7 us average, up to 115 us, 7.5 % of the C2
// step: profiles/r04_c2_kernel_stats.csv).  Phase 1: workgroup (slice, b) merges the chunks of image b (redundantly: a few KB) and computes 32
// hidden units over 8 channel slices; phase 2: workgroup (slice, b) computes 64 gate channels over 4 hidden slices.  Each reads 128 KB of weights.
template <int PHASE>
__global__ __launch_bounds__(256) void gca_final_split_kernel(const ImagenGcaFinalParams p) {
  __shared__ float s_in[1024], s_wgt[1024], s_red[kGcaScratchFloats];
  const int tid = threadIdx.x, b = blockIdx.y;
  if constexpr (PHASE == 1) {
    gca_merge(p.part + (size_t)b * p.chunks * (p.C + 2), p.chunks, p.C, s_in, s_wgt, s_red);   // s_in = ctx[C]
    const int o = blockIdx.x * 32 + (tid & 31), sl = tid >> 5;                                 // 8 slices of the channels
    float a = 0.f;
    if (o < p.hidden) {
      const float* w = p.w1t + o;
#pragma unroll 8
      for (int c = sl; c < p.C; c += 8) a += w[(size_t)c * p.hidden] * s_in[c];
    }
    s_red[tid] = a;
    __syncthreads();
    if (tid < 32 && o < p.hidden) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += s_red[q * 32 + tid];
      p.hid[(size_t)b * p.hidden + o] = silu_f(t + p.b1[o]);
    }
  } else {
    for (int i = tid; i < p.hidden; i += 256) s_in[i] = p.hid[(size_t)b * p.hidden + i];
    __syncthreads();
    const int c = blockIdx.x * 64 + (tid & 63), sl = tid >> 6;                                 // 4 slices of the hidden units
    float a = 0.f;
    if (c < p.C) {
      const float* w = p.w2t + c;
#pragma unroll 8
      for (int o = sl; o < p.hidden; o += 4) a += w[(size_t)o * p.C] * s_in[o];
    }
    s_red[tid] = a;
    __syncthreads();
    if (tid < 64 && c < p.C) p.gate[(size_t)b * p.C + c] = sigmoid_f(s_red[tid] + s_red[64 + tid] + s_red[128 + tid] + s_red[192 + tid] + p.b2[c]);
  }
}

// ------------------------------------------------------------------------------------------------ gca_tail
// The tail of an identity ResnetBlock in one launch (ImagenGcaTailParams): workgroup (slab, b) finalises the GlobalContext gate of image b
// in LDS, then streams its slab of rows: out = h * gate + res (+ per-row statistics, + the next Block's activated input).  One lane = 8
// channels of a row; a row = C / 8 consecutive lanes (a power of two <= 64: the reductions are shuffles inside the wave).  The first
// pass's loads are issued BEFORE the finalisation — its ~5 dependent round trips then overlap the first rows' HBM latency.
__global__ __launch_bounds__(kGcaFastThreads) void gca_tail_kernel(const ImagenGcaTailParams p) {
  GCA_STAMP(0);
  __shared__ float s_gate[1024];
  const int t = threadIdx.x, b = blockIdx.y;
  const int C = p.C, lpr = C >> 3;                         // lanes per row
  const int rpp = kGcaFastThreads / lpr;                   // rows per pass
  const int per = (p.HW + (int)gridDim.x - 1) / (int)gridDim.x;
  const int r_begin = blockIdx.x * per, r_end = min(r_begin + per, p.HW);
  const int g = t & (lpr - 1), sub = t / lpr;
  const f16* hb = reinterpret_cast<const f16*>(p.h) + (size_t)b * p.HW * p.ld_h + g * 8;
  const f16* rb = reinterpret_cast<const f16*>(p.res) + (size_t)b * p.HW * p.ld_res + g * 8;
  f16* ob = reinterpret_cast<f16*>(p.out) + (size_t)b * p.HW * p.ld_out + g * 8;
  f16* ab = p.act_out ? reinterpret_cast<f16*>(p.act_out) + (size_t)b * p.HW * p.ld_act + g * 8 : nullptr;
  int r = r_begin + sub;
  f16x8 hv = {}, rv = {};
  if (r < r_end) {
    hv = *reinterpret_cast<const f16x8*>(hb + (size_t)r * p.ld_h);
    rv = *reinterpret_cast<const f16x8*>(rb + (size_t)r * p.ld_res);
  }
  // ---- the gate of image b -> LDS
  if (p.part) {
    gca_final_fast_body<6>(p.part, p.w1t, p.b1, p.w2t, p.b2, b, C, p.hidden, p.chunks, s_gate, (p.gate && blockIdx.x == 0) ? p.gate + (size_t)b * C : nullptr);
  } else {
    if (t < C) s_gate[t] = p.gate_in ? p.gate_in[(size_t)b * C + t] : 1.0f;
    __syncthreads();
  }
  float4 pa0 = make_float4(0.f, 0.f, 0.f, 0.f), pa1 = pa0;
  if (p.act_pa) {   // (an L2 hit after the first workgroups; loaded behind the finalisation to keep its register budget)
    pa0 = *reinterpret_cast<const float4*>(p.act_pa + g * 8);
    pa1 = *reinterpret_cast<const float4*>(p.act_pa + g * 8 + 4);
  }
  const float4 g0 = *reinterpret_cast<const float4*>(s_gate + g * 8), g1 = *reinterpret_cast<const float4*>(s_gate + g * 8 + 4);
  const float gt[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float pa[8] = {pa0.x, pa0.y, pa0.z, pa0.w, pa1.x, pa1.y, pa1.z, pa1.w};
  const float inv_c = 1.0f / (float)C;
  while (r < r_end) {   // (rows are dealt to whole lane groups: the loop condition is uniform inside a group)
    const int rn = r + rpp;
    f16x8 hn = {}, rnx = {};
    if (rn < r_end) {   // next pass in flight while this one is reduced and stored
      hn = *reinterpret_cast<const f16x8*>(hb + (size_t)rn * p.ld_h);
      rnx = *reinterpret_cast<const f16x8*>(rb + (size_t)rn * p.ld_res);
    }
    f16x8 o;
    float vr[8], ssq = 0.f, sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = (f16)((float)hv[j] * gt[j] + (float)rv[j]);
      vr[j] = (float)o[j];   // statistics of the value a consumer will read back
      ssq += vr[j] * vr[j];
      sum += vr[j];
    }
    *reinterpret_cast<f16x8*>(ob + (size_t)r * p.ld_out) = o;
    const size_t row = (size_t)b * p.HW + r;
    if (p.ssq_out || ab) {
      const float tot = group_sum(ssq, lpr);
      if (p.ssq_out && g == 0) p.ssq_out[row] = tot;
      if (ab) {
        const float rs = __builtin_amdgcn_rsqf(fmaxf(tot, 1e-24f));
        f16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (f16)silu_f(vr[j] * rs * pa[j]);
        *reinterpret_cast<f16x8*>(ab + (size_t)r * p.ld_act) = a;
      }
    }
    if (p.mu_out) {
      const float mean = group_sum(sum, lpr) * inv_c;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = vr[j] - mean; q += d * d; }
      const float var = group_sum(q, lpr) * inv_c;
      if (g == 0) {
        p.mu_out[row] = mean;
        p.rs_out[row] = rsqrtf(var + p.eps);
      }
    }
    hv = hn;
    rv = rnx;
    r = rn;
  }
  GCA_STAMP(8);
}

// ------------------------------------------------------------------------------------------------ embeddings / affine
__global__ __launch_bounds__(256) void time_embed_kernel(const ImagenTimeEmbedParams p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * p.out_dim) return;
  const int b = i / p.out_dim, o = i - b * p.out_dim;
  int st = p.step_ptr ? *p.step_ptr : 0;
  if (p.steps > 0) st = min(max(st, 0), p.steps - 1);
  const float x = p.step_ptr ? p.coef[(size_t)st * 8 + 6] : p.times[b];
  const int nf = 2 * p.half_dim + 1;
  const float* w = p.w + (size_t)o * nf;
  float a = p.bias[o] + w[0] * x;
  for (int k = 0; k < p.half_dim; ++k) {
    const float f = x * p.freqs[k] * 6.283185307179586f;
    a += w[1 + k] * sinf(f) + w[1 + p.half_dim + k] * cosf(f);
  }
  reinterpret_cast<f16*>(p.hid)[(size_t)b * p.ld_hid + o] = (f16)silu_f(a);
}

__global__ __launch_bounds__(256) void scale_shift_kernel(const ImagenScaleShiftParams p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * p.total_c) return;
  const int b = i / p.total_c, c = i - b * p.total_c;
  float sc, sh;
  if (p.ss_f32) {   // LINEAR_F32's rows
    const float* ss = reinterpret_cast<const float*>(p.ss) + (size_t)b * p.ld_ss;
    sc = ss[p.idx_scale[c]];
    sh = ss[p.idx_shift[c]];
  } else {
    const f16* ss = reinterpret_cast<const f16*>(p.ss) + (size_t)b * p.ld_ss;
    sc = (float)ss[p.idx_scale[c]];
    sh = (float)ss[p.idx_shift[c]];
  }
  p.pa[i] = p.gamma_s[c] * (sc + 1.0f);
  p.ps[i] = sh;
}

// ---- LINEAR_F32: nn.Linear on per-sample vectors, fp32 end to end (include/imagen_hip.h: the timestep-conditioning chain).  No matrix pipe:
// a thread owns one output channel and 16 rows, the activated rows sit in LDS (every lane reads the same word: a broadcast), the
// transposed weight is read once per row block, coalesced over the output channels.  R rows per step (a 28-workgroup, weight-latency-bound
// launch like the GEMM it replaces), or NR * R rows once per request in table mode (16000 x 128 x 7K: ~0.5 ms of fp32 FMAs).
constexpr int kLfRows = 16, kLfKc = 128;

__global__ __launch_bounds__(256) void linear_f32_kernel(const ImagenLinearF32Params p) {
  __shared__ float xs[kLfRows][kLfKc];
  const int o = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * kLfRows;
  const int nr = min(kLfRows, p.rows - r0);
  float acc[kLfRows];
#pragma unroll
  for (int j = 0; j < kLfRows; ++j) acc[j] = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += kLfKc) {
    const int kc = min(kLfKc, p.K - k0);
    __syncthreads();   // (the previous chunk has been consumed)
    for (int i = threadIdx.x; i < kLfRows * kLfKc; i += 256) {
      const int j = i / kLfKc, k = i - j * kLfKc;
      float v = 0.f;
      if (j < nr && k < kc) {
        const size_t at = (size_t)(r0 + j) * p.ld_x + k0 + k;
        v = p.x_f32 ? reinterpret_cast<const float*>(p.x)[at] : (float)reinterpret_cast<const f16*>(p.x)[at];
        if (p.act_in == IMAGEN_ACT_SILU) v = silu_f(v);
      }
      xs[j][k] = v;
    }
    __syncthreads();
    if (o < p.Cout) {
      const float* w = p.wt + (size_t)k0 * p.Cout + o;
      for (int k = 0; k < kc; ++k) {
        const float wv = w[(size_t)k * p.Cout];
#pragma unroll
        for (int j = 0; j < kLfRows; ++j) acc[j] = fmaf(xs[j][k], wv, acc[j]);
      }
    }
  }
  if (o >= p.Cout) return;
  const float b = p.bias ? p.bias[o] : 0.f;
  for (int j = 0; j < nr; ++j) {
    float v = acc[j] + b;
    if (p.res) v += (float)reinterpret_cast<const f16*>(p.res)[(size_t)(r0 + j) * p.ld_res + o];
    p.y[(size_t)(r0 + j) * p.ld_y + o] = v;
  }
}

__global__ __launch_bounds__(256) void pack_image_kernel(const ImagenPackImageParams p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int HW = p.H * p.W;
  if (i >= p.B * p.Brep * HW) return;
  const int bo = i / HW, px = i - bo * HW;
  const int bs = bo % p.B;
  f16* out = reinterpret_cast<f16*>(p.out) + (size_t)i * p.Cpad;
  for (int c = 0; c < p.Cpad; ++c) {
    float v = 0.f;
    if (c < p.Ca) v = p.a[((size_t)bs * p.Ca + c) * HW + px];
    else if (c - p.Ca < p.Cb) v = p.b[((size_t)bs * p.Cb + (c - p.Ca)) * HW + px];
    out[c] = (f16)v;
  }
}

__global__ __launch_bounds__(256) void rows_copy_kernel(const ImagenRowsCopyParams p) {
  const int groups = p.C >> 3;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * p.rows * groups) return;
  const int g = i % groups;
  const int r = (i / groups) % p.rows;
  const int b = i / (groups * p.rows);
  const f16* s = reinterpret_cast<const f16*>(p.src) + (size_t)b * p.src_bs + (size_t)r * p.src_rs + g * 8;
  f16* d = reinterpret_cast<f16*>(p.dst) + (size_t)b * p.dst_bs + (size_t)r * p.dst_rs + g * 8;
  *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
}

__global__ __launch_bounds__(256) void select_rows_kernel(const ImagenSelectRowsParams p) {
  const int groups = p.C >> 3;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.R * p.L * groups) return;
  const int g = i % groups;
  const int l = (i / groups) % p.L;
  const int r = i / (groups * p.L);
  const int sb = p.src[r];
  const bool take = p.keep[r] && (p.mask == nullptr || p.mask[(size_t)sb * p.L + l]);
  const f16* s = take ? reinterpret_cast<const f16*>(p.a) + ((size_t)sb * p.L + l) * p.C + g * 8
                      : reinterpret_cast<const f16*>(p.nul) + (size_t)l * p.C + g * 8;
  *reinterpret_cast<uint4*>(reinterpret_cast<f16*>(p.dst) + ((size_t)r * p.L + l) * p.C + g * 8) = *reinterpret_cast<const uint4*>(s);
}

__global__ __launch_bounds__(256) void mean_rows_kernel(const ImagenMeanRowsParams p) {
  const int groups = p.C >> 3;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * groups) return;
  const int b = i / groups, g = i - b * groups;
  const f16* x = reinterpret_cast<const f16*>(p.x) + (size_t)b * p.bs_x + g * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < p.rows; ++r) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(x + (size_t)r * p.ld_x);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
  }
  f16x8 o;
  const float inv = 1.0f / (float)p.rows;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (f16)(acc[j] * inv);
  *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.out) + (size_t)b * p.ld_out + g * 8) = o;
}

__global__ __launch_bounds__(256) void memset32_kernel(uint32_t* dst, uint32_t value, int count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) dst[i] = value;
}

int host_lpr(int groups) {
  int l = 1;
  while (l < groups && l < 64) l <<= 1;
  return l;
}

// ------------------------------------------------------------------------------------------------ act_prep
// one lane = 8 channels (16 B) of one pixel, U items per thread in flight (the pass is a pure latency chain otherwise); the per-pixel
// statistics and the per-(batch, channel) affine are L2-resident
__global__ __launch_bounds__(256) void act_prep_kernel(const ImagenActPrepParams p) {
  constexpr int U = 4;
  const int gpp = (p.C1 + p.C2) >> 3;                    // 8-channel groups per pixel
  const long total = (long)p.rows * gpp;
  const f16* x1 = reinterpret_cast<const f16*>(p.x1);
  const f16* x2 = reinterpret_cast<const f16*>(p.x2);
  f16* y = reinterpret_cast<f16*>(p.y);
  const long stride = (long)gridDim.x * 256;
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += stride * U) {
    f16x8 in[U];
    float rs[U], mu[U], qb[U];
    float4 a0[U], a1[U], s0[U], s1[U];
    f16* dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      const bool ok = i < total;
      const long ii = ok ? i : 0;
      const int r = (int)(ii / gpp), g = (int)(ii - (long)r * gpp);
      const int b = r / p.rows_per_batch, rr = r - b * p.rows_per_batch;
      const int c0 = g * 8;
      const f16* src = c0 < p.C1 ? x1 + (size_t)b * p.bs1 + (size_t)rr * p.ld1 + c0 : x2 + (size_t)b * p.bs2 + (size_t)rr * p.ld2 + (c0 - p.C1);
      in[u] = *reinterpret_cast<const f16x8*>(src);
      rs[u] = p.rs ? p.rs[r] : (p.ssq_a ? p.ssq_a[r] : 1.0f);
      qb[u] = (!p.rs && p.ssq_b) ? p.ssq_b[r] : 0.0f;
      mu[u] = p.mu ? p.mu[r] : 0.0f;
      const float* pa = p.pa ? p.pa + (size_t)b * p.pstride + c0 : nullptr;
      const float* ps = p.ps ? p.ps + (size_t)b * p.pstride + c0 : nullptr;
      a0[u] = pa ? *reinterpret_cast<const float4*>(pa) : make_float4(1.f, 1.f, 1.f, 1.f);
      a1[u] = pa ? *reinterpret_cast<const float4*>(pa + 4) : make_float4(1.f, 1.f, 1.f, 1.f);
      s0[u] = ps ? *reinterpret_cast<const float4*>(ps) : make_float4(0.f, 0.f, 0.f, 0.f);
      s1[u] = ps ? *reinterpret_cast<const float4*>(ps + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      dst[u] = ok ? y + (size_t)b * p.bsy + (size_t)rr * p.ldy + c0 : nullptr;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!dst[u]) continue;
      float sc = rs[u];
      if (!p.rs && p.ssq_a) sc = __builtin_amdgcn_rsqf(fmaxf(rs[u] + p.ssq_wb * qb[u], 1e-24f));
      const float a[8] = {a0[u].x, a0[u].y, a0[u].z, a0[u].w, a1[u].x, a1[u].y, a1[u].z, a1[u].w};
      const float sh[8] = {s0[u].x, s0[u].y, s0[u].z, s0[u].w, s1[u].x, s1[u].y, s1[u].z, s1[u].w};
      f16x8 out;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = ((float)in[u][j] - mu[u]) * sc * a[j] + sh[j];
        if (p.act_in == IMAGEN_ACT_SILU) v = silu_f(v);
        else if (p.act_in == IMAGEN_ACT_GELU) v = gelu_f(v);
        out[j] = (f16)v;
      }
      *reinterpret_cast<f16x8*>(dst[u]) = out;
    }
  }
}

// self_stat form: L lanes per pixel (L = 16 / 32 / 64 >= 8-channel groups of the pixel), one 16-byte group per lane; the sum of squares of
// x1's channels is reduced over the pixel's lanes (butterfly inside the wave), so the input is read once and nothing precedes the launch
template <int L>
__global__ __launch_bounds__(256) void act_prep_stat_kernel(const ImagenActPrepParams p) {
  const int gpp = (p.C1 + p.C2) >> 3;
  const int g = threadIdx.x % L;
  const int r = blockIdx.x * (256 / L) + threadIdx.x / L;
  const bool live = r < p.rows && g < gpp;
  const int rc = min(r, p.rows - 1), gc = min(g, gpp - 1);
  const int b = rc / p.rows_per_batch, rr = rc - b * p.rows_per_batch;
  const int c0 = gc * 8;
  const f16* x1 = reinterpret_cast<const f16*>(p.x1);
  const f16* x2 = reinterpret_cast<const f16*>(p.x2);
  const f16* src = c0 < p.C1 ? x1 + (size_t)b * p.bs1 + (size_t)rr * p.ld1 + c0 : x2 + (size_t)b * p.bs2 + (size_t)rr * p.ld2 + (c0 - p.C1);
  const f16x8 in = *reinterpret_cast<const f16x8*>(src);
  const float qb = p.ssq_b ? p.ssq_b[rc] : 0.0f;
  const float* pa = p.pa ? p.pa + (size_t)b * p.pstride + c0 : nullptr;
  const float* ps = p.ps ? p.ps + (size_t)b * p.pstride + c0 : nullptr;
  const float4 a0 = pa ? *reinterpret_cast<const float4*>(pa) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 a1 = pa ? *reinterpret_cast<const float4*>(pa + 4) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 s0 = ps ? *reinterpret_cast<const float4*>(ps) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 s1 = ps ? *reinterpret_cast<const float4*>(ps + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float own = 0.0f;
  if (live && c0 < p.C1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) own += (float)in[j] * (float)in[j];
  }
#pragma unroll
  for (int o = L / 2; o >= 1; o >>= 1) own += __shfl_xor(own, o);
  if (!live) return;
  const float sc = __builtin_amdgcn_rsqf(fmaxf(own + p.ssq_wb * qb, 1e-24f));
  const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  const float sh[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  f16x8 out;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = (float)in[j] * sc * a[j] + sh[j];
    if (p.act_in == IMAGEN_ACT_SILU) v = silu_f(v);
    else if (p.act_in == IMAGEN_ACT_GELU) v = gelu_f(v);
    out[j] = (f16)v;
  }
  *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.y) + (size_t)b * p.bsy + (size_t)rr * p.ldy + c0) = out;
}

// ------------------------------------------------------------------------------------------------ step_slice
__global__ __launch_bounds__(256) void step_slice_kernel(const ImagenStepSliceParams p) {
  int st = *p.step_ptr;
  if (p.steps > 0) st = min(max(st, 0), p.steps - 1);
  const size_t step = (size_t)st;
  const int w0 = p.words0, w1 = w0 + p.words1, w2 = w1 + p.words2, tot = w2 + p.words3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < tot; i += gridDim.x * 256) {
    const uint4* s;
    uint4* d;
    int j, w;
    if (i < w0) { s = reinterpret_cast<const uint4*>(p.src0); d = reinterpret_cast<uint4*>(p.dst0); j = i; w = p.words0; }
    else if (i < w1) { s = reinterpret_cast<const uint4*>(p.src1); d = reinterpret_cast<uint4*>(p.dst1); j = i - w0; w = p.words1; }
    else if (i < w2) { s = reinterpret_cast<const uint4*>(p.src2); d = reinterpret_cast<uint4*>(p.dst2); j = i - w1; w = p.words2; }
    else { s = reinterpret_cast<const uint4*>(p.src3); d = reinterpret_cast<uint4*>(p.dst3); j = i - w2; w = p.words3; }
    d[j] = s[step * (size_t)w + j];
  }
}

}  // namespace

#ifdef GCA_TRACE
extern "C" int imagen_debug_gca_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gca_trace), &buf, sizeof(buf)); }
#endif

int launch_step_slice(const ImagenStepSliceParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->step_ptr && p->words0 > 0 && p->src0 && p->dst0, "step_slice: null step_ptr / first segment");
  IMAGEN_CHECK((!p->words1 || (p->src1 && p->dst1)) && (!p->words2 || (p->src2 && p->dst2)) && (!p->words3 || (p->src3 && p->dst3)),
               "step_slice: a segment with words but no pointers");
  const long tot = (long)p->words0 + p->words1 + p->words2 + p->words3;
  hipLaunchKernelGGL(step_slice_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 2048L)), dim3(256), 0, s, *p);
  return imagen_hip_status("step_slice");
}

int launch_rowstat(const ImagenRowstatParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->C1 % 8 == 0 && p->C2 % 8 == 0 && p->ld1 % 8 == 0, "rowstat: channels/stride must be multiples of 8");
  IMAGEN_CHECK(p->rows > 0 && p->rows_per_batch > 0, "rowstat: empty");
  const int lpr = host_lpr((p->C1 + (p->x2 ? p->C2 : 0)) / 8);
  const int rpb = 256 / lpr;
  hipLaunchKernelGGL(rowstat_kernel, dim3((p->rows + rpb - 1) / rpb), dim3(256), 0, s, *p, lpr);
  return imagen_hip_status("rowstat");
}

int launch_gate_residual(const ImagenGateResidualParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->C % 8 == 0, "gate_residual: C %% 8");
  const int lpr = host_lpr(p->C / 8);
  const int rpb = 256 / lpr;
  hipLaunchKernelGGL(gate_residual_kernel, dim3((p->rows + rpb - 1) / rpb), dim3(256), 0, s, *p, lpr);
  return imagen_hip_status("gate_residual");
}

int launch_ln_residual(const ImagenLnResidualParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->C % 8 == 0, "ln_residual: C %% 8");
  const int lpr = host_lpr(p->C / 8);
  const int rpb = 256 / lpr;
  hipLaunchKernelGGL(ln_residual_kernel, dim3((p->rows + rpb - 1) / rpb), dim3(256), 0, s, *p, lpr);
  return imagen_hip_status("ln_residual");
}

int launch_qnorm(const ImagenQnormParams* p, hipStream_t s) {
  const int items = p->rows * p->heads;
  hipLaunchKernelGGL(qnorm_kernel, dim3((items + 31) / 32), dim3(256), 0, s, *p);
  return imagen_hip_status("qnorm");
}

int launch_kv_prep(const ImagenKvPrepParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->rows > 0, "kv_prep: empty");
  hipLaunchKernelGGL(kv_prep_kernel, dim3((p->rows + 31) / 32, p->B * p->heads), dim3(256), 0, s, *p);
  return imagen_hip_status("kv_prep");
}

int launch_kv_prep_multi(const ImagenKvPrepMultiParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->jobs && p->n > 0 && p->max_rows > 0 && p->max_bh > 0, "kv_prep_multi: empty");
  hipLaunchKernelGGL(kv_prep_multi_kernel, dim3((p->max_rows + 31) / 32, p->max_bh, p->n), dim3(256), 0, s, *p);
  return imagen_hip_status("kv_prep_multi");
}

int launch_gca_partial(const ImagenGcaPartialParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->C % 8 == 0 && p->C / 8 <= 256, "gca: unsupported C %d", p->C);
  const int chunk_px = (p->HW + p->chunks - 1) / p->chunks;
  const int groups = p->C / 8;
  if (p->w1t) {
    IMAGEN_CHECK((groups & (groups - 1)) == 0 && groups <= 64, "gca: in-kernel finalisation needs a power-of-two C/8 (C = %d)", p->C);
    IMAGEN_CHECK(p->b1 && p->w2t && p->b2 && p->gate && p->hidden > 0, "gca: incomplete finalisation parameters");
    IMAGEN_CHECK(p->chunks == 1, "gca: in-kernel finalisation needs one chunk per image (got %d)", p->chunks);
    IMAGEN_CHECK(p->C + p->hidden + p->chunks + kGcaScratchFloats <= 2048, "gca: finalisation scratch too large (C %d hidden %d chunks %d)", p->C,
                 p->hidden, p->chunks);
  }
  if ((groups & (groups - 1)) == 0 && groups <= 64) {
    hipLaunchKernelGGL(gca_partial_online_kernel, dim3(p->chunks, p->B), dim3(256), 0, s, *p, chunk_px);
    return imagen_hip_status("gca_partial");
  }
  IMAGEN_CHECK(chunk_px <= kGcaMaxChunk, "gca: chunk of %d pixels too large", chunk_px);
  hipLaunchKernelGGL(gca_partial_kernel, dim3(p->chunks, p->B), dim3(256), 0, s, *p, chunk_px);
  return imagen_hip_status("gca_partial");
}

int launch_gca_final(const ImagenGcaFinalParams* p, hipStream_t s) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  IMAGEN_CHECK(p->phase >= 0 && p->phase <= 2, "gca_final: phase %d", p->phase);
  if (p->phase != 0) {
    IMAGEN_CHECK(p->hid && p->C <= 1024 && p->hidden <= 1024 && p->chunks <= 1024 && p->C % 2 == 0, "gca_final: the two-phase finalisation needs hid, C / hidden / chunks <= 1024 (got %d / %d / %d)", p->C, p->hidden, p->chunks);
    if (p->phase == 1) hipLaunchKernelGGL(gca_final_split_kernel<1>, dim3((p->hidden + 31) / 32, p->B), dim3(256), 0, s, *p);
    else hipLaunchKernelGGL(gca_final_split_kernel<2>, dim3((p->C + 63) / 64, p->B), dim3(256), 0, s, *p);
    return imagen_hip_status("gca_final");
  }
  if (pow2(p->C) && pow2(p->hidden) && p->C >= 4 && p->C <= 1024 && p->hidden >= 4 && p->hidden <= 1024 && p->chunks <= 1024) {
    hipLaunchKernelGGL(gca_final_fast_kernel, dim3(p->B), dim3(kGcaFastThreads), 0, s, *p);
    return imagen_hip_status("gca_final");
  }
  const size_t sm = (size_t)(p->C + p->hidden + p->chunks + kGcaScratchFloats) * sizeof(float);
  hipLaunchKernelGGL(gca_final_kernel, dim3(p->B), dim3(256), sm, s, *p);
  return imagen_hip_status("gca_final");
}

int launch_gca_tail(const ImagenGcaTailParams* p, hipStream_t s) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  IMAGEN_CHECK(p->h && p->res && p->out && p->B > 0 && p->HW > 0, "gca_tail: null tensors / empty problem");
  IMAGEN_CHECK(pow2(p->C) && p->C >= 8 && p->C <= 512, "gca_tail: C %d must be a power of two in [8, 512]", p->C);
  IMAGEN_CHECK(!p->part || (p->w1t && p->b1 && p->w2t && p->b2 && pow2(p->hidden) && p->hidden >= 4 && p->hidden <= 1024 && p->chunks >= 1 && p->chunks <= 1024),
               "gca_tail: GlobalContext finalisation needs the squeeze MLP, a power-of-two hidden width (%d) and 1..1024 chunks (%d)", p->hidden, p->chunks);
  IMAGEN_CHECK(p->ld_h % 8 == 0 && p->ld_res % 8 == 0 && p->ld_out % 8 == 0 && (!p->act_out || (p->act_pa && p->ld_act % 8 == 0)),
               "gca_tail: row strides must be multiples of 8 (act_out needs act_pa)");
  IMAGEN_CHECK(!p->mu_out == !p->rs_out, "gca_tail: mu_out and rs_out come together");
  IMAGEN_CHECK(p->slabs >= 1 && p->slabs <= 65535, "gca_tail: slabs %d", p->slabs);
  hipLaunchKernelGGL(gca_tail_kernel, dim3(p->slabs, p->B), dim3(kGcaFastThreads), 0, s, *p);
  return imagen_hip_status("gca_tail");
}

int launch_time_embed(const ImagenTimeEmbedParams* p, hipStream_t s) {
  const int n = p->B * p->out_dim;
  hipLaunchKernelGGL(time_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *p);
  return imagen_hip_status("time_embed");
}

int launch_scale_shift(const ImagenScaleShiftParams* p, hipStream_t s) {
  const int n = p->B * p->total_c;
  hipLaunchKernelGGL(scale_shift_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *p);
  return imagen_hip_status("scale_shift");
}

int launch_linear_f32(const ImagenLinearF32Params* p, hipStream_t s) {
  IMAGEN_CHECK(p->x && p->wt && p->y, "linear_f32: null pointer");
  IMAGEN_CHECK(p->rows > 0 && p->K > 0 && p->Cout > 0 && p->ld_x >= p->K && p->ld_y >= p->Cout && (!p->res || p->ld_res >= p->Cout), "linear_f32: bad shape");
  IMAGEN_CHECK(p->act_in == IMAGEN_ACT_NONE || p->act_in == IMAGEN_ACT_SILU, "linear_f32: input activation none | SiLU");
  IMAGEN_CHECK((p->rows + kLfRows - 1) / kLfRows <= 65535, "linear_f32: %d rows", p->rows);
  hipLaunchKernelGGL(linear_f32_kernel, dim3((p->Cout + 255) / 256, (p->rows + kLfRows - 1) / kLfRows), dim3(256), 0, s, *p);
  return imagen_hip_status("linear_f32");
}

int launch_pack_image(const ImagenPackImageParams* p, hipStream_t s) {
  const int n = p->B * p->Brep * p->H * p->W;
  hipLaunchKernelGGL(pack_image_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *p);
  return imagen_hip_status("pack_image");
}

int launch_rows_copy(const ImagenRowsCopyParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->C % 8 == 0, "rows_copy: C %% 8");
  const int n = p->B * p->rows * (p->C / 8);
  hipLaunchKernelGGL(rows_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *p);
  return imagen_hip_status("rows_copy");
}

int launch_select_rows(const ImagenSelectRowsParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->C % 8 == 0, "select_rows: C %% 8");
  const int n = p->R * p->L * (p->C / 8);
  hipLaunchKernelGGL(select_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *p);
  return imagen_hip_status("select_rows");
}

int launch_mean_rows(const ImagenMeanRowsParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->C % 8 == 0 && p->rows > 0, "mean_rows: bad shape");
  const int n = p->B * (p->C / 8);
  hipLaunchKernelGGL(mean_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *p);
  return imagen_hip_status("mean_rows");
}

int launch_memset32(const ImagenMemset32Params* p, hipStream_t s) {
  hipLaunchKernelGGL(memset32_kernel, dim3((p->count + 255) / 256), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p->dst), p->value,
                     p->count);
  return imagen_hip_status("memset32");
}

int launch_act_prep(const ImagenActPrepParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->x1 && p->y && p->rows > 0 && p->rows_per_batch > 0, "act_prep: null x1/y or empty problem");
  IMAGEN_CHECK(p->C1 % 8 == 0 && p->C2 % 8 == 0 && p->ld1 % 8 == 0 && p->ldy % 8 == 0 && (!p->x2 || p->ld2 % 8 == 0) && (p->x2 || p->C2 == 0),
               "act_prep: channel counts / strides must be multiples of 8 (C1=%d C2=%d ld1=%d ld2=%d ldy=%d)", p->C1, p->C2, p->ld1, p->ld2, p->ldy);
  const long total = (long)p->rows * ((p->C1 + p->C2) >> 3);
  const int blocks = (int)std::min<long>((total + 1023) / 1024, 256L * 8);   // 4 items per thread and pass
  if (p->self_stat) {
    IMAGEN_CHECK(!p->rs && !p->mu && !p->ssq_a, "act_prep: self_stat excludes rs / mu / ssq_a");
    const int gpp = (p->C1 + p->C2) / 8;
    IMAGEN_CHECK(gpp <= 64, "act_prep: self_stat handles at most 512 channels per pixel (got %d)", 8 * gpp);
    if (gpp <= 16) hipLaunchKernelGGL(act_prep_stat_kernel<16>, dim3((p->rows + 15) / 16), dim3(256), 0, s, *p);
    else if (gpp <= 32) hipLaunchKernelGGL(act_prep_stat_kernel<32>, dim3((p->rows + 7) / 8), dim3(256), 0, s, *p);
    else hipLaunchKernelGGL(act_prep_stat_kernel<64>, dim3((p->rows + 3) / 4), dim3(256), 0, s, *p);
    return imagen_hip_status("act_prep (self_stat)");
  }
  hipLaunchKernelGGL(act_prep_kernel, dim3(blocks), dim3(256), 0, s, *p);
  return imagen_hip_status("act_prep");
}
