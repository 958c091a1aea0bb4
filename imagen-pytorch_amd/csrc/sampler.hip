// sampler.hip — the per-timestep DDPM epilogue of Imagen.p_sample (ip.py:2042-2165) as graph-capturable
// kernels: classifier-free-guidance combine + x0 prediction, exact per-sample 0.95-quantile of |x0|
// (torch.quantile semantics: fp32 rank, linear interpolation), dynamic thresholding, posterior mean /
// variance and ancestral noise (injected tensor for parity runs, counter-based Philox4x32-10 otherwise).
// Every kernel reads the current step from a device counter so one captured graph replays for all T steps.
#include "common.h"

namespace {

// coef row layout (host-built, fp32): [alpha, sigma, alpha_next, sigma_next, c, nonzero, log_snr, 0]
constexpr int kCoefStride = 8;

__global__ __launch_bounds__(256) void cfg_x0_kernel(const ImagenCfgX0Params p) {
  const size_t n = (size_t)p.B * p.n_per_sample;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* cf = p.coef + (size_t)(*p.step_ptr) * kCoefStride;
  const float alpha = cf[0], sigma = cf[1];
  float eps;
  if (p.cfg) {
    const float cond = p.pred[i], nul = p.pred[n + i];
    eps = nul + (cond - nul) * p.cond_scale;  // ip.py:1522
  } else {
    eps = p.pred[i];
  }
  float x0;
  if (p.objective == 0) x0 = (p.x[i] - sigma * eps) / fmaxf(alpha, 1e-8f);  // noise, ip.py:314-318
  else if (p.objective == 1) x0 = eps;                                      // x_start, ip.py:2087-2088
  else x0 = alpha * p.x[i] - sigma * eps;                                   // v, ip.py:308-312
  p.x0[i] = x0;
  p.absx0[i] = fabsf(x0);
}

// ---- exact quantile: 4 x 8-bit MSB-first radix select on the (non-negative) float bit patterns ------------
// scratch per sample (uint32): hist[4][256] | cnt_le | min_gt | unused...
constexpr int kScratch = IMAGEN_QUANTILE_SCRATCH_WORDS;

struct Narrow { uint32_t prefix; uint32_t k; };

// Re-derive (prefix, remaining rank) from the histograms of the passes already done — by the whole 256-thread workgroup:
// thread t owns bin t, a block-wide exclusive scan finds the bin that holds rank k (a single thread walking 4 x 256 counters
// in global memory was most of the old quantile's 150 us).  s_scan: 8 words of LDS.  All threads get the result.
__device__ Narrow narrow_block(const uint32_t* hist, int passes_done, uint32_t k, uint32_t* s_scan) {
  Narrow nr{0u, k};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int ps = 0; ps < passes_done; ++ps) {
    const uint32_t c = hist[ps * 256 + tid];
    uint32_t v = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(v, off);
      if (lane >= off) v += t;
    }
    if (lane == 63) s_scan[wave] = v;
    if (tid == 0) { s_scan[4] = 255u; s_scan[5] = 0xFFFFFFFFu; }   // fallback: last bin (rank beyond the counted keys: cannot happen)
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += s_scan[w];
    const uint32_t incl = v + base, excl = incl - c;
    if (excl <= nr.k && nr.k < incl) { s_scan[4] = (uint32_t)tid; s_scan[5] = excl; }
    __syncthreads();
    const uint32_t bin = s_scan[4];
    uint32_t cum = s_scan[5];
    if (cum == 0xFFFFFFFFu) cum = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3] - hist[ps * 256 + 255];
    __syncthreads();   // s_scan is rewritten by the next pass
    nr.prefix |= bin << (24 - 8 * ps);
    nr.k -= cum;
  }
  return nr;
}

__device__ __forceinline__ uint32_t rank_below(int n, float q) {
  const float rank = q * (float)(n - 1);  // fp32 rank, as torch.quantile computes it
  return (uint32_t)floorf(rank);
}

__global__ __launch_bounds__(256) void quantile_hist_kernel(const ImagenQuantileParams p, int pass, int blocks_per_sample) {
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_scan[8];
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  uint32_t* scratch = p.scratch + (size_t)b * kScratch;
  s_hist[threadIdx.x] = 0;
  const Narrow nr = narrow_block(scratch, pass, rank_below(p.n, p.q), s_scan);
  __syncthreads();
  const uint32_t prefix = nr.prefix;
  const uint32_t himask = pass == 0 ? 0u : (0xFFFFFFFFu << (32 - 8 * pass));
  const int shift = 24 - 8 * pass;
  const uint32_t* keys = reinterpret_cast<const uint32_t*>(p.absx0) + (size_t)b * p.n;
  for (int i = blk * 256 + threadIdx.x; i < p.n; i += blocks_per_sample * 256) {
    const uint32_t key = keys[i];
    if ((key & himask) == (prefix & himask)) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  const uint32_t c = s_hist[threadIdx.x];
  if (c) atomicAdd(&scratch[pass * 256 + threadIdx.x], c);
}

// After 4 passes the rank-k key is fully known; count keys <= it and find the smallest key above it.
__global__ __launch_bounds__(256) void quantile_tail_kernel(const ImagenQuantileParams p, int blocks_per_sample) {
  __shared__ uint32_t s_scan[8];
  __shared__ uint32_t s_cnt, s_min;
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  uint32_t* scratch = p.scratch + (size_t)b * kScratch;
  if (threadIdx.x == 0) {
    s_cnt = 0;
    s_min = 0xFFFFFFFFu;
  }
  const Narrow nr = narrow_block(scratch, 4, rank_below(p.n, p.q), s_scan);
  __syncthreads();
  const uint32_t vlo = nr.prefix;
  const uint32_t* keys = reinterpret_cast<const uint32_t*>(p.absx0) + (size_t)b * p.n;
  uint32_t cnt = 0, mn = 0xFFFFFFFFu;
  for (int i = blk * 256 + threadIdx.x; i < p.n; i += blocks_per_sample * 256) {
    const uint32_t key = keys[i];
    if (key <= vlo) ++cnt;
    else mn = min(mn, key);
  }
  atomicAdd(&s_cnt, cnt);
  atomicMin(&s_min, mn);
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&scratch[1024], s_cnt);
    atomicMin(&scratch[1025], s_min);
  }
}

// One workgroup per sample.  Leaves the sample's scratch zeroed (min slot = ~0) for the next call, so a replayed plan needs
// no separate clearing launch; the host initialises it once when the op is built (ops.quantile).
__global__ __launch_bounds__(256) void quantile_final_kernel(const ImagenQuantileParams p) {
  __shared__ uint32_t s_scan[8];
  const int b = blockIdx.x;
  uint32_t* scratch = p.scratch + (size_t)b * kScratch;
  const float rank = p.q * (float)(p.n - 1);
  const uint32_t k = (uint32_t)floorf(rank);
  const float w = rank - floorf(rank);
  const Narrow nr = narrow_block(scratch, 4, k, s_scan);
  const uint32_t cnt_le = scratch[1024], min_gt = scratch[1025];
  __syncthreads();   // everyone has read the scratch
  for (int i = threadIdx.x; i < 1026; i += 256) scratch[i] = (i == 1025) ? 0xFFFFFFFFu : 0u;
  if (threadIdx.x == 0) {
    const uint32_t hi_key = (cnt_le >= k + 2u || min_gt == 0xFFFFFFFFu) ? nr.prefix : min_gt;
    const float lo = __uint_as_float(nr.prefix), hi = __uint_as_float(hi_key);
    // torch lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
    const float d = hi - lo;
    p.out[b] = (w < 0.5f) ? (lo + w * d) : (hi - d * (1.0f - w));
  }
}

// ---- Philox4x32-10 + Box-Muller ------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
  const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
  const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ void philox_normal4(uint32_t ctr0, uint32_t ctr1, uint32_t ctr2, uint32_t ctr3, uint32_t seed_lo, uint32_t seed_hi,
                                               float out[4]) {
  uint32_t c0 = ctr0, c1 = ctr1, c2 = ctr2, c3 = ctr3, k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float inv = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)c0 + 0.5f) * inv, u1 = ((float)c1 + 0.5f) * inv;
  const float u2 = ((float)c2 + 0.5f) * inv, u3 = ((float)c3 + 0.5f) * inv;
  const float r0 = sqrtf(-2.0f * __logf(fmaxf(u0, 1e-12f))), r1 = sqrtf(-2.0f * __logf(fmaxf(u2, 1e-12f)));
  float s0, c0f, s1, c1f;
  __sincosf(6.283185307179586f * u1, &s0, &c0f);
  __sincosf(6.283185307179586f * u3, &s1, &c1f);
  out[0] = r0 * c0f; out[1] = r0 * s0; out[2] = r1 * c1f; out[3] = r1 * s1;
}

__global__ __launch_bounds__(256) void ddpm_update_kernel(const ImagenDdpmUpdateParams p) {
  // n_per_sample % 4 == 0 (checked at launch): a 4-element group never straddles two samples
  const size_t n = (size_t)p.B * p.n_per_sample;
  const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const int step = *p.step_ptr;
  const float* cf = p.coef + (size_t)step * kCoefStride;
  const float alpha = cf[0], alpha_next = cf[2], sigma_next = cf[3], c = cf[4], nonzero = cf[5];
  const float var = sigma_next * sigma_next * c;                // ip.py:268
  const float stdev = sqrtf(fmaxf(var, 1e-20f));                 // = exp(0.5*log(clamp(var,1e-20))), ip.py:269, 2164
  float z[4];
  if (p.noise) {
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = (i4 + e < n) ? p.noise[i4 + e] : 0.f;
  } else {
    const int bs = (int)(i4 / p.n_per_sample);
    const uint32_t within = (uint32_t)((i4 - (size_t)bs * p.n_per_sample) >> 2);
    uint32_t k0, k1, sidx;
    if (p.row_keys) {   // merged requests: every row has the key and the sample index of its own request
      k0 = p.row_keys[4 * bs], k1 = p.row_keys[4 * bs + 1], sidx = p.row_keys[4 * bs + 2];
    } else {
      k0 = p.seed_ptr ? p.seed_ptr[0] : p.seed_lo, k1 = p.seed_ptr ? p.seed_ptr[1] : p.seed_hi, sidx = (uint32_t)(p.sample_offset + bs);
    }
    philox_normal4(within, (uint32_t)step, p.stream_id, sidx, k0, k1, z);
  }
  const bool last = step + 1 >= p.total_steps;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const size_t i = i4 + e;
    if (i >= n) break;
    const int b = (int)(i / p.n_per_sample);
    float x0 = p.x0[i];
    if (p.dynamic_threshold) {
      const float s = fmaxf(p.quant[b], 1.0f);                  // ip.py:2103
      x0 = fminf(fmaxf(x0, -s), s) / s;                         // ip.py:2105
    } else {
      x0 = fminf(fmaxf(x0, -1.0f), 1.0f);                       // ip.py:2107
    }
    if (p.x0_thr) p.x0_thr[i] = x0;
    const float xt = p.x[i];
    const float mean = alpha_next * (xt * (1.0f - c) / alpha + c * x0);  // ip.py:265
    const float xn = mean + nonzero * stdev * z[e];                       // ip.py:2164
    p.x[i] = xn;
    if (last && p.final_out) p.final_out[i] = (fminf(fmaxf(xn, -1.0f), 1.0f) + 1.0f) * 0.5f;  // ip.py:2281-2288
  }
}

__global__ void step_advance_kernel(int32_t* step_ptr) { *step_ptr += 1; }

__global__ __launch_bounds__(256) void randn_kernel(const ImagenRandnParams p) {
  const size_t n = (size_t)p.B * p.n_per_sample;
  const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const int bs = (int)(i4 / p.n_per_sample);
  const uint32_t within = (uint32_t)((i4 - (size_t)bs * p.n_per_sample) >> 2);
  float z[4];
  philox_normal4(within, p.tag, p.stream_id, (uint32_t)(p.sample_offset + bs), p.seed_lo, p.seed_hi, z);
#pragma unroll
  for (int e = 0; e < 4; ++e) p.out[i4 + e] = z[e];
}

__global__ __launch_bounds__(256) void lowres_prep_kernel(const ImagenLowresPrepParams p) {
  const size_t n = (size_t)p.B * p.C * p.Hout * p.Wout;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int xo = (int)(i % p.Wout);
  const int yo = (int)((i / p.Wout) % p.Hout);
  const size_t bc = i / ((size_t)p.Wout * p.Hout);
  // F.interpolate(mode='nearest'): src = floor(dst * in/out)
  const int ys = min((int)floorf((float)yo * ((float)p.Hin / (float)p.Hout)), p.Hin - 1);
  const int xs = min((int)floorf((float)xo * ((float)p.Win / (float)p.Wout)), p.Win - 1);
  const float v = p.img[(bc * p.Hin + ys) * p.Win + xs] * 2.0f - 1.0f;
  p.out[i] = p.alpha * v + p.sigma * p.noise[i];
}

// ElucidatedImagen state updates (see ImagenLincombParams).  n_per_sample % 4 == 0: a 4-element group stays inside one sample.
__global__ __launch_bounds__(256) void lincomb_kernel(const ImagenLincombParams p) {
  const size_t n = (size_t)p.B * p.n_per_sample;
  const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const int step = *p.step_ptr;
  const float* w = p.coef + (size_t)step * 8;
  const float w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], w5 = w[5];
  const int b = (int)(i4 / p.n_per_sample);
  float z[4] = {0.f, 0.f, 0.f, 0.f};
  if (w4 != 0.0f) {
    const uint32_t within = (uint32_t)((i4 - (size_t)b * p.n_per_sample) >> 2);
    const uint32_t k0 = p.seed_ptr ? p.seed_ptr[0] : p.seed_lo, k1 = p.seed_ptr ? p.seed_ptr[1] : p.seed_hi;
    philox_normal4(within, (uint32_t)step, p.stream_id, (uint32_t)(p.sample_offset + b), k0, k1, z);
  }
  const float s1 = (p.thr_mode == 1 && p.q1) ? fmaxf(p.q1[b], 1.0f) : 1.0f;
  const float s3 = (p.thr_mode == 1 && p.q3) ? fmaxf(p.q3[b], 1.0f) : 1.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const size_t i = i4 + e;
    float v = w0 * p.t0[i] + w4 * z[e];
    if (p.t1) {
      float t = p.t1[i];
      if (p.thr_mode) t = fminf(fmaxf(t, -s1), s1) / s1;
      v += w1 * t;
    }
    if (p.t2) v += w2 * p.t2[i];
    if (p.t3) {
      float t = p.t3[i];
      if (p.thr_mode) t = fminf(fmaxf(t, -s3), s3) / s3;
      v += w3 * t;
    }
    if (p.mask && p.mask[i] == 0.0f) v = p.mask_else[i];
    p.out[i] = v;
    if (p.out2) p.out2[i] = w5 * v;
    if (p.final && p.final_out) p.final_out[i] = (fminf(fmaxf(v, -1.0f), 1.0f) + 1.0f) * 0.5f;
  }
}

}  // namespace

int launch_lincomb(const ImagenLincombParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->t0 && p->out && p->coef && p->step_ptr, "lincomb: t0 / out / coef / step_ptr required");
  IMAGEN_CHECK(p->n_per_sample % 4 == 0 && p->B > 0, "lincomb: n_per_sample %% 4");
  IMAGEN_CHECK(!p->mask || p->mask_else, "lincomb: mask needs mask_else");
  const size_t n = (size_t)p->B * p->n_per_sample;
  hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, *p);
  if (p->advance) hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, p->step_ptr);
  return imagen_hip_status("lincomb");
}

int launch_cfg_x0(const ImagenCfgX0Params* p, hipStream_t s) {
  IMAGEN_CHECK(p->step_ptr && p->coef, "cfg_x0: coef table / step counter required");
  const size_t n = (size_t)p->B * p->n_per_sample;
  hipLaunchKernelGGL(cfg_x0_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *p);
  return imagen_hip_status("cfg_x0");
}

int launch_quantile(const ImagenQuantileParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->n >= 2 && p->B > 0, "quantile: need n >= 2");
  int bps = (p->n + 8191) / 8192;  // blocks per sample: ~8k keys each
  if (bps < 1) bps = 1;
  if (bps > 64) bps = 64;
  // the scratch arrives clean: initialised by the host when the op is built, re-cleaned by quantile_final_kernel after each use
  for (int pass = 0; pass < 4; ++pass)
    hipLaunchKernelGGL(quantile_hist_kernel, dim3(p->B * bps), dim3(256), 0, s, *p, pass, bps);
  hipLaunchKernelGGL(quantile_tail_kernel, dim3(p->B * bps), dim3(256), 0, s, *p, bps);
  hipLaunchKernelGGL(quantile_final_kernel, dim3(p->B), dim3(256), 0, s, *p);
  return imagen_hip_status("quantile");
}

int launch_randn(const ImagenRandnParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->n_per_sample % 4 == 0, "randn: n_per_sample must be a multiple of 4");
  const size_t n4 = ((size_t)p->B * p->n_per_sample) / 4;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, *p);
  return imagen_hip_status("randn");
}

int launch_lowres_prep(const ImagenLowresPrepParams* p, hipStream_t s) {
  const size_t n = (size_t)p->B * p->C * p->Hout * p->Wout;
  hipLaunchKernelGGL(lowres_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *p);
  return imagen_hip_status("lowres_prep");
}

int launch_ddpm_update(const ImagenDdpmUpdateParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->step_ptr && p->coef, "ddpm_update: coef table / step counter required");
  IMAGEN_CHECK(p->n_per_sample % 4 == 0, "ddpm_update: n_per_sample must be a multiple of 4");
  IMAGEN_CHECK(!p->dynamic_threshold || p->quant, "ddpm_update: dynamic thresholding needs the quantile");
  const size_t n4 = ((size_t)p->B * p->n_per_sample + 3) / 4;
  hipLaunchKernelGGL(ddpm_update_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, *p);
  if (!p->no_advance) hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, p->step_ptr);
  return imagen_hip_status("ddpm_update");
}
