// rowchain.hip — IMAGEN_OP_ROWCHAIN: the token layers of the <= 32^2 levels as ONE launch per chain (round 5, the small-map execution unit).
//
// Why.  The in-graph profile of the sampling loop (profiles/r04_graph_profile.txt) shows the transformer / cross-attention layers of the
// denoiser as chains of 5-10 launches of ~10 us each for a few MFLOP per row: a launch boundary plus two or three dependent round trips
// per op, i.e. latency, not work — 248 launches x 9.5 us per unet1 step at 3.4 % of the matrix peak.  Every op of these chains maps a token
// row to a token row (LayerNorm, Linear, GELU, residual; the keys / values of a cross-attention are constants of the image), so nothing
// but the launch structure forces the round trips through HBM / L2.
//
// What.  A workgroup (8 waves) owns a tile of 32 (or 64) consecutive rows of one image and walks the chain:
//   * GEMM stage:  D[cout][row] += W[cout][k] . X[k][row] on v_mfma_f32_32x32x16_f16 — A fragments (weights) are 16-byte loads straight
//     from the packed weight buffer (L2-resident: every workgroup streams the same few hundred KB), four K steps ahead in a register
//     ring with unconditional, clamped addresses (so the compiler's vmcnt waits are counted, never 0); B fragments (rows) are
//     ds_read_b128 from an LDS row tile of pitch 2 (K + 8) bytes (conflict-free: 4 x odd dwords).  The waves split the output channels;
//     layers with fewer than 8 cout tiles split K over the spare waves and add the partial accumulators through LDS.
//   * row pass:    the LayerNorm / residual / statistics arithmetic of LN_RESIDUAL and ROWSTAT on the fp16 LDS tile the GEMM stage wrote,
//     16 (8) lanes per row, results back into LDS as the next stage's B operand or out to global memory in 16-byte pieces.
//   * cross-attention (mode 2): the q GEMM gives wave h the 64 output channels of head h, so Q^ never leaves the wave — the accumulator
//     registers ARE the B fragments of S^T = K^ . Q^T in a permuted dim order (attention.hip's P trick applied to Q), K^ and V^T
//     fragments are 8-byte loads from the site's operand buffers (16 KB per image and head, L2), online softmax as attention_kernel.
// The rounding points are those of the launches it replaces (fp16 where they stored fp16), so parity is that of the unfused path.
#include "common.h"

namespace {

constexpr int kThreads = 512;
constexpr int kDh = 64;

struct Geo {           // LDS geometry of a launch (launcher and kernel agree through this)
  int p0_cols, p1_cols, p2_cols;
  __host__ __device__ static int pitch(int cols) { return (cols + 8) * 2; }
};

__host__ __device__ inline Geo chain_geo(const ImagenRowchainParams& p) {
  Geo g{};
  if (p.mode == IMAGEN_CHAIN_FF) {
    g.p0_cols = p.inner > p.hidden ? p.inner : p.hidden;
    g.p1_cols = p.C;
    g.p2_cols = p.C;
  } else if (p.mode == IMAGEN_CHAIN_XATTN) {
    g.p0_cols = p.inner;
    g.p1_cols = p.C;
    g.p2_cols = 0;
  } else {
    g.p0_cols = p.inner + 2 * kDh;
    g.p1_cols = p.C;
    g.p2_cols = 0;
  }
  return g;
}

__host__ __device__ inline size_t chain_lds_bytes(const ImagenRowchainParams& p, int rows) {
  const Geo g = chain_geo(p);
  size_t n = (size_t)rows * Geo::pitch(g.p0_cols) + (size_t)rows * Geo::pitch(g.p1_cols) + (g.p2_cols ? (size_t)rows * Geo::pitch(g.p2_cols) : 0);
  return (n + 15) & ~(size_t)15;
}

// how a GEMM stage with T cout tiles of 32 and `ksteps` K = 16 steps is dealt to the 8 waves
struct Part {
  int tile0, nt;        // this wave's cout tiles [tile0, tile0 + nt)
  int wk, kpart;        // K split: wk parts, this wave's part (kpart >= wk: idle)
  int s0, s1;           // this wave's K steps [s0, s1)
  int T;
};

__device__ __forceinline__ Part make_part(int T, int ksteps, int wave) {
  Part q;
  q.T = T;
  if (T >= 8) {
    q.nt = T >> 3;
    q.tile0 = wave * q.nt;
    q.wk = 1;
    q.kpart = 0;
    q.s0 = 0;
    q.s1 = ksteps;
  } else {
    q.nt = 1;
    q.tile0 = wave % T;
    int wk = 8 / T;
    if (wk > ksteps) wk = ksteps;
    q.wk = wk;
    q.kpart = wave / T;
    const int per = ksteps / wk;
    q.s0 = q.kpart < wk ? q.kpart * per : 0;
    q.s1 = q.kpart < wk ? q.s0 + per : 0;
  }
  return q;
}

template <int NT, int RB>
__device__ __forceinline__ void acc_zero(f32x16 (*acc)[RB]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
}

// acc[t][j] += W[tiles tile0 + t][k steps s0 .. s1) . X[rows of row block j]: weights four steps ahead in a register ring (every load
// unconditional, past-the-end steps re-read the last one), rows from the LDS tile xs.
template <int NT, int RB>
__device__ __forceinline__ void gemm_rows(f32x16 (*acc)[RB], const f16* __restrict__ w, int cout_pad, int tile0, int s0, int s1,
                                          const char* xs, int pitch, int lane) {
  const int n = s1 - s0;
  if (n <= 0) return;
  const int half = lane >> 5, l31 = lane & 31;
  const f16* wl = w + ((size_t)half * cout_pad + (size_t)tile0 * 32 + l31) * 8;
  const size_t step = (size_t)2 * cout_pad * 8;
  const char* xl = xs + l31 * pitch + 16 * half;
  const int n4 = n & ~3;
  if (n4 > 0) {
    f16x8 a[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) a[i][t] = *reinterpret_cast<const f16x8*>(wl + (size_t)(s0 + i) * step + t * 256);
    for (int sb = 0; sb < n4; sb += 4) {   // straight-line body: four steps, each followed by the request of the step four ahead
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = sb + i;
        f16x8 b[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) b[j] = *reinterpret_cast<const f16x8*>(xl + (size_t)j * 32 * pitch + (s0 + s) * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < RB; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][t], b[j], acc[t][j], 0, 0, 0);
        const int sn = s0 + (s + 4 < n ? s + 4 : n - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) a[i][t] = *reinterpret_cast<const f16x8*>(wl + (size_t)sn * step + t * 256);
        __builtin_amdgcn_sched_barrier(0);   // pins the request here (the scheduler otherwise sinks look-ahead loads to their use)
      }
    }
  }
  for (int s = n4; s < n; ++s) {           // 1 - 3 remaining steps (the K-split parts of the narrowest layers)
    f16x8 a[NT], b[RB];
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t] = *reinterpret_cast<const f16x8*>(wl + (size_t)(s0 + s) * step + t * 256);
#pragma unroll
    for (int j = 0; j < RB; ++j) b[j] = *reinterpret_cast<const f16x8*>(xl + (size_t)j * 32 * pitch + (s0 + s) * 32);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < RB; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[j], acc[t][j], 0, 0, 0);
  }
}

// K-split: the partial accumulators of the waves with kpart > 0 are added into the tile's owner (kpart 0) through `scratch` (an LDS
// buffer nobody reads any more once every wave has left its K loop).  Called by ALL threads.  NT == 1 whenever wk > 1.
template <int RB>
__device__ __forceinline__ void ksplit_reduce(f32x16 (&acc)[RB], const Part& q, char* scratch, int lane) {
  if (q.wk <= 1) return;            // (uniform over the workgroup)
  __syncthreads();
  if (q.kpart > 0 && q.kpart < q.wk) {
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      f32x4* dst = reinterpret_cast<f32x4*>(scratch + ((size_t)((q.kpart - 1) * q.T + q.tile0) * RB + j) * 4096);
#pragma unroll
      for (int g = 0; g < 4; ++g) dst[g * 64 + lane] = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
    }
  }
  __syncthreads();
  if (q.kpart == 0) {
    for (int k = 1; k < q.wk; ++k) {
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const f32x4* src = reinterpret_cast<const f32x4*>(scratch + ((size_t)((k - 1) * q.T + q.tile0) * RB + j) * 4096);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = src[g * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][4 * g + e] += v[e];
        }
      }
    }
  }
  __syncthreads();                  // (the owners' tile stores may land in the scratch buffer itself)
}

// accumulator tile (cout tile `tile`, row block j) -> fp16 LDS row tile: lane = row, four consecutive channels per 8-byte store
template <int ACT>
__device__ __forceinline__ void store_tile(const f32x16& acc, char* dst, int pitch, int tile, int j, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  char* row = dst + (size_t)(32 * j + l31) * pitch + (tile * 32 + 4 * half) * 2;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = acc[4 * g + e];
      if (ACT == IMAGEN_ACT_GELU) t = gelu_f(t);
      v[e] = (f16)t;
    }
    *reinterpret_cast<f16x4*>(row + g * 16) = v;
  }
}

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int off = LPR >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// LayerNorm statistics (two-pass, as ROWSTAT mode 1 / LN_RESIDUAL) of one fp16 LDS row by its LPR lanes
template <int LPR>
__device__ __forceinline__ void row_ln_stats(const char* row, int C, int li, float eps, float& mean, float& rstd) {
  const int np = C >> 3;
  float s = 0.f;
  for (int g = li; g < np; g += LPR) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(row + g * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)v[e];
  }
  mean = row_sum<LPR>(s) / (float)C;
  float q = 0.f;
  for (int g = li; g < np; g += LPR) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(row + g * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = (float)v[e] - mean;
      q += d * d;
    }
  }
  rstd = rsqrtf(row_sum<LPR>(q) / (float)C + eps);
}

// x rows (global) -> fp16((x - mean) * rstd * g) into the LDS tile: the LayerNorm prologue of a GEMM (IGEMM's (x - mu) * rs * pa)
template <int RB>
__device__ __forceinline__ void load_ln_rows(const ImagenRowchainParams& p, int row0, char* dst, int pitch, int tid) {
  constexpr int LPR = kThreads / (32 * RB);
  const int r = tid / LPR, li = tid % LPR;
  const int C = p.C, np = C >> 3;
  const f16* x = reinterpret_cast<const f16*>(p.x) + (size_t)(row0 + r) * p.ld_x;
  float mean, rstd;
  if (p.mu) {
    mean = p.mu[row0 + r];
    rstd = p.rs[row0 + r];
  } else {
    float s = 0.f;
    for (int g = li; g < np; g += LPR) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(x + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[e];
    }
    mean = row_sum<LPR>(s) / (float)C;
    float q = 0.f;
    for (int g = li; g < np; g += LPR) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(x + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[e] - mean;
        q += d * d;
      }
    }
    rstd = rsqrtf(row_sum<LPR>(q) / (float)C + p.eps);
  }
  char* drow = dst + (size_t)r * pitch;
  for (int g = li; g < np; g += LPR) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(x + g * 8);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)(((float)v[e] - mean) * rstd * p.g0[g * 8 + e]);
    *reinterpret_cast<f16x8*>(drow + g * 16) = o;
  }
}

// y (fp16 LDS tile) -> out = fp16(LN(y) * g + res) -> global rows (16-byte pieces) + the row's sum of squares: LN_RESIDUAL with ssq_out
template <int RB>
__device__ __forceinline__ void ln_res_out_rows(const ImagenRowchainParams& p, int row0, const char* src, int pitch, const float* gain,
                                                const f16* resbase, int ld_res, int tid) {
  constexpr int LPR = kThreads / (32 * RB);
  const int r = tid / LPR, li = tid % LPR;
  const int C = p.C, np = C >> 3;
  const char* srow = src + (size_t)r * pitch;
  float mean, rstd;
  row_ln_stats<LPR>(srow, C, li, p.eps, mean, rstd);
  const f16* res = resbase + (size_t)(row0 + r) * ld_res;
  f16* out = reinterpret_cast<f16*>(p.out) + (size_t)(row0 + r) * p.ld_out;
  float ssq = 0.f;
  for (int g = li; g < np; g += LPR) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(srow + g * 16);
    const f16x8 rv = *reinterpret_cast<const f16x8*>(res + g * 8);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[e] = (f16)(((float)v[e] - mean) * rstd * gain[g * 8 + e] + (float)rv[e]);
      const float t = (float)o[e];
      ssq += t * t;
    }
    *reinterpret_cast<f16x8*>(out + g * 8) = o;
  }
  ssq = row_sum<LPR>(ssq);
  if (p.ssq_out && li == 0) p.ssq_out[row0 + r] = ssq;
}

// one GEMM stage with its K-split reduction; the owners' accumulators stay in acc (first q.nt tiles)
template <int RB>
__device__ __forceinline__ void gemm_stage(f32x16 (&acc)[2][RB], const Part& q, const f16* w, int cout_pad, const char* xs, int pitch, char* scratch,
                                           int lane) {
  acc_zero<2, RB>(acc);
  if (q.nt == 2) gemm_rows<2, RB>(acc, w, cout_pad, q.tile0, q.s0, q.s1, xs, pitch, lane);
  else gemm_rows<1, RB>(acc, w, cout_pad, q.tile0, q.s0, q.s1, xs, pitch, lane);
  ksplit_reduce<RB>(acc[0], q, scratch, lane);
}

template <int ACT, int RB>
__device__ __forceinline__ void store_stage(const f32x16 (&acc)[2][RB], const Part& q, char* dst, int pitch, int lane) {
  if (q.kpart != 0) return;
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < q.nt)
#pragma unroll
      for (int j = 0; j < RB; ++j) store_tile<ACT>(acc[t][j], dst, pitch, q.tile0 + t, j, lane);
}

// ------------------------------------------------------------------------------------------------ mode 1: FF
template <int RB>
__device__ __forceinline__ void chain_ff(const ImagenRowchainParams& p, char* smem, int row0) {
  constexpr int ROWS = 32 * RB, LPR = kThreads / ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo geo = chain_geo(p);
  const int C = p.C, inner = p.inner, hidden = p.hidden;
  const int pitch0o = Geo::pitch(inner), pitch0h = Geo::pitch(hidden), pitch1 = Geo::pitch(C);
  char* P0 = smem;
  char* P1 = P0 + (size_t)ROWS * Geo::pitch(geo.p0_cols);
  char* P2 = P1 + (size_t)ROWS * pitch1;
  // ---- o rows -> P0
  {
    const int npr = inner >> 3;
    const f16* x = reinterpret_cast<const f16*>(p.x);
    for (int i = tid; i < ROWS * npr; i += kThreads) {
      const int r = i / npr, g = i - r * npr;
      *reinterpret_cast<uint4*>(P0 + (size_t)r * pitch0o + g * 16) = *reinterpret_cast<const uint4*>(x + (size_t)(row0 + r) * p.ld_x + g * 8);
    }
  }
  __syncthreads();
  f32x16 acc[2][RB];
  // ---- y = o W_out^T -> P1 (fp16)
  const Part q0 = make_part(C >> 5, inner >> 4, wave);
  gemm_stage<RB>(acc, q0, reinterpret_cast<const f16*>(p.w0), p.w_cout_pad0, P0, pitch0o, P0, lane);
  store_stage<IMAGEN_ACT_NONE, RB>(acc, q0, P1, pitch1, lane);
  __syncthreads();
  // ---- row pass: x1 = fp16(LN(y) * g0 + res) -> P2;  a0 = fp16((x1 - mean x1) * rstd x1 * g1) -> P1
  {
    const int r = tid / LPR, li = tid % LPR, np = C >> 3;
    char* yrow = P1 + (size_t)r * pitch1;
    char* xrow = P2 + (size_t)r * pitch1;
    float mean, rstd;
    row_ln_stats<LPR>(yrow, C, li, p.eps, mean, rstd);
    const f16* res = reinterpret_cast<const f16*>(p.res) + (size_t)(row0 + r) * p.ld_res;
    float so = 0.f;
    for (int g = li; g < np; g += LPR) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(yrow + g * 16);
      const f16x8 rv = *reinterpret_cast<const f16x8*>(res + g * 8);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (f16)(((float)v[e] - mean) * rstd * p.g0[g * 8 + e] + (float)rv[e]);
        so += (float)o[e];
      }
      *reinterpret_cast<f16x8*>(xrow + g * 16) = o;
    }
    const float mo = row_sum<LPR>(so) / (float)C;
    float qo = 0.f;
    for (int g = li; g < np; g += LPR) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(xrow + g * 16);   // (this lane's own pieces: no barrier needed)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[e] - mo;
        qo += d * d;
      }
    }
    const float ro = rsqrtf(row_sum<LPR>(qo) / (float)C + p.eps);
    for (int g = li; g < np; g += LPR) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(xrow + g * 16);
      f16x8 a;
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = (f16)(((float)v[e] - mo) * ro * p.g1[g * 8 + e]);
      *reinterpret_cast<f16x8*>(yrow + g * 16) = a;
    }
  }
  __syncthreads();
  // ---- hid = fp16(gelu(a0 W1^T)) -> P0
  const Part q1 = make_part(hidden >> 5, C >> 4, wave);
  gemm_stage<RB>(acc, q1, reinterpret_cast<const f16*>(p.w1), p.w_cout_pad1, P1, pitch1, P0, lane);
  store_stage<IMAGEN_ACT_GELU, RB>(acc, q1, P0, pitch0h, lane);
  __syncthreads();
  // ---- row pass: a1 = fp16((hid - mean) * rstd * g2), in place
  {
    const int r = tid / LPR, li = tid % LPR, np = hidden >> 3;
    char* hrow = P0 + (size_t)r * pitch0h;
    float mean, rstd;
    row_ln_stats<LPR>(hrow, hidden, li, p.eps, mean, rstd);
    for (int g = li; g < np; g += LPR) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(hrow + g * 16);
      f16x8 a;
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = (f16)(((float)v[e] - mean) * rstd * p.g2[g * 8 + e]);
      *reinterpret_cast<f16x8*>(hrow + g * 16) = a;
    }
  }
  __syncthreads();
  // ---- out = fp16(a1 W2^T + x1) -> P1 -> global
  {
    const Part q2 = make_part(C >> 5, hidden >> 4, wave);
    gemm_stage<RB>(acc, q2, reinterpret_cast<const f16*>(p.w2), p.w_cout_pad2, P0, pitch0h, P0, lane);
    if (q2.kpart == 0) {
      const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (t < q2.nt)
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const char* xr = P2 + (size_t)(32 * j + l31) * pitch1 + ((q2.tile0 + t) * 32 + 4 * half) * 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f16x4 xv = *reinterpret_cast<const f16x4*>(xr + g * 16);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[t][j][4 * g + e] += (float)xv[e];
            }
          }
    }
    store_stage<IMAGEN_ACT_NONE, RB>(acc, q2, P1, pitch1, lane);
  }
  __syncthreads();
  {
    const int r = tid / LPR, li = tid % LPR, np = C >> 3;
    const char* srow = P1 + (size_t)r * pitch1;
    f16* out = reinterpret_cast<f16*>(p.out) + (size_t)(row0 + r) * p.ld_out;
    float ssq = 0.f;
    for (int g = li; g < np; g += LPR) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(srow + g * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) ssq += (float)v[e] * (float)v[e];
      *reinterpret_cast<f16x8*>(out + g * 8) = v;
    }
    ssq = row_sum<LPR>(ssq);
    if (p.ssq_out && li == 0) p.ssq_out[row0 + r] = ssq;
  }
}

// ------------------------------------------------------------------------------------------------ mode 2: XATTN
// wave-local cross attention of head `wave` for one 32-row block: qa = the wave's two q accumulator tiles (dims 0-31 | 32-63 of the head)
__device__ __forceinline__ void head_attention(const ImagenRowchainParams& p, const f32x16 (&qa)[2], int b, int hd, char* orow /* P0 row of this lane */,
                                               int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  // ---- Q^: fp16(q) -> l2norm * q_scale * q_mult (ATTENTION's fused QNORM), as B fragments in the accumulator's own dim order:
  // fragment (t, s), element e  <->  dim 32 t + 16 s + 4 half + (e & 3) + 8 (e >> 2)
  f16 q16[2][16];
  float ssq = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      q16[t][r] = (f16)qa[t][r];
      ssq += (float)q16[t][r] * (float)q16[t][r];
    }
  ssq += __shfl_xor(ssq, 32);
  const float inv = p.q_mult / fmaxf(sqrtf(ssq), 1e-12f);
  f16x8 qf[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
      qf[t][r >> 3][r & 7] = (f16)((float)q16[t][r] * inv * p.q_scale[d]);
    }
  const f16* kg = reinterpret_cast<const f16*>(p.khat) + (size_t)b * p.k_bs + (size_t)hd * p.k_hs;
  const f16* vg = reinterpret_cast<const f16*>(p.vt) + (size_t)b * p.vt_bs + (size_t)hd * p.vt_hs;
  f32x16 oacc[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;
  const int ntiles = (p.J + 31) >> 5;
  for (int kt = 0; kt < ntiles; ++kt) {
    // ---- S^T[key][row] = K^ . Q^T: A fragment of key 32 kt + l31 in the same permuted dim order (two 8-byte pieces)
    const f16* krow = kg + (size_t)(32 * kt + l31) * p.k_rs + 4 * half;
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint2 lo = *reinterpret_cast<const uint2*>(krow + 32 * t + 16 * s);
        const uint2 hi = *reinterpret_cast<const uint2*>(krow + 32 * t + 16 * s + 8);
        uint4 pk = make_uint4(lo.x, lo.y, hi.x, hi.y);
        const f16x8 kf = *reinterpret_cast<const f16x8*>(&pk);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[t][s], sacc, 0, 0, 0);
      }
    // ---- online softmax (attention_kernel's): lane = row, this lane holds 16 of the tile's 32 keys
    const int kbase = 32 * kt + 4 * half;
    float mx = -1.0e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kbase + (r & 3) + 8 * (r >> 2);
      if (key >= p.J) sacc[r] = -1.0e30f;
      mx = fmaxf(mx, sacc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    f16x8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = exp2f(sacc[r] - m_new);
      psum += e;
      pf[r >> 3][r & 7] = (f16)e;
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    // ---- O^T[d][row] += V^T . P (k-step s covers the keys of accumulator registers 8 s .. 8 s + 7)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const f16* vrow = vg + (size_t)(32 * db + l31) * p.vt_ds + 32 * kt + 16 * s + 4 * half;
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 8);
        uint4 pk = make_uint4(lo.x, lo.y, hi.x, hi.y);
        const f16x8 vf = *reinterpret_cast<const f16x8*>(&pk);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], oacc[db], 0, 0, 0);
      }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float il = 1.0f / l_tot;
  // o[row][hd * 64 + 32 db + 8 g + 4 half + e] -> the P0 row tile (the B operand of the out-projection)
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (f16)(oacc[db][4 * g + e] * il);
      *reinterpret_cast<f16x4*>(orow + (hd * 64 + 32 * db + 8 * g + 4 * half) * 2) = v;
    }
}

template <int RB>
__device__ __forceinline__ void chain_xattn(const ImagenRowchainParams& p, char* smem, int row0) {
  constexpr int ROWS = 32 * RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31;
  const int C = p.C, inner = p.inner;
  const int pitch0 = Geo::pitch(inner), pitch1 = Geo::pitch(C);
  char* P0 = smem;
  char* P1 = P0 + (size_t)ROWS * pitch0;
  load_ln_rows<RB>(p, row0, P1, pitch1, tid);
  __syncthreads();
  f32x16 acc[2][RB];
  // ---- q = a Wq^T: wave h owns the 64 output channels of head h (two cout tiles), all K steps
  {
    Part q;
    q.T = inner >> 5;
    q.nt = 2;
    q.tile0 = 2 * wave;
    q.wk = 1;
    q.kpart = 0;
    q.s0 = 0;
    q.s1 = C >> 4;
    acc_zero<2, RB>(acc);
    gemm_rows<2, RB>(acc, reinterpret_cast<const f16*>(p.w0), p.w_cout_pad0, q.tile0, q.s0, q.s1, P1, pitch1, lane);
  }
  const int b = row0 / p.rows_per_batch;
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const f32x16 qa[2] = {acc[0][j], acc[1][j]};
    head_attention(p, qa, b, wave, P0 + (size_t)(32 * j + l31) * pitch0, lane);
  }
  __syncthreads();
  // ---- y = o W_out^T -> P1 (the normalised input rows are dead)
  const Part q1 = make_part(C >> 5, inner >> 4, wave);
  gemm_stage<RB>(acc, q1, reinterpret_cast<const f16*>(p.w1), p.w_cout_pad1, P0, pitch0, P0, lane);
  store_stage<IMAGEN_ACT_NONE, RB>(acc, q1, P1, pitch1, lane);
  __syncthreads();
  // ---- out = fp16(LN(y) * g1 + x)
  const f16* resbase = p.res ? reinterpret_cast<const f16*>(p.res) : reinterpret_cast<const f16*>(p.x);
  ln_res_out_rows<RB>(p, row0, P1, pitch1, p.g1, resbase, p.res ? p.ld_res : p.ld_x, tid);
}

// ------------------------------------------------------------------------------------------------ mode 3: QKV
template <int RB>
__device__ __forceinline__ void chain_qkv(const ImagenRowchainParams& p, char* smem, int row0) {
  constexpr int ROWS = 32 * RB, LPR = kThreads / ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = p.C, inner = p.inner, nout = inner + 2 * kDh;
  const int pitch0 = Geo::pitch(nout), pitch1 = Geo::pitch(C);
  char* P0 = smem;
  char* P1 = P0 + (size_t)ROWS * pitch0;
  load_ln_rows<RB>(p, row0, P1, pitch1, tid);
  __syncthreads();
  // ---- y = a [Wq | Wkv]^T: 20 cout tiles — every wave two (q head `wave`), waves 0-3 one of the k | v tiles on top
  f32x16 acc[2][RB];
  acc_zero<2, RB>(acc);
  gemm_rows<2, RB>(acc, reinterpret_cast<const f16*>(p.w0), p.w_cout_pad0, 2 * wave, 0, C >> 4, P1, pitch1, lane);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < RB; ++j) store_tile<IMAGEN_ACT_NONE>(acc[t][j], P0, pitch0, 2 * wave + t, j, lane);
  if (wave < 4) {
    f32x16 a1[1][RB];
    acc_zero<1, RB>(a1);
    gemm_rows<1, RB>(a1, reinterpret_cast<const f16*>(p.w0), p.w_cout_pad0, 16 + wave, 0, C >> 4, P1, pitch1, lane);
#pragma unroll
    for (int j = 0; j < RB; ++j) store_tile<IMAGEN_ACT_NONE>(a1[0][j], P0, pitch0, 16 + wave, j, lane);
  }
  __syncthreads();
  // ---- row pass: q pieces -> out rows; K^ = l2norm(k) * k_scale -> khat row; v -> V^T column
  {
    const int r = tid / LPR, li = tid % LPR;
    const char* srow = P0 + (size_t)r * pitch0;
    f16* out = reinterpret_cast<f16*>(p.out) + (size_t)(row0 + r) * p.ld_out;
    const int npq = inner >> 3;
    for (int g = li; g < npq; g += LPR) *reinterpret_cast<uint4*>(out + g * 8) = *reinterpret_cast<const uint4*>(srow + g * 16);
    const int b = row0 / p.rows_per_batch, n = row0 + r - b * p.rows_per_batch;
    // k: 8 pieces of 8 dims by lanes 0-7 of the row's group (LPR >= 8)
    {
      const bool on = li < 8;
      const f16x8 v = *reinterpret_cast<const f16x8*>(srow + (inner + (on ? li : 0) * 8) * 2);
      float ssq = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ssq += (float)v[e] * (float)v[e];
      if (!on) ssq = 0.f;
      ssq = row_sum<8>(ssq);                 // (lanes 0-7 of the group are an aligned 8-lane block)
      const float inv = 1.0f / fmaxf(sqrtf(ssq), 1e-12f);
      if (on) {
        f16x8 ko;
#pragma unroll
        for (int e = 0; e < 8; ++e) ko[e] = (f16)((float)v[e] * inv * p.k_scale[li * 8 + e]);
        f16* kh = reinterpret_cast<f16*>(p.khat) + (size_t)b * p.k_bs + (size_t)(p.r0 + n) * p.k_rs + li * 8;
        *reinterpret_cast<f16x8*>(kh) = ko;
      }
    }
    // v: dims li, li + LPR, ... of the row -> V^T[d][r0 + n]
    {
      f16* vt = reinterpret_cast<f16*>(p.vt) + (size_t)b * p.vt_bs + (p.r0 + n);
      const f16* vrow = reinterpret_cast<const f16*>(srow + (inner + kDh) * 2);
      for (int d = li; d < kDh; d += LPR) vt[(size_t)d * p.vt_ds] = vrow[d];
    }
  }
}

template <int MODE, int RB>
__global__ __launch_bounds__(kThreads) void rowchain_kernel(const ImagenRowchainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int row0 = blockIdx.x * 32 * RB;
  if (MODE == IMAGEN_CHAIN_FF) chain_ff<RB>(p, smem, row0);
  else if (MODE == IMAGEN_CHAIN_XATTN) chain_xattn<RB>(p, smem, row0);
  else chain_qkv<RB>(p, smem, row0);
}

template <int MODE, int RB>
int launch_one(const ImagenRowchainParams& p, hipStream_t s) {
  const size_t lds = chain_lds_bytes(p, 32 * RB);
  IMAGEN_CHECK(lds <= 160 * 1024, "rowchain: %zu bytes of LDS", lds);
  static bool attr_set[16] = {};   // per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rowchain_kernel<MODE, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((rowchain_kernel<MODE, RB>), dim3((unsigned)(p.rows / (32 * RB))), dim3(kThreads), lds, s, p);
  return imagen_hip_status("rowchain");
}

}  // namespace

int launch_rowchain(const ImagenRowchainParams* pp, hipStream_t s) {
  const ImagenRowchainParams& p = *pp;
  IMAGEN_CHECK(p.mode >= IMAGEN_CHAIN_FF && p.mode <= IMAGEN_CHAIN_QKV, "rowchain: mode %d", p.mode);
  IMAGEN_CHECK(p.x && p.out && p.w0 && p.g0, "rowchain: null pointer");
  const int tile = p.tile64 ? 64 : 32;
  IMAGEN_CHECK(p.rows > 0 && p.rows_per_batch > 0 && p.rows % p.rows_per_batch == 0 && p.rows_per_batch % tile == 0,
               "rowchain: %d rows, %d per image, %d-row tiles", p.rows, p.rows_per_batch, tile);
  IMAGEN_CHECK(p.C >= 32 && p.C <= 256 && (p.C & (p.C - 1)) == 0, "rowchain: C = %d (a power of two in 32 .. 256: 1, 2, 4 or 8 cout tiles)", p.C);
  IMAGEN_CHECK(p.inner == 512 && p.heads * kDh == p.inner, "rowchain: heads x 64 == 512 (got %d heads, inner %d)", p.heads, p.inner);
  IMAGEN_CHECK(p.ld_x % 8 == 0 && p.ld_out % 8 == 0 && ((size_t)p.x & 15) == 0 && ((size_t)p.out & 15) == 0, "rowchain: 16-byte aligned rows");
  IMAGEN_CHECK((p.mu == nullptr) == (p.rs == nullptr), "rowchain: mu and rs come together");
  if (p.mode == IMAGEN_CHAIN_FF) {
    IMAGEN_CHECK(p.res && p.w1 && p.w2 && p.g1 && p.g2 && p.ld_res % 8 == 0 && ((size_t)p.res & 15) == 0, "rowchain FF: null / unaligned operand");
    IMAGEN_CHECK(p.hidden >= 32 && p.hidden <= 512 && (p.hidden & (p.hidden - 1)) == 0, "rowchain FF: hidden = %d (a power of two in 32 .. 512)", p.hidden);
    IMAGEN_CHECK(p.w_cout_pad0 >= p.C && p.w_cout_pad1 >= p.hidden && p.w_cout_pad2 >= p.C, "rowchain FF: weight padding");
    return p.tile64 ? launch_one<IMAGEN_CHAIN_FF, 2>(p, s) : launch_one<IMAGEN_CHAIN_FF, 1>(p, s);
  }
  if (p.mode == IMAGEN_CHAIN_XATTN) {
    IMAGEN_CHECK(p.w1 && p.g1 && p.khat && p.vt && p.q_scale && p.J > 0, "rowchain XATTN: null operand");
    IMAGEN_CHECK(p.k_rs % 4 == 0 && p.k_hs % 4 == 0 && p.k_bs % 4 == 0 && p.vt_ds % 4 == 0 && p.vt_hs % 4 == 0 && p.vt_bs % 4 == 0 &&
                     ((size_t)p.khat & 7) == 0 && ((size_t)p.vt & 7) == 0,
                 "rowchain XATTN: 8-byte aligned operand rows");
    IMAGEN_CHECK(p.w_cout_pad0 >= p.inner && p.w_cout_pad1 >= p.C, "rowchain XATTN: weight padding");
    IMAGEN_CHECK(!p.res || (p.ld_res % 8 == 0 && ((size_t)p.res & 15) == 0), "rowchain XATTN: unaligned residual");
    return p.tile64 ? launch_one<IMAGEN_CHAIN_XATTN, 2>(p, s) : launch_one<IMAGEN_CHAIN_XATTN, 1>(p, s);
  }
  IMAGEN_CHECK(p.khat && p.vt && p.k_scale, "rowchain QKV: null operand");
  IMAGEN_CHECK(p.k_rs % 8 == 0 && p.k_bs % 8 == 0 && ((size_t)p.khat & 15) == 0, "rowchain QKV: 16-byte aligned K^ rows");
  IMAGEN_CHECK(p.w_cout_pad0 >= p.inner + 2 * kDh, "rowchain QKV: weight padding");
  return p.tile64 ? launch_one<IMAGEN_CHAIN_QKV, 2>(p, s) : launch_one<IMAGEN_CHAIN_QKV, 1>(p, s);
}
