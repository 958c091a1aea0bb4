// attention.hip — flash-style cosine-similarity attention for the Imagen denoiser on CDNA4 MFMA.
// Replaces the materialised (b, h, n, n+39) fp32 similarity + softmax + AV of ip.py:559-590 (self attention,
// ONE k/v head shared by all 8 query heads), ip.py:812-833 (cross attention, per-head k/v, 39|41 keys) and
// PerceiverAttention ip.py:424-444.
//
// Inputs are prepared by QNORM (or the fused q_scale path below) / KV_PREP: q rows l2-normalised * q_scale * 8 * log2(e)
// (so softmax is exp2 of the raw dot product), k rows l2-normalised * k_scale, V stored transposed (V^T[d][key]) and both
// K / V^T zero-padded to a multiple of 32 keys.  For the shared-k/v self attention the host passes
// heads = 1 and rows = n*8: the (token, head) pairs are just 8n query rows over one key set.
//
// Workgroup = 4 wave64 = 128 query rows of one (batch, head); wave = 32 query rows.  Per 32-key tile:
//   S^T[key][q]  = mfma_32x32x16(A = K tile rows (LDS, ds_read_b128), B = Q (registers))      4 MFMA
//   online softmax in registers: lane = query column, 16 keys per lane + 1 cross-half shuffle
//   O^T[d][q]   += mfma_32x32x16(A = V^T rows (LDS, 2x ds_read_b64), B = P (registers, fp16))  4 MFMA
// The key order of the PV contraction is chosen to match the S^T accumulator layout, so P never moves
// between lanes.  K / V^T tiles are register-prefetched one tile ahead and double-buffered in LDS.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

constexpr int KT = 32;        // keys per tile
constexpr int KSTR = 144;     // LDS bytes per K row at head dim 64 (128 + 16: conflict-free ds_read_b128)
constexpr int VSTR = 72;      // LDS bytes per V^T row (64 + 8: conflict-free ds_read_b64)

// D = head dim: 64 (every README config) or 32 (the reference's UnetConfig default, configs.py:48-49)
template <int D>
__global__ __launch_bounds__(256) void attention_kernel(const ImagenAttentionParams p) {
  constexpr int KS = D / 16;            // K=16 steps of the S^T contraction
  constexpr int DB = D / 32;            // 32-dim blocks of O^T
  constexpr int KSTRD = 2 * D + 16;     // LDS bytes per K row
  constexpr int KBYTES = KT * KSTRD;
  constexpr int VBYTES = D * VSTR;
  __shared__ __attribute__((aligned(16))) char smem[2 * (KBYTES + VBYTES)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int row = blockIdx.x * 128 + wave * 32 + l31;
  const int row_c = row < p.rows ? row : p.rows - 1;

  const f16* q = reinterpret_cast<const f16*>(p.q) + (size_t)b * p.q_bs + (size_t)hd * p.q_hs + (size_t)row_c * p.q_rs;
  f16x8 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const f16x8*>(q + 16 * s + 8 * half);
  if (p.q_scale) {   // fused QNORM (ip.py:559-560): this lane holds half of the row's dims, lane ^ 32 the other half
    float ssq = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) ssq += (float)qf[s][j] * (float)qf[s][j];
    ssq += __shfl_xor(ssq, 32);
    const float inv = p.q_mult / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 g0 = *reinterpret_cast<const float4*>(p.q_scale + 16 * s + 8 * half);
      const float4 g1 = *reinterpret_cast<const float4*>(p.q_scale + 16 * s + 8 * half + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (f16)((float)qf[s][j] * inv * g[j]);
    }
  }

  const f16* kg = reinterpret_cast<const f16*>(p.k) + (size_t)b * p.k_bs + (size_t)hd * p.k_hs;
  const f16* vg = reinterpret_cast<const f16*>(p.vt) + (size_t)b * p.vt_bs + (size_t)hd * p.vt_hs;
  // staging roles: K tile 32 keys x D/8 groups of 8 dims; V^T tile D dims x 4 groups of 8 keys (D = 32: threads 0-127 only)
  const bool stager = tid < 4 * D;
  const int sk_key = tid / (D / 8), sk_dg = tid % (D / 8);
  const int sv_d = (tid >> 2) & (D - 1), sv_kg = tid & 3;

  uint4 k_stage = make_uint4(0, 0, 0, 0), v_stage = make_uint4(0, 0, 0, 0);
  auto tile_load = [&](int kt0) {
    if (stager) {
      k_stage = *reinterpret_cast<const uint4*>(kg + (size_t)(kt0 + sk_key) * p.k_rs + sk_dg * 8);
      v_stage = *reinterpret_cast<const uint4*>(vg + (size_t)sv_d * p.vt_ds + kt0 + sv_kg * 8);
    }
  };
  auto tile_store = [&](char* buf) {
    if (stager) {
      *reinterpret_cast<uint4*>(buf + sk_key * KSTRD + sk_dg * 16) = k_stage;
      uint2* vd = reinterpret_cast<uint2*>(buf + KBYTES + sv_d * VSTR + sv_kg * 16);
      vd[0] = make_uint2(v_stage.x, v_stage.y);
      vd[1] = make_uint2(v_stage.z, v_stage.w);
    }
  };

  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int ntiles = (p.J + KT - 1) / KT;
  tile_load(0);
  tile_store(smem);
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) tile_load((t + 1) * KT);
    const char* kb = smem + cur * (KBYTES + VBYTES);
    const char* vb = kb + KBYTES;

    // ---- S^T = K . Q^T
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f16x8 kf = *reinterpret_cast<const f16x8*>(kb + l31 * KSTRD + (16 * s + 8 * half) * 2);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc, 0, 0, 0);
    }
    // ---- mask the ragged last tile, online softmax (lane = query; this lane holds 16 of the tile's 32 keys)
    const int kbase = t * KT + 4 * half;
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
    // P enters the PV product as fp16 hi + lo pairs (round 6; the contract keeps P in fp32): this kernel serves the once-per-request
    // Perceiver attention pooling of the text tokens, whose bare-fp16 P was the most COHERENT deviation of a denoiser from its contract (one
    // vector on every row: profiles/r05_t_op_audit_c5_null_row.txt), and the sites of fewer than 256 query rows, where the MFMAs are free
    f16x8 pf[2], pl[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = exp2f(sacc[r] - m_new);
      psum += e;
      const f16 eh = (f16)e;
      pf[r >> 3][r & 7] = eh;
      pl[r >> 3][r & 7] = (f16)(e - (float)eh);
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;

    // ---- O^T += V^T . P^T   (k-step s covers the keys of accumulator registers 8s..8s+7)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const char* vrow = vb + (32 * db + l31) * VSTR + (16 * s + 4 * half) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16);
        uint4 packed = make_uint4(lo.x, lo.y, hi.x, hi.y);
        const f16x8 vf = *reinterpret_cast<const f16x8*>(&packed);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pl[s], oacc[db], 0, 0, 0);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], oacc[db], 0, 0, 0);
      }
    }

    if (more) tile_store(smem + (cur ^ 1) * (KBYTES + VBYTES));
    __syncthreads();
    cur ^= 1;
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (row < p.rows) {
    f16* o = reinterpret_cast<f16*>(p.o) + (size_t)b * p.o_bs + (size_t)hd * p.o_hs + (size_t)row * p.o_rs;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (f16)(oacc[db][4 * qd + e] * inv);
        *reinterpret_cast<f16x4*>(o + 32 * db + 8 * qd + 4 * half) = v;
      }
  }
}

// ---- second tiling: 8 waves (256 query rows) per workgroup, 64-key tiles.
// Per tile and wave 8 + 8 MFMAs behind ONE workgroup barrier (the 4-wave / 32-key kernel above: 4 + 4), the K / V^T tile is staged
// once for twice the queries, the ragged-tile mask is applied to the last tile only, exponentials are raw v_exp_f32, and the
// accumulator rescale (32 multiplies per lane) is skipped while no lane of the wave has seen a new maximum — after the first few tiles
// of a row that is almost always.  Used when a (batch, head) has at least 256 query rows.
constexpr int KT2 = 64;
constexpr int VSTR2 = 144;                 // LDS bytes per V^T row (128 + 16: conflict-free ds_read_b128)
constexpr int KBYTES2 = KT2 * KSTR;
constexpr int VBYTES2 = 64 * VSTR2;

// MINW: waves per SIMD the register budget is capped for (4: two workgroups per CU; 2: one, no spills).  SUB: 64-key tiles per
// workgroup barrier — a staging buffer holds SUB tiles, so with SUB = 2 the 8 waves meet half as often (counters of SUB = 1 on the
// 1024-token site: 37 % of the wave cycles parked at s_waitcnt / s_barrier, round-2 call gpu_r2_y.sh).
template <int MINW, int SUB>
__global__ __launch_bounds__(512, MINW) void attention_kernel_w8(const ImagenAttentionParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 buffers x SUB tiles x (KBYTES2 + VBYTES2)
  constexpr int SLOT = KBYTES2 + VBYTES2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int row = blockIdx.x * 256 + wave * 32 + l31;
  const int row_c = row < p.rows ? row : p.rows - 1;

  const f16* q = reinterpret_cast<const f16*>(p.q) + (size_t)b * p.q_bs + (size_t)hd * p.q_hs + (size_t)row_c * p.q_rs;
  f16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const f16x8*>(q + 16 * s + 8 * half);
  if (p.q_scale) {   // fused QNORM (ip.py:559-560)
    float ssq = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) ssq += (float)qf[s][j] * (float)qf[s][j];
    ssq += __shfl_xor(ssq, 32);
    const float inv = p.q_mult / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 g0 = *reinterpret_cast<const float4*>(p.q_scale + 16 * s + 8 * half);
      const float4 g1 = *reinterpret_cast<const float4*>(p.q_scale + 16 * s + 8 * half + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (f16)((float)qf[s][j] * inv * g[j]);
    }
  }

  const f16* kg = reinterpret_cast<const f16*>(p.k) + (size_t)b * p.k_bs + (size_t)hd * p.k_hs;
  const f16* vg = reinterpret_cast<const f16*>(p.vt) + (size_t)b * p.vt_bs + (size_t)hd * p.vt_hs;
  // staging roles (512 threads, one 16-byte item each per tile): K tile 64 keys x 8 groups of 8 dims; V^T tile 64 dims x 8 groups of 8
  // keys.  K / V^T are zero-padded to a multiple of 32 keys by KV_PREP: the second half of the last 64-key tile may lie beyond the
  // padding, its loads are redirected to the tile's first half (masked below, never used); tiles past the end re-read tile 0
  const int sk_key = tid >> 3, sk_dg = tid & 7;
  const int sv_d = tid >> 3, sv_kg = tid & 7;
  const int Jpad = (p.J + 31) & ~31;
  const int ntiles = (p.J + KT2 - 1) / KT2;

  uint4 ks0, vs0, ks1, vs1;   // staged tile(s) (scalars, not arrays: the lambdas below would otherwise pin them in scratch)
  auto one_load = [&](int t, uint4& ks, uint4& vs) __attribute__((always_inline)) {
    const int kt0 = t < ntiles ? t * KT2 : 0;
    const int kk = kt0 + sk_key < Jpad ? kt0 + sk_key : kt0;
    const int kv = kt0 + sv_kg * 8 < Jpad ? kt0 + sv_kg * 8 : kt0;
    ks = *reinterpret_cast<const uint4*>(kg + (size_t)kk * p.k_rs + sk_dg * 8);
    vs = *reinterpret_cast<const uint4*>(vg + (size_t)sv_d * p.vt_ds + kv);
  };
  auto one_store = [&](char* buf, const uint4& ks, const uint4& vs) __attribute__((always_inline)) {
    *reinterpret_cast<uint4*>(buf + sk_key * KSTR + sk_dg * 16) = ks;
    // keys of a 16-key group are stored in the order [0-3, 8-11, 4-7, 12-15]: the 8 keys a lane contracts in one PV step (the keys
    // of 8 consecutive S^T accumulator registers: 4h + {0-3, 8-11}) are then contiguous — one ds_read_b128 instead of two b64
    char* vrow = buf + KBYTES2 + sv_d * VSTR2 + (sv_kg >> 1) * 32 + (sv_kg & 1) * 8;
    *reinterpret_cast<uint2*>(vrow) = make_uint2(vs.x, vs.y);
    *reinterpret_cast<uint2*>(vrow + 16) = make_uint2(vs.z, vs.w);
  };
  auto group_load = [&](int t0) __attribute__((always_inline)) {
    one_load(t0, ks0, vs0);
    if constexpr (SUB > 1) one_load(t0 + 1, ks1, vs1);
  };
  auto group_store = [&](char* buf) __attribute__((always_inline)) {
    one_store(buf, ks0, vs0);
    if constexpr (SUB > 1) one_store(buf + SLOT, ks1, vs1);
  };

  f32x16 oacc[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  group_load(0);
  group_store(smem);
  __syncthreads();
  int cur = 0;
  for (int t0 = 0; t0 < ntiles; t0 += SUB) {
    const bool more = t0 + SUB < ntiles;
    if (more) group_load(t0 + SUB);
#pragma unroll
    for (int u = 0; u < SUB; ++u) {
      const int t = t0 + u;
      if (u > 0 && t >= ntiles) continue;   // (wave-uniform)
      const char* kb = smem + (cur * SUB + u) * SLOT;
      const char* vb = kb + KBYTES2;

      // ---- S^T = K . Q^T for the two 32-key halves of the tile
      f32x16 sacc[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[h2][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const f16x8 kf = *reinterpret_cast<const f16x8*>(kb + (32 * h2 + l31) * KSTR + (16 * s + 8 * half) * 2);
          sacc[h2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[h2], 0, 0, 0);
        }
      }
      if (t == ntiles - 1) {   // ragged last tile: keys >= J contribute nothing
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KT2 + 32 * h2 + 4 * half + (r & 3) + 8 * (r >> 2);
            if (key >= p.J) sacc[h2][r] = -1.0e30f;
          }
      }
      // ---- online softmax (lane = query; this lane holds 32 of the tile's 64 keys, lane ^ 32 the others)
      float mx = sacc[0][0];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[h2][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      if (__any(m_new > m_run)) {   // wave-uniform: rescale only when some row of the wave has a new maximum
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        m_run = m_new;
      }
      float psum = 0.f;
      f16x8 pf[4];   // (fp16 P: hi + lo pairs as in attention_kernel above spill inside this tiling's loop at its 128-register budget — round 6, not kept)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(sacc[h2][r] - m_run);
          psum += e;
          pf[2 * h2 + (r >> 3)][r & 7] = (f16)e;
        }
      l_run += psum;

      // ---- O^T += V^T . P^T   (k-step s of key half h2 covers the keys of accumulator registers 8s..8s+7 of sacc[h2])
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const f16x8 vf = *reinterpret_cast<const f16x8*>(vb + (32 * db + l31) * VSTR2 + (32 * h2 + 16 * s + 8 * half) * 2);
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[2 * h2 + s], oacc[db], 0, 0, 0);
          }
        }
    }
    if (more) group_store(smem + (cur ^ 1) * SUB * SLOT);
    __syncthreads();
    cur ^= 1;
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (row < p.rows) {
    f16* o = reinterpret_cast<f16*>(p.o) + (size_t)b * p.o_bs + (size_t)hd * p.o_hs + (size_t)row * p.o_rs;
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

// ---- third tiling: bounded logits, no running maximum (ImagenAttentionParams.softmax_mode = 1).
// q and k rows are l2-normalised and scaled by fixed parameter vectors, so every logit lies in [-B, B] with B known when the plan is built
// (ops.attention: q_mult * max_d |q_scale_d k_scale_d|, in log2 units).  With B <= 14 the weights exp2(s) of ALL keys fit the NORMAL fp16
// range at once (2^-14 .. 2^14), so the softmax needs no row maximum, no subtraction, no accumulator rescale and no cross-tile dependency:
// p = v_exp_f32(s) straight from the MFMA result, the row sum and one conversion per element are all that is left on the VALU (the online
// kernel above: + max, subtract, the rescale test).  Without the serial m_run chain the three stages of a 32-key half tile —
// S^T(i+1) = K.Q^T (4 MFMA), exp2 of S^T(i) (16 v_exp_f32 per lane), O^T += V^T.P(i-1) (4 MFMA) — are independent of each other, and a
// wave interleaves them instruction by instruction.  The K / V^T fragments of half step i+1 are read from LDS during half step i (after a
// workgroup barrier all eight waves would otherwise ask the LDS for their 8 KB at once and wait for it together: 512 cycles of LDS time
// per round, measured as a third of the kernel in round-4 call E).  Tiles of 64 keys move through a ring of four LDS slots, three tiles
// ahead of the half step that multiplies them; one workgroup barrier per tile.
constexpr int BND_RING = 4;

template <class F, int... I>
__device__ __forceinline__ void att_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void att_static_for(F&& f) {
  att_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

struct AttFrag { f16x8 k[4], v[4]; };   // one half step's operands: K rows (4 K=16 steps) and V^T rows (2 key steps x 2 dim blocks)

__device__ __forceinline__ void att_frag_load(AttFrag& f, const char* krow, const char* vrow) {
#pragma unroll
  for (int s = 0; s < 4; ++s) f.k[s] = *reinterpret_cast<const f16x8*>(krow + 32 * s);
#pragma unroll
  for (int k = 0; k < 4; ++k) f.v[k] = *reinterpret_cast<const f16x8*>(vrow + (k & 1) * 32 * VSTR2 + (k >> 1) * 32);
}

// One half step: Sn = K.Q^T out of fc.k (QK), pc = exp2(Sc) (+ row sums), oacc += V^T.pp out of fc.v (PV); fn <- the next half step's fragments
template <bool QK, bool PV>
__device__ __forceinline__ void att_bnd_step(f32x16& Sn, f32x16& Sc, f16x8 (&pc)[2], const f16x8 (&pp)[2], f32x16 (&oacc)[2], float& sum0, float& sum1,
                                             const f16x8 (&qf)[4], const AttFrag& fc, AttFrag& fn, const char* krow_next, const char* vrow_next) {
  // One scheduling region per half step.  The exponentials are pure VALU work the instruction selector is free to place anywhere after
  // their inputs exist (it put them right behind the previous step's last MFMA): the empty asm makes S^T(i) opaque HERE, so they stay
  // inside this region, and the sched_group_barrier pipeline below deals them between the matrix instructions.
  __builtin_amdgcn_sched_barrier(0);
  IMAGEN_OPAQUE(Sc);
  att_frag_load(fn, krow_next, vrow_next);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float e = __builtin_amdgcn_exp2f(Sc[r]);
    if (r & 1) sum1 += e; else sum0 += e;
    pc[r >> 3][r & 7] = (f16)e;
  }
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if constexpr (QK) Sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(fc.k[k], qf[k], k == 0 ? zero : Sn, 0, 0, 0);
    if constexpr (PV) oacc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fc.v[k], pp[k >> 1], oacc[k & 1], 0, 0, 0);
  }
  constexpr int NM = (QK ? 4 : 0) + (PV ? 4 : 0);
  att_static_for<NM>([&](auto kc) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 8 / NM, 0);           // the next half step's fragment reads
    __builtin_amdgcn_sched_group_barrier(0x400, 16 / NM, 0);          // transcendentals
    __builtin_amdgcn_sched_group_barrier(0x002, NM == 8 ? 3 : 6, 0);  // other VALU (sums, conversions)
  });
  __builtin_amdgcn_sched_barrier(0);
}

// NW waves (32 query rows each) per workgroup.  NW = 4: two workgroups share a CU (2 x 72 KB of LDS, one wave per SIMD each) — their
// barriers are independent and one's prologue / epilogue (a third of a workgroup's life on the 1024-token site: round-4 call I, loop
// removed: 18 of 60 us) runs behind the other's tile steps; each thread then stages two 16-byte items of K and of V^T per tile.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attention_kernel_bnd(const ImagenAttentionParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // BND_RING slots of one 64-key tile (K rows | V^T rows)
  constexpr int SLOT = KBYTES2 + VBYTES2;
  constexpr int NT = 64 * NW, IT = 512 / NT;   // staging items (16 bytes of K + 16 bytes of V^T each) per thread and tile
  static_assert(IT * NT == 512 && IT <= 2, "512 items per tile, one or two per thread");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int row = blockIdx.x * (32 * NW) + wave * 32 + l31;
  const int row_c = row < p.rows ? row : p.rows - 1;

  const f16* q = reinterpret_cast<const f16*>(p.q) + (size_t)b * p.q_bs + (size_t)hd * p.q_hs + (size_t)row_c * p.q_rs;
  f16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const f16x8*>(q + 16 * s + 8 * half);
  if (p.q_scale) {   // fused QNORM (ip.py:559-560)
    float ssq = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) ssq += (float)qf[s][j] * (float)qf[s][j];
    ssq += __shfl_xor(ssq, 32);
    const float inv = p.q_mult / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 g0 = *reinterpret_cast<const float4*>(p.q_scale + 16 * s + 8 * half);
      const float4 g1 = *reinterpret_cast<const float4*>(p.q_scale + 16 * s + 8 * half + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (f16)((float)qf[s][j] * inv * g[j]);
    }
  }

  const f16* kg = reinterpret_cast<const f16*>(p.k) + (size_t)b * p.k_bs + (size_t)hd * p.k_hs;
  const f16* vg = reinterpret_cast<const f16*>(p.vt) + (size_t)b * p.vt_bs + (size_t)hd * p.vt_hs;
  // staging roles (item i = tid + j NT: K key i >> 3, dims 8 (i & 7)..; V^T dim i >> 3, keys 8 (i & 7)..), padding rules and the V^T key
  // order inside a 16-key group: those of attention_kernel_w8
  const int Jpad = (p.J + 31) & ~31;
  const int ntiles = (p.J + KT2 - 1) / KT2;
  // two register sets: a tile is requested TWO tile steps before it is written to its slot (one step is ~700 cycles of work, an L2 round
  // trip under load 1-2k: with one set every tile step ended waiting for its own request)
  struct Stage { uint4 k0, v0, k1, v1; };   // (named members, not arrays: indexed through the lambdas below an array lands in scratch)
  Stage st0, st1;
  auto load_item = [&](int kt0, int i, uint4& kd, uint4& vd) __attribute__((always_inline)) {
    const int key = i >> 3, g8 = i & 7;
    const int kk = kt0 + key < Jpad ? kt0 + key : kt0;
    const int kv = kt0 + g8 * 8 < Jpad ? kt0 + g8 * 8 : kt0;
    kd = *reinterpret_cast<const uint4*>(kg + (size_t)kk * p.k_rs + g8 * 8);
    vd = *reinterpret_cast<const uint4*>(vg + (size_t)key * p.vt_ds + kv);
  };
  auto store_item = [&](char* buf, int i, const uint4& kd, const uint4& vd) __attribute__((always_inline)) {
    const int key = i >> 3, g8 = i & 7;
    *reinterpret_cast<uint4*>(buf + key * KSTR + g8 * 16) = kd;
    char* vrow = buf + KBYTES2 + key * VSTR2 + (g8 >> 1) * 32 + (g8 & 1) * 8;
    *reinterpret_cast<uint2*>(vrow) = make_uint2(vd.x, vd.y);
    *reinterpret_cast<uint2*>(vrow + 16) = make_uint2(vd.z, vd.w);
  };
  auto one_load = [&](int t, Stage& st) __attribute__((always_inline)) {   // (tiles past the end re-read the last one: stored to a dead slot, never multiplied)
    const int kt0 = (t < ntiles ? t : ntiles - 1) * KT2;
    load_item(kt0, tid, st.k0, st.v0);
    if constexpr (IT > 1) load_item(kt0, tid + NT, st.k1, st.v1);
  };
  auto one_store = [&](char* buf, const Stage& st) __attribute__((always_inline)) {
    store_item(buf, tid, st.k0, st.v0);
    if constexpr (IT > 1) store_item(buf, tid + NT, st.k1, st.v1);
  };
  auto slot = [&](int t) __attribute__((always_inline)) { return smem + (t & (BND_RING - 1)) * SLOT; };
  // this lane's fragment rows inside a slot: K row (32 hh + l31), dims 8 half..; V^T row l31 (+ 32 db), keys 32 hh + 8 half..
  const int koff = l31 * KSTR + 16 * half, voff = KBYTES2 + l31 * VSTR2 + 16 * half;
  auto krow = [&](int t, int hh) __attribute__((always_inline)) { return slot(t) + koff + hh * 32 * KSTR; };
  auto vrow = [&](int t, int hh) __attribute__((always_inline)) { return slot(t) + voff + hh * 64; };
  auto mask_last = [&](f32x16& S, int t, int hh) __attribute__((always_inline)) {   // ragged last tile: keys >= J weigh nothing
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = t * KT2 + 32 * hh + 4 * half + (r & 3) + 8 * (r >> 2);
      if (key >= p.J) S[r] = -1.0e30f;
    }
  };

  f32x16 oacc[2], S0, S1;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  f16x8 pA[2], pB[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) pA[s][j] = pB[s][j] = (f16)0.f;
  float sum0 = 0.f, sum1 = 0.f;
  AttFrag fA, fB;

  {   // tiles 0-2 -> their slots (all three requested before the first is written: one round trip, not three), tile 3 left in flight
    Stage st2;
    one_load(0, st0);
    one_load(1, st1);
    one_load(2, st2);
    one_store(slot(0), st0);
    one_store(slot(1), st1);
    one_store(slot(2), st2);
    one_load(3, st1);   // (written to its slot at the end of tile step 0)
  }
  __syncthreads();
  // half steps: A(t) = S^T(t, 1) | exp2 S^T(t, 0) | O^T += V^T P(t - 1, 1);   B(t) = S^T(t + 1, 0) | exp2 S^T(t, 1) | O^T += V^T P(t, 0)
  // fragment sets: A steps multiply fA and read fB for the B step behind them, B steps the other way round
  {   // S^T(0, 0), then A(0) without a predecessor
#pragma unroll
    for (int r = 0; r < 16; ++r) S0[r] = 0.f;
    att_frag_load(fA, krow(0, 0), vrow(0, 0));
#pragma unroll
    for (int s = 0; s < 4; ++s) S0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fA.k[s], qf[s], S0, 0, 0, 0);
    if (ntiles == 1) mask_last(S0, 0, 0);
    att_frag_load(fA, krow(0, 1), vrow(0, 0));
    att_bnd_step<true, false>(S1, S0, pA, pB, oacc, sum0, sum1, qf, fA, fB, krow(ntiles > 1 ? 1 : 0, 0), vrow(0, 0));
    if (ntiles == 1) mask_last(S1, 0, 1);
  }
  // B(t), A(t + 1): slots t and t + 1 are multiplied, slots t + 1 and t + 2 read for the steps behind, tile t + 3 arrives.  The iteration
  // that computes the LAST tile's S^T is a separate copy with the ragged-tile mask applied unconditionally: a branch around the mask lets
  // the compiler hoist the exponentials into it
  auto iter = [&](int t, auto last_c, auto par_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    constexpr int PAR = decltype(par_c)::value;     // = t & 1: tile t + 4 is requested into set PAR, tile t + 3 written from set PAR ^ 1
    if constexpr (!LAST) {
      if constexpr (PAR == 0) one_load(t + 4, st0); else one_load(t + 4, st1);
    }
    att_bnd_step<true, true>(S0, S1, pB, pA, oacc, sum0, sum1, qf, fB, fA, krow(t + 1, 1), vrow(t, 1));
    if constexpr (LAST) mask_last(S0, t + 1, 0);
    att_bnd_step<true, true>(S1, S0, pA, pB, oacc, sum0, sum1, qf, fA, fB, krow(LAST ? t + 1 : t + 2, 0), vrow(t + 1, 0));
    if constexpr (LAST) mask_last(S1, t + 1, 1);
    if constexpr (!LAST) {
      if constexpr (PAR == 0) one_store(slot(t + 3), st1); else one_store(slot(t + 3), st0);
    }
    __syncthreads();
  };
  for (int t = 0; t + 2 < ntiles; t += 2) {
    iter(t, std::false_type{}, std::integral_constant<int, 0>{});
    if (t + 3 < ntiles) iter(t + 1, std::false_type{}, std::integral_constant<int, 1>{});
  }
  if (ntiles > 1) iter(ntiles - 2, std::true_type{}, std::integral_constant<int, 0>{});
  {   // B(last) without a successor, then O^T += V^T P(last, 1)
    att_bnd_step<false, true>(S0, S1, pB, pA, oacc, sum0, sum1, qf, fB, fA, krow(ntiles - 1, 1), vrow(ntiles - 1, 1));
#pragma unroll
    for (int k = 0; k < 4; ++k) oacc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fA.v[k], pB[k >> 1], oacc[k & 1], 0, 0, 0);
  }

  float l_run = sum0 + sum1;
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (row < p.rows) {
    f16* o = reinterpret_cast<f16*>(p.o) + (size_t)b * p.o_bs + (size_t)hd * p.o_hs + (size_t)row * p.o_rs;
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

}  // namespace

int launch_attention(const ImagenAttentionParams* p, hipStream_t s) {
  IMAGEN_CHECK(p->rows > 0 && p->J > 0 && p->B > 0 && p->heads > 0, "attention: empty problem");
  IMAGEN_CHECK(p->q_rs % 8 == 0 && p->k_rs % 8 == 0 && p->vt_ds % 8 == 0 && p->o_rs % 4 == 0,
               "attention: strides must keep 16B alignment");
  IMAGEN_CHECK(p->head_dim == 0 || p->head_dim == 64 || p->head_dim == 32, "attention: head_dim %d (64 or 32)", p->head_dim);
  IMAGEN_CHECK(p->softmax_mode == 0 || p->softmax_mode == 1, "attention: softmax_mode %d", p->softmax_mode);
  if (p->head_dim != 32 && p->rows >= 256) {
    dim3 grid((p->rows + 255) / 256, p->heads, p->B);
    auto launch = [&](auto kern, int lds, int threads = 512) {
      static bool attr_done[16] = {};   // (per kernel instantiation: one lambda instantiation per `kern` type; per device)
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (dev < 0 || dev >= 16 || !attr_done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
      }
      hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, *p);
    };
    constexpr int slot = KBYTES2 + VBYTES2;
    if (p->softmax_mode == 1) {
      IMAGEN_CHECK(p->logit_bound > 0.0f && p->logit_bound <= 14.0f, "attention: softmax_mode 1 needs a logit bound in (0, 14] log2 units (got %g): use softmax_mode 0", (double)p->logit_bound);
      grid = dim3((p->rows + 127) / 128, p->heads, p->B);
      launch(attention_kernel_bnd<4>, BND_RING * slot, 256);
      return imagen_hip_status("attention");
    }
    if (p->J <= 2 * KT2) launch(attention_kernel_w8<4, 1>, 2 * slot);   // one or two tiles: nothing to gain from staging two at a time
    else launch(attention_kernel_w8<4, 2>, 4 * slot);
    return imagen_hip_status("attention");
  }
  dim3 grid((p->rows + 127) / 128, p->heads, p->B);
  if (p->head_dim == 32) hipLaunchKernelGGL(attention_kernel<32>, grid, dim3(256), 0, s, *p);
  else hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(256), 0, s, *p);
  return imagen_hip_status("attention");
}
