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
#include <cstdio>
#include "common.h"

namespace {

constexpr int kThreads = 512;
constexpr int kDh = 64;
#ifndef ROWCHAIN_RING
#define ROWCHAIN_RING 8       // A fragments in flight per wave.  Call C (profiles/r05_c_chain_bench_variants.jsonl: every chain shape of the benchmark, 4 / 8 / 16
                              // deep, one or two workgroups per CU, the round's first kernel beside them): 8 is best or within 1 us of the best on 17 of
                              // 20 shapes; 16 costs the 16384-row launches 20-30 % (registers), 4 the small grids 5-15 % (tools/chain_bench.py builds the variants)
#endif
#ifndef ROWCHAIN_MINW
#define ROWCHAIN_MINW 1       // minimum waves per SIMD the register allocation leaves room for (4: two 512-thread workgroups per CU)
#endif
constexpr int kRing = ROWCHAIN_RING;
#ifndef ROWCHAIN_PHASES
#define ROWCHAIN_PHASES 4
#endif
constexpr int kPhases = ROWCHAIN_PHASES;

// -DROWCHAIN_TRACE (tools/chain_bench.py --trace, a throw-away variant library: never in the product build): thread 0 of every workgroup
// stamps s_memtime at the phase boundaries into a buffer handed over by imagen_debug_rowchain_trace()
#ifdef ROWCHAIN_TRACE
__device__ unsigned long long* g_rowchain_trace = nullptr;
#define RC_STAMP(i)                                                                                                  \
  do {                                                                                                               \
    if (threadIdx.x == 0 && g_rowchain_trace) g_rowchain_trace[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define RC_STAMP(i) ((void)0)
#endif

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
  } else if (p.mode == IMAGEN_CHAIN_QKV) {
    g.p0_cols = p.inner + 2 * kDh;
    g.p1_cols = p.C;
    g.p2_cols = 0;
  } else {   // RESPREP: the concatenated input rows | the output rows (+ gate and bias of the image behind them)
    g.p0_cols = p.inner + p.C2;
    g.p1_cols = p.C;
    g.p2_cols = 0;
  }
  return g;
}

// behind the row tiles: the per-channel gain vectors of the chain (fp32: g0 | g1 | g2 | q_scale or k_scale) and, for the cross-attention, every
// wave's private K^ / V^T staging of one 32-key tile of its head
constexpr int kGainG1 = 256, kGainG2 = 512, kGainQs = 1024, kGainFloats = 1088;
constexpr int kKPitch = 136, kVPitch = 72;                 // bytes per staged K^ row (64 dims) / V^T row (32 keys): conflict-free ds_read_b64
constexpr int kKvWave = 32 * kKPitch + 64 * kVPitch;       // 8960 bytes per wave

__host__ __device__ inline size_t chain_tiles_bytes(const ImagenRowchainParams& p, int rows) {
  const Geo g = chain_geo(p);
  return (size_t)rows * Geo::pitch(g.p0_cols) + (size_t)rows * Geo::pitch(g.p1_cols) + (g.p2_cols ? (size_t)rows * Geo::pitch(g.p2_cols) : 0);
}

__host__ __device__ inline size_t chain_lds_bytes(const ImagenRowchainParams& p, int rows) {
  size_t n = chain_tiles_bytes(p, rows);
  if (p.mode != IMAGEN_CHAIN_RESPREP) n += kGainFloats * sizeof(float);
  if (p.mode == IMAGEN_CHAIN_XATTN) n += (size_t)8 * kKvWave;
  if (p.mode == IMAGEN_CHAIN_RESPREP) {
    n += (size_t)2 * p.C * sizeof(float);
    const int T = p.C >> 5, ks = (p.inner + p.C2) >> 4;
    if (T < 8) {   // the K-split partials of the narrow layers go through the row tiles (dead by then): room for (wk - 1) * T tiles of 4 KB per row block
      int wk = 8 / T;
      if (wk > ks) wk = ks;
      const size_t need = (size_t)(wk - 1) * T * (rows / 32) * 4096;
      if (need > n) n = need;
    }
  }
  return (n + 15) & ~(size_t)15;
}

// how a GEMM stage with T cout tiles of 32 and `ksteps` K = 16 steps is dealt to the 8 waves
struct Part {
  int tile0, nt;        // this wave's cout tiles [tile0, tile0 + nt)
  int wk, kpart;        // K split: wk parts, this wave's part (kpart >= wk: idle)
  int s0, s1;           // this wave's K steps [s0, s1)
  int T;
  int rot;              // the wave walks them as s0 + (i + rot) mod (s1 - s0): see make_part
};

// rotation of a wave's walk over n K steps: the row tile's phase (0 .. kPhases - 1) x n / kPhases; short slices walk from the head
// (call N, per launch: the small grids gain 0.8 - 1.6 us — qkv 256 x 128: 11.1 -> 9.5, FF 64 x 256: 24.8 -> 23.3 —, the 256- / 512-workgroup
// launches lose as much: their tiles are spread in time by the dispatch already.  Grids of at most 128 tiles only.)
__device__ __forceinline__ int phase_rot(int n) { return (n >= 2 * kPhases && gridDim.x <= 128u) ? (int)(blockIdx.x % kPhases) * n / kPhases : 0; }

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
  // phase-shifted walk (conv_small.hip, call M: K loop 10.3k -> 6.6k cycles): every row tile streams the same weights, in lockstep when they all
  // start at the head of the slice — each ring refill is then a cold miss for all of them.  Neighbouring tiles of a small grid start at 0, 1/4,
  // 1/2, 3/4 of it instead (wrapping): behind the first quarter a wave meets lines a neighbour pulled into L2 a quarter earlier.
  q.rot = phase_rot(q.s1 - q.s0);
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

// ---- the weight stream of a GEMM stage.  A wave's A fragments are 16-byte loads from the packed weight buffer; kRing of them (8 K steps of
// one cout tile, or 4 of two) are kept in flight in a register ring, and the ring of the NEXT stage is requested before the current stage's
// row pass and barriers (fill and run are separate calls): the weights are cold when a launch starts (last touched a whole denoiser step
// ago).  Measured (call C): the depth matters little — a chain is ~7 dependent phases of 1.5-3 us each, whatever feeds the matrix pipe.
// Addressing: a wave-uniform byte offset (cout tile, K step: scalar registers) + ONE 32-bit per-lane offset that never changes — the loads
// then take their base from SGPRs and the whole ring costs a single address VGPR (64-bit per-load addresses cost 2 x 16 of them).
struct WeightStream {
  const char* base;      // packed weight buffer (kernel argument: uniform)
  unsigned lane_off;     // (half * cout_pad + l31) * 16 bytes
  size_t tile_off;       // tile0 * 512 bytes (uniform)
  size_t step_bytes;     // 2 * cout_pad * 16 bytes per K = 16 step (uniform)
};

__device__ __forceinline__ WeightStream weight_stream(const void* w, int cout_pad, int tile0, int lane) {
  WeightStream ws;
  ws.base = reinterpret_cast<const char*>(w);
  ws.lane_off = ((unsigned)(lane >> 5) * (unsigned)cout_pad + (unsigned)(lane & 31)) * 16u;
  ws.tile_off = (size_t)tile0 * 512;
  ws.step_bytes = (size_t)cout_pad * 32;
  return ws;
}

__device__ __forceinline__ f16x8 weight_frag(const WeightStream& ws, int s, int t) {
  const char* ub = ws.base + (ws.tile_off + (size_t)s * ws.step_bytes + (size_t)t * 512);   // uniform
  return *reinterpret_cast<const f16x8*>(ub + ws.lane_off);
}

__device__ __forceinline__ int walk_step(int i, int rot, int n) {   // i < n
  const int x = i + rot;
  return x >= n ? x - n : x;
}

template <int NT>
__device__ __forceinline__ void ring_fill(f16x8 (&ring)[kRing], const WeightStream& ws, int s0, int n, int rot) {
  constexpr int D = kRing / NT;
  if (n <= 0) return;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const int s = s0 + walk_step(i < n ? i : n - 1, rot, n);
#pragma unroll
    for (int t = 0; t < NT; ++t) ring[i * NT + t] = weight_frag(ws, s, t);
  }
}

// acc[t][j] += W[cout tiles tile0 + t][K steps s0 .. s0 + n) . X[rows of row block j]: the ring holds steps s0 .. s0 + D - 1 on entry
template <int NT, int RB>
__device__ __forceinline__ void gemm_run(f32x16 (*acc)[RB], f16x8 (&ring)[kRing], const WeightStream& ws, int s0, int n, int rot, const char* xs, int pitch,
                                         int lane) {
  constexpr int D = kRing / NT;
  if (n <= 0) return;
  const char* xl = xs + (lane & 31) * pitch + 16 * (lane >> 5);
  int sb = 0;
  for (; sb + D < n; sb += D) {   // straight-line body: D steps, each followed by the request of the step D ahead (clamped to the last one)
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const int s = sb + i;
      f16x8 b[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j) b[j] = *reinterpret_cast<const f16x8*>(xl + (size_t)j * 32 * pitch + (s0 + walk_step(s, rot, n)) * 32);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < RB; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[i * NT + t], b[j], acc[t][j], 0, 0, 0);
      const int sn = s0 + walk_step(s + D < n ? s + D : n - 1, rot, n);
#pragma unroll
      for (int t = 0; t < NT; ++t) ring[i * NT + t] = weight_frag(ws, sn, t);
      __builtin_amdgcn_sched_barrier(0);   // pins the request here (the scheduler otherwise sinks look-ahead loads to their use)
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {   // the last (up to D) steps are in the ring: no further requests
    if (sb + i < n) {
      f16x8 b[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j) b[j] = *reinterpret_cast<const f16x8*>(xl + (size_t)j * 32 * pitch + (s0 + walk_step(sb + i, rot, n)) * 32);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < RB; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[i * NT + t], b[j], acc[t][j], 0, 0, 0);
    }
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

// the chain's gain vectors -> LDS (visible behind the next workgroup barrier): the row passes then read them without a global round trip
__device__ __forceinline__ float* stage_gains(const ImagenRowchainParams& p, char* smem, size_t tiles_bytes, int tid) {
  float* sg = reinterpret_cast<float*>(smem + tiles_bytes);
  const int C = p.C;
  for (int i = tid; i < kGainFloats; i += kThreads) {
    float v = 0.f;
    if (i < kGainG1) v = i < C ? p.g0[i] : 0.f;
    else if (i < kGainG2) v = (p.g1 && i - kGainG1 < C) ? p.g1[i - kGainG1] : 0.f;
    else if (i < kGainQs) v = (p.g2 && i - kGainG2 < p.hidden) ? p.g2[i - kGainG2] : 0.f;
    else v = p.q_scale ? p.q_scale[i - kGainQs] : (p.k_scale ? p.k_scale[i - kGainQs] : 0.f);
    sg[i] = v;
  }
  return sg;
}

// this thread's pieces (8 channels each) of row r = tid / LPR of a [rows][C] global tensor, C <= 256: pieces li, li + LPR, ... (at most PMAX)
template <int RB>
struct RowPieces {
  static constexpr int LPR = kThreads / (32 * RB);
  static constexpr int PMAX = 32 / LPR;
  f16x8 v[PMAX];
};

template <int RB>
__device__ __forceinline__ void load_row_pieces(RowPieces<RB>& rp, const void* base, int ld, int row, int C, int li) {
  constexpr int LPR = RowPieces<RB>::LPR;
  const f16* x = reinterpret_cast<const f16*>(base) + (size_t)row * ld;
  const int np = C >> 3;
#pragma unroll
  for (int k = 0; k < RowPieces<RB>::PMAX; ++k) {
    const int g = li + k * LPR;
    rp.v[k] = *reinterpret_cast<const f16x8*>(x + (g < np ? g : 0) * 8);   // (unconditional: a piece past the row re-reads piece 0)
  }
}

// the x rows (already in registers) -> fp16((x - mean) * rstd * g) into the LDS tile: the LayerNorm prologue of a GEMM (IGEMM's (x - mu) * rs * pa)
template <int RB>
__device__ __forceinline__ void ln_rows_to_lds(const ImagenRowchainParams& p, const RowPieces<RB>& rp, int row0, char* dst, int pitch, int tid) {
  constexpr int LPR = RowPieces<RB>::LPR, PMAX = RowPieces<RB>::PMAX;
  const int r = tid / LPR, li = tid % LPR;
  const int C = p.C, np = C >> 3;
  float4 ga[PMAX][2];     // the gain of this thread's pieces: requested with the rows, one round trip
#pragma unroll
  for (int k = 0; k < PMAX; ++k) {
    const int g = li + k * LPR < np ? li + k * LPR : 0;
    ga[k][0] = *reinterpret_cast<const float4*>(p.g0 + g * 8);
    ga[k][1] = *reinterpret_cast<const float4*>(p.g0 + g * 8 + 4);
  }
  float mean, rstd;
  if (p.mu) {
    mean = p.mu[row0 + r];
    rstd = p.rs[row0 + r];
  } else {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
      if (li + k * LPR < np)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)rp.v[k][e];
    mean = row_sum<LPR>(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
      if (li + k * LPR < np)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)rp.v[k][e] - mean;
          q += d * d;
        }
    rstd = rsqrtf(row_sum<LPR>(q) / (float)C + p.eps);
  }
  char* drow = dst + (size_t)r * pitch;
#pragma unroll
  for (int k = 0; k < PMAX; ++k) {
    const int g = li + k * LPR;
    if (g < np) {
      const float gv[8] = {ga[k][0].x, ga[k][0].y, ga[k][0].z, ga[k][0].w, ga[k][1].x, ga[k][1].y, ga[k][1].z, ga[k][1].w};
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)(((float)rp.v[k][e] - mean) * rstd * gv[e]);
      *reinterpret_cast<f16x8*>(drow + g * 16) = o;
    }
  }
}

// y (fp16 LDS tile) -> out = fp16(LN(y) * g + res) -> global rows (16-byte pieces) + the row's sum of squares: LN_RESIDUAL with ssq_out;
// res: this thread's pieces of the residual rows, requested long before (RowPieces)
template <int RB>
__device__ __forceinline__ void ln_res_out_rows(const ImagenRowchainParams& p, int row0, const char* src, int pitch, const float* gain,
                                                const RowPieces<RB>& res, int tid) {
  constexpr int LPR = RowPieces<RB>::LPR, PMAX = RowPieces<RB>::PMAX;
  const int r = tid / LPR, li = tid % LPR;
  const int C = p.C, np = C >> 3;
  const char* srow = src + (size_t)r * pitch;
  float mean, rstd;
  row_ln_stats<LPR>(srow, C, li, p.eps, mean, rstd);
  f16* out = reinterpret_cast<f16*>(p.out) + (size_t)(row0 + r) * p.ld_out;
  float ssq = 0.f;
#pragma unroll
  for (int k = 0; k < PMAX; ++k) {
    const int g = li + k * LPR;
    if (g < np) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(srow + g * 16);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (f16)(((float)v[e] - mean) * rstd * gain[g * 8 + e] + (float)res.v[k][e]);
        const float t = (float)o[e];
        ssq += t * t;
      }
      *reinterpret_cast<f16x8*>(out + g * 8) = o;
    }
  }
  ssq = row_sum<LPR>(ssq);
  if (p.ssq_out && li == 0) p.ssq_out[row0 + r] = ssq;
}

// one GEMM stage: the request of its first ring-full (stage_fill: as early as the caller can place it) and the K loop + K-split reduction
// (stage_run); the owners' accumulators stay in acc (first q.nt tiles)
__device__ __forceinline__ void stage_fill(f16x8 (&ring)[kRing], const Part& q, const void* w, int cout_pad, int lane) {
  const WeightStream ws = weight_stream(w, cout_pad, q.tile0, lane);
  if (q.nt == 2) ring_fill<2>(ring, ws, q.s0, q.s1 - q.s0, q.rot);
  else ring_fill<1>(ring, ws, q.s0, q.s1 - q.s0, q.rot);
}

template <int RB>
__device__ __forceinline__ void stage_run(f32x16 (&acc)[2][RB], f16x8 (&ring)[kRing], const Part& q, const void* w, int cout_pad, const char* xs, int pitch,
                                          char* scratch, int lane) {
  const WeightStream ws = weight_stream(w, cout_pad, q.tile0, lane);
  acc_zero<2, RB>(acc);
  if (q.nt == 2) gemm_run<2, RB>(acc, ring, ws, q.s0, q.s1 - q.s0, q.rot, xs, pitch, lane);
  else gemm_run<1, RB>(acc, ring, ws, q.s0, q.s1 - q.s0, q.rot, xs, pitch, lane);
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
  constexpr int ROWS = 32 * RB, LPR = RowPieces<RB>::LPR, PMAX = RowPieces<RB>::PMAX;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo geo = chain_geo(p);
  const int C = p.C, inner = p.inner, hidden = p.hidden;
  const int pitch0o = Geo::pitch(inner), pitch0h = Geo::pitch(hidden), pitch1 = Geo::pitch(C);
  char* P0 = smem;
  char* P1 = P0 + (size_t)ROWS * Geo::pitch(geo.p0_cols);
  char* P2 = P1 + (size_t)ROWS * pitch1;
  const int r = tid / LPR, li = tid % LPR;
  f16x8 ring[kRing];
  f32x16 acc[2][RB];
  // ---- every request that depends on nothing: the out-projection's first weights, the residual rows, the o rows
  RC_STAMP(0);
  const Part q0 = make_part(C >> 5, inner >> 4, wave);
  stage_fill(ring, q0, p.w0, p.w_cout_pad0, lane);
  RowPieces<RB> res;
  load_row_pieces<RB>(res, p.res, p.ld_res, row0 + r, C, li);
  {   // the o rows: every piece of this thread requested before the first is stored (inner == 512: 4 RB pieces per thread)
    constexpr int NP = 4 * RB;
    const int npr = inner >> 3;
    const f16* x = reinterpret_cast<const f16*>(p.x);
    uint4 t[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int i = tid + k * kThreads, rr = i / npr, g = i - rr * npr;
      t[k] = *reinterpret_cast<const uint4*>(x + (size_t)(row0 + rr) * p.ld_x + g * 8);
    }
    const float* sg0 = stage_gains(p, smem, chain_tiles_bytes(p, ROWS), tid);
    (void)sg0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int i = tid + k * kThreads, rr = i / npr, g = i - rr * npr;
      *reinterpret_cast<uint4*>(P0 + (size_t)rr * pitch0o + g * 16) = t[k];
    }
  }
  const float* sg = reinterpret_cast<const float*>(smem + chain_tiles_bytes(p, ROWS));
  __syncthreads();
  RC_STAMP(1);
  // ---- y = o W_out^T -> P1 (fp16)
  stage_run<RB>(acc, ring, q0, p.w0, p.w_cout_pad0, P0, pitch0o, P0, lane);
  RC_STAMP(2);
  store_stage<IMAGEN_ACT_NONE, RB>(acc, q0, P1, pitch1, lane);
  const Part q1 = make_part(hidden >> 5, C >> 4, wave);
  stage_fill(ring, q1, p.w1, p.w_cout_pad1, lane);          // (in flight across the barrier and the row pass)
  __syncthreads();
  RC_STAMP(3);
  // ---- row pass: x1 = fp16(LN(y) * g0 + res) -> P2;  a0 = fp16((x1 - mean x1) * rstd x1 * g1) -> P1
  {
    const int np = C >> 3;
    char* yrow = P1 + (size_t)r * pitch1;
    char* xrow = P2 + (size_t)r * pitch1;
    float mean, rstd;
    row_ln_stats<LPR>(yrow, C, li, p.eps, mean, rstd);
    f16x8 x1[PMAX];
    float so = 0.f;
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
      const int g = li + k * LPR;
      if (g < np) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(yrow + g * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          x1[k][e] = (f16)(((float)v[e] - mean) * rstd * sg[g * 8 + e] + (float)res.v[k][e]);
          so += (float)x1[k][e];
        }
        *reinterpret_cast<f16x8*>(xrow + g * 16) = x1[k];
      }
    }
    const float mo = row_sum<LPR>(so) / (float)C;
    float qo = 0.f;
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
      if (li + k * LPR < np)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)x1[k][e] - mo;
          qo += d * d;
        }
    const float ro = rsqrtf(row_sum<LPR>(qo) / (float)C + p.eps);
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
      const int g = li + k * LPR;
      if (g < np) {
        f16x8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (f16)(((float)x1[k][e] - mo) * ro * sg[kGainG1 + g * 8 + e]);
        *reinterpret_cast<f16x8*>(yrow + g * 16) = a;
      }
    }
  }
  __syncthreads();
  RC_STAMP(4);
  // ---- hid = fp16(gelu(a0 W1^T)) -> P0
  stage_run<RB>(acc, ring, q1, p.w1, p.w_cout_pad1, P1, pitch1, P0, lane);
  RC_STAMP(5);
  store_stage<IMAGEN_ACT_GELU, RB>(acc, q1, P0, pitch0h, lane);
  const Part q2 = make_part(C >> 5, hidden >> 4, wave);
  stage_fill(ring, q2, p.w2, p.w_cout_pad2, lane);
  __syncthreads();
  RC_STAMP(6);
  // ---- row pass: a1 = fp16((hid - mean) * rstd * g2), in place
  {
    const int np = hidden >> 3;
    char* hrow = P0 + (size_t)r * pitch0h;
    float mean, rstd;
    row_ln_stats<LPR>(hrow, hidden, li, p.eps, mean, rstd);
    for (int g = li; g < np; g += LPR) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(hrow + g * 16);
      f16x8 a;
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = (f16)(((float)v[e] - mean) * rstd * sg[kGainG2 + g * 8 + e]);
      *reinterpret_cast<f16x8*>(hrow + g * 16) = a;
    }
  }
  __syncthreads();
  RC_STAMP(7);
  // ---- out = fp16(a1 W2^T + x1) -> P1 -> global
  stage_run<RB>(acc, ring, q2, p.w2, p.w_cout_pad2, P0, pitch0h, P0, lane);
  RC_STAMP(8);
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
  __syncthreads();
  RC_STAMP(9);
  {
    const int np = C >> 3;
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
  RC_STAMP(10);
}

// ------------------------------------------------------------------------------------------------ mode 2: XATTN
// The K^ / V^T fragments of ONE 32-key tile of head hd of image b, in the permuted dim / key orders of the two contractions below: A fragments
// built from two 8-byte pieces each (16 requests per tile, issued together).
struct KvTile {
  f16x8 k[2][2];   // [32-dim block t][K step s]: key 32 kt + l31, dims 32 t + 16 s + 4 half + {0..3, 8..11}
  f16x8 v[2][2];   // [K step s][32-dim block db]: dim 32 db + l31, keys 32 kt + 16 s + 4 half + {0..3, 8..11}
};

// The tile travels global -> registers (16-byte pieces, consecutive lanes on consecutive addresses) -> the wave's private LDS staging -> fragments
// (two ds_read_b64 each).  Round 5, call D (profiles/r05_d_rowchain_phase_timeline.json): with the fragments taken straight from global memory
// — 8-byte pieces of 32 different rows per load instruction — the attention of a 32-row block took 17.6k cycles, a third of the launch: the
// CU's address coalescer serves one cache line per cycle, and 8 waves x 32 instructions x 32 lines is 16k of them.
struct KvRegs { uint4 k[4], v[4]; };

__device__ __forceinline__ void kv_request(KvRegs& r, const ImagenRowchainParams& p, int b, int hd, int kt, int lane) {
  const f16* kg = reinterpret_cast<const f16*>(p.khat) + (size_t)b * p.k_bs + (size_t)hd * p.k_hs + (size_t)(32 * kt) * p.k_rs;
  const f16* vg = reinterpret_cast<const f16*>(p.vt) + (size_t)b * p.vt_bs + (size_t)hd * p.vt_hs + 32 * kt;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    r.k[i] = *reinterpret_cast<const uint4*>(kg + (size_t)(c >> 3) * p.k_rs + (c & 7) * 8);      // key c / 8, dims 8 (c % 8) ..
    r.v[i] = *reinterpret_cast<const uint4*>(vg + (size_t)(c >> 2) * p.vt_ds + (c & 3) * 8);     // dim c / 4, keys 8 (c % 4) ..
  }
}

__device__ __forceinline__ void kv_stage(const KvRegs& r, char* kvs, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    uint2* kd = reinterpret_cast<uint2*>(kvs + (c >> 3) * kKPitch + (c & 7) * 16);
    kd[0] = make_uint2(r.k[i].x, r.k[i].y);
    kd[1] = make_uint2(r.k[i].z, r.k[i].w);
    uint2* vd = reinterpret_cast<uint2*>(kvs + 32 * kKPitch + (c >> 2) * kVPitch + (c & 3) * 16);
    vd[0] = make_uint2(r.v[i].x, r.v[i].y);
    vd[1] = make_uint2(r.v[i].z, r.v[i].w);
  }
}

__device__ __forceinline__ f16x8 two_pieces(const char* ptr) {
  const uint2 lo = *reinterpret_cast<const uint2*>(ptr);
  const uint2 hi = *reinterpret_cast<const uint2*>(ptr + 16);
  uint4 pk = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return *reinterpret_cast<const f16x8*>(&pk);
}

__device__ __forceinline__ void kv_frags(KvTile& kv, const char* kvs, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const char* krow = kvs + l31 * kKPitch + 8 * half;
  const char* vbase = kvs + 32 * kKPitch + 8 * half;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) kv.k[t][s] = two_pieces(krow + (32 * t + 16 * s) * 2);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int db = 0; db < 2; ++db) kv.v[s][db] = two_pieces(vbase + (32 * db + l31) * kVPitch + 32 * s);
}

// Q^ of this lane's row from the wave's two q accumulator tiles (dims 0-31 | 32-63 of the head): fp16(q) -> l2norm * q_scale * q_mult
// (ATTENTION's fused QNORM), as B fragments in the accumulator's own dim order: fragment (t, s), element e <-> dim 32 t + 16 s + 4 half + (e & 3) + 8 (e >> 2)
__device__ __forceinline__ void make_qhat(f16x8 (&qf)[2][2], const ImagenRowchainParams& p, const float* s_qscale, const f32x16& q0, const f32x16& q1, int lane) {
  const int half = lane >> 5;
  f16 q16[2][16];
  float ssq = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    q16[0][r] = (f16)q0[r];
    q16[1][r] = (f16)q1[r];
    ssq += (float)q16[0][r] * (float)q16[0][r] + (float)q16[1][r] * (float)q16[1][r];
  }
  ssq += __shfl_xor(ssq, 32);
  const float inv = p.q_mult / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
      qf[t][r >> 3][r & 7] = (f16)((float)q16[t][r] * inv * s_qscale[d]);
    }
}

struct Softmax {   // attention_kernel's online softmax state of one row (lane = row; the two half-waves hold the two halves of a key tile)
  f32x16 oacc[2];
  float m_run, l_run;
};

__device__ __forceinline__ void attn_tile(Softmax& st, const KvTile& kv, const f16x8 (&qf)[2][2], int kt, int J, int lane) {
  const int half = lane >> 5;
  f32x16 sacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kv.k[t][s], qf[t][s], sacc, 0, 0, 0);
  const int kbase = 32 * kt + 4 * half;
  float mx = -1.0e30f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = kbase + (r & 3) + 8 * (r >> 2);
    if (key >= J) sacc[r] = -1.0e30f;
    mx = fmaxf(mx, sacc[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float m_new = fmaxf(st.m_run, mx);
  const float alpha = exp2f(st.m_run - m_new);
  st.m_run = m_new;
  float psum = 0.f;
  // The softmax weights enter the PV product as fp16 hi + lo pairs (round 6): the contract keeps P in fp32, and a bare fp16 P was the one
  // rounding of this chain the contract does not have — 3.4-3.8e-4 per chain launch against the interpreter from identical inputs
  // (profiles/r05_t_op_audit_c5_null_row.txt), every other launch of the denoiser at ~5e-6.  Four more MFMAs per 32-key tile.
  f16x8 pf[2], pl[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float e = exp2f(sacc[r] - m_new);
    psum += e;
    const f16 eh = (f16)e;
    pf[r >> 3][r & 7] = eh;
#ifndef ROWCHAIN_P16
    pl[r >> 3][r & 7] = (f16)(e - (float)eh);
#endif
  }
  st.l_run = st.l_run * alpha + psum;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) st.oacc[db][r] *= alpha;
#pragma unroll
  for (int s = 0; s < 2; ++s)     // O^T[d][row] += V^T . P (k-step s covers the keys of accumulator registers 8 s .. 8 s + 7)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#ifndef ROWCHAIN_P16
      st.oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kv.v[s][db], pl[s], st.oacc[db], 0, 0, 0);
#endif
      st.oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kv.v[s][db], pf[s], st.oacc[db], 0, 0, 0);
    }
}

// o[row][hd * 64 + 32 db + 8 g + 4 half + e] -> the P0 row tile (the B operand of the out-projection)
__device__ __forceinline__ void store_o(const Softmax& st, int hd, char* orow, int lane) {
  const int half = lane >> 5;
  const float l_tot = st.l_run + __shfl_xor(st.l_run, 32);
  const float il = 1.0f / l_tot;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (f16)(st.oacc[db][4 * g + e] * il);
      *reinterpret_cast<f16x4*>(orow + (hd * 64 + 32 * db + 8 * g + 4 * half) * 2) = v;
    }
}

template <int RB>
__device__ __forceinline__ void chain_xattn(const ImagenRowchainParams& p, char* smem, int row0) {
  constexpr int ROWS = 32 * RB, LPR = RowPieces<RB>::LPR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31;
  const int C = p.C, inner = p.inner;
  const int pitch0 = Geo::pitch(inner), pitch1 = Geo::pitch(C);
  char* P0 = smem;
  char* P1 = P0 + (size_t)ROWS * pitch0;
  const int r = tid / LPR, li = tid % LPR;
  f16x8 ring[kRing];
  f32x16 acc[2][RB];
  // ---- q = a Wq^T: wave h owns the 64 output channels of head h (two cout tiles), all K steps; its first weights are requested first
  Part qq;
  qq.T = inner >> 5;
  qq.nt = 2;
  qq.tile0 = 2 * wave;
  qq.wk = 1;
  qq.kpart = 0;
  qq.s0 = 0;
  qq.s1 = C >> 4;
  qq.rot = phase_rot(qq.s1);
  RC_STAMP(0);
  stage_fill(ring, qq, p.w0, p.w_cout_pad0, lane);
  {
    RowPieces<RB> xr;   // the block input rows: the LayerNorm input
    load_row_pieces<RB>(xr, p.x, p.ld_x, row0 + r, C, li);
    ln_rows_to_lds<RB>(p, xr, row0, P1, pitch1, tid);
  }
  const float* sg = stage_gains(p, smem, chain_tiles_bytes(p, ROWS), tid);
  RC_STAMP(1);
  __syncthreads();
  RC_STAMP(2);
  {
    const WeightStream ws = weight_stream(p.w0, p.w_cout_pad0, qq.tile0, lane);
    acc_zero<2, RB>(acc);
    gemm_run<2, RB>(acc, ring, ws, 0, qq.s1, qq.rot, P1, pitch1, lane);
  }
  RC_STAMP(3);
  // ---- the attention of head `wave`, wave-local.  Q^ of every row block first (the fp32 accumulators are dead after it); the key tiles are
  // requested before that arithmetic
  const int b = row0 / p.rows_per_batch;
  const int ntiles = (p.J + 31) >> 5;
  char* kvs = smem + chain_tiles_bytes(p, ROWS) + kGainFloats * sizeof(float) + (size_t)wave * kKvWave;
  KvRegs kr0, kr1;      // the first two key tiles (every README / BASELINE site: J = 39 | 41) in ONE round trip
  kv_request(kr0, p, b, wave, 0, lane);
  kv_request(kr1, p, b, wave, ntiles > 1 ? 1 : 0, lane);
  f16x8 qf[RB][2][2];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    make_qhat(qf[j], p, sg + kGainQs, acc[0][j], acc[1][j], lane);
    __builtin_amdgcn_sched_barrier(0);
  }
  const Part q1 = make_part(C >> 5, inner >> 4, wave);
  auto stage = [&](const KvRegs& r) __attribute__((always_inline)) {
    // the staging is wave-private: the wave's earlier fragment reads precede these stores, the stores precede the reads behind them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    kv_stage(r, kvs, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    Softmax st;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) st.oacc[db][e] = 0.f;
    st.m_run = -1.0e30f;
    st.l_run = 0.f;
    KvTile kv;
    stage(kr0);
    kv_frags(kv, kvs, lane);
    attn_tile(st, kv, qf[j], 0, p.J, lane);
    if (ntiles > 1) {
      stage(kr1);
      kv_frags(kv, kvs, lane);
      attn_tile(st, kv, qf[j], 1, p.J, lane);
    }
    for (int kt = 2; kt < ntiles; ++kt) {   // (long contexts: tile after tile, through a third register set)
      KvRegs kr;
      kv_request(kr, p, b, wave, kt, lane);
      stage(kr);
      kv_frags(kv, kvs, lane);
      attn_tile(st, kv, qf[j], kt, p.J, lane);
    }
    store_o(st, wave, P0 + (size_t)(32 * j + l31) * pitch0, lane);
    __builtin_amdgcn_sched_barrier(0);   // (one row block after the other: interleaved, their softmax states do not fit the register file)
  }
  __builtin_amdgcn_sched_barrier(0);   // (the out-projection's ring is 64 registers: requested here, not hoisted across the attention)
  stage_fill(ring, q1, p.w1, p.w_cout_pad1, lane);
  RowPieces<RB> xr;     // the residual rows of the last row pass (the block input again unless the caller names another tensor), requested
  if (p.res) load_row_pieces<RB>(xr, p.res, p.ld_res, row0 + r, C, li);   // behind the attention: 16 registers it could not spare
  else load_row_pieces<RB>(xr, p.x, p.ld_x, row0 + r, C, li);
  RC_STAMP(4);
  __syncthreads();
  RC_STAMP(5);
  // ---- y = o W_out^T -> P1 (the normalised input rows are dead)
  stage_run<RB>(acc, ring, q1, p.w1, p.w_cout_pad1, P0, pitch0, P0, lane);
  RC_STAMP(6);
  store_stage<IMAGEN_ACT_NONE, RB>(acc, q1, P1, pitch1, lane);
  __syncthreads();
  RC_STAMP(7);
  // ---- out = fp16(LN(y) * g1 + x)
  ln_res_out_rows<RB>(p, row0, P1, pitch1, sg + kGainG1, xr, tid);
  RC_STAMP(8);
}

// ------------------------------------------------------------------------------------------------ mode 3: QKV
template <int RB>
__device__ __forceinline__ void chain_qkv(const ImagenRowchainParams& p, char* smem, int row0) {
  constexpr int ROWS = 32 * RB, LPR = RowPieces<RB>::LPR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = p.C, inner = p.inner, nout = inner + 2 * kDh;
  const int pitch0 = Geo::pitch(nout), pitch1 = Geo::pitch(C);
  char* P0 = smem;
  char* P1 = P0 + (size_t)ROWS * pitch0;
  const int r = tid / LPR, li = tid % LPR;
  // ---- y = a [Wq | Wkv]^T: 20 cout tiles — every wave two (q head `wave`), waves 0-3 one of the k | v tiles on top
  f16x8 ring[kRing];
  WeightStream ws = weight_stream(p.w0, p.w_cout_pad0, 2 * wave, lane);
  const int rotq = phase_rot(C >> 4);
  ring_fill<2>(ring, ws, 0, C >> 4, rotq);
  {
    RowPieces<RB> xr;
    load_row_pieces<RB>(xr, p.x, p.ld_x, row0 + r, C, li);
    ln_rows_to_lds<RB>(p, xr, row0, P1, pitch1, tid);
  }
  __syncthreads();
  f32x16 acc[2][RB];
  acc_zero<2, RB>(acc);
  gemm_run<2, RB>(acc, ring, ws, 0, C >> 4, rotq, P1, pitch1, lane);
  if (wave < 4) {   // (requested before the first two tiles are stored)
    ws = weight_stream(p.w0, p.w_cout_pad0, 16 + wave, lane);
    ring_fill<1>(ring, ws, 0, C >> 4, rotq);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < RB; ++j) store_tile<IMAGEN_ACT_NONE>(acc[t][j], P0, pitch0, 2 * wave + t, j, lane);
  if (wave < 4) {
    acc_zero<1, RB>(acc);
    gemm_run<1, RB>(acc, ring, ws, 0, C >> 4, rotq, P1, pitch1, lane);
#pragma unroll
    for (int j = 0; j < RB; ++j) store_tile<IMAGEN_ACT_NONE>(acc[0][j], P0, pitch0, 16 + wave, j, lane);
  }
  __syncthreads();
  // ---- row pass: q pieces -> out rows; K^ = l2norm(k) * k_scale -> khat row; v -> V^T column
  {
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

// ------------------------------------------------------------------------------------------------ mode 4: RESPREP
template <int RB>
__device__ __forceinline__ void chain_resprep(const ImagenRowchainParams& p, char* smem, int row0) {
  constexpr int ROWS = 32 * RB, LPR = RowPieces<RB>::LPR, PMAX = RowPieces<RB>::PMAX;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = p.C, C1 = p.inner, K = C1 + p.C2;
  const int pitch0 = Geo::pitch(K), pitch1 = Geo::pitch(C);
  char* P0 = smem;
  char* P1 = P0 + (size_t)ROWS * pitch0;
  float* s_gate = reinterpret_cast<float*>(P1 + (size_t)ROWS * pitch1);
  float* s_bias = s_gate + C;
  const int r = tid / LPR, li = tid % LPR;
  const int b = row0 / p.rows_per_batch;
  f16x8 ring[kRing];
  f32x16 acc[2][RB];
  const Part q0 = make_part(C >> 5, K >> 4, wave);
  stage_fill(ring, q0, p.w0, p.w_cout_pad0, lane);
  // ---- the requests that depend on nothing: the next Block's second input rows and their statistics, the addend rows of this wave's tiles
  RowPieces<RB> nx;
  float nssq = 0.f;
  if (p.prep_out) {
    if (p.prep_x2) load_row_pieces<RB>(nx, p.prep_x2, p.ld_prep_x2, row0 + r, p.prep_C2, li);
    if (p.prep_ssq_b) nssq = p.prep_ssq_b[row0 + r];
  }
  f16x4 add[2][RB][4];
  const bool owner = q0.kpart == 0;
  if (p.addend && owner) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < q0.nt)
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const f16* ar = reinterpret_cast<const f16*>(p.addend) + (size_t)(row0 + 32 * j + l31) * p.ld_add + (q0.tile0 + t) * 32 + 4 * half;
#pragma unroll
          for (int g = 0; g < 4; ++g) add[t][j][g] = *reinterpret_cast<const f16x4*>(ar + g * 8);
        }
  }
  for (int c = tid; c < C; c += kThreads) {
    s_gate[c] = p.gate ? p.gate[(size_t)b * p.gate_stride + c] : 1.f;   // (no gate: the plain residual add of a block without GlobalContext)
    s_bias[c] = p.bias ? p.bias[c] : 0.f;
  }
  {   // concat(x, x2) rows -> P0: every piece of this thread requested before the first is stored (K <= 512: at most 4 RB pieces per thread)
    constexpr int NP = 4 * RB;
    const int npr = K >> 3, np1 = C1 >> 3, total = ROWS * npr;
    const f16* x1 = reinterpret_cast<const f16*>(p.x);
    const f16* x2 = reinterpret_cast<const f16*>(p.x2);
    uint4 t[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int i = tid + k * kThreads < total ? tid + k * kThreads : 0;
      const int rr = i / npr, g = i - rr * npr;
      const f16* src = g < np1 ? x1 + (size_t)(row0 + rr) * p.ld_x + g * 8 : x2 + (size_t)(row0 + rr) * p.ld_x2 + (g - np1) * 8;
      t[k] = *reinterpret_cast<const uint4*>(src);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int i = tid + k * kThreads;
      if (i < total) {
        const int rr = i / npr, g = i - rr * npr;
        *reinterpret_cast<uint4*>(P0 + (size_t)rr * pitch0 + g * 16) = t[k];
      }
    }
  }
  __syncthreads();
  // ---- out = fp16(in Wres^T + bias + addend * gate) -> P1
  stage_run<RB>(acc, ring, q0, p.w0, p.w_cout_pad0, P0, pitch0, P0, lane);
  if (owner) {
    const int half = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < q0.nt)
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c = (q0.tile0 + t) * 32 + 8 * g + 4 * half + e;
              float v = acc[t][j][4 * g + e] + s_bias[c];
              if (p.addend) v += (float)add[t][j][g][e] * s_gate[c];
              acc[t][j][4 * g + e] = v;
            }
  }
  store_stage<IMAGEN_ACT_NONE, RB>(acc, q0, P1, pitch1, lane);
  __syncthreads();
  // ---- row pass: out rows -> global (+ their sum of squares), then the next Block's activated input silu(concat(out, prep_x2) * rs * pa)
  {
    const int np = C >> 3;
    const char* srow = P1 + (size_t)r * pitch1;
    f16* out = reinterpret_cast<f16*>(p.out) + (size_t)(row0 + r) * p.ld_out;
    f16x8 ov[PMAX];
    float ssq = 0.f;
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
      const int g = li + k * LPR;
      if (g < np) {
        ov[k] = *reinterpret_cast<const f16x8*>(srow + g * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) ssq += (float)ov[k][e] * (float)ov[k][e];
        *reinterpret_cast<f16x8*>(out + g * 8) = ov[k];
      }
    }
    ssq = row_sum<LPR>(ssq);
    if (p.ssq_out && li == 0) p.ssq_out[row0 + r] = ssq;
    if (p.prep_out) {
      const float sc = __builtin_amdgcn_rsqf(fmaxf(ssq + p.prep_ssq_wb * nssq, 1e-24f));
      f16* ya = reinterpret_cast<f16*>(p.prep_out) + (size_t)(row0 + r) * p.ld_prep;
#pragma unroll
      for (int k = 0; k < PMAX; ++k) {
        const int g = li + k * LPR;
        if (g < np) {
          f16x8 a;
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = (f16)silu_f((float)ov[k][e] * sc * p.prep_pa[g * 8 + e]);
          *reinterpret_cast<f16x8*>(ya + g * 8) = a;
        }
      }
      const int np2 = p.prep_C2 >> 3;
#pragma unroll
      for (int k = 0; k < PMAX; ++k) {
        const int g = li + k * LPR;
        if (g < np2) {
          f16x8 a;
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = (f16)silu_f((float)nx.v[k][e] * sc * p.prep_pa[C + g * 8 + e]);
          *reinterpret_cast<f16x8*>(ya + C + g * 8) = a;
        }
      }
    }
  }
}

template <int MODE, int RB>
__global__ __launch_bounds__(kThreads, ROWCHAIN_MINW) void rowchain_kernel(const ImagenRowchainParams p, unsigned code_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the kernel's own code range read as data, one parallel round trip (common.h: consecutive launches of a step run different kernels)
  const unsigned warm = imagen_code_warm(code_bytes, threadIdx.x, kThreads);
  const int row0 = blockIdx.x * 32 * RB;
  if (MODE == IMAGEN_CHAIN_FF) chain_ff<RB>(p, smem, row0);
  else if (MODE == IMAGEN_CHAIN_XATTN) chain_xattn<RB>(p, smem, row0);
  else if (MODE == IMAGEN_CHAIN_QKV) chain_qkv<RB>(p, smem, row0);
  else chain_resprep<RB>(p, smem, row0);
  imagen_code_warm_sink(warm);
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
  static const unsigned code_bytes = [] {
    char name[160];
    snprintf(name, sizeof(name), "_ZN12_GLOBAL__N_115rowchain_kernelILi%dELi%dEEEv20ImagenRowchainParamsj", MODE, RB);
    return imagen_kernel_code_bytes(name);
  }();
  hipLaunchKernelGGL((rowchain_kernel<MODE, RB>), dim3((unsigned)(p.rows / (32 * RB))), dim3(kThreads), lds, s, p, code_bytes);
  return imagen_hip_status("rowchain");
}

}  // namespace

#ifdef ROWCHAIN_TRACE
extern "C" int imagen_debug_rowchain_trace(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rowchain_trace), &buf, sizeof(buf));
}
#endif

int launch_rowchain(const ImagenRowchainParams* pp, hipStream_t s) {
  const ImagenRowchainParams& p = *pp;
  IMAGEN_CHECK(p.mode >= IMAGEN_CHAIN_FF && p.mode <= IMAGEN_CHAIN_RESPREP, "rowchain: mode %d", p.mode);
  IMAGEN_CHECK(p.x && p.out && p.w0 && (p.g0 || p.mode == IMAGEN_CHAIN_RESPREP), "rowchain: null pointer");
  const int tile = p.tile64 ? 64 : 32;
  IMAGEN_CHECK(p.rows > 0 && p.rows_per_batch > 0 && p.rows % p.rows_per_batch == 0 && p.rows_per_batch % tile == 0,
               "rowchain: %d rows, %d per image, %d-row tiles", p.rows, p.rows_per_batch, tile);
  IMAGEN_CHECK(p.C >= 32 && p.C <= 256 && (p.C & (p.C - 1)) == 0, "rowchain: C = %d (a power of two in 32 .. 256: 1, 2, 4 or 8 cout tiles)", p.C);
  if (p.mode == IMAGEN_CHAIN_RESPREP) {
    IMAGEN_CHECK(p.inner % 32 == 0 && p.inner > 0 && p.C2 % 32 == 0 && p.C2 >= 0 && p.inner + p.C2 <= 512 && (p.C2 == 0 || p.x2),
                 "rowchain RESPREP: inputs in 32-channel chunks, C1 + C2 <= 512 (got %d + %d)", p.inner, p.C2);
    IMAGEN_CHECK(p.ld_x % 8 == 0 && p.ld_out % 8 == 0 && (p.C2 == 0 || (p.ld_x2 % 8 == 0 && ((size_t)p.x2 & 15) == 0)) && ((size_t)p.x & 15) == 0 &&
                     ((size_t)p.out & 15) == 0,
                 "rowchain RESPREP: 16-byte aligned rows");
    IMAGEN_CHECK((p.addend || !p.gate) && (!p.addend || (p.ld_add % 4 == 0 && ((size_t)p.addend & 7) == 0)),
                 "rowchain RESPREP: a gate needs its addend (8-byte aligned rows)");
    IMAGEN_CHECK(!p.prep_out || (p.prep_pa && p.ld_prep % 8 == 0 && ((size_t)p.prep_out & 15) == 0 && p.prep_C2 % 8 == 0 && p.prep_C2 <= 256 &&
                                 (p.prep_C2 == 0 || (p.prep_x2 && p.ld_prep_x2 % 8 == 0 && ((size_t)p.prep_x2 & 15) == 0))),
                 "rowchain RESPREP: the activated output needs its gain vector, 16-byte aligned rows, prep_C2 <= 256");
    IMAGEN_CHECK(p.w_cout_pad0 >= p.C, "rowchain RESPREP: weight padding");
    return p.tile64 ? launch_one<IMAGEN_CHAIN_RESPREP, 2>(p, s) : launch_one<IMAGEN_CHAIN_RESPREP, 1>(p, s);
  }
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
    // (always 32-row tiles: with 64 the two row blocks' softmax states and the K^ / V^T register sets spill, and call E measured no gain —
    // 28.2 vs 26.8 us at 16384 rows of 32 channels; `tile64` is ignored here)
    return launch_one<IMAGEN_CHAIN_XATTN, 1>(p, s);
  }
  IMAGEN_CHECK(p.khat && p.vt && p.k_scale, "rowchain QKV: null operand");
  IMAGEN_CHECK(p.k_rs % 8 == 0 && p.k_bs % 8 == 0 && ((size_t)p.khat & 15) == 0, "rowchain QKV: 16-byte aligned K^ rows");
  IMAGEN_CHECK(p.w_cout_pad0 >= p.inner + 2 * kDh, "rowchain QKV: weight padding");
  return p.tile64 ? launch_one<IMAGEN_CHAIN_QKV, 2>(p, s) : launch_one<IMAGEN_CHAIN_QKV, 1>(p, s);
}
