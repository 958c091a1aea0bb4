// conv_stream.hip — the streaming 3x3 convolution of the 32-channel levels (fourth igemm family, gfx950): C_out <= 32, C_in = 32 (one
// input) or 32 + 32 (the up path's concat of x and the skip connection), stride 1, pad 1, NHWC fp16 (Block, ip.py:671-691, at the
// 256^2 / 128^2 levels of the README super-resolution unet and the 64^2 level of the base unet).
//
// Why it exists (profiles/r02_pmc_SQ_mfma_busy.json, round-2 probe stream_probe.py): these layers move 134-201 MB per launch for 19-39 GFLOP —
// HBM-bound by a factor of five — yet ran at 0.11-0.35 of the HBM rate.  The wave-specialised kernel (igemm.hip) stages through
// registers with four producer waves, six 16-byte items each per tile; the all-DMA kernel (conv_dma.hip) starts one workgroup per tile,
// reloads the weights for every tile and has a single halo load in flight per workgroup.  Measured here (MI355X, batch 16, 256^2):
// 32->32 raw 52 -> 37 us (3.7 TB/s of algorithmic traffic), with the prologue 67-76 -> 56-62 us.  This kernel is
//   * PERSISTENT: a workgroup walks a strided list of 16x16-pixel tiles, and the halo tile (18x18 pixels x 32 channels per input, dense
//     in LDS with the source-side bank swizzle of conv_dma.hip) of tile t+1 is copied global -> LDS by global_load_lds_dwordx4 while
//     tile t is transformed, multiplied and stored — a whole tile period of load latency hidden, no VGPR round trip;
//   * WEIGHT-STATIONARY: the 18 (36) KB of packed weights are copied to LDS once per workgroup; a K=16 step reads its A fragment with
//     one conflict-free ds_read_b128 (the packed layout IS the fragment order);
//   * the Block prologue (ChanRMSNorm statistics from the producers' per-pixel sums of squares, per-(batch, channel) affine, SiLU) runs
//     IN PLACE on the landed tile (ds_read_b128 -> fp32 math -> ds_write_b128) by all eight waves, with the per-pixel statistics
//     of tile t+1 prefetched into registers together with its DMA.
// Contract, packed weight layout and epilogue are those of the other families (ImagenIgemmParams; conv_epilogue.h).
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int CS_TW = 16, CS_TH = 16, CS_ITW = 18, CS_ITH = 18;
constexpr int CS_PITCH = CS_ITW * 64;            // bytes per halo row: 4 x 16 B per pixel, dense
constexpr int CS_NPX = CS_ITH * CS_ITW;          // 324 halo pixels
constexpr int CS_NSLOT = CS_NPX * 4;             // 1296 16-byte slots
constexpr int CS_NDMA = (CS_NSLOT + 63) / 64;    // 21 one-KiB DMA pieces per (tile, input)
constexpr int CS_NW = 8;                         // waves per workgroup: wave w owns the 32 pixels [32 w, 32 w + 32) of the tile
constexpr int CS_NJ = (CS_NDMA + CS_NW - 1) / CS_NW;   // <= 3 DMA pieces per wave, tile and input
constexpr int CS_ABUF = CS_NJ * CS_NW * 1024;    // 24 KiB per (tile, input) buffer (pieces 21-23 are never written)
constexpr int CS_WCH = 18 * 1024;                // packed weights of one 32-channel chunk: 18 K steps x 1 KiB
constexpr int CS_EP_RED = CS_NW * 32;             // floats
constexpr int CS_EP_PAR = 5 * 32 + 8 * 32 * 2 + 64;   // floats (conv_epilogue.h: 4 * BN + WM * WN * 32 * MI + 8 + WM * (BN + 4) with the GlobalContext partials = 680 here)

typedef unsigned cs_u32x4 __attribute__((ext_vector_type(4)));

// direct-to-LDS copy (lane l -> LDS bytes [dst + 16 l, +16)) and LDS base: lds_dma.h
#define cs_dma16 IMAGEN_DMA16
#define CS_LDS_BASE IMAGEN_LDS_BASE

constexpr size_t cs_lds_bytes(int nch) {
  return (size_t)2 * nch * CS_ABUF + (size_t)nch * CS_WCH + (size_t)(CS_EP_RED + CS_EP_PAR) * sizeof(float) + (size_t)2 * 2 * 64 * sizeof(float) + 16;
}

// s_waitcnt vmcnt(min(n, 12)) where n counts the lane's output stores behind its copies (lds_dma.h)
#define cs_wait_vm IMAGEN_WAIT_VM_STORES

// NCH: 32-channel inputs (1: x1 only; 2: x1 | x2).  PRO: Block prologue on the inputs.  GEN: generic epilogue (conv_epilogue.h).
//
// What bounds it (rocprofv3 SQ counters, round-2 call gpu_r2_y.sh; variants in the history of this file): every wave issues ~24 % of the time
// and a SIMD holds four of them — the SIMD's instruction issue is saturated.  Deeper prefetch (tile t+2 / t+3 staged in registers, or
// the LDS buffer refilled right after the MFMA phase with hand-counted vmcnt waits) changed nothing; fewer instructions did.  Hence
//   * the tile list is walked incrementally (no integer division per tile: (b, ty, tx) advance by the grid stride with two wrap checks);
//   * an INTERIOR tile (halo inside the image — all but the border ring) is addressed as one scalar base + one per-lane constant offset
//     per slot: two VALU per DMA piece instead of bounds checks, selects and 64-bit multiplies; border tiles take the general path;
//   * the accumulators start at the bias (one ds_read_b128 per channel quad) instead of being zeroed and biased afterwards;
//   * the plain / post_pa epilogue of an interior tile stores through a per-lane constant output offset (anything else: conv_epilogue.h,
//     with its per-channel operands preloaded once per image — a global load in the epilogue would be younger than the next tile's
//     copies, and waiting for it would drain them);
//   * the wait for a landed tile leaves the previous tile's output stores in flight (vmcnt retires in issue order and counts stores;
//     the copies are older): vmcnt(number of stores known to have been issued behind them).
template <int NCH, bool PRO, bool GEN>
__global__ __launch_bounds__(64 * CS_NW, NCH == 1 ? 4 : 2) void conv_stream_kernel(const ImagenIgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const acts = smem;                                       // [2 tiles][NCH][CS_ABUF]
  char* const wlds = smem + 2 * NCH * CS_ABUF;                   // [NCH][18][1 KiB]
  float* const ep_red = reinterpret_cast<float*>(wlds + NCH * CS_WCH);
  float* const ep_par = ep_red + CS_EP_RED;                      // [bias 32 | post_pa 32 | post_ps 32 | ...]
  float* const aff = ep_par + CS_EP_PAR;                         // [2 tiles][pa 64 | ps 64]
  const unsigned lds0 = CS_LDS_BASE(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int tilesX = (p.OW + CS_TW - 1) / CS_TW, tilesY = (p.OH + CS_TH - 1) / CS_TH;
  const int total = p.B * tilesY * tilesX;
  const int G = gridDim.x;
  // tile cursor (workgroup-uniform): advanced by G tiles with wrap checks instead of div / mod per tile
  const int stepX = G % tilesX, stepY = (G / tilesX) % tilesY, stepB = G / (tilesX * tilesY);
  struct Cur { int t, b, ty, tx; };
  auto advance = [&](Cur c) __attribute__((always_inline)) -> Cur {
    c.t += G;
    c.tx += stepX;
    if (c.tx >= tilesX) { c.tx -= tilesX; ++c.ty; }
    c.ty += stepY;
    if (c.ty >= tilesY) { c.ty -= tilesY; ++c.b; }
    c.b += stepB;
    return c;
  };
  auto tile_of = [&](const Cur& c) __attribute__((always_inline)) -> ClTile {
    ClTile tc;
    tc.b = c.b;
    tc.oy0 = c.ty * CS_TH;
    tc.ox0 = c.tx * CS_TW;
    tc.n0 = 0;
    return tc;
  };
  auto interior = [&](const ClTile& tc) __attribute__((always_inline)) -> bool {   // the 18x18 halo lies inside the image
    return tc.oy0 >= 1 && tc.ox0 >= 1 && tc.oy0 + CS_TH + 1 <= p.H && tc.ox0 + CS_TW + 1 <= p.W;
  };

  const size_t wrow = (size_t)p.Cout_pad * 16;
  const char* const zero_src = reinterpret_cast<const char*>(p.w) + (size_t)(NCH * 36) * wrow;   // the packed buffer's zero tail

  // ---- per-lane constants of this lane's DMA slots (see the first form); interior tiles: byte offset from halo pixel (0, 0)
  int s_r[CS_NJ], s_hx[CS_NJ], off1[CS_NJ], off2[CS_NJ], offs[CS_NJ];
#pragma unroll
  for (int j = 0; j < CS_NJ; ++j) {
    const int hp = ((wave + CS_NW * j) * 64 + lane) >> 2;
    const int r = (hp * 3641) >> 16;
    const bool in_tile = hp < CS_NPX;
    s_r[j] = in_tile ? r : -100000;
    s_hx[j] = hp - r * CS_ITW;
    const int kg = (lane & 3) ^ ((s_hx[j] >> 1) & 3);
    const int px = in_tile ? r * p.W + s_hx[j] : 0;   // (slots past the tile re-read halo pixel 0: never consumed)
    off1[j] = (px * p.ld1 + kg * 8) * 2;
    off2[j] = NCH == 2 ? (px * p.ld2 + kg * 8) * 2 : 0;
    offs[j] = px * 4;
  }
  const int pos = lane & 3;

  float sa[CS_NJ], sb[CS_NJ];   // per-pixel statistics of the NEXT tile's slots (PRO)
  unsigned okm = 0;             // in-image mask of the next tile's slots
  float aff_next = 0.f;
  const bool has_b = NCH == 2 && p.ssq_b != nullptr;

  auto issue_tile = [&](const ClTile& tc, int buf) __attribute__((always_inline)) {
    if (interior(tc)) {   // (workgroup-uniform) scalar base + per-lane constant
      const size_t o0 = (size_t)(tc.oy0 - 1) * p.W + (tc.ox0 - 1);
      const char* b1 = reinterpret_cast<const char*>(reinterpret_cast<const f16*>(p.x1) + (size_t)tc.b * p.bs1 + o0 * p.ld1);
      const char* b2 = NCH == 2 ? reinterpret_cast<const char*>(reinterpret_cast<const f16*>(p.x2) + (size_t)tc.b * p.bs2 + o0 * p.ld2) : nullptr;
      const float* bs_a = PRO ? p.ssq_a + (size_t)tc.b * (p.H * p.W) + o0 : nullptr;
      const float* bs_b = (PRO && has_b) ? p.ssq_b + (size_t)tc.b * (p.H * p.W) + o0 : nullptr;
#pragma unroll
      for (int j = 0; j < CS_NJ; ++j) {
        const int d = wave + CS_NW * j;
        if (d < CS_NDMA) {
          cs_dma16(b1 + off1[j], __builtin_amdgcn_readfirstlane(lds0 + (buf * NCH) * CS_ABUF + d * 1024));
          if constexpr (NCH == 2) cs_dma16(b2 + off2[j], __builtin_amdgcn_readfirstlane(lds0 + (buf * NCH + 1) * CS_ABUF + d * 1024));
        }
        if constexpr (PRO) {
          sa[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bs_a) + offs[j]);
          sb[j] = has_b ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bs_b) + offs[j]) : 0.f;
        }
      }
      okm = 0xffffffffu;
    } else {
      okm = 0;
#pragma unroll
      for (int j = 0; j < CS_NJ; ++j) {
        const int d = wave + CS_NW * j;
        const int gy = tc.oy0 - 1 + s_r[j], gx = tc.ox0 - 1 + s_hx[j];
        const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const int gp = ok ? gy * p.W + gx : 0;
        const int kg = pos ^ ((s_hx[j] >> 1) & 3);
        if (d < CS_NDMA) {
          const f16* x1 = reinterpret_cast<const f16*>(p.x1) + (size_t)tc.b * p.bs1;
          cs_dma16(ok ? reinterpret_cast<const char*>(x1 + (size_t)gp * p.ld1 + kg * 8) : zero_src,
                   __builtin_amdgcn_readfirstlane(lds0 + (buf * NCH) * CS_ABUF + d * 1024));
          if constexpr (NCH == 2) {
            const f16* x2 = reinterpret_cast<const f16*>(p.x2) + (size_t)tc.b * p.bs2;
            cs_dma16(ok ? reinterpret_cast<const char*>(x2 + (size_t)gp * p.ld2 + kg * 8) : zero_src,
                     __builtin_amdgcn_readfirstlane(lds0 + (buf * NCH + 1) * CS_ABUF + d * 1024));
          }
        }
        if constexpr (PRO) {
          const size_t sp = (size_t)tc.b * (p.H * p.W) + gp;
          sa[j] = p.ssq_a[sp];
          sb[j] = has_b ? p.ssq_b[sp] : 0.f;
          if (ok) okm |= 1u << j;
        }
      }
    }
    if constexpr (PRO) {
      const int ch = tid & 63;
      const size_t o = (size_t)tc.b * p.pstride + ch;
      aff_next = 0.f;
      if (tid < 64) aff_next = ch < 32 * NCH ? p.pa[o] : 0.f;
      else if (tid < 128 && p.ps) aff_next = ch < 32 * NCH ? p.ps[o] : 0.f;
    }
  };

  // ---- MFMA side: wave w owns the 32 pixels [32 w, 32 w + 32) x all 32 output channels
  int pix_y[1], pix_x[1], bA[6];
  {
    const int tp = wave * 32 + l31;
    const int py = tp / CS_TW, px = tp - py * CS_TW;
    pix_y[0] = py;
    pix_x[0] = px;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = px + dx;
      const int a0 = py * CS_PITCH + hx * 64 + ((half ^ ((hx >> 1) & 3)) << 4);
      bA[2 * dx] = a0;            // K step 0 (channel groups 0 / 1)
      bA[2 * dx + 1] = a0 ^ 32;   // K step 1 (groups 2 / 3)
    }
  }
  // output offsets of this lane (bytes from the tile's first output pixel).  A lane holds channel quads 8q + 4*half + {0..3}; before
  // the stores the two half-waves exchange quads (v_permlane32_swap) so that lanes 0-31 own channels 0-15 and lanes 32-63 channels
  // 16-31 of their pixel as two 16-byte pieces: half the store instructions, twice the bytes per request
  const int yoff = ((pix_y[0] * p.OW + pix_x[0]) * p.ldy + 16 * half) * 2;
  const int qoff = (pix_y[0] * p.OW + pix_x[0]) * 4;
  // quads (q, q + 2) are exchanged dword-wise: afterwards a lane of the lower half-wave holds [own q | partner's q] = channels 8q .. 8q+7,
  // a lane of the upper one [partner's q + 2 | own q + 2] = channels 16 + 8q .. 16 + 8q + 7 — the same register order in both halves
  auto store_quads = [&](char* yb, const f16x4 (&oq)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint2 lo = __builtin_bit_cast(uint2, oq[q]), hi = __builtin_bit_cast(uint2, oq[q + 2]);
      const auto r0 = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
      *reinterpret_cast<cs_u32x4*>(yb + 16 * q) = cs_u32x4{r0[0], r1[0], r0[1], r1[1]};
    }
  };

  Cur c;
  c.t = blockIdx.x;
  if (c.t >= total) return;
  {
    int tt = c.t;
    c.tx = tt % tilesX;
    tt /= tilesX;
    c.ty = tt % tilesY;
    c.b = tt / tilesY;
  }
  for (int s = wave; s < 18 * NCH; s += CS_NW) {   // weights -> LDS once
    const int ch = s / 18, ks = s - 18 * ch;
    const char* src = reinterpret_cast<const char*>(p.w) + ((size_t)(ch * 36 + 2 * ks + half) * p.Cout_pad + l31) * 16;
    cs_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + 2 * NCH * CS_ABUF + s * 1024));
  }
  ClTile tc = tile_of(c);
  issue_tile(tc, 0);
  int cur = 0;
  int ep_b = -1;
  int stores_behind = 0;
  // store instructions of the plain / post epilogue of conv_epilogue.h per wave (a lower bound is what the wait needs): one 16-byte piece
  // per pair of channel quads when Cout is a multiple of 8, else one 8-byte store per quad below Cout
  const int quads = GEN ? 0 : ((p.Cout & 7) == 0 ? 1 + (p.Cout > 8 ? 1 : 0) : min((p.Cout + 7) >> 3, 4));
  const bool lean_ok = !GEN && p.Cout == 32;   // the lean epilogue writes all four channel quads

  while (true) {
    float sa_c[CS_NJ], sb_c[CS_NJ];
    unsigned okm_c = 0;
    if constexpr (PRO) {
#pragma unroll
      for (int j = 0; j < CS_NJ; ++j) { sa_c[j] = sa[j]; sb_c[j] = sb[j]; }
      okm_c = okm;
      if (tid < 128) aff[cur * 128 + tid] = aff_next;
    }
    if (tc.b != ep_b) {   // (workgroup-uniform, once per image) the epilogue's per-channel operands; published by the barrier below
      __syncthreads();    // (the previous tile's epilogue may still be reading them)
      if (tid < 32) cl_epilogue_params<32>(p, tc.b, 0, ep_par, tid);
      ep_b = tc.b;
      stores_behind = 0;
    }
    cs_wait_vm(stores_behind);
    __syncthreads();
    const Cur cn = advance(c);
    const bool more = cn.t < total;
    const ClTile tn = tile_of(cn);
    if (more) issue_tile(tn, cur ^ 1);

    if constexpr (PRO) {
      const float* pa_l = aff + cur * 128;
      const bool silu = p.act_in == IMAGEN_ACT_SILU;
#pragma unroll
      for (int j = 0; j < CS_NJ; ++j) {
        const int d = wave + CS_NW * j;
        if (d >= CS_NDMA) continue;
        const int S = d * 64 + lane;
        const int kg = pos ^ ((s_hx[j] >> 1) & 3);
        const float q = sa_c[j] + p.ssq_wb * sb_c[j];
        const float rs = __builtin_amdgcn_rsqf(fmaxf(q, 1e-24f));
        const bool ok = (okm_c >> j) & 1u;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          char* a = acts + (cur * NCH + ch) * CS_ABUF + S * 16;
          const f16x8 in = *reinterpret_cast<const f16x8*>(a);
          const float* pl = pa_l + ch * 32 + kg * 8;
          f16x8 out;
#pragma unroll
          for (int h4 = 0; h4 < 2; ++h4) {
            const float4 aq = *reinterpret_cast<const float4*>(pl + 4 * h4);
            const float4 sq = *reinterpret_cast<const float4*>(pl + 64 + 4 * h4);
            const float av[4] = {aq.x, aq.y, aq.z, aq.w}, sv[4] = {sq.x, sq.y, sq.z, sq.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float tt = (float)in[4 * h4 + i] * rs * av[i] + sv[i];
              const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * tt));
              out[4 * h4 + i] = (f16)(silu ? tt * sg : tt);
            }
          }
          cs_u32x4 ow = __builtin_bit_cast(cs_u32x4, out);
          if (!ok) ow = cs_u32x4{0u, 0u, 0u, 0u};
          *reinterpret_cast<cs_u32x4*>(a) = ow;
        }
      }
      __syncthreads();
    }

    // ---- accumulators: start at the bias where the lean epilogue follows (it then adds none), else at zero
    const bool lean = lean_ok && tc.oy0 + CS_TH <= p.OH && tc.ox0 + CS_TW <= p.OW;   // (workgroup-uniform) all 256 pixels inside the image
    f32x16 acc[1][1];
    if (lean) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bq = *reinterpret_cast<const float4*>(ep_par + 8 * q + 4 * half);
        acc[0][0][4 * q] = bq.x; acc[0][0][4 * q + 1] = bq.y; acc[0][0][4 * q + 2] = bq.z; acc[0][0][4 * q + 3] = bq.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const char* ab = acts + (cur * NCH + ch) * CS_ABUF;
      const char* wb = wlds + ch * CS_WCH + lane * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const f16x8 af = *reinterpret_cast<const f16x8*>(wb + (tap * 2 + ks) * 1024);
          const f16x8 bf = *reinterpret_cast<const f16x8*>(ab + bA[2 * dx + ks] + dy * CS_PITCH);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[0][0], 0, 0, 0);
        }
      }
    }

    if (lean) {
      // ---- lean epilogue (plain / post_pa, every pixel and channel of the tile stored): one scalar base + per-lane constant offsets
      char* yb = reinterpret_cast<char*>(reinterpret_cast<f16*>(p.y) + (size_t)tc.b * p.bsy + ((size_t)tc.oy0 * p.OW + tc.ox0) * p.ldy) + yoff;
      if (p.post_pa) {   // output-side Block prologue (conv_epilogue.h): v / ||v|| * pa + ps -> SiLU
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) tot += acc[0][0][r] * acc[0][0][r];
        tot += __shfl_xor(tot, 32);
        const float rsn = __builtin_amdgcn_rsqf(fmaxf(tot, 1e-24f));
        f16x4 oq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 pa = *reinterpret_cast<const float4*>(ep_par + 32 + 8 * q + 4 * half);
          const float4 ps = *reinterpret_cast<const float4*>(ep_par + 64 + 8 * q + 4 * half);
          const float pav[4] = {pa.x, pa.y, pa.z, pa.w}, psv[4] = {ps.x, ps.y, ps.z, ps.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) oq[q][e] = (f16)silu_f(acc[0][0][4 * q + e] * rsn * pav[e] + psv[e]);
        }
        store_quads(yb, oq);
      } else {
        float ssq = 0.f;
        f16x4 oq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            oq[q][e] = (f16)acc[0][0][4 * q + e];
            const float r = (float)oq[q][e];
            ssq += r * r;
          }
        }
        store_quads(yb, oq);
        if (p.ssq_out) {
          ssq += __shfl_xor(ssq, 32);
          if (half == 0)
            *reinterpret_cast<float*>(reinterpret_cast<char*>(p.ssq_out + (size_t)tc.b * (p.OH * p.OW) + (size_t)tc.oy0 * p.OW + tc.ox0) + qoff) = ssq;
        }
      }
    } else {
      cl_epilogue<1, 1, CS_NW, 1, GEN, true>(p, tc, acc, pix_y, pix_x, ep_red, ep_par, wave, 0, half, l31);
    }

    if (!more) break;
    stores_behind = lean ? 2 : ((tc.oy0 + CS_TH <= p.OH && tc.ox0 + CS_TW <= p.OW) ? quads : 0);
    c = cn;
    tc = tn;
    cur ^= 1;
  }
}

template <int NCH, bool PRO, bool GEN>
int cs_launch_gen(const ImagenIgemmParams& p, hipStream_t s) {
  auto kern = conv_stream_kernel<NCH, PRO, GEN>;
  constexpr size_t lds = cs_lds_bytes(NCH);
  static bool attr_done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("conv_stream: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  const int tilesX = (p.OW + CS_TW - 1) / CS_TW, tilesY = (p.OH + CS_TH - 1) / CS_TH;
  const int total = p.B * tilesX * tilesY;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int per_cu = NCH == 1 ? 2 : 1;
  const int resident = std::max(1, std::max(1, cus) * per_cu);
  int gx = total;
  if (total > resident) {   // even rounds: every workgroup walks the same number of tiles (+-1)
    const int rounds = (total + resident - 1) / resident;
    gx = (total + rounds - 1) / rounds;
  }
  hipLaunchKernelGGL(kern, dim3(gx), dim3(64 * CS_NW), lds, s, p);
  return imagen_hip_status("conv_stream launch");
}

}  // namespace

// ---- family interface (igemm.hip lists this family behind the all-DMA one)
int imagen_conv_stream_num_configs() { return 1; }

int imagen_conv_stream_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (idx != 0) return -1;
  if (tile_pixels) *tile_pixels = CS_TH * CS_TW;
  if (tile_cout) *tile_cout = 32;
  if (kgroups) *kgroups = 4;
  return 0;
}

long imagen_conv_stream_lds_bytes(int idx, int KH, int KW, int TH, int TW) {
  if (idx != 0 || KH != 3 || KW != 3 || TH != CS_TH || TW != CS_TW) return -1;
  return (long)cs_lds_bytes(2);
}

int launch_conv_stream(const ImagenIgemmParams* pp, int idx, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  IMAGEN_CHECK(idx == 0, "conv_stream: bad cfg");
  IMAGEN_CHECK(p.TH == CS_TH && p.TW == CS_TW, "conv_stream: 16x16 tiles (got %dx%d)", p.TH, p.TW);
  IMAGEN_CHECK(p.stride == 1 && p.KH == 3 && p.KW == 3 && p.pad == 1, "conv_stream: 3x3 stride-1 convolutions only");
  IMAGEN_CHECK(p.C1 == 32 && (p.C2 == 0 || (p.C2 == 32 && p.x2)) && p.Cin_pad == p.C1 + p.C2, "conv_stream: inputs of 32 (+ 32) channels (C1 %d C2 %d)",
               p.C1, p.C2);
  IMAGEN_CHECK(p.ld1 % 8 == 0 && (p.C2 == 0 || p.ld2 % 8 == 0), "conv_stream: row strides must keep 16-byte alignment");
  IMAGEN_CHECK(p.Cout <= 32 && p.Cout_pad % 32 == 0, "conv_stream: at most 32 output channels (Cout %d)", p.Cout);
  IMAGEN_CHECK(!p.mu && !p.rs, "conv_stream: the prologue takes its statistics from ssq_a / ssq_b (no mu / rs)");
  const bool pro = p.ssq_a != nullptr || p.pa != nullptr || p.ps != nullptr || p.act_in != IMAGEN_ACT_NONE;
  IMAGEN_CHECK(!pro || (p.ssq_a && p.pa && (p.act_in == IMAGEN_ACT_NONE || p.act_in == IMAGEN_ACT_SILU)),
               "conv_stream: the prologue needs ssq_a and pa (act_in NONE | SILU)");
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NCHW_F32 || p.Cout % 4 == 0, "conv_stream: Cout %d must be a multiple of 4", p.Cout);
  IMAGEN_CHECK(p.out_mode != IMAGEN_OUT_PIXEL_SHUFFLE || p.Cout % 16 == 0, "conv_stream: pixel-shuffle needs Cout %% 16 == 0");
  IMAGEN_CHECK(!p.post_pa || (p.post_ps && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res && !p.ssq_out && p.act_out == IMAGEN_ACT_NONE && p.Cout % 4 == 0),
               "conv_stream: post_pa needs post_ps and a plain NHWC output");
  IMAGEN_CHECK(!(p.addend && p.res), "conv_stream: addend and residual are mutually exclusive");
  IMAGEN_CHECK(!p.gca_part, "conv_stream: no GlobalContext partials from this family (its eight-wave epilogue with three barriers per tile cost more than the pass it replaced: round 4, call F) — families 2, 5, 7, 8 emit them");
  IMAGEN_CHECK(!p.ssq_out || p.out_mode == IMAGEN_OUT_NHWC, "conv_stream: ssq_out needs NHWC output");
  const bool plain = p.act_out == IMAGEN_ACT_NONE && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res;
  const int key = (p.C2 ? 4 : 0) | (pro ? 2 : 0) | (plain ? 0 : 1);
  switch (key) {
    case 0: return cs_launch_gen<1, false, false>(p, s);
    case 1: return cs_launch_gen<1, false, true>(p, s);
    case 2: return cs_launch_gen<1, true, false>(p, s);
    case 3: return cs_launch_gen<1, true, true>(p, s);
    case 4: return cs_launch_gen<2, false, false>(p, s);
    case 5: return cs_launch_gen<2, false, true>(p, s);
    case 6: return cs_launch_gen<2, true, false>(p, s);
    default: return cs_launch_gen<2, true, true>(p, s);
  }
}
