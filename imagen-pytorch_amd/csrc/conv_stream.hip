// conv_stream.hip — the streaming 3x3 convolution of the 32-channel levels (fourth igemm family, gfx950): C_out <= 32, C_in = 32 (one
// input) or 32 + 32 (the up path's concat of x and the skip connection), stride 1, pad 1, NHWC fp16 (Block, ip.py:671-691, at the
// 256^2 / 128^2 levels of the README super-resolution unet and the 64^2 level of the base unet).
//
// Why it exists (profiles/r02_pmc_SQ_mfma_busy.json, tools/igemm_probe.py raw8): these layers move 134-201 MB per launch for 19-39 GFLOP —
// HBM-bound by a factor of five — yet ran at 0.11-0.35 of the HBM rate.  The wave-specialised kernel (igemm.hip) stages through
// registers with four producer waves (~600 cycles per 16-byte item and wave, six items per tile: the staging chain, not the memory,
// sets the pace) and the all-DMA kernel (conv_dma.hip) starts one workgroup per tile, whose single 26 KB load is all it ever has in
// flight: two resident workgroups per CU cover a fraction of the ~2.5 us load latency.  This kernel is
//   * PERSISTENT: a workgroup walks a strided list of 16x16-pixel tiles, and the halo tile (18x18 pixels x 32 channels per input, dense
//     in LDS with the source-side bank swizzle of conv_dma.hip) of tile t+1 is copied global -> LDS by global_load_lds_dwordx4 while
//     tile t is transformed, multiplied and stored — a whole tile period of load latency hidden, no VGPR round trip;
//   * WEIGHT-STATIONARY: the 18 (36) KB of packed weights are copied to LDS once per workgroup; a K=16 step reads its A fragment with
//     one conflict-free ds_read_b128 (the packed layout IS the fragment order);
//   * the Block prologue (ChanRMSNorm statistics from the producers' per-pixel sums of squares, per-(batch, channel) affine, SiLU) runs
//     IN PLACE on the landed tile (ds_read_b128 -> fp32 math -> ds_write_b128) by all eight waves — the VALU is idle in an HBM-bound
//     layer — with the per-pixel statistics of tile t+1 prefetched into registers together with its DMA.
// Contract, packed weight layout and epilogue are those of the other families (ImagenIgemmParams; conv_epilogue.h).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>
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
constexpr int CS_EP_PAR = 5 * 32 + 8 * 32 * 2;   // floats (conv_epilogue.h: 5 * BN + 8 * 32 * MI, MI = 1 here: ample)

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) — register arrays indexed inside stay in registers
template <class F, int... I>
__device__ __forceinline__ void cs_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cs_static_for(F&& f) {
  cs_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

typedef unsigned cs_u32x4 __attribute__((ext_vector_type(4)));   // (a native vector: HIP's uint4 is a struct and lands in scratch when carried across the loop)

__device__ __forceinline__ void cs_dma16(const void* gsrc, unsigned lds_dst) {   // lane l -> LDS bytes [dst + 16 l, +16)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr size_t cs_lds_bytes(int nch) {
  return (size_t)2 * nch * CS_ABUF + (size_t)nch * CS_WCH + (size_t)(CS_EP_RED + CS_EP_PAR) * sizeof(float) + (size_t)2 * 2 * 64 * sizeof(float) +
         (size_t)2 * 2 * 6 * 64 * sizeof(float) + 16;   // + the prologue's affine row, + the per-pixel statistics tables (third form)
}

// NCH: 32-channel inputs (1: x1 only; 2: x1 | x2).  PRO: Block prologue on the inputs.  GEN: generic epilogue (conv_epilogue.h).
template <int NCH, bool PRO, bool GEN>
__global__ __launch_bounds__(64 * CS_NW, NCH == 1 ? 4 : 2) void conv_stream_kernel(const ImagenIgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const acts = smem;                                       // [2 tiles][NCH][CS_ABUF]
  char* const wlds = smem + 2 * NCH * CS_ABUF;                   // [NCH][18][1 KiB]
  float* const ep_red = reinterpret_cast<float*>(wlds + NCH * CS_WCH);
  float* const ep_par = ep_red + CS_EP_RED;
  float* const aff = ep_par + CS_EP_PAR;                         // [2 tiles][pa 64 | ps 64]: prologue affine of the tile's batch row
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int tilesX = (p.OW + CS_TW - 1) / CS_TW, tilesY = (p.OH + CS_TH - 1) / CS_TH;
  const int total = p.B * tilesY * tilesX;
  auto decode = [&](int t) __attribute__((always_inline)) -> ClTile {
    ClTile c;
    const int tx = t % tilesX;
    t /= tilesX;
    const int ty = t % tilesY;
    c.b = t / tilesY;
    c.oy0 = ty * CS_TH;
    c.ox0 = tx * CS_TW;
    c.n0 = 0;
    return c;
  };

  const size_t wrow = (size_t)p.Cout_pad * 16;   // bytes per packed 8-channel-group row
  const char* const zero_src = reinterpret_cast<const char*>(p.w) + (size_t)(NCH * 36) * wrow;   // the packed buffer's zero tail

  // ---- tile-independent geometry of this lane's DMA slots: slot S = (wave + 4 j) * 64 + lane = (halo pixel S >> 2, position S & 3);
  //      the lane fetches channel group (S & 3) ^ ((hx >> 1) & 3) of its pixel (source-side swizzle: the B-fragment reads below are
  //      conflict-free), or 16 zero bytes outside the image / the tile
  int s_r[CS_NJ], s_hx[CS_NJ];
#pragma unroll
  for (int j = 0; j < CS_NJ; ++j) {
    const int hp = ((wave + CS_NW * j) * 64 + lane) >> 2;
    const int r = (hp * 3641) >> 16;           // hp / 18 for hp < 2048
    s_r[j] = hp < CS_NPX ? r : -100000;        // slots past the tile: never in the image
    s_hx[j] = hp - r * CS_ITW;
  }
  const int pos = lane & 3;

  float sa[CS_NJ], sb[CS_NJ];   // per-pixel statistics of the NEXT tile's slots (PRO)
  unsigned okm = 0;             // in-image mask of the next tile's slots
  float aff_next = 0.f;         // this thread's element of the next tile's affine row (threads 0-127: pa | ps of 64 channels)

  auto issue_tile = [&](const ClTile& tc, int buf) __attribute__((always_inline)) {
    okm = 0;
#pragma unroll
    for (int j = 0; j < CS_NJ; ++j) {
      const int d = wave + CS_NW * j;
      const int gy = tc.oy0 - 1 + s_r[j], gx = tc.ox0 - 1 + s_hx[j];
      const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const int gp = ok ? gy * p.W + gx : 0;
      const int kg = pos ^ ((s_hx[j] >> 1) & 3);
      if (d < CS_NDMA) {   // (wave-uniform)
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
        sb[j] = (NCH == 2 && p.ssq_b) ? p.ssq_b[sp] : 0.f;
        if (ok) okm |= 1u << j;
      }
    }
    if constexpr (PRO) {
      // threads 0-63: pa of channel tid, 64-127: ps (or 0)
      const int ch = tid & 63;
      const size_t o = (size_t)tc.b * p.pstride + ch;
      aff_next = 0.f;
      if (tid < 64) aff_next = ch < 32 * NCH ? p.pa[o] : 0.f;
      else if (tid < 128 && p.ps) aff_next = ch < 32 * NCH ? p.ps[o] : 0.f;
    }
  };

  // ---- MFMA side: wave w owns pixels [32 w, 32 w + 32) of the tile (one 32-pixel fragment = two tile rows) x all 32 output channels
  int pix_y[1], pix_x[1], bP[1][3];
#pragma unroll
  for (int mi = 0; mi < 1; ++mi) {
    const int tp = wave * 32 + l31;
    const int py = tp / CS_TW, px = tp - py * CS_TW;
    pix_y[mi] = py;
    pix_x[mi] = px;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = px + dx;
      bP[mi][dx] = py * CS_PITCH + hx * 64 + ((half ^ ((hx >> 1) & 3)) << 4);   // K step 0 (groups 0 / 1); step 1: ^ 32
    }
  }

  // ---- prologue of the workgroup: weights -> LDS (once), first tile
  int t = blockIdx.x;
  if (t >= total) return;
  {
    // chunk c, K step s: lanes 0-31 copy packed group row c*36 + 2 s, lanes 32-63 row 2 s + 1 (32 couts x 16 B each)
    for (int s = wave; s < 18 * NCH; s += CS_NW) {
      const int c = s / 18, ks = s - 18 * c;
      const char* src = reinterpret_cast<const char*>(p.w) + ((size_t)(c * 36 + 2 * ks + half) * p.Cout_pad + l31) * 16;
      cs_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + 2 * NCH * CS_ABUF + s * 1024));
    }
  }
  ClTile tc = decode(t);
  issue_tile(tc, 0);
  int cur = 0;
  int ep_b = -1;           // batch row whose epilogue operands sit in ep_par
  int stores_behind = 0;   // lower bound of the store instructions this wave issued AFTER the DMA pieces being waited for
  const int quads = (GEN || CL_DBG(8)) ? 0 : min((p.Cout + 7) >> 3, 4);   // plain / post epilogue: one store per channel quad (both lane halves at once)

  while (true) {
    // ---- tile t has landed in buffer `cur` (and, first time round, the weights)
    float sa_c[CS_NJ], sb_c[CS_NJ];
    unsigned okm_c = 0;
    if constexpr (PRO) {
#pragma unroll
      for (int j = 0; j < CS_NJ; ++j) { sa_c[j] = sa[j]; sb_c[j] = sb[j]; }
      okm_c = okm;
      if (tid < 128) aff[cur * 128 + tid] = aff_next;
    }
    // The DMA pieces of this tile are OLDER than the previous tile's output stores (vmcnt retires in issue order and counts stores):
    // waiting for vmcnt(0) would park every wave until the stores are acknowledged (~2-4k cycles, every tile).  Where the number of
    // store instructions behind the pieces has a known lower bound — the plain / post_pa epilogue of a tile that lies inside the image
    // issues one 8-byte store per channel quad below Cout — the wait leaves that many operations outstanding.
    if (stores_behind >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (stores_behind >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (stores_behind >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (stores_behind >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int t_next = t + gridDim.x;
    const bool more = t_next < total;
    ClTile tn = tc;
    if (more) {
      tn = decode(t_next);
      issue_tile(tn, cur ^ 1);   // in flight while this tile is transformed, multiplied and stored
    }

    // ---- Block prologue in place (ip.py:675-684): x * rsqrt(ssq) * pa (+ ps) -> SiLU, zero outside the image
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
        for (int c = 0; c < NCH; ++c) {
          char* a = acts + (cur * NCH + c) * CS_ABUF + S * 16;
          const f16x8 in = *reinterpret_cast<const f16x8*>(a);
          const float4 a0 = *reinterpret_cast<const float4*>(pa_l + c * 32 + kg * 8);
          const float4 a1 = *reinterpret_cast<const float4*>(pa_l + c * 32 + kg * 8 + 4);
          const float4 s0 = *reinterpret_cast<const float4*>(pa_l + 64 + c * 32 + kg * 8);
          const float4 s1 = *reinterpret_cast<const float4*>(pa_l + 64 + c * 32 + kg * 8 + 4);
          const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          float v[8], e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = (float)in[i] * rs * av[i] + sv[i];
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(-1.4426950408889634f * v[i]);
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_rcpf(1.0f + e[i]);
          f16x8 out;
#pragma unroll
          for (int i = 0; i < 8; ++i) out[i] = (f16)(silu ? v[i] * e[i] : v[i]);
          uint4 ow = *reinterpret_cast<const uint4*>(&out);
          ow = ok ? ow : make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(a) = ow;
        }
      }
      __syncthreads();
    }

    // ---- 9 taps x 2 K steps per 32-channel input
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const char* ab = acts + (cur * NCH + c) * CS_ABUF;
      const char* wb = wlds + c * CS_WCH + lane * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const f16x8 af = *reinterpret_cast<const f16x8*>(wb + (tap * 2 + ks) * 1024);
          const f16x8 bf = *reinterpret_cast<const f16x8*>(ab + (ks ? bP[0][dx] ^ 32 : bP[0][dx]) + dy * CS_PITCH);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[0][0], 0, 0, 0);
        }
      }
    }

    if (tc.b != ep_b) {   // (workgroup-uniform, once per image) the epilogue's per-channel operands: bias, post_pa / post_ps of this batch row
      __syncthreads();
      if (tid < 32) cl_epilogue_params<32>(p, tc.b, 0, ep_par, tid);
      __syncthreads();
      ep_b = tc.b;
    }
    cl_epilogue<1, 1, CS_NW, 1, GEN, true>(p, tc, acc, pix_y, pix_x, ep_red, ep_par, wave, 0, half, l31);   // (no global load, no barrier inside)

    if (!more) break;
    // (this tile's stores were issued after the next tile's pieces; every lane of an interior tile stores, so no store is branched over)
    stores_behind = (tc.oy0 + CS_TH <= p.OH && tc.ox0 + CS_TW <= p.OW) ? quads : 0;
    t = t_next;
    tc = tn;
    cur ^= 1;
  }
}

// ---- third form: the DMA form with a TWO-deep pipeline.
// Counters of the first form (tools/gpu_r2_y.sh): 48 % of the wave cycles parked at s_waitcnt / s_barrier — the copy of tile t+1 is issued
// when tile t starts and needed one tile period (~5k cycles) later, which is about the loaded memory latency.  Here a buffer is refilled
// as soon as its tile has been MULTIPLIED (one extra barrier), i.e. with tile t+2 before the epilogue of tile t: every copy has a full
// period plus an epilogue of lead, and two tiles per workgroup are in flight.  All per-tile traffic is direct-to-LDS — also the per-pixel
// statistics of the prologue (global_load_lds_dword into a small LDS table), so the loop holds no compiler-counted load whose wait would
// drain the copy queue; the waits are counted by hand: vmcnt retires in issue order, and the operations younger than the copy being
// awaited are the next tile's pieces (a per-wave constant) and the output stores in between (lower bound, see above).
constexpr int CS_NST = 6;   // 256-byte pieces of a per-pixel statistics table (324 halo pixels -> 384 slots)

__device__ __forceinline__ void cs_wait_vm(int n) {   // s_waitcnt vmcnt(min(n, 12)): fewer outstanding than allowed is always safe
  switch (n < 12 ? n : 12) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
}

template <int NCH, bool PRO, bool GEN>
__global__ __launch_bounds__(64 * CS_NW, NCH == 1 ? 4 : 2) void conv_stream3_kernel(const ImagenIgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const acts = smem;                                       // [2 tiles][NCH][CS_ABUF]
  char* const wlds = smem + 2 * NCH * CS_ABUF;                   // [NCH][18][1 KiB]
  float* const ep_red = reinterpret_cast<float*>(wlds + NCH * CS_WCH);
  float* const ep_par = ep_red + CS_EP_RED;
  float* const aff = ep_par + CS_EP_PAR;                         // [pa 64 | ps 64] of batch row aff_b (+ 128 spare floats)
  float* const stats = aff + 256;                                // [2 tiles][ssq_a | ssq_b][CS_NST * 64] per-pixel statistics of the halo pixels
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned stats0 = lds0 + (unsigned)(reinterpret_cast<char*>(stats) - smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int tilesX = (p.OW + CS_TW - 1) / CS_TW, tilesY = (p.OH + CS_TH - 1) / CS_TH;
  const int total = p.B * tilesY * tilesX;
  auto decode = [&](int t) __attribute__((always_inline)) -> ClTile {
    ClTile c;
    const int tx = t % tilesX;
    t /= tilesX;
    const int ty = t % tilesY;
    c.b = t / tilesY;
    c.oy0 = ty * CS_TH;
    c.ox0 = tx * CS_TW;
    c.n0 = 0;
    return c;
  };
  const size_t wrow = (size_t)p.Cout_pad * 16;
  const char* const zero_src = reinterpret_cast<const char*>(p.w) + (size_t)(NCH * 36) * wrow;   // the packed buffer's zero tail

  int s_r[CS_NJ], s_hx[CS_NJ];
#pragma unroll
  for (int j = 0; j < CS_NJ; ++j) {
    const int hp = ((wave + CS_NW * j) * 64 + lane) >> 2;
    const int r = (hp * 3641) >> 16;
    s_r[j] = hp < CS_NPX ? r : -100000;
    s_hx[j] = hp - r * CS_ITW;
  }
  const int pos = lane & 3;
  // statistics piece of this wave (waves 0-5): halo pixel hp = wave * 64 + lane
  const int st_hp = wave * 64 + lane;
  const int st_r = st_hp < CS_NPX ? (st_hp * 3641) >> 16 : -100000;
  const int st_hx = st_hp - ((st_hp * 3641) >> 16) * CS_ITW;
  const bool has_b = PRO && NCH == 2 && p.ssq_b != nullptr;
  // VMEM operations this wave issues per tile (wave-uniform): the hand-counted waits depend on it
  const int ppt = NCH * (wave < CS_NDMA - 2 * CS_NW ? 3 : 2) + ((PRO && wave < CS_NST) ? (has_b ? 2 : 1) : 0);

  auto issue_tile = [&](const ClTile& tc, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < CS_NJ; ++j) {
      const int d = wave + CS_NW * j;
      if (d >= CS_NDMA) continue;   // (wave-uniform)
      const int gy = tc.oy0 - 1 + s_r[j], gx = tc.ox0 - 1 + s_hx[j];
      const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const int gp = ok ? gy * p.W + gx : 0;
      const int kg = pos ^ ((s_hx[j] >> 1) & 3);
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
      if (wave < CS_NST) {   // (wave-uniform)
        const int gy = tc.oy0 - 1 + st_r, gx = tc.ox0 - 1 + st_hx;
        const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const size_t sp = (size_t)tc.b * (p.H * p.W) + (ok ? gy * p.W + gx : 0);
        const unsigned dst = __builtin_amdgcn_readfirstlane(stats0 + (unsigned)((buf * 2) * CS_NST * 64 + wave * 64) * 4u);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(p.ssq_a + sp), "s"(dst) : "memory");
        if (has_b) {
          const unsigned dstb = __builtin_amdgcn_readfirstlane(stats0 + (unsigned)((buf * 2 + 1) * CS_NST * 64 + wave * 64) * 4u);
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(p.ssq_b + sp), "s"(dstb) : "memory");
        }
      }
    }
  };

  int pix_y[1], pix_x[1], bP[1][3];
  {
    const int tp = wave * 32 + l31;
    const int py = tp / CS_TW, px = tp - py * CS_TW;
    pix_y[0] = py;
    pix_x[0] = px;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = px + dx;
      bP[0][dx] = py * CS_PITCH + hx * 64 + ((half ^ ((hx >> 1) & 3)) << 4);
    }
  }

  const int G = gridDim.x;
  int t = blockIdx.x;
  if (t >= total) return;
  for (int s = wave; s < 18 * NCH; s += CS_NW) {   // weights -> LDS once
    const int c = s / 18, ks = s - 18 * c;
    const char* src = reinterpret_cast<const char*>(p.w) + ((size_t)(c * 36 + 2 * ks + half) * p.Cout_pad + l31) * 16;
    cs_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + 2 * NCH * CS_ABUF + s * 1024));
  }
  ClTile tc = decode(t);
  issue_tile(tc, 0);
  bool next_issued = t + G < total;
  if (next_issued) issue_tile(decode(t + G), 1);
  int cur = 0;
  int ep_b = -1, aff_b = -1;
  int sb1 = 0, sb2 = 0;   // lower bounds of the store instructions of the previous tile's / the tile before's epilogue
  const int quads = (GEN || CL_DBG(8)) ? 0 : min((p.Cout + 7) >> 3, 4);

  while (true) {
    // ---- tile t has landed in buffer `cur`: everything younger than its copy may stay in flight
    cs_wait_vm(sb2 + (next_issued ? ppt : 0) + sb1);
    __syncthreads();

    if constexpr (PRO) {
      if (tc.b != aff_b) {   // (workgroup-uniform, once per image) the prologue's per-channel affine of this batch row
        if (tid < 128) {
          const int ch = tid & 63;
          const size_t o = (size_t)tc.b * p.pstride + ch;
          float a = 0.f;
          if (ch < 32 * NCH) a = tid < 64 ? p.pa[o] : (p.ps ? p.ps[o] : 0.f);
          aff[tid] = a;
        }
        __syncthreads();
        aff_b = tc.b;
      }
      // ---- Block prologue in place (ip.py:675-684): x * rsqrt(ssq) * pa (+ ps) -> SiLU, zero outside the image
      const bool silu = p.act_in == IMAGEN_ACT_SILU;
      const float* st_a = stats + (cur * 2) * CS_NST * 64;
      const float* st_b = st_a + CS_NST * 64;
#pragma unroll
      for (int j = 0; j < CS_NJ; ++j) {
        const int d = wave + CS_NW * j;
        if (d >= CS_NDMA) continue;
        const int S = d * 64 + lane;
        const int hp = S >> 2;
        const int kg = pos ^ ((s_hx[j] >> 1) & 3);
        const int gy = tc.oy0 - 1 + s_r[j], gx = tc.ox0 - 1 + s_hx[j];
        const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const int hpc = hp < CS_NST * 64 ? hp : 0;
        const float q = st_a[hpc] + (has_b ? p.ssq_wb * st_b[hpc] : 0.f);
        const float rs = __builtin_amdgcn_rsqf(fmaxf(q, 1e-24f));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          char* a = acts + (cur * NCH + c) * CS_ABUF + S * 16;
          const f16x8 in = *reinterpret_cast<const f16x8*>(a);
          const float* pa_l = aff + c * 32 + kg * 8;
          f16x8 out;
#pragma unroll
          for (int h4 = 0; h4 < 2; ++h4) {
            const float4 aq = *reinterpret_cast<const float4*>(pa_l + 4 * h4);
            const float4 sq = *reinterpret_cast<const float4*>(pa_l + 64 + 4 * h4);
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

    // ---- 9 taps x 2 K steps per 32-channel input
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const char* ab = acts + (cur * NCH + c) * CS_ABUF;
      const char* wb = wlds + c * CS_WCH + lane * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const f16x8 af = *reinterpret_cast<const f16x8*>(wb + (tap * 2 + ks) * 1024);
          const f16x8 bf = *reinterpret_cast<const f16x8*>(ab + (ks ? bP[0][dx] ^ 32 : bP[0][dx]) + dy * CS_PITCH);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[0][0], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();   // everybody is done with buffer `cur`: refill it with tile t + 2G before this tile's epilogue
    const int t2 = t + 2 * G;
    const bool issue2 = t2 < total;
    if (issue2) issue_tile(decode(t2), cur);

    if (tc.b != ep_b) {   // (workgroup-uniform, once per image) the epilogue's per-channel operands: bias, post_pa / post_ps of this batch row
      __syncthreads();
      if (tid < 32) cl_epilogue_params<32>(p, tc.b, 0, ep_par, tid);
      __syncthreads();
      ep_b = tc.b;
    }
    cl_epilogue<1, 1, CS_NW, 1, GEN, true>(p, tc, acc, pix_y, pix_x, ep_red, ep_par, wave, 0, half, l31);   // (no global load, no barrier inside)

    if (!next_issued) break;
    // next iteration awaits tile t + G (copied one iteration ago): younger than it are the previous tile's stores, the pieces of tile
    // t + 2G just issued and this tile's stores
    sb2 = sb1;
    sb1 = (tc.oy0 + CS_TH <= p.OH && tc.ox0 + CS_TW <= p.OW) ? quads : 0;
    t += G;
    tc = decode(t);
    next_issued = issue2;
    cur ^= 1;
  }
}

// ---- fourth form: the first form on an instruction diet.
// Counters of forms 1-3 (tools/gpu_r2_y.sh): every wave is issuing 24 % of the time and a SIMD holds four of them — the instruction
// issue of the SIMD is saturated by ~320 instructions per wave and tile (134 VALU + 122 SALU + LDS / MFMA / VMEM), which is why the
// deeper pipelines above change nothing: at 32 pixels x 32 channels per wave the per-tile bookkeeping IS the run time.  This form
//   * walks the tile list incrementally (no integer division per tile: (b, ty, tx) advance by the grid stride with two wrap checks);
//   * addresses an INTERIOR tile (halo inside the image — all but the border ring) as one scalar base + one per-lane constant offset
//     per slot: two VALU per DMA piece instead of bounds checks, selects and 64-bit multiplies; border tiles take the general path;
//   * starts the accumulators at the bias (one ds_read_b128 per channel quad) instead of zeroing them and adding the bias afterwards;
//   * stores through a per-lane constant output offset (plain / post_pa epilogue of interior tiles; anything else: conv_epilogue.h).
template <int NCH, bool PRO, bool GEN>
__global__ __launch_bounds__(64 * CS_NW, NCH == 1 ? 4 : 2) void conv_stream4_kernel(const ImagenIgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const acts = smem;                                       // [2 tiles][NCH][CS_ABUF]
  char* const wlds = smem + 2 * NCH * CS_ABUF;                   // [NCH][18][1 KiB]
  float* const ep_red = reinterpret_cast<float*>(wlds + NCH * CS_WCH);
  float* const ep_par = ep_red + CS_EP_RED;                      // [bias 32 | post_pa 32 | post_ps 32 | ...]
  float* const aff = ep_par + CS_EP_PAR;                         // [2 tiles][pa 64 | ps 64]
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

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
  // output offsets of this lane (bytes from the tile's first output pixel): quad q of the 32 couts at + 16 q
  const int yoff = ((pix_y[0] * p.OW + pix_x[0]) * p.ldy + 4 * half) * 2;
  const int qoff = (pix_y[0] * p.OW + pix_x[0]) * 4;

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
  const int quads = (GEN || CL_DBG(8)) ? 0 : min((p.Cout + 7) >> 3, 4);
  const bool lean_ok = !GEN && p.Cout == 32 && !CL_DBG(8);   // the lean epilogue writes all four channel quads

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
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 pa = *reinterpret_cast<const float4*>(ep_par + 32 + 8 * q + 4 * half);
          const float4 ps = *reinterpret_cast<const float4*>(ep_par + 64 + 8 * q + 4 * half);
          const float pav[4] = {pa.x, pa.y, pa.z, pa.w}, psv[4] = {ps.x, ps.y, ps.z, ps.w};
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)silu_f(acc[0][0][4 * q + e] * rsn * pav[e] + psv[e]);
          *reinterpret_cast<f16x4*>(yb + 16 * q) = o;
        }
      } else {
        float ssq = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (f16)acc[0][0][4 * q + e];
            const float r = (float)o[e];
            ssq += r * r;
          }
          *reinterpret_cast<f16x4*>(yb + 16 * q) = o;
        }
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
    stores_behind = lean ? 4 : ((tc.oy0 + CS_TH <= p.OH && tc.ox0 + CS_TW <= p.OW) ? quads : 0);
    c = cn;
    tc = tn;
    cur ^= 1;
  }
}

// ---- second form: REGISTER-staged, two tiles ahead.
// The DMA form above has one tile per workgroup in flight (the second LDS buffer): 41 KB per CU, which at the loaded memory latency
// sustains ~3 TB/s of reads (tools/stream_probe.py: the same rate with and without output stores, with one or two inputs).  LDS cannot
// hold more buffers, registers can: every wave loads its 3 sixteen-byte slots per input of tile t+2 and t+3 into VGPRs (plain global
// loads, the compiler counts them), and hands a set to LDS one tile ahead of its use — raw, or through the Block prologue IN REGISTERS
// (the in-place LDS read-modify-write of the DMA form disappears).  Twice the bytes in flight, no manual vmcnt bookkeeping: a wait for
// a set is a wait for loads issued two iterations ago, older than every store in between.
template <int NCH, bool PRO, bool GEN>
__global__ __launch_bounds__(64 * CS_NW, NCH == 1 ? 4 : 2) void conv_stream2_kernel(const ImagenIgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const acts = smem;                                       // [2 tiles][NCH][CS_ABUF]
  char* const wlds = smem + 2 * NCH * CS_ABUF;                   // [NCH][18][1 KiB]
  float* const ep_red = reinterpret_cast<float*>(wlds + NCH * CS_WCH);
  float* const ep_par = ep_red + CS_EP_RED;
  float* const aff = ep_par + CS_EP_PAR;                         // [pa 64 | ps 64] of batch row aff_b
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int tilesX = (p.OW + CS_TW - 1) / CS_TW, tilesY = (p.OH + CS_TH - 1) / CS_TH;
  const int total = p.B * tilesY * tilesX;
  auto decode = [&](int t) __attribute__((always_inline)) -> ClTile {
    ClTile c;
    const int tx = t % tilesX;
    t /= tilesX;
    const int ty = t % tilesY;
    c.b = t / tilesY;
    c.oy0 = ty * CS_TH;
    c.ox0 = tx * CS_TW;
    c.n0 = 0;
    return c;
  };
  const size_t wrow = (size_t)p.Cout_pad * 16;
  const char* const zero_src = reinterpret_cast<const char*>(p.w) + (size_t)(NCH * 36) * wrow;   // the packed buffer's zero tail

  int s_r[CS_NJ], s_hx[CS_NJ];
#pragma unroll
  for (int j = 0; j < CS_NJ; ++j) {
    const int hp = ((wave + CS_NW * j) * 64 + lane) >> 2;
    const int r = (hp * 3641) >> 16;
    s_r[j] = hp < CS_NPX ? r : -100000;
    s_hx[j] = hp - r * CS_ITW;
  }
  const int pos = lane & 3;

  // a staged tile in registers: 16-byte slots v[input][j], the per-pixel statistics of the slots' pixels (PRO), the in-image mask
  // (plain arrays / scalars handed to the lambdas by reference: a struct of arrays ends up in scratch memory)
#define CS_STAGE_DECL(X) cs_u32x4 v##X[NCH][CS_NJ]; float sa##X[CS_NJ], sb##X[CS_NJ]; unsigned okm##X = 0; int b##X = 0; bool valid##X = false
#define CS_STAGE_ARGS(X) v##X, sa##X, sb##X, okm##X, b##X, valid##X
  auto load_stage = [&](cs_u32x4 (&v)[NCH][CS_NJ], float (&sa)[CS_NJ], float (&sb)[CS_NJ], unsigned& okm, int& sb_b, bool& valid, int t)
      __attribute__((always_inline)) {
    valid = t < total;
    if (!valid) return;   // (workgroup-uniform)
    const ClTile tc = decode(t);
    sb_b = tc.b;
    okm = 0;
    cs_static_for<CS_NJ>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      const int gy = tc.oy0 - 1 + s_r[j], gx = tc.ox0 - 1 + s_hx[j];
      const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const int gp = ok ? gy * p.W + gx : 0;
      const int kg = pos ^ ((s_hx[j] >> 1) & 3);
      const f16* x1 = reinterpret_cast<const f16*>(p.x1) + (size_t)tc.b * p.bs1;
      v[0][j] = *reinterpret_cast<const cs_u32x4*>(ok ? reinterpret_cast<const char*>(x1 + (size_t)gp * p.ld1 + kg * 8) : zero_src);
      if constexpr (NCH == 2) {
        const f16* x2 = reinterpret_cast<const f16*>(p.x2) + (size_t)tc.b * p.bs2;
        v[NCH - 1][j] = *reinterpret_cast<const cs_u32x4*>(ok ? reinterpret_cast<const char*>(x2 + (size_t)gp * p.ld2 + kg * 8) : zero_src);
      }
      if constexpr (PRO) {
        const size_t sp = (size_t)tc.b * (p.H * p.W) + gp;
        sa[j] = p.ssq_a[sp];
        sb[j] = (NCH == 2 && p.ssq_b) ? p.ssq_b[sp] : 0.f;
        if (ok) okm |= 1u << j;
      }
    });
  };
  int aff_b = -1;   // batch row whose (pa | ps) sit in LDS
  auto write_stage = [&](cs_u32x4 (&v)[NCH][CS_NJ], float (&sa)[CS_NJ], float (&sb)[CS_NJ], unsigned& okm, int& S_b, bool& valid, int buf)
      __attribute__((always_inline)) {
    if (!valid) return;
    if constexpr (PRO) {
      if (S_b != aff_b) {   // (workgroup-uniform, once per image) refresh the affine row
        __syncthreads();
        if (tid < 128) {
          const int ch = tid & 63;
          const size_t o = (size_t)S_b * p.pstride + ch;
          float a = 0.f;
          if (ch < 32 * NCH) a = tid < 64 ? p.pa[o] : (p.ps ? p.ps[o] : 0.f);
          aff[tid] = a;
        }
        __syncthreads();
        aff_b = S_b;
      }
    }
    const bool silu = p.act_in == IMAGEN_ACT_SILU;
    cs_static_for<CS_NJ>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      const int d = wave + CS_NW * j;
      if (d >= CS_NDMA) return;   // (wave-uniform)
      const int off = (d * 64 + lane) * 16;
      if constexpr (!PRO) {
        cs_static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
          constexpr int c = decltype(cc)::value;
          *reinterpret_cast<cs_u32x4*>(acts + (buf * NCH + c) * CS_ABUF + off) = v[c][j];
        });
      } else {
        const int kg = pos ^ ((s_hx[j] >> 1) & 3);
        const float q = sa[j] + p.ssq_wb * sb[j];
        const float rs = __builtin_amdgcn_rsqf(fmaxf(q, 1e-24f));
        const bool ok = (okm >> j) & 1u;
        cs_static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
          constexpr int c = decltype(cc)::value;
          const f16x8 in = __builtin_bit_cast(f16x8, v[c][j]);
          const float* pa_l = aff + c * 32 + kg * 8;
          f16x8 out;
          // (four elements at a time: the 16 affine operands and 8 + 8 temporaries of a whole item at once do not fit beside two staged
          // tiles under the 128-register budget of two resident workgroups)
#pragma unroll
          for (int h4 = 0; h4 < 2; ++h4) {
            const float4 aq = *reinterpret_cast<const float4*>(pa_l + 4 * h4);
            const float4 sq = *reinterpret_cast<const float4*>(pa_l + 64 + 4 * h4);
            const float av[4] = {aq.x, aq.y, aq.z, aq.w}, sv[4] = {sq.x, sq.y, sq.z, sq.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float t = (float)in[4 * h4 + i] * rs * av[i] + sv[i];
              const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
              out[4 * h4 + i] = (f16)(silu ? t * sg : t);
            }
          }
          cs_u32x4 ow = __builtin_bit_cast(cs_u32x4, out);
          if (!ok) ow = cs_u32x4{0u, 0u, 0u, 0u};
          *reinterpret_cast<cs_u32x4*>(acts + (buf * NCH + c) * CS_ABUF + off) = ow;
        });
      }
    });
  };

  int pix_y[1], pix_x[1], bP[1][3];
  {
    const int tp = wave * 32 + l31;
    const int py = tp / CS_TW, px = tp - py * CS_TW;
    pix_y[0] = py;
    pix_x[0] = px;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = px + dx;
      bP[0][dx] = py * CS_PITCH + hx * 64 + ((half ^ ((hx >> 1) & 3)) << 4);
    }
  }
  int ep_b = -1;   // batch row whose epilogue operands sit in ep_par
  auto compute_store = [&](int t, int buf) __attribute__((always_inline)) {
    const ClTile tc = decode(t);
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const char* ab = acts + (buf * NCH + c) * CS_ABUF;
      const char* wb = wlds + c * CS_WCH + lane * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const f16x8 af = *reinterpret_cast<const f16x8*>(wb + (tap * 2 + ks) * 1024);
          const f16x8 bf = *reinterpret_cast<const f16x8*>(ab + (ks ? bP[0][dx] ^ 32 : bP[0][dx]) + dy * CS_PITCH);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[0][0], 0, 0, 0);
        }
        if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);   // at most one tap row of fragments (12 x 4 registers) hoisted: the staged tiles need the rest
      }
    }
    if (tc.b != ep_b) {   // (workgroup-uniform, once per image) the epilogue's per-channel operands: bias, post_pa / post_ps of this batch row
      __syncthreads();
      if (tid < 32) cl_epilogue_params<32>(p, tc.b, 0, ep_par, tid);
      __syncthreads();
      ep_b = tc.b;
    }
    // no global load and no barrier inside: a load here would be younger than the staged tiles' loads, and waiting for it would drain them
    cl_epilogue<1, 1, CS_NW, 1, GEN, true>(p, tc, acc, pix_y, pix_x, ep_red, ep_par, wave, 0, half, l31);
  };

  const int t0 = blockIdx.x, G = gridDim.x;
  if (t0 >= total) return;
  for (int s = wave; s < 18 * NCH; s += CS_NW) {   // weights -> LDS once (direct-to-LDS copies)
    const int c = s / 18, ks = s - 18 * c;
    const char* src = reinterpret_cast<const char*>(p.w) + ((size_t)(c * 36 + 2 * ks + half) * p.Cout_pad + l31) * 16;
    cs_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + 2 * NCH * CS_ABUF + s * 1024));
  }
  CS_STAGE_DECL(A);   // A: tiles t0, t0 + 2G, ... (buffer 0); B: t0 + G, t0 + 3G, ... (buffer 1)
  CS_STAGE_DECL(B);
  load_stage(CS_STAGE_ARGS(A), t0);
  load_stage(CS_STAGE_ARGS(B), t0 + G);
  write_stage(CS_STAGE_ARGS(A), 0);
  load_stage(CS_STAGE_ARGS(A), t0 + 2 * G);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight copies (asm: invisible to the compiler's own counting)
  __syncthreads();
  for (int t = t0; t < total; t += 2 * G) {
    // ---- tile t (buffer 0); hand tile t + G to buffer 1 and start loading tile t + 3G into its registers
    write_stage(CS_STAGE_ARGS(B), 1);
    load_stage(CS_STAGE_ARGS(B), t + 3 * G);
    compute_store(t, 0);
    __syncthreads();   // everybody is done with buffer 0 and with the epilogue scratch; buffer 1 is complete
    if (t + G >= total) break;
    // ---- tile t + G (buffer 1); hand tile t + 2G to buffer 0, load tile t + 4G
    write_stage(CS_STAGE_ARGS(A), 0);
    load_stage(CS_STAGE_ARGS(A), t + 4 * G);
    compute_store(t + G, 1);
    __syncthreads();
  }
}

template <int NCH, bool PRO, bool GEN>
int cs_launch_gen(const ImagenIgemmParams& p, hipStream_t s) {
  static const int form = [] { const char* e = getenv("IMAGEN_STREAM_FORM"); return e ? atoi(e) : 4; }();   // A/B: 1 = DMA, 2 = register-staged, 3 = two-deep DMA, 4 = lean DMA
  auto kern = form == 1 ? conv_stream_kernel<NCH, PRO, GEN> : form == 2 ? conv_stream2_kernel<NCH, PRO, GEN>
              : form == 3 ? conv_stream3_kernel<NCH, PRO, GEN> : conv_stream4_kernel<NCH, PRO, GEN>;
  constexpr size_t lds = cs_lds_bytes(NCH);
  static bool attr_done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_done[dev]) {
    for (auto k : {conv_stream_kernel<NCH, PRO, GEN>, conv_stream2_kernel<NCH, PRO, GEN>, conv_stream3_kernel<NCH, PRO, GEN>,
                   conv_stream4_kernel<NCH, PRO, GEN>}) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { imagen_set_error("conv_stream: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    }
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  const int tilesX = (p.OW + CS_TW - 1) / CS_TW, tilesY = (p.OH + CS_TH - 1) / CS_TH;
  const int total = p.B * tilesX * tilesY;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  static const int per_cu_env = [] { const char* e = getenv("IMAGEN_STREAM_WG_PER_CU"); return e ? atoi(e) : 0; }();   // probe knob
  const int per_cu = per_cu_env > 0 ? per_cu_env : (NCH == 1 ? 2 : 1);
  const int resident = std::max(1, cus) * per_cu;
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
  IMAGEN_CHECK(!p.gca_part, "conv_stream: GlobalContext partials are emitted by the other families only");
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
