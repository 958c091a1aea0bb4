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

__device__ __forceinline__ void cs_dma16(const void* gsrc, unsigned lds_dst) {   // lane l -> LDS bytes [dst + 16 l, +16)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr size_t cs_lds_bytes(int nch) {
  return (size_t)2 * nch * CS_ABUF + (size_t)nch * CS_WCH + (size_t)(CS_EP_RED + CS_EP_PAR) * sizeof(float) + (size_t)2 * 2 * 64 * sizeof(float) + 16;
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
  int stores_behind = 0;   // lower bound of the store instructions this wave issued AFTER the DMA pieces being waited for
  const int quads = GEN ? 0 : min((p.Cout + 7) >> 3, 4);   // plain / post epilogue: one store per channel quad (both lane halves at once)

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

    cl_epilogue<1, 1, CS_NW, 1, GEN>(p, tc, acc, pix_y, pix_x, ep_red, ep_par, wave, 0, half, l31);

    if (!more) break;
    // (this tile's stores were issued after the next tile's pieces; every lane of an interior tile stores, so no store is branched over)
    stores_behind = (tc.oy0 + CS_TH <= p.OH && tc.ox0 + CS_TW <= p.OW) ? quads : 0;
    t = t_next;
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
