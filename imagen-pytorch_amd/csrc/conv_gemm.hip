// conv_gemm.hip — the pointwise (1x1) GEMM of the token / small-map layers (eighth igemm family, gfx950): every nn.Linear of the transformer
// blocks (qkv, to_q, to_out, FeedForward: ip.py:521-532, 782-791, 972-980), the res_conv of the 32^2 level (ip.py:741) and the pixel-shuffle
// GEMMs, i.e. 1x1 stride-1 launches with 128-640 input channels and 128-640 output channels over 8k-64k rows.
//
// Why it exists (round 4, profiles/r04_bench_n1.json: per_kernel "hbm:cfg10"): on the wave-specialised kernel these 35 launches are the
// largest group of the step pair — 0.79 ms at 142 TFLOP/s / 1.08 TB/s, 0.14 of either roof; a [256 -> 640 @16k rows] GEMM (5.4 GFLOP,
// 30 MB) takes 30 us.  DESIGN 9.1 measured why: that kernel's skeleton (persistent roles, a barrier hand-over per 64-pixel phase, a generic
// epilogue with dependent round trips) costs more than the work.  conv_pw.hip fixed the same problem for K <= 192 by keeping the whole
// weight matrix in registers; these layers do not fit.  This kernel is the plain tiled GEMM instead:
//   * one workgroup of 4 waves per (128 rows x 128 output channels) tile, three workgroups per CU (32 KB of LDS, 160 VGPRs each), no persistence —
//     128-640 workgroups per launch, the slabs of a row tile next to each other on one XCD (the rows are re-read from that XCD's L2);
//   * wave = 64 rows x 64 couts (2 x 2 fragments of v_mfma_f32_32x32x16_f16: 1 KB of ds_read_b128 per MFMA);
//   * K loop over 32-channel chunks, both operands register-staged TWO chunks ahead (two register sets, chunk c + 2 requested while chunk c
//     is multiplied and chunk c + 1 written) into a two-slot LDS ring, one barrier per chunk; rows land in conv_pw.hip's swizzled order,
//     weights as contiguous [8-channel group][128 couts][8] images — both fragment reads conflict-free;
//   * the LayerNorm prologue (x - mu[row]) * rs[row] * pa[c] + ps[c] (each factor optional) is applied in registers between the global
//     load and the LDS write: once per element, 16 elements per thread and chunk;
//   * every epilogue of the contract through conv_epilogue.h (bias, GELU / SiLU, gate * addend, residual, pixel shuffle, fp32 NCHW,
//     ssq_out / post_pa / GlobalContext partials where one tile covers all couts).
// Contract: ImagenIgemmParams with KH = KW = 1, stride 1, pad 0; C1, C2 multiples of 32, Cin_pad = C1 + C2; act_in NONE; no ssq_a
// statistics (mu / rs only).  The packed weight layout of a 1x1 layer is the same for every G >= 2 (consecutive 8-channel group rows).
#include <algorithm>
#include <type_traits>
#include <utility>
#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int CG_TP = 128, CG_BN = 128;          // rows (pixels) and output channels per workgroup tile
constexpr int CG_ABYTES = CG_TP * 64;            // one 32-channel chunk of the tile's rows
constexpr int CG_WBYTES = 4 * CG_BN * 16;        // one 32-channel chunk of the slab's weights: [4 groups][128 couts][8 halves]
constexpr int CG_SLOT = CG_ABYTES + CG_WBYTES;   // 16 KB
constexpr int CG_EP_PAR = 4 * CG_BN + 2 * 2 * 64 + 8 + 2 * (CG_BN + 4);   // floats (conv_epilogue.h, MI = NI = WM = WN = 2)
constexpr int CG_EP_RED = 2 * 2 * 64;                                      // floats

__device__ __forceinline__ int cg_swz(int px) { return (px >> 2) & 3; }

constexpr size_t cg_lds_bytes(int K, bool aff) {
  const size_t ring = 2 * (size_t)CG_SLOT, ep = (size_t)(CG_EP_PAR + CG_EP_RED) * sizeof(float);
  return (ring > ep ? ring : ep) + (aff ? (size_t)2 * K * sizeof(float) : 0);
}

// PRO: the LayerNorm prologue on the rows.  GEN: generic epilogue (conv_epilogue.h).
template <bool PRO, bool GEN>
__global__ __launch_bounds__(256, 3) void conv_gemm_kernel(const ImagenIgemmParams p) {
  constexpr int MI = 2, NI = 2, WM = 2, WN = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const aff = reinterpret_cast<float*>(smem + 2 * CG_SLOT);   // [pa K | ps K] of this tile's batch row (PRO)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile of this workgroup: contiguous ranges per XCD (blockIdx round-robin over 8 XCDs), cout slab fastest
  const int tilesX = (p.OW + p.TW - 1) / p.TW, tilesY = (p.OH + p.TH - 1) / p.TH;
  const int tilesN = (p.Cout + CG_BN - 1) / CG_BN;
  ClTile tc;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int nt = t % tilesN;
    t /= tilesN;
    const int tx = t % tilesX;
    t /= tilesX;
    const int ty = t % tilesY;
    tc.b = t / tilesY;
    tc.oy0 = ty * p.TH;
    tc.ox0 = tx * p.TW;
    tc.n0 = nt * CG_BN;
  }
  const int NC = p.Cin_pad >> 5;
  const int n1 = p.C1 >> 5;

  // ---- staging roles: slots S = tid + 256 j (j < 2).  Rows: pixel S >> 2 of the tile, position S & 3 holds channel group (S & 3) ^ swz;
  //      weights: group S >> 7, cout S & 127 of the slab
  int a_px[2], a_goff[2];
  const f16* a_row1[2];
  const f16* a_row2[2];
  bool a_ok[2];
  float a_mu[2], a_rs[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int S = tid + 256 * j;
    const int tp = S >> 2;
    const int py = tp / p.TW, px = tp - py * p.TW;
    const int oy = tc.oy0 + py, ox = tc.ox0 + px;
    a_ok[j] = oy < p.OH && ox < p.OW;
    const int pix = a_ok[j] ? oy * p.OW + ox : 0;
    a_px[j] = tp;
    a_goff[j] = ((S & 3) ^ cg_swz(tp)) * 8;
    a_row1[j] = reinterpret_cast<const f16*>(p.x1) + (size_t)tc.b * p.bs1 + (size_t)pix * p.ld1 + a_goff[j];
    a_row2[j] = p.x2 ? reinterpret_cast<const f16*>(p.x2) + (size_t)tc.b * p.bs2 + (size_t)pix * p.ld2 + a_goff[j] : a_row1[j];
    a_mu[j] = 0.f;
    a_rs[j] = 1.f;
    if constexpr (PRO) {
      const size_t row = (size_t)tc.b * (p.OH * p.OW) + pix;
      if (p.mu) a_mu[j] = p.mu[row];
      if (p.rs) a_rs[j] = p.rs[row];
    }
  }
  const char* w_src[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int S = tid + 256 * j;
    w_src[j] = reinterpret_cast<const char*>(p.w) + ((size_t)(S >> 7) * p.Cout_pad + tc.n0 + (S & 127)) * 16;
  }
  const size_t w_chunk = (size_t)4 * p.Cout_pad * 16;   // bytes between the same group of consecutive chunks

  struct Regs { uint4 a0, a1, w0, w1; };
  auto request = [&](Regs& R, int c) __attribute__((always_inline)) {   // (chunks past the end re-read the last one: written to a slot nobody multiplies)
    const int cc = c < NC ? c : NC - 1;
    const bool from1 = cc < n1;                                          // (workgroup-uniform)
    const int coff = (from1 ? cc : cc - n1) * 32;
    R.a0 = *reinterpret_cast<const uint4*>((from1 ? a_row1[0] : a_row2[0]) + coff);
    R.a1 = *reinterpret_cast<const uint4*>((from1 ? a_row1[1] : a_row2[1]) + coff);
    R.w0 = *reinterpret_cast<const uint4*>(w_src[0] + (size_t)cc * w_chunk);
    R.w1 = *reinterpret_cast<const uint4*>(w_src[1] + (size_t)cc * w_chunk);
  };
  auto transform = [&](uint4 raw, int j, int c) __attribute__((always_inline)) -> uint4 {
    if (!a_ok[j] || c >= NC) return make_uint4(0, 0, 0, 0);   // (outside the image / past the last chunk: zero rows)
    if constexpr (!PRO) {
      return raw;
    } else {
      const int cc = c < NC ? c : NC - 1;
      const f16x8 x = __builtin_bit_cast(f16x8, raw);
      const float* pa = aff + cc * 32 + a_goff[j];
      const float* ps = pa + p.Cin_pad;
      const float4 g0 = *reinterpret_cast<const float4*>(pa), g1 = *reinterpret_cast<const float4*>(pa + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(ps), s1 = *reinterpret_cast<const float4*>(ps + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, sh[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)(((float)x[e] - a_mu[j]) * a_rs[j] * g[e] + sh[e]);
      return __builtin_bit_cast(uint4, o);
    }
  };
  auto stage = [&](const Regs& R, int c, int slot) __attribute__((always_inline)) {
    char* sb = smem + slot * CG_SLOT;
    *reinterpret_cast<uint4*>(sb + tid * 16) = transform(R.a0, 0, c);
    *reinterpret_cast<uint4*>(sb + (tid + 256) * 16) = transform(R.a1, 1, c);
    *reinterpret_cast<uint4*>(sb + CG_ABYTES + tid * 16) = R.w0;
    *reinterpret_cast<uint4*>(sb + CG_ABYTES + (tid + 256) * 16) = R.w1;
  };

  // ---- MFMA side: lane = row (B operand) / cout (A operand)
  int pix_y[MI], pix_x[MI], b_off[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int tp = wm * 64 + mi * 32 + l31;
    pix_y[mi] = tp / p.TW;
    pix_x[mi] = tp - pix_y[mi] * p.TW;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_off[mi][ks] = tp * 64 + (((2 * ks + half) ^ cg_swz(tp)) << 4);
  }
  const int a_off = CG_ABYTES + (half * CG_BN + wn * 64 + l31) * 16;   // + ks * 2 * CG_BN * 16 + ni * 512

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.0f;

  Regs R0, R1;
  request(R0, 0);
  request(R1, 1);
  if constexpr (PRO) {   // the per-channel affine of this batch row -> LDS (absent factors: neutral constants)
    for (int i = tid; i < p.Cin_pad; i += 256) {
      aff[i] = p.pa ? p.pa[(size_t)tc.b * p.pstride + i] : 1.0f;
      aff[p.Cin_pad + i] = p.ps ? p.ps[(size_t)tc.b * p.pstride + i] : 0.0f;
    }
    __syncthreads();
  }
  stage(R0, 0, 0);
  __syncthreads();

  // chunk step i: request chunk i + 2 into the set chunk i came from, multiply slot i & 1, write chunk i + 1 (requested a step ago) to the
  // other slot, barrier
  auto step = [&](int i, Regs& Rload, const Regs& Rwrite) __attribute__((always_inline)) {
    request(Rload, i + 2);
    __builtin_amdgcn_sched_barrier(0);   // (the requests stay HERE: the scheduler otherwise sinks them behind the LDS writes below — one step of latency cover instead of two)
    const char* sb = smem + (i & 1) * CG_SLOT;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 af[NI], bf[MI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) af[ni] = *reinterpret_cast<const f16x8*>(sb + a_off + ks * 2 * CG_BN * 16 + ni * 512);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) bf[mi] = *reinterpret_cast<const f16x8*>(sb + b_off[mi][ks]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ni], bf[mi], acc[ni][mi], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    stage(Rwrite, i + 1, (i + 1) & 1);
    IMAGEN_LGKM0_BARRIER();   // (LDS traffic retired + workgroup barrier; the requests stay in flight)
  };
  // (two steps per trip with the sets swapped by NAME: a conditional second step lets the compiler roll the pair into one body that rotates
  // the sets through v_mov / v_cndmask — of registers with loads in flight, i.e. vmcnt(0) at the top of every step)
  // An odd chunk count runs one step more on zero rows (stage() writes zeros past the end): a separate tail step and a loop that may
  // run zero times each cost a copy of the 64 accumulator registers at the merge in front of the generic epilogue — 130 dwords of scratch.
  int i = 0;
  do {
    step(i, R0, R1);
    step(i + 1, R1, R0);
    i += 2;
  } while (i < NC);

  // ---- epilogue (its scratch aliases the ring: every wave is behind the last step's barrier, the stray write of that step included)
  float* ep_par = reinterpret_cast<float*>(smem);
  float* ep_red = ep_par + CG_EP_PAR;
  cl_epilogue<MI, NI, WM, WN, GEN>(p, tc, acc, pix_y, pix_x, ep_red, ep_par, wm, wn, half, l31);
}

template <bool PRO, bool GEN>
int cg_launch(const ImagenIgemmParams& p, hipStream_t s) {
  auto kern = conv_gemm_kernel<PRO, GEN>;
  const size_t lds = cg_lds_bytes(p.Cin_pad, PRO);
  static bool attr_done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("conv_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  const int tilesX = (p.OW + p.TW - 1) / p.TW, tilesY = (p.OH + p.TH - 1) / p.TH, tilesN = (p.Cout + CG_BN - 1) / CG_BN;
  const int total = p.B * tilesY * tilesX * tilesN;
  hipLaunchKernelGGL(kern, dim3(total), dim3(256), lds, s, p);
  return imagen_hip_status("conv_gemm launch");
}

}  // namespace

int imagen_conv_gemm_num_configs() { return 1; }

int imagen_conv_gemm_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (idx != 0) return -1;
  if (tile_pixels) *tile_pixels = CG_TP;
  if (tile_cout) *tile_cout = CG_BN;
  if (kgroups) *kgroups = 4;
  return 0;
}

long imagen_conv_gemm_lds_bytes(int idx, int KH, int KW, int TH, int TW) {
  if (idx != 0 || KH != 1 || KW != 1 || TH * TW != CG_TP) return -1;
  return (long)cg_lds_bytes(1024, true);   // (upper bound: the prologue's affine tables grow with Cin)
}

int launch_conv_gemm(const ImagenIgemmParams* pp, int idx, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  IMAGEN_CHECK(idx == 0, "conv_gemm: bad cfg index %d", idx);
  IMAGEN_CHECK(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.OH == p.H && p.OW == p.W, "conv_gemm: 1x1 stride-1 convolutions only");
  IMAGEN_CHECK(p.TH * p.TW == CG_TP && p.TH >= 1 && p.TW >= 1, "conv_gemm: 128-pixel tiles (got %dx%d)", p.TH, p.TW);
  IMAGEN_CHECK(p.C1 % 32 == 0 && p.C2 % 32 == 0 && p.C1 > 0 && p.Cin_pad == p.C1 + p.C2 && (p.C2 == 0 || p.x2) && p.Cin_pad <= 2048,
               "conv_gemm: inputs in 32-channel chunks, Cin_pad = C1 + C2 <= 2048 (got %d + %d, padded %d)", p.C1, p.C2, p.Cin_pad);
  IMAGEN_CHECK(p.ld1 % 8 == 0 && p.bs1 % 8 == 0 && (p.C2 == 0 || (p.ld2 % 8 == 0 && p.bs2 % 8 == 0)) && ((size_t)p.x1 & 15) == 0 && ((size_t)p.x2 & 15) == 0,
               "conv_gemm: input rows must be 16-byte aligned");
  IMAGEN_CHECK(p.Cout_pad % CG_BN == 0, "conv_gemm: Cout_pad %d not a multiple of %d", p.Cout_pad, CG_BN);
  IMAGEN_CHECK(!p.ssq_a && !p.ssq_b && p.act_in == IMAGEN_ACT_NONE, "conv_gemm: the prologue is (x - mu) * rs * pa + ps (no ssq statistics, no input activation)");
  IMAGEN_CHECK(!p.mu || p.rs, "conv_gemm: mu needs rs");
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NCHW_F32 || p.Cout % 4 == 0, "conv_gemm: Cout %d must be a multiple of 4", p.Cout);
  IMAGEN_CHECK(p.out_mode != IMAGEN_OUT_PIXEL_SHUFFLE || p.Cout % 16 == 0, "conv_gemm: pixel-shuffle needs Cout %% 16 == 0");
  IMAGEN_CHECK(!(p.addend && p.res), "conv_gemm: addend and residual are mutually exclusive");
  IMAGEN_CHECK(!p.addend || p.gate, "conv_gemm: addend requires gate");
  const bool full = p.ssq_out || p.post_pa || p.gca_part;
  IMAGEN_CHECK(!full || p.Cout <= CG_BN, "conv_gemm: ssq_out / post_pa / gca_part need one tile over all %d couts", p.Cout);
  IMAGEN_CHECK(!p.post_pa || (p.post_ps && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res && !p.ssq_out && p.act_out == IMAGEN_ACT_NONE),
               "conv_gemm: post_pa needs post_ps and a plain NHWC output");
  IMAGEN_CHECK(!p.gca_part || (p.gca_wk && !p.post_pa && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res && p.act_out == IMAGEN_ACT_NONE),
               "conv_gemm: gca_part needs gca_wk and a plain NHWC output");
  const bool pro = p.mu || p.rs || p.pa || p.ps;
  const bool plain = p.act_out == IMAGEN_ACT_NONE && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res;
  IMAGEN_CHECK((long)cg_lds_bytes(p.Cin_pad, pro) <= 160 * 1024, "conv_gemm: LDS");
  if (pro) return plain ? cg_launch<true, false>(p, s) : cg_launch<true, true>(p, s);
  return plain ? cg_launch<false, false>(p, s) : cg_launch<false, true>(p, s);
}
