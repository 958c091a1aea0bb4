// conv_pw.hip — the streaming pointwise (1x1) convolution of the up path's res_conv (ip.py:732, 753-757) at the large feature maps
// (fifth igemm family, gfx950):  y = conv1x1(cat(x1, x2)) + bias + gate[b, :] * addend  (| + res), NHWC fp16, optional per-pixel sum of
// squares of the stored output.  32 | 64 output channels from 64 | 96 input channels: 60-120 FLOP per byte — HBM-bound by a factor of
// three to five — yet the wave-specialised kernel ran these launches at 3.3-3.8 TB/s of algorithmic traffic.
//
// Why it exists (round 3, calls I / J: profiles/r03_j_graph_profile_skeleton.txt): with no producer work, no MFMAs and no stores a
// [64->32 k1 @256^2] launch of igemm.hip keeps 47 of its 75 us — sixteen tiles per persistent workgroup, each a barrier hand-over plus a
// generic epilogue with its own dependent round trip for the gate * addend operand, two workgroups per CU.  Nothing in that kernel can
// request the operand early: the consumers' loads retire in order with their weight ring.  Here
//   * the WEIGHTS LIVE IN REGISTERS (K <= 96 x N <= 64: 4-12 A fragments per wave, loaded once per workgroup), so a wave's memory queue
//     carries nothing but its own tile traffic and the compiler's counted waits are exact (no inline-asm copies in this file);
//   * every wave requests the NEXT tile's input rows AND its epilogue operand rows before it touches the current tile: a whole tile
//     period (MFMAs + epilogue + stores) of latency hidden for both, one workgroup barrier per tile;
//   * the input tile (256 consecutive pixels x 32-channel chunks) goes registers -> LDS in the bank-swizzled order the MFMA B fragment
//     reads back conflict-free (slot = channel group ^ ((pixel >> 2) & 3): checked exhaustively for all eight waves);
//   * operands and outputs move in 16-byte pieces (imagen_pair_quads / imagen_unpair_quads, common.h).
// Contract: ImagenIgemmParams with KH = KW = 1, stride 1, no prologue, plain NHWC output, C1 and C2 multiples of 32.
#include <algorithm>
#include "common.h"

namespace {

constexpr int PW_NW = 8;                 // waves per workgroup
// WN: waves that split the output channels of a pixel block (1: every wave owns 32 pixels x all couts; 2: 32 pixels x half the couts —
// twice the couts per workgroup at half the pixels per tile, for the 128-channel res_conv of the 64^2 level).  A tile is 32 * 8 / WN
// consecutive pixels of the image's row-major order.

__device__ __forceinline__ int pw_swz(int px) { return (px >> 2) & 3; }

template <int NI, int KCH, int WN>
__global__ __launch_bounds__(64 * PW_NW, 2) void conv_pw_kernel(const ImagenIgemmParams p) {
  constexpr int PW_TP = 32 * PW_NW / WN;   // pixels per tile
  constexpr int PW_CH = PW_TP * 64;        // LDS bytes of one 32-channel chunk of a tile
  constexpr int NJ = PW_TP * 4 / 512;      // 16-byte staging slots per thread and chunk
  constexpr int BN = 32 * NI * WN;         // output channels per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const acts = smem;                                                       // [2 tiles][KCH][PW_CH]
  float* const par = reinterpret_cast<float*>(smem + 2 * KCH * PW_CH);           // [bias BN | gate BN | ssq partials PW_TP (WN = 2)]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wpx = wave / WN, wn = wave % WN;      // pixel block / cout group of this wave
  const int co0 = wn * 32 * NI;                   // first output channel of this wave
  const int HW = p.OH * p.OW;
  const int tiles_img = (HW + PW_TP - 1) / PW_TP;
  const int total = p.B * tiles_img;
  int t = blockIdx.x;
  if (t >= total) return;

  // ---- weights: A fragment of K step s (input channels 16 s .. 16 s + 15) and cout block ni, straight from the packed buffer
  f16x8 areg[2 * KCH][NI];
  {
    const f16x8* wl = reinterpret_cast<const f16x8*>(p.w) + (size_t)half * p.Cout_pad + co0 + l31;
#pragma unroll
    for (int s = 0; s < 2 * KCH; ++s)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) areg[s][ni] = wl[(size_t)(2 * s) * p.Cout_pad + ni * 32];
  }

  // ---- staging: thread -> slots S = tid + 512 j (j < NJ) of every chunk: halo-free, so slot = (pixel S >> 2, position S & 3) and the
  //      thread fetches channel group (S & 3) ^ swz(pixel) of that pixel's chunk
  const int n1 = p.C1 >> 5;                       // chunks that come from x1
  int s_px[NJ], s_goff[NJ];                       // pixel inside the tile, channel offset of the group inside the chunk (elements)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int S = tid + 512 * j;
    s_px[j] = S >> 2;
    s_goff[j] = ((S & 3) ^ pw_swz(S >> 2)) * 8;
  }
  const f16* const x1 = reinterpret_cast<const f16*>(p.x1);
  const f16* const x2 = reinterpret_cast<const f16*>(p.x2);
  const f16* const eop = p.addend ? reinterpret_cast<const f16*>(p.addend) : reinterpret_cast<const f16*>(p.res);
  const int eld = p.addend ? p.ld_add : p.ld_res;
  const int ebs = p.addend ? p.bs_add : p.bs_res;

  struct Next {
    uint4 x[KCH][NJ];             // the tile's input rows, as staged
    imagen_u32x4 op[NI][2];       // this lane's epilogue operand pieces: couts co0 + ni * 32 + 8 qp + 16 half .. + 7 of its pixel
  };
  auto request = [&](Next& N, int b, int pix0) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
      const bool from1 = c < n1;                  // (workgroup-uniform)
      const f16* base = from1 ? x1 + (size_t)b * p.bs1 + c * 32 : x2 + (size_t)b * p.bs2 + (c - n1) * 32;
      const int ld = from1 ? p.ld1 : p.ld2;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int px = pix0 + s_px[j];
        const bool ok = px < HW;
        const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)(ok ? px : 0) * ld + s_goff[j]);
        N.x[c][j] = ok ? v : make_uint4(0, 0, 0, 0);
      }
    }
    if (eop) {
      const int px = pix0 + wpx * 32 + l31;
      const f16* row = eop + (size_t)b * ebs + (size_t)(px < HW ? px : 0) * eld;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int cx = co0 + ni * 32 + 8 * qp + 16 * half;
          N.op[ni][qp] = *reinterpret_cast<const imagen_u32x4*>(row + (cx < p.Cout ? cx : 0));
        }
    }
  };
  auto stage = [&](const Next& N, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < KCH; ++c)
#pragma unroll
      for (int j = 0; j < NJ; ++j) *reinterpret_cast<uint4*>(acts + (buf * KCH + c) * PW_CH + (tid + 512 * j) * 16) = N.x[c][j];
  };

  const int px_l = wpx * 32 + l31;                                    // this lane's pixel inside the tile (MFMA N dimension)
  const int b_off0 = px_l * 64 + ((half ^ pw_swz(px_l)) << 4);        // B fragment of K step 0 of a chunk; K step 1: channel groups 2 + half
  const int b_off1 = px_l * 64 + (((2 + half) ^ pw_swz(px_l)) << 4);

  int b = t / tiles_img, pix0 = (t - b * tiles_img) * PW_TP;
  Next cur_n;
  request(cur_n, b, pix0);
  stage(cur_n, 0);
  int ep_b = -1, cur = 0;
  imagen_u32x4 op_cur[NI][2];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) op_cur[ni][qp] = cur_n.op[ni][qp];

  while (true) {
    if (b != ep_b) {   // (workgroup-uniform, once per image) per-channel epilogue operands of this image
      __syncthreads();
      if (tid < BN) {
        par[tid] = (p.bias && tid < p.Cout_pad) ? p.bias[tid] : 0.0f;
        par[BN + tid] = (p.addend && tid < p.Cout) ? p.gate[(size_t)b * p.gate_stride + tid] : 1.0f;
      }
      ep_b = b;
    }
    __syncthreads();   // the tile staged at the end of the last iteration (and the operands above) are visible
    const int tn = t + gridDim.x;
    const bool more = tn < total;
    const int bn = more ? tn / tiles_img : b;
    const int pixn = more ? (tn - bn * tiles_img) * PW_TP : pix0;
    Next nx;
    if (more) request(nx, bn, pixn);    // next tile's rows and operands: in flight across this tile's MFMAs, epilogue and stores

    f32x16 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][r] = 0.0f;
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
      const char* ab = acts + (cur * KCH + c) * PW_CH;
      const f16x8 b0 = *reinterpret_cast<const f16x8*>(ab + b_off0);
      const f16x8 b1 = *reinterpret_cast<const f16x8*>(ab + b_off1);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[2 * c][ni], b0, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[2 * c + 1][ni], b1, acc[ni], 0, 0, 0);
      }
    }

    // ---- epilogue: lane = pixel; register quad q of a 32-cout block holds couts 8 q + 4 half + {0..3}
    {
      const int px = pix0 + px_l;
      const bool px_ok = px < HW;
      f16* yrow = reinterpret_cast<f16*>(p.y) + (size_t)b * p.bsy + (size_t)(px_ok ? px : 0) * p.ldy;
      float ssq = 0.0f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          f16x4 ad[2] = {f16x4{}, f16x4{}};
          if (eop) imagen_unpair_quads(op_cur[ni][qp], ad[0], ad[1]);     // (uniform branch; the exchange runs on all lanes)
          f16x4 o[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int q = qp + 2 * h;
            const int cl = co0 + ni * 32 + 8 * q + 4 * half;
            const float4 bq = *reinterpret_cast<const float4*>(par + cl);
            const float4 gq = *reinterpret_cast<const float4*>(par + BN + cl);
            const float bb[4] = {bq.x, bq.y, bq.z, bq.w}, gg[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[ni][4 * q + e] + bb[e];
              if (eop) v += (float)ad[h][e] * gg[e];
              o[h][e] = (f16)v;
              const float r = (float)o[h][e];    // statistics of the value the consumer will read back
              ssq += (cl + e < p.Cout) ? r * r : 0.0f;
            }
          }
          const imagen_u32x4 v16 = imagen_pair_quads(o[0], o[1]);
          const int cx = co0 + ni * 32 + 8 * qp + 16 * half;
          if (px_ok && cx < p.Cout) *reinterpret_cast<imagen_u32x4*>(yrow + cx) = v16;
        }
      if (p.ssq_out) {
        ssq += __shfl_xor(ssq, 32);
        if constexpr (WN == 1) {
          if (half == 0 && px_ok) p.ssq_out[(size_t)b * HW + px] = ssq;
        } else {   // the two waves of a pixel block each hold half the channels: one hop through LDS (workgroup-uniform branch)
          float* red = par + 2 * BN;
          if (wn == 1 && half == 0) red[px_l] = ssq;
          __syncthreads();
          if (wn == 0 && half == 0 && px_ok) p.ssq_out[(size_t)b * HW + px] = ssq + red[px_l];
        }
      }
    }

    if (!more) break;
    stage(nx, cur ^ 1);     // (waits for the next tile's rows; this tile's stores stay in flight behind them)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) op_cur[ni][qp] = nx.op[ni][qp];
    t = tn;
    b = bn;
    pix0 = pixn;
    cur ^= 1;
  }
}

template <int NI, int KCH, int WN>
int pw_launch(const ImagenIgemmParams& p, hipStream_t s) {
  constexpr int PW_TP = 32 * PW_NW / WN, PW_CH = PW_TP * 64;
  const size_t lds = (size_t)2 * KCH * PW_CH + (size_t)(2 * 32 * NI * WN + PW_TP) * sizeof(float);
  auto kern = conv_pw_kernel<NI, KCH, WN>;
  static bool attr_done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("conv_pw: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  const int tiles_img = (p.OH * p.OW + PW_TP - 1) / PW_TP;
  const int total = p.B * tiles_img;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  const int resident = std::max(1, cus * per_cu);
  int gx = total;
  if (total > resident) {   // even rounds: every workgroup walks the same number of tiles (+-1)
    const int rounds = (total + resident - 1) / resident;
    gx = (total + rounds - 1) / rounds;
  }
  hipLaunchKernelGGL(kern, dim3(gx), dim3(64 * PW_NW), lds, s, p);
  return imagen_hip_status("conv_pw launch");
}

struct PwCfg { int NI, KCH, WN; };
constexpr PwCfg kPwCfgs[] = {
    {1, 2, 1},   // 0: 256 px,  64 -> <= 32 channels
    {2, 3, 1},   // 1: 256 px,  96 -> <= 64 channels
    {2, 2, 1},   // 2: 256 px,  64 -> <= 64 channels
    {1, 3, 1},   // 3: 256 px,  96 -> <= 32 channels
    {2, 6, 2},   // 4: 128 px, 192 -> <= 128 channels (two waves per pixel block, 64 couts each: 24 A fragments per wave)
    {2, 4, 2},   // 5: 128 px, 128 -> <= 128 channels
};
constexpr int kNumPwCfgs = sizeof(kPwCfgs) / sizeof(kPwCfgs[0]);

}  // namespace

int imagen_conv_pw_num_configs() { return kNumPwCfgs; }

// tile_cout: output channels one workgroup covers; kgroups: the number of 32-channel input chunks the instantiation is built for (the
// packed weight layout of a 1x1 layer is the same for every G >= 2: consecutive 8-channel group rows)
int imagen_conv_pw_config_info(int idx, int* tile_pixels, int* tile_cout, int* kchunks) {
  if (idx < 0 || idx >= kNumPwCfgs) return -1;
  if (tile_pixels) *tile_pixels = 32 * PW_NW / kPwCfgs[idx].WN;
  if (tile_cout) *tile_cout = 32 * kPwCfgs[idx].NI * kPwCfgs[idx].WN;
  if (kchunks) *kchunks = kPwCfgs[idx].KCH;
  return 0;
}

long imagen_conv_pw_lds_bytes(int idx, int KH, int KW, int TH, int TW) {
  if (idx < 0 || idx >= kNumPwCfgs || KH != 1 || KW != 1) return -1;
  const PwCfg c = kPwCfgs[idx];
  const int tp = 32 * PW_NW / c.WN;
  if (TH * TW != tp) return -1;
  return 2L * c.KCH * tp * 64 + (2L * 32 * c.NI * c.WN + tp) * (long)sizeof(float);
}

int launch_conv_pw(const ImagenIgemmParams* pp, int idx, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  IMAGEN_CHECK(idx >= 0 && idx < kNumPwCfgs, "conv_pw: bad cfg index %d", idx);
  const PwCfg c = kPwCfgs[idx];
  IMAGEN_CHECK(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.OH == p.H && p.OW == p.W, "conv_pw: 1x1 stride-1 convolutions only");
  IMAGEN_CHECK(!p.mu && !p.rs && !p.pa && !p.ps && !p.ssq_a && p.act_in == IMAGEN_ACT_NONE && p.act_out == IMAGEN_ACT_NONE && !p.post_pa && !p.gca_part,
               "conv_pw: no prologue, no output activation, no post_pa / gca_part");
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NHWC, "conv_pw: NHWC output only");
  IMAGEN_CHECK(p.C1 % 32 == 0 && p.C2 % 32 == 0 && (p.C1 + p.C2) == 32 * c.KCH && p.Cin_pad == p.C1 + p.C2 && (p.C2 == 0 || p.x2),
               "conv_pw: cfg %d takes %d input channels in 32-channel chunks (got %d + %d)", idx, 32 * c.KCH, p.C1, p.C2);
  IMAGEN_CHECK(p.Cout <= 32 * c.NI * c.WN && p.Cout % 8 == 0 && p.Cout_pad >= 32 * c.NI * c.WN, "conv_pw: cfg %d covers %d output channels (Cout %d, padded %d)",
               idx, 32 * c.NI * c.WN, p.Cout, p.Cout_pad);
  IMAGEN_CHECK(p.ld1 % 8 == 0 && (p.C2 == 0 || p.ld2 % 8 == 0) && p.ldy % 8 == 0 && p.bsy % 8 == 0 && ((size_t)p.y & 15) == 0,
               "conv_pw: strides must keep 16-byte alignment");
  IMAGEN_CHECK(!(p.addend && p.res), "conv_pw: addend and residual are mutually exclusive");
  IMAGEN_CHECK(!p.addend || (p.gate && p.ld_add % 8 == 0 && p.bs_add % 8 == 0 && ((size_t)p.addend & 15) == 0), "conv_pw: addend needs gate and 16-byte aligned rows");
  IMAGEN_CHECK(!p.res || (p.ld_res % 8 == 0 && p.bs_res % 8 == 0 && ((size_t)p.res & 15) == 0), "conv_pw: residual rows must be 16-byte aligned");
  switch (idx) {
    case 0: return pw_launch<1, 2, 1>(p, s);
    case 1: return pw_launch<2, 3, 1>(p, s);
    case 2: return pw_launch<2, 2, 1>(p, s);
    case 3: return pw_launch<1, 3, 1>(p, s);
    case 4: return pw_launch<2, 6, 2>(p, s);
    case 5: return pw_launch<2, 4, 2>(p, s);
  }
  return -1;
}
