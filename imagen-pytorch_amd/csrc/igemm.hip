// igemm.hip — implicit-GEMM convolution / linear layer on CDNA4 MFMA (v_mfma_f32_32x32x16_f16) with the
// Imagen block prologue (ChanRMSNorm / LayerNorm statistics + per-(batch,channel) affine + SiLU) fused into
// the activation staging and bias / activation / gate*addend / residual / pixel-shuffle fused into the
// epilogue.  Replaces the ATen sequences cited at ImagenIgemmParams in include/imagen_hip.h.
//
// Data flow per workgroup (256 threads = 4 wave64, one output tile of TP pixels x BN output channels):
//   HBM (NHWC fp16) --16B/lane loads--> VGPR --prologue in fp32--> LDS halo tile [pixels][8*G ch], padded rows
//   LDS --ds_read_b128 per (tap, 8-channel group)--> MFMA B operand (pixels are the N/lane dimension)
//   packed weights (L2-resident, fragment order) --16B/lane loads--> MFMA A operand (output channels = rows)
//   accumulators D[cout][pixel]: lane = pixel, 4 consecutive couts per register quad -> 8B NHWC stores.
// The k dimension runs over channel chunks of 8*G channels; inside a chunk over (tap, 8-channel group) pairs;
// two consecutive groups (lane>>5 selects) feed one K=16 MFMA.  Staging of chunk c+1 (global loads issued
// before, LDS writes after the MFMAs of chunk c) overlaps the matrix work of chunk c.
#include <type_traits>
#include <utility>
#include "common.h"
#include "gca_device.h"

namespace {

struct TileCfg { int MI, NI, WM, WN, G; };
constexpr int kMaxItems = 6;  // 16B staging items per thread per chunk

constexpr TileCfg kCfgs[] = {
    {2, 1, 4, 1, 4},  // 0: 256 px x  32 co, 32-ch chunks   (C_out = 32 layers, 256^2/128^2 levels)
    // wave tilings put ALL pixels of the tile on every wave and split the output channels across waves where possible:
    // a weight fragment (streamed from L2, 16 B/lane) then feeds MI MFMAs and no two waves fetch the same one
    {4, 1, 1, 4, 4},  // 1: 128 px x 128 co
    {4, 1, 2, 2, 4},  // 2: 256 px x  64 co
    {2, 1, 1, 4, 4},  // 3:  64 px x 128 co                (small feature maps, token GEMMs)
    {1, 1, 4, 1, 4},  // 4: 128 px x  32 co
    {2, 1, 4, 1, 1},  // 5: 256 px x  32 co,  8-ch chunks   (15x15 cross-embed conv, C_in = 3|6 padded to 8)
    {1, 1, 2, 2, 4},  // 6:  64 px x  64 co
    {1, 2, 2, 2, 1},  // 7:  64 px x 128 co,  8-ch chunks   (channel counts that are only multiples of 8)
    {1, 1, 2, 2, 1},  // 8:  64 px x  64 co,  8-ch chunks
    {1, 1, 4, 1, 1},  // 9: 128 px x  32 co,  8-ch chunks
    // 1x1 convolutions / linear layers: no halo, so the k-chunk can be 64 or 128 channels deep (G = 8 | 16): 4-8 K=16 steps
    // per barrier instead of 2, and 2-4x the bytes in flight per staging round trip
    {2, 1, 1, 4, 16},  // 10:  64 px x 128 co, 128-ch chunks
    {4, 1, 1, 4, 8},   // 11: 128 px x 128 co,  64-ch chunks
    {1, 1, 2, 2, 16},  // 12:  64 px x  64 co, 128-ch chunks
    {1, 1, 4, 1, 8},   // 13: 128 px x  32 co,  64-ch chunks
    {2, 1, 1, 4, 8},   // 14:  64 px x 128 co,  64-ch chunks
    {1, 1, 2, 2, 8},   // 15:  64 px x  64 co,  64-ch chunks
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) — keeps every register-array index static
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int G> struct Geo {
  static constexpr int KC = 8 * G;                          // channels per chunk
  static constexpr int PS = (G == 1) ? 16 : (G * 16 + 16);  // LDS bytes per staged pixel (16B pad: conflict-free ds_read_b128)
};

// KSC > 0: the number of K=16 steps per channel chunk is a compile-time constant (18 for 3x3 convs with 32-channel
// chunks, 2 for 1x1 / linear, 8 for the 2x2 stride-2 downsample): the k-loop is fully unrolled so the weight-fragment ring is
// statically indexed (no register copies of in-flight loads, which would force a vmcnt wait every step).  KSC == 0: generic loop.
// occupancy target (waves per SIMD): 2 for the 64-pixel-per-wave tilings (<= 256 VGPR+AGPR), 1 for the 128-pixel ones
template <int MI, int NI, int WM, int WN, int G, int KSC>
__global__ __launch_bounds__(256, (MI * NI <= 2 ? 2 : 1)) void igemm_kernel(const ImagenIgemmParams p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int BN = 32 * NI * WN;
  constexpr int KC = Geo<G>::KC;
  constexpr int PS = Geo<G>::PS;
  constexpr int LOG2G = (G == 1) ? 0 : (G == 2) ? 1 : (G == 4) ? 2 : (G == 8) ? 3 : 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  // ---- which output tile
  const int tilesX = (p.OW + p.TW - 1) / p.TW;
  const int tilesY = (p.OH + p.TH - 1) / p.TH;
  int t = blockIdx.x;
  const int tile_x = t % tilesX;
  t /= tilesX;
  const int tile_y = t % tilesY;
  const int b = t / tilesY;
  const int oy0 = tile_y * p.TH, ox0 = tile_x * p.TW;
  const int n0 = blockIdx.y * BN;

  const int ITW = (p.TW - 1) * p.stride + p.KW;
  const int ITH = (p.TH - 1) * p.stride + p.KH;
  const int IT = ITH * ITW;
  const int items = IT << LOG2G;
  const float inv_itw = 1.0f / (float)ITW;
  const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
  const int buf_bytes = IT * PS;

  const int ntap = p.KH * p.KW;
  const int KG = ntap * G;            // 8-channel groups per chunk
  const int KS = KSC > 0 ? KSC : (KG + 1) >> 1;  // K=16 MFMA steps per chunk (odd KG: last half-step is zero weights)
  const int NC = p.Cin_pad / KC;      // chunks

  // ---- per-lane output pixel coordinates (lane = pixel in the MFMA N dimension)
  int a_base[MI];
  int opix_y[MI], opix_x[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int tp = (wm * MI + mi) * 32 + l31;
    const int py = tp / p.TW, px = tp - py * p.TW;
    opix_y[mi] = oy0 + py;
    opix_x[mi] = ox0 + px;
    a_base[mi] = ((py * p.stride) * ITW + px * p.stride) * PS;
  }

  // ---- weights: packed [chunk*KGP + kg][Cout_pad] x (8 halves); this lane's rows
  const int KGP = KS * 2;
  const f16x8* wbase = reinterpret_cast<const f16x8*>(p.w) + (size_t)half * p.Cout_pad + n0 + wn * (NI * 32) + l31;
  const size_t wstep = (size_t)2 * p.Cout_pad;  // one K=16 step
  (void)KGP;

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.0f;

  // ---- staging state
  uint4 raw[kMaxItems];
  float st_rs[kMaxItems], st_mu[kMaxItems];
  unsigned inb_mask = 0;

  const f16* x1 = reinterpret_cast<const f16*>(p.x1);
  const f16* x2 = reinterpret_cast<const f16*>(p.x2);

  // per-chunk prologue affine of THIS thread's 8-channel group: every item of a thread has the same group (256 % G == 0) and
  // the tile lies in one batch row, so the 8 + 8 floats are loaded once per chunk, together with the activations
  float st_a[8], st_s[8];
  const int my_cg = tid & (G - 1);

  auto stage_load = [&](int chunk) __attribute__((always_inline)) {
    inb_mask = 0;
    const int cc = chunk * KC + my_cg * 8;
    if (p.pa) {
      const float4* q = reinterpret_cast<const float4*>(p.pa + (size_t)b * p.pstride + cc);
      const float4 q0 = q[0], q1 = q[1];
      st_a[0] = q0.x; st_a[1] = q0.y; st_a[2] = q0.z; st_a[3] = q0.w; st_a[4] = q1.x; st_a[5] = q1.y; st_a[6] = q1.z; st_a[7] = q1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) st_a[j] = 1.0f;
    }
    if (p.ps) {
      const float4* q = reinterpret_cast<const float4*>(p.ps + (size_t)b * p.pstride + cc);
      const float4 q0 = q[0], q1 = q[1];
      st_s[0] = q0.x; st_s[1] = q0.y; st_s[2] = q0.z; st_s[3] = q0.w; st_s[4] = q1.x; st_s[5] = q1.y; st_s[6] = q1.z; st_s[7] = q1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) st_s[j] = 0.0f;
    }
#pragma unroll
    for (int it = 0; it < kMaxItems; ++it) {
      const int idx = tid + it * 256;
      raw[it] = make_uint4(0, 0, 0, 0);
      st_rs[it] = 1.0f;
      st_mu[it] = 0.0f;
      if (idx < items) {
        const int pix = idx >> LOG2G;
        const int iy = (int)(((float)pix + 0.5f) * inv_itw);
        const int ix = pix - iy * ITW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
          const int gp = gy * p.W + gx;
          const f16* src = nullptr;
          if (cc < p.C1) src = x1 + (size_t)b * p.bs1 + (size_t)gp * p.ld1 + cc;
          else if (cc - p.C1 < p.C2) src = x2 + (size_t)b * p.bs2 + (size_t)gp * p.ld2 + (cc - p.C1);
          if (src) {
            if (!(p.dbg & 16)) raw[it] = *reinterpret_cast<const uint4*>(src);   // dbg 16: ablate the activation loads
            inb_mask |= 1u << it;
            const int sp = b * (p.H * p.W) + gp;
            if (p.rs) st_rs[it] = p.rs[sp];
            else if (p.ssq_a) {  // ChanRMSNorm statistics straight from the producers' per-pixel sums of squares
              float q = p.ssq_a[sp];
              if (p.ssq_b) q += p.ssq_wb * p.ssq_b[sp];
              st_rs[it] = 1.0f / fmaxf(sqrtf(q), 1e-12f);
            }
            if (p.mu) st_mu[it] = p.mu[sp];
          }
        }
      }
    }
  };

  // transform + LDS write of ONE staged item (it is a compile-time index at every call site)
  auto stage_write_item = [&](int it, char* buf) __attribute__((always_inline)) {
    const int idx = tid + it * 256;
    if (idx < items) {
      const int pix = idx >> LOG2G;
      f16x8 out;
      if (inb_mask & (1u << it)) {
        const f16x8 in = *reinterpret_cast<const f16x8*>(&raw[it]);
        const float rs = st_rs[it], mu = st_mu[it];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = ((float)in[j] - mu) * rs * st_a[j] + st_s[j];
          if (p.act_in == IMAGEN_ACT_SILU) v = silu_f(v);
          out[j] = (f16)v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = (f16)0.0f;
      }
      *reinterpret_cast<f16x8*>(buf + pix * PS + my_cg * 16) = out;
    }
  };

  auto stage_write = [&](char* buf) __attribute__((always_inline)) {
    static_for<kMaxItems>([&](auto ic) __attribute__((always_inline)) { stage_write_item(decltype(ic)::value, buf); });
  };

  // ---- weight fragment pipeline (continuous over chunks)
  // wq[j] holds the fragments of K=16 step (gstep + j); wptr always points at step (gstep + kLookAhead)
  constexpr int kLookAhead = KSC == 0 ? 1 : (KSC >= 6 ? 6 : KSC);
  f16x8 wq[kLookAhead][NI];
  const f16x8* wptr = wbase;
#pragma unroll
  for (int j = 0; j < kLookAhead; ++j) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wq[j][ni] = wptr[ni * 32];
    wptr += wstep;
  }
  static_assert(kLookAhead <= 8, "packed weights carry a zero tail of 8 steps");

  // wbuf != nullptr: the LDS writes of the NEXT chunk's staged items are interleaved with the last kMaxItems MFMA steps, so the
  // prologue VALU work runs under the matrix pipe instead of serialising behind it (only when the k-loop is unrolled)
  auto compute = [&](const char* buf, char* wbuf) __attribute__((always_inline)) {
    // (dy, dx, group) walk of this lane's 8-channel group: kg = 2*ks + half
    int dy = 0, dx = 0, cgp = 0;  // G >= 2: uniform walk, group = 2*cgp + half
    int tap_l = half;             // G == 1: per-lane tap walk (tap = 2*ks + half), kept as (ty_l, tx_l) incrementally
    int ty_l = half / p.KW, tx_l = half - ty_l * p.KW;
    auto next_aoff = [&]() __attribute__((always_inline)) -> int {  // LDS offset of this lane's fragment for the next K=16 step, advancing the walk
      int aoff;
      if (G == 1) {
        // padded half-step (tap_l == ntap): any valid address works, its weights are zero
        aoff = tap_l < ntap ? (ty_l * ITW + tx_l) * PS : 0;
        tap_l += 2;
        tx_l += 2;
        while (tx_l >= p.KW) { tx_l -= p.KW; ++ty_l; }
      } else {
        aoff = (dy * ITW + dx) * PS + (2 * cgp + half) * 16;
        if (++cgp == G / 2) {
          cgp = 0;
          if (++dx == p.KW) { dx = 0; ++dy; }
        }
      }
      return aoff;
    };
    // activation fragments are register-prefetched one step ahead: the ds_read latency of step k+1 hides under the MFMAs of k
    f16x8 afrag[MI], anext[MI];
    {
      const int a0 = next_aoff();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) afrag[mi] = *reinterpret_cast<const f16x8*>(buf + a_base[mi] + a0);
    }
    auto do_step = [&](int ks) __attribute__((always_inline)) {
      // prefetch the weight fragments kLookAhead steps ahead (L2 latency ~ 2-3 MFMA groups)
      // (unconditional: the packed buffer carries a zero tail of kTailSteps steps, so reads past the last step stay in bounds)
      f16x8 wnew[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wnew[ni] = wptr[ni * 32];
      wptr += wstep;
      if (ks + 1 < KS) {
        const int a1 = next_aoff();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) anext[mi] = *reinterpret_cast<const f16x8*>(buf + a_base[mi] + a1);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[0][ni], afrag[mi], acc[ni][mi], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) afrag[mi] = anext[mi];
#pragma unroll
      for (int j = 0; j + 1 < kLookAhead; ++j)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wq[j][ni] = wq[j + 1][ni];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wq[kLookAhead - 1][ni] = wnew[ni];
    };
    if constexpr (KSC > 0) {
      static_for<KSC>([&](auto ic) __attribute__((always_inline)) {
        constexpr int ks = decltype(ic)::value;
        do_step(ks);
        if constexpr (KSC >= kMaxItems && ks >= KSC - kMaxItems) {
          if (wbuf != nullptr) stage_write_item(ks - (KSC - kMaxItems), wbuf);
        }
      });
    } else {
      for (int ks = 0; ks < KS; ++ks) do_step(ks);
    }
  };

  // ---- main loop over channel chunks (double-buffered LDS)
  stage_load(0);
  stage_write(smem);
  __syncthreads();
  int cur = 0;
  for (int chunk = 0; chunk < NC; ++chunk) {
    const bool more = chunk + 1 < NC;
    const bool restage = more && !(p.dbg & 1);
    char* nbuf = smem + (cur ^ 1) * buf_bytes;
    constexpr bool kInterleave = KSC >= kMaxItems;
    if (restage) stage_load(chunk + 1);
    if (!(p.dbg & 2)) compute(smem + cur * buf_bytes, (kInterleave && restage) ? nbuf : nullptr);
    if (restage && (!kInterleave || (p.dbg & 2))) stage_write(nbuf);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane = pixel; register quad q holds couts 8q + 4*half + {0..3} of each 32-cout fragment
  const f16* addend = reinterpret_cast<const f16*>(p.addend);
  const f16* res = reinterpret_cast<const f16*>(p.res);
  float ssq_px[MI];  // per-pixel sum of squares of this wave's stored channels (for the consumer's ChanRMSNorm)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    ssq_px[mi] = 0.0f;
    const int oy = opix_y[mi], ox = opix_x[mi];
    if (oy >= p.OH || ox >= p.OW) continue;
    const int op = oy * p.OW + ox;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
        if (co >= p.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[ni][mi][4 * q + e];
          if (p.bias) x += p.bias[co + e];  // bias is padded to Cout_pad by the host
          if (p.act_out == IMAGEN_ACT_SILU) x = silu_f(x);
          else if (p.act_out == IMAGEN_ACT_GELU) x = gelu_f(x);
          v[e] = x;
        }
        if (p.out_mode == IMAGEN_OUT_NCHW_F32) {
          float* y = reinterpret_cast<float*>(p.y);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.Cout) y[((size_t)b * p.Cout + co + e) * (p.OH * p.OW) + op] = v[e];
          continue;
        }
        if (addend) {
          const f16x4 ad = *reinterpret_cast<const f16x4*>(addend + (size_t)b * p.bs_add + (size_t)op * p.ld_add + co);
          const float4 g = *reinterpret_cast<const float4*>(p.gate + (size_t)b * p.gate_stride + co);
          v[0] += (float)ad[0] * g.x; v[1] += (float)ad[1] * g.y; v[2] += (float)ad[2] * g.z; v[3] += (float)ad[3] * g.w;
        }
        if (res) {
          const f16x4 rr = *reinterpret_cast<const f16x4*>(res + (size_t)b * p.bs_res + (size_t)op * p.ld_res + co);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
        }
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (f16)v[e];
          const float r = (float)o[e];  // statistics of the value the consumer will read back
          ssq_px[mi] += r * r;
        }
        f16* y = reinterpret_cast<f16*>(p.y);
        if (p.out_mode == IMAGEN_OUT_PIXEL_SHUFFLE) {
          // output channels are packed (s1, s2, c): cout = (2*s1 + s2) * Cq + c   (PixelShuffle(2), ip.py:616)
          const int Cq = p.Cout >> 2;
          const int sub = co / Cq, c = co - sub * Cq;
          const int yy = 2 * oy + (sub >> 1), xx = 2 * ox + (sub & 1);
          *reinterpret_cast<f16x4*>(y + (size_t)b * p.bsy + ((size_t)yy * (2 * p.OW) + xx) * p.ldy + c) = o;
        } else {
          if (!(p.dbg & 8)) *reinterpret_cast<f16x4*>(y + (size_t)b * p.bsy + (size_t)op * p.ldy + co) = o;   // dbg 8: ablate the stores
        }
      }
    }
  }
  // ---- optional: emit the per-pixel sum of squares (launcher guarantees one workgroup covers all Cout: gridDim.y == 1)
  if (p.ssq_out) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) ssq_px[mi] += __shfl_xor(ssq_px[mi], 32);  // both lane halves hold disjoint channel quads
    if (WN == 1) {
      if (half == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          if (opix_y[mi] < p.OH && opix_x[mi] < p.OW) p.ssq_out[(size_t)b * (p.OH * p.OW) + opix_y[mi] * p.OW + opix_x[mi]] = ssq_px[mi];
      }
    } else {
      float* red = reinterpret_cast<float*>(smem);  // [WN][pixels of this wave row]; LDS is free after the main loop's last barrier
      constexpr int PXW = 32 * MI;                  // pixels per wave
      if (half == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) red[(wm * WN + wn) * PXW + mi * 32 + l31] = ssq_px[mi];
      }
      __syncthreads();
      if (wn == 0 && half == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          float tot = 0.0f;
#pragma unroll
          for (int w = 0; w < WN; ++w) tot += red[(wm * WN + w) * PXW + mi * 32 + l31];
          if (opix_y[mi] < p.OH && opix_x[mi] < p.OW) p.ssq_out[(size_t)b * (p.OH * p.OW) + opix_y[mi] * p.OW + opix_x[mi]] = tot;
        }
      }
    }
  }

  // ---- optional: fused GlobalContext partials + last-workgroup finalisation (see ImagenIgemmParams.gca_*)
  if (p.gca_wk) {
    constexpr int PXW = 32 * MI;
    float* lds = reinterpret_cast<float*>(smem) + 1024;   // past the ssq scratch; LDS is free after the main loop's last barrier
    float* red = lds;                                     // [4 waves][PXW]
    float* s_w = lds + 4 * PXW;                           // [8] per-wave scalars
    float* chan = s_w + 8;                                // [BN] channel sums
    int* s_flag = reinterpret_cast<int*>(chan + BN);
    const int tiles_img = tilesX * tilesY;
    const int tile_in_img = tile_y * tilesX + tile_x;
    bool valid[MI];
    float lg[MI];
    // 1. logit per pixel: sum over this wave's channels, both lane halves, then all WN waves
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      valid[mi] = opix_y[mi] < p.OH && opix_x[mi] < p.OW;
      float d = 0.0f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
          if (co < p.Cout) {
#pragma unroll
            for (int e = 0; e < 4; ++e) d += (acc[ni][mi][4 * q + e] + (p.bias ? p.bias[co + e] : 0.0f)) * p.gca_wk[co + e];
          }
        }
      d += __shfl_xor(d, 32);
      lg[mi] = d;
    }
    if (WN > 1) {
      if (half == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) red[(wm * WN + wn) * PXW + mi * 32 + l31] = lg[mi];
      }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < WN; ++w) tot += red[(wm * WN + w) * PXW + mi * 32 + l31];
        lg[mi] = tot;
      }
    }
    // 2. tile max
    float mx = -3.0e38f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      lg[mi] += p.gca_bk;
      if (valid[mi]) mx = fmaxf(mx, lg[mi]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) s_w[wave] = mx;
    if (tid < BN) chan[tid] = 0.0f;
    __syncthreads();
    const float m_blk = fmaxf(fmaxf(s_w[0], s_w[1]), fmaxf(s_w[2], s_w[3]));
    // 3. exp weights and their sum (every pixel counted once: wn == 0 waves, lower lane half)
    float ew[MI];
    float se = 0.0f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ew[mi] = valid[mi] ? __expf(lg[mi] - m_blk) : 0.0f;
      se += ew[mi];
    }
    if (wn != 0 || half != 0) se = 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) se += __shfl_xor(se, off);
    __syncthreads();   // everyone has read s_w (max) before it is reused for the sums
    if (lane == 0) s_w[wave] = se;
    // 4. exp-weighted channel sums: in-lane over MI pixels, across the 32 pixel lanes by shuffles, across WM waves by LDS atomics
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = 0.0f;
          if (co < p.Cout) {
            const float bb = p.bias ? p.bias[co + e] : 0.0f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) t += ew[mi] * (acc[ni][mi][4 * q + e] + bb);
          }
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) t += __shfl_xor(t, off);
          if (l31 == 0 && co < p.Cout) atomicAdd(&chan[co - n0 + e], t);
        }
      }
    __syncthreads();
    float* part = p.gca_part + ((size_t)b * tiles_img + tile_in_img) * (p.Cout + 2);
    if (tid < p.Cout) part[2 + tid] = chan[tid];
    if (tid == 0) {
      part[0] = m_blk;
      part[1] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
    // 5. ticket: the last workgroup of this image merges all tiles (placement-independent release / acquire, agent scope)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int old = __hip_atomic_fetch_add(p.gca_counter + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (old == tiles_img - 1) ? 1 : 0;
      if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      *s_flag = last;
    }
    __syncthreads();
    const int i_am_last = *s_flag;
    __syncthreads();   // the finalisation below reuses this LDS region
    if (i_am_last) {
      gca_finalize(p.gca_part + (size_t)b * tiles_img * (p.Cout + 2), tiles_img, p.Cout, p.gca_hidden, p.gca_w1t, p.gca_b1, p.gca_w2t,
                   p.gca_b2, p.gca_gate + (size_t)b * p.Cout, lds);
      if (tid == 0) __hip_atomic_store(p.gca_counter + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int MI, int NI, int WM, int WN, int G, int KSC>
int launch_ksc(const ImagenIgemmParams& p, hipStream_t s) {
  constexpr int TP = 32 * MI * WM, BN = 32 * NI * WN;
  const int ITW = (p.TW - 1) * p.stride + p.KW, ITH = (p.TH - 1) * p.stride + p.KH;
  const int IT = ITH * ITW;
  IMAGEN_CHECK(p.TH * p.TW == TP, "igemm: tile %dx%d does not match cfg %d (%d pixels)", p.TH, p.TW, p.cfg, TP);
  IMAGEN_CHECK(IT * G <= kMaxItems * 256, "igemm: halo tile too large (%d px x %d groups)", IT, G);
  IMAGEN_CHECK(p.Cout_pad % BN == 0, "igemm: Cout_pad %d not a multiple of %d", p.Cout_pad, BN);
  IMAGEN_CHECK(p.Cin_pad % (8 * G) == 0, "igemm: Cin_pad %d not a multiple of %d", p.Cin_pad, 8 * G);
  IMAGEN_CHECK(p.C1 % 8 == 0 && p.C2 % 8 == 0 && p.ld1 % 8 == 0 && (p.x2 == nullptr || p.ld2 % 8 == 0),
               "igemm: channel counts / strides must be multiples of 8 (C1=%d C2=%d ld1=%d ld2=%d)", p.C1, p.C2, p.ld1, p.ld2);
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NCHW_F32 || p.Cout % 4 == 0, "igemm: Cout %d must be a multiple of 4", p.Cout);
  IMAGEN_CHECK(p.out_mode != IMAGEN_OUT_PIXEL_SHUFFLE || p.Cout % 16 == 0, "igemm: pixel-shuffle needs Cout %% 16 == 0");
  if (p.gca_wk) {
    IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN && p.act_out == IMAGEN_ACT_NONE && !p.addend && !p.res,
                 "igemm: fused GlobalContext needs a plain NHWC conv output and one workgroup covering all %d channels", p.Cout);
    IMAGEN_CHECK(p.gca_part && p.gca_counter && p.gca_w1t && p.gca_b1 && p.gca_w2t && p.gca_b2 && p.gca_gate && p.gca_hidden > 0,
                 "igemm: incomplete gca_* parameters");
    const int tiles_img = ((p.OW + p.TW - 1) / p.TW) * ((p.OH + p.TH - 1) / p.TH);
    const size_t need = (size_t)(1024 + p.Cout + p.gca_hidden + tiles_img + kGcaScratchFloats + 4 * 32 * MI + 8 + BN + 4) * sizeof(float);
    IMAGEN_CHECK(need <= (size_t)2 * IT * Geo<G>::PS, "igemm: fused GlobalContext scratch (%zu B) exceeds the tile LDS", need);
  }
  IMAGEN_CHECK(!p.ssq_out || (p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN),
               "igemm: ssq_out needs NHWC output and one workgroup covering all %d output channels (tile has %d)", p.Cout, BN);
  const size_t lds = (size_t)2 * IT * Geo<G>::PS;
  IMAGEN_CHECK(lds <= 160 * 1024, "igemm: LDS tile %zu bytes too large", lds);
  auto kern = igemm_kernel<MI, NI, WM, WN, G, KSC>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("igemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const int tilesX = (p.OW + p.TW - 1) / p.TW, tilesY = (p.OH + p.TH - 1) / p.TH;
  dim3 grid(p.B * tilesX * tilesY, (p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  return imagen_hip_status("igemm launch");
}

template <int MI, int NI, int WM, int WN, int G>
int launch_cfg(const ImagenIgemmParams& p, hipStream_t s) {
  const int ks = (p.KH * p.KW * G + 1) / 2;
  if (G == 4) {
    if (ks == 18) return launch_ksc<MI, NI, WM, WN, G, (G == 4 ? 18 : 0)>(p, s);
    if (ks == 2) return launch_ksc<MI, NI, WM, WN, G, (G == 4 ? 2 : 0)>(p, s);
    if (ks == 8) return launch_ksc<MI, NI, WM, WN, G, (G == 4 ? 8 : 0)>(p, s);
  }
  if (G == 8 && ks == 4) return launch_ksc<MI, NI, WM, WN, G, (G == 8 ? 4 : 0)>(p, s);
  if (G == 8 && ks == 36) return launch_ksc<MI, NI, WM, WN, G, (G == 8 ? 36 : 0)>(p, s);   // 3x3 with 64-channel chunks
  if (G == 16 && ks == 8) return launch_ksc<MI, NI, WM, WN, G, (G == 16 ? 8 : 0)>(p, s);
  return launch_ksc<MI, NI, WM, WN, G, 0>(p, s);
}

}  // namespace

int launch_igemm(const ImagenIgemmParams* pp, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  IMAGEN_CHECK(p.cfg >= 0 && p.cfg < kNumCfgs, "igemm: bad cfg %d", p.cfg);
  IMAGEN_CHECK(p.x1 && p.w && p.y, "igemm: null x1/w/y");
  IMAGEN_CHECK(!p.addend || p.gate, "igemm: addend requires gate");
  switch (p.cfg) {
    case 0: return launch_cfg<2, 1, 4, 1, 4>(p, s);
    case 1: return launch_cfg<4, 1, 1, 4, 4>(p, s);
    case 2: return launch_cfg<4, 1, 2, 2, 4>(p, s);
    case 3: return launch_cfg<2, 1, 1, 4, 4>(p, s);
    case 4: return launch_cfg<1, 1, 4, 1, 4>(p, s);
    case 5: return launch_cfg<2, 1, 4, 1, 1>(p, s);
    case 6: return launch_cfg<1, 1, 2, 2, 4>(p, s);
    case 7: return launch_cfg<1, 2, 2, 2, 1>(p, s);
    case 8: return launch_cfg<1, 1, 2, 2, 1>(p, s);
    case 9: return launch_cfg<1, 1, 4, 1, 1>(p, s);
    case 10: return launch_cfg<2, 1, 1, 4, 16>(p, s);
    case 11: return launch_cfg<4, 1, 1, 4, 8>(p, s);
    case 12: return launch_cfg<1, 1, 2, 2, 16>(p, s);
    case 13: return launch_cfg<1, 1, 4, 1, 8>(p, s);
    case 14: return launch_cfg<2, 1, 1, 4, 8>(p, s);
    case 15: return launch_cfg<1, 1, 2, 2, 8>(p, s);
  }
  return -1;
}

extern "C" int imagen_igemm_num_configs(void) { return kNumCfgs; }

extern "C" int imagen_igemm_config_info(int cfg, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (cfg < 0 || cfg >= kNumCfgs) return -1;
  const TileCfg& c = kCfgs[cfg];
  if (tile_pixels) *tile_pixels = 32 * c.MI * c.WM;
  if (tile_cout) *tile_cout = 32 * c.NI * c.WN;
  if (kgroups) *kgroups = c.G;
  return 0;
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
constexpr int kTailSteps = 8;  // >= the kernel's weight look-ahead

extern "C" size_t imagen_igemm_packed_elems(int G, int Cin, int Cout_pad, int KH, int KW) {
  if (G != 1 && G != 2 && G != 4 && G != 8 && G != 16) return 0;
  const int KC = 8 * G;
  const int NC = round_up(Cin, KC) / KC;
  const int KGP = ((KH * KW * G + 1) / 2) * 2;
  return (size_t)(NC * KGP + kTailSteps * 2) * Cout_pad * 8;  // + zero tail: the kernel prefetches up to kTailSteps K=16 steps past the end
}

// Host-side packing into MFMA A-operand fragment order: element (chunk, tap, group cg, cout, j) holds
// W[cout][chunk*KC + cg*8 + j][tap] (* in_scale[c]) at ((chunk*KGP + tap*G + cg) * Cout_pad + cout) * 8 + j.
// The layout depends only on G (8-channel groups per k-chunk) and Cout_pad, not on the tile configuration.
extern "C" int imagen_pack_igemm_weights(int G, const float* w_in, const float* in_scale, int Cin, int Cout, int Cout_pad,
                                         int KH, int KW, uint16_t* w_out) {
  if (G != 1 && G != 2 && G != 4 && G != 8 && G != 16) { imagen_set_error("pack: bad G %d", G); return -1; }
  if (Cout_pad < Cout || Cout_pad % 32) { imagen_set_error("pack: bad Cout_pad %d", Cout_pad); return -1; }
  const int KC = 8 * G;
  const int Cin_pad = round_up(Cin, KC);
  const int NC = Cin_pad / KC, ntap = KH * KW;
  const int KGP = ((ntap * G + 1) / 2) * 2;
  const size_t total = (size_t)(NC * KGP + kTailSteps * 2) * Cout_pad * 8;
  f16* out = reinterpret_cast<f16*>(w_out);
  for (size_t i = 0; i < total; ++i) out[i] = (f16)0.0f;
  for (int chunk = 0; chunk < NC; ++chunk)
    for (int tap = 0; tap < ntap; ++tap)
      for (int cg = 0; cg < G; ++cg)
        for (int co = 0; co < Cout; ++co)
          for (int j = 0; j < 8; ++j) {
            const int ci = chunk * KC + cg * 8 + j;
            if (ci >= Cin) continue;
            float v = w_in[((size_t)co * Cin + ci) * ntap + tap];
            if (in_scale) v *= in_scale[ci];
            out[((size_t)(chunk * KGP + tap * G + cg) * Cout_pad + co) * 8 + j] = (f16)v;
          }
  return 0;
}
