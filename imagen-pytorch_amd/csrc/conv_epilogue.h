// conv_epilogue.h — the epilogue of the all-DMA conv kernel (conv_dma.hip): accumulators D[cout][pixel]
// (lane = pixel; register quad q of a 32x32 fragment holds couts 8q + 4*half + {0..3}) -> bias / activation / gate*addend / residual
// / pixel-shuffle / fp32-NCHW / output-side Block prologue (post_pa) / per-pixel sum of squares, as documented at ImagenIgemmParams.
#pragma once
#include "common.h"

struct ClTile { int b, oy0, ox0, n0; };

// ep_par: LDS scratch of 4 * BN + WM * WN * 32 * MI + 8 + WM * (BN + 4) floats, ep_red: WM * WN * 32 * MI floats, dead staging memory of the caller (all its LDS traffic retired: call behind a barrier).  The
// per-channel epilogue operands (bias, post_pa, post_ps) are fetched ONCE per workgroup into it: a dependent global load per channel
// quad costs ~1-2 us each under load (igemm.hip measured 14k cycles for four such rounds), one cooperative fetch + barrier ~1 us.
// sum (or max) over the 32 lanes of each half-wave on the VALU: inclusive scan inside the 16-lane rows by DPP row shifts, then row
// broadcast 15 into the odd rows — the total of lanes 0-31 ends up in lane 31, of lanes 32-63 in lane 63 (other lanes hold partials)
template <bool MAX>
__device__ __forceinline__ float cl_half_reduce(float v) {
  const int id = __builtin_bit_cast(int, MAX ? -3.0e38f : 0.0f);
#define CL_DPP_STEP(ctrl, row_mask)                                                                                                    \
  do {                                                                                                                                 \
    const float y_ = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(id, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false)); \
    v = MAX ? fmaxf(v, y_) : v + y_;                                                                                                   \
  } while (0)
  CL_DPP_STEP(0x111, 0xf);   // row_shr:1
  CL_DPP_STEP(0x112, 0xf);   // row_shr:2
  CL_DPP_STEP(0x114, 0xf);   // row_shr:4
  CL_DPP_STEP(0x118, 0xf);   // row_shr:8
  CL_DPP_STEP(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
#undef CL_DPP_STEP
  return v;
}

// cl_epilogue_params: the per-channel operands of batch row `b` -> ep_par (called by the threads i < BN; barrier by the caller)
template <int BN>
__device__ __forceinline__ void cl_epilogue_params(const ImagenIgemmParams& p, int b, int n0, float* ep_par, int i) {
  const int co = n0 + i;   // < Cout_pad (the bias is padded by the host; post_pa / post_ps are not)
  ep_par[i] = p.bias ? p.bias[co] : 0.0f;
  if (p.post_pa) {
    ep_par[BN + i] = co < p.Cout ? p.post_pa[(size_t)b * p.post_pstride + co] : 0.0f;
    ep_par[2 * BN + i] = co < p.Cout ? p.post_ps[(size_t)b * p.post_pstride + co] : 0.0f;
  }
  if (p.gca_part) ep_par[3 * BN + i] = co < p.Cout ? p.gca_wk[co] : 0.0f;
}

// PRELOADED: ep_par already holds the operands of tc.b (persistent kernels refresh it when the batch row changes): the epilogue then
// contains NO global load — a load here is younger than the caller's in-flight prefetch, and waiting for it drains that whole queue
// STG: the 16-byte output pieces of the plain / post_pa paths go to an LDS image of the tile first ([tile pixel][BN couts] fp16, pitch
// 2 BN + 16 bytes: a lane = pixel writes conflict-free), and the workgroup then stores it in memory order — 16 consecutive lanes write one
// pixel's 2 BN contiguous bytes.  Direct stores are one 16-byte piece per lane at pixel stride: every store instruction touches 64 cache
// lines with a 16-byte partial write, and the L2 takes one write per channel and clock whatever its size (conv_big timing ablation, round
// 3 call Q: the epilogue cost 6 of 37 us).  stg: TP * (2 BN + 16) bytes of dead LDS that overlap neither ep_par nor ep_red.
template <int TP, int BN, int NTHREADS>
__device__ __forceinline__ void cl_copy_out(const ImagenIgemmParams& p, const ClTile& tc, const char* stg) {
  constexpr int PPR = BN / 8, PITCH = 2 * BN + 16;
  __syncthreads();
  f16* y = reinterpret_cast<f16*>(p.y) + (size_t)tc.b * p.bsy;
  for (int j = threadIdx.x; j < TP * PPR; j += NTHREADS) {
    const int pix = j / PPR, pc = j - pix * PPR;
    const int py = pix / p.TW, px = pix - py * p.TW;
    const int oy = tc.oy0 + py, ox = tc.ox0 + px, co = tc.n0 + pc * 8;
    if (oy < p.OH && ox < p.OW && co < p.Cout)
      *reinterpret_cast<imagen_u32x4*>(y + (size_t)(oy * p.OW + ox) * p.ldy + co) = *reinterpret_cast<const imagen_u32x4*>(stg + pix * PITCH + pc * 16);
  }
}

template <int MI, int NI, int WM, int WN, bool GEN, bool PRELOADED = false, bool STG = false>
__device__ __forceinline__ void cl_epilogue(const ImagenIgemmParams& p, const ClTile& tc, f32x16 (&acc)[NI][MI], const int (&pix_y)[MI],
                                            const int (&pix_x)[MI], float* ep_red, float* ep_par, int wm, int wn, int half, int l31,
                                            char* stg = nullptr) {
  constexpr int PXW = 32 * MI;
  constexpr int BN = 32 * NI * WN;
  constexpr int STG_PITCH = 2 * BN + 16;
  if constexpr (!PRELOADED) {
    const int i = threadIdx.x;
    if (i < BN) {
      const int co = tc.n0 + i;   // < Cout_pad (the bias is padded by the host; post_pa / post_ps are not)
      ep_par[i] = p.bias ? p.bias[co] : 0.0f;
      if (p.post_pa) {
        ep_par[BN + i] = co < p.Cout ? p.post_pa[(size_t)tc.b * p.post_pstride + co] : 0.0f;
        ep_par[2 * BN + i] = co < p.Cout ? p.post_ps[(size_t)tc.b * p.post_pstride + co] : 0.0f;
      }
      if (p.gca_part) ep_par[3 * BN + i] = co < p.Cout ? p.gca_wk[co] : 0.0f;
    }
    __syncthreads();
  }
  // lane = pixel; register quad q holds couts 8q + 4*half + {0..3} of each 32-cout fragment
  const int b = tc.b, n0 = tc.n0;
  const f16* addend = reinterpret_cast<const f16*>(p.addend);
  const f16* res = reinterpret_cast<const f16*>(p.res);
  int op[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int oy = tc.oy0 + pix_y[mi], ox = tc.ox0 + pix_x[mi];
    op[mi] = (oy < p.OH && ox < p.OW) ? oy * p.OW + ox : -1;
  }
  auto load_bias = [&](int co) __attribute__((always_inline)) -> float4 { return *reinterpret_cast<const float4*>(ep_par + (co - n0)); };

  if (p.post_pa) {
    // ---- output-side Block prologue: norm over all Cout of the pixel, then activate + store
    float tot[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) tot[mi] = 0.0f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
        if (co >= p.Cout) continue;
        const float4 bq = load_bias(co);
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[ni][mi][4 * q + e] + bb[e];
            acc[ni][mi][4 * q + e] = v;
            tot[mi] += v * v;
          }
      }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) tot[mi] += __shfl_xor(tot[mi], 32);
    if (WN > 1) {
      if (half == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ep_red[(wm * WN + wn) * PXW + mi * 32 + l31] = tot[mi];
      }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < WN; ++w) t += ep_red[(wm * WN + w) * PXW + mi * 32 + l31];
        tot[mi] = t;
      }
    }
    float rsn[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) rsn[mi] = __builtin_amdgcn_rsqf(fmaxf(tot[mi], 1e-24f));
    f16* y = reinterpret_cast<f16*>(p.y) + (size_t)b * p.bsy;
    const bool wide = (p.Cout & 7) == 0;   // 16-byte pieces (imagen_pair_quads, common.h): quads q and q + 2 are produced together
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        f16x4 o2[2][MI];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = qp + 2 * h;
          const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
          const int cl = min(co - n0, BN - 4);
          const float4 pa = *reinterpret_cast<const float4*>(ep_par + BN + cl);
          const float4 ps = *reinterpret_cast<const float4*>(ep_par + 2 * BN + cl);
          const float pav[4] = {pa.x, pa.y, pa.z, pa.w}, psv[4] = {ps.x, ps.y, ps.z, ps.w};
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o2[h][mi][e] = (f16)silu_f(acc[ni][mi][4 * q + e] * rsn[mi] * pav[e] + psv[e]);
            if (!wide && co < p.Cout && op[mi] >= 0) *reinterpret_cast<f16x4*>(y + (size_t)op[mi] * p.ldy + co) = o2[h][mi];
          }
        }
        if (wide) {
          const int co16 = n0 + (wn * NI + ni) * 32 + 8 * qp + 16 * half;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const imagen_u32x4 v = imagen_pair_quads(o2[0][mi], o2[1][mi]);
            if constexpr (STG) *reinterpret_cast<imagen_u32x4*>(stg + ((wm * MI + mi) * 32 + l31) * STG_PITCH + (co16 - n0) * 2) = v;
            else if (co16 < p.Cout && op[mi] >= 0) *reinterpret_cast<imagen_u32x4*>(y + (size_t)op[mi] * p.ldy + co16) = v;
          }
        }
      }
    if constexpr (STG) {
      if (wide) cl_copy_out<32 * MI * WM, BN, 64 * WM * WN>(p, tc, stg);
    }
    return;
  }

  float ssq_px[MI], gca_k[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) ssq_px[mi] = gca_k[mi] = 0.0f;

  if constexpr (!GEN) {   // plain NHWC output (optionally + ssq_out): branch-free
    f16* y = reinterpret_cast<f16*>(p.y) + (size_t)b * p.bsy;
    const bool wide = (p.Cout & 7) == 0;   // 16-byte pieces (imagen_pair_quads, common.h): quads q and q + 2 are produced together
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        __builtin_amdgcn_sched_barrier(0);   // one pair of channel quads at a time (register footprint)
        f16x4 o2[2][MI];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = qp + 2 * h;
          const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
          const float4 bq = load_bias(co);   // co < Cout_pad always
          const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
          const float4 wq = p.gca_part ? *reinterpret_cast<const float4*>(ep_par + 3 * BN + (co - n0)) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float ww[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o2[h][mi][e] = (f16)(acc[ni][mi][4 * q + e] + bb[e]);
              const float r = (float)o2[h][mi][e];
              ssq_px[mi] += r * r;
              acc[ni][mi][4 * q + e] = r;          // (the GlobalContext block below reads the stored values back from here)
              gca_k[mi] += r * ww[e];
            }
            if (!wide && co < p.Cout && op[mi] >= 0) *reinterpret_cast<f16x4*>(y + (size_t)op[mi] * p.ldy + co) = o2[h][mi];
          }
        }
        if (wide) {
          const int co16 = n0 + (wn * NI + ni) * 32 + 8 * qp + 16 * half;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const imagen_u32x4 v = imagen_pair_quads(o2[0][mi], o2[1][mi]);
            if constexpr (STG) *reinterpret_cast<imagen_u32x4*>(stg + ((wm * MI + mi) * 32 + l31) * STG_PITCH + (co16 - n0) * 2) = v;
            else if (co16 < p.Cout && op[mi] >= 0) *reinterpret_cast<imagen_u32x4*>(y + (size_t)op[mi] * p.ldy + co16) = v;
          }
        }
      }
    if constexpr (STG) {
      if (wide) cl_copy_out<32 * MI * WM, BN, 64 * WM * WN>(p, tc, stg);
    }
    if (p.gca_part) {
      // ---- GlobalContext partials of this tile (ip.py:965-968; launcher: one tile covers all Cout): logit[px] = h[px, :].wk + bk,
      // (max, sum exp, sum exp * h[px, c]) over the tile's pixels -> part[b][tile][C + 2], merged over the tiles by GCA_FINAL.
      // h = the fp16 values just stored (kept in the accumulator registers by the store loop above).  Replaces the GCA_PARTIAL
      // launch and its re-read of the whole tensor.  Reductions over the 32 pixel lanes of a half-wave run on the VALU (DPP row
      // shifts + row broadcast: 5 instructions per value, result in lanes 31 / 63), not through the LDS crossbar.
      float* gk = ep_par + 4 * BN;            // [WM * WN waves][PXW] logit partials
      float* gm = gk + WM * WN * PXW;         // [8] wave-row maxima
      float* gs = gm + 8;                     // [WM][BN + 4] weighted channel sums (+ sum exp at [BN])
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) gca_k[mi] += __shfl_xor(gca_k[mi], 32);
      if (WN > 1) {
        if (half == 0) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) gk[(wm * WN + wn) * PXW + mi * 32 + l31] = gca_k[mi];
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          float t = 0.0f;
#pragma unroll
          for (int w = 0; w < WN; ++w) t += gk[(wm * WN + w) * PXW + mi * 32 + l31];
          gca_k[mi] = t;
        }
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        gca_k[mi] = op[mi] >= 0 ? gca_k[mi] + p.gca_bk : -3.0e38f;
        mx = fmaxf(mx, gca_k[mi]);
      }
      mx = __builtin_amdgcn_readlane(cl_half_reduce<true>(mx), 63) ;   // (both half-waves hold the same pixels)
      if (WM > 1) {
        if (l31 == 0 && half == 0 && wn == 0) gm[wm] = mx;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < WM; ++w) mx = fmaxf(mx, gm[w]);
      }
      float ew[MI], se = 0.0f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        ew[mi] = op[mi] >= 0 ? __expf(gca_k[mi] - mx) : 0.0f;
        se += ew[mi];
      }
      se = cl_half_reduce<false>(se);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cl = (wn * NI + ni) * 32 + 8 * q + 4 * half;
          float sv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = 0.0f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) t += ew[mi] * acc[ni][mi][4 * q + e];
            sv[e] = cl_half_reduce<false>(t);
          }
          if (l31 == 31) *reinterpret_cast<float4*>(gs + wm * (BN + 4) + cl) = make_float4(sv[0], sv[1], sv[2], sv[3]);
        }
      if (l31 == 31 && half == 0 && wn == 0) gs[wm * (BN + 4) + BN] = se;
      __syncthreads();
      const int tilesX = (p.OW + p.TW - 1) / p.TW, tilesY = (p.OH + p.TH - 1) / p.TH;
      float* out = p.gca_part + ((size_t)(b * tilesY + tc.oy0 / p.TH) * tilesX + tc.ox0 / p.TW) * (p.Cout + 2);
      const int i = threadIdx.x;
      if (i < BN && n0 + i < p.Cout) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < WM; ++w) t += gs[w * (BN + 4) + i];
        out[2 + n0 + i] = t;
      }
      if (i == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < WM; ++w) t += gs[w * (BN + 4) + BN];
        out[0] = mx;
        out[1] = t;
      }
    }
  } else {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
        if (co >= p.Cout) continue;
        const float4 bq = load_bias(co);
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (addend) g = *reinterpret_cast<const float4*>(p.gate + (size_t)b * p.gate_stride + co);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          if (op[mi] < 0) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * q + e] + bb[e];
          if (p.act_out == IMAGEN_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
          } else if (p.act_out == IMAGEN_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
          }
          if (p.out_mode == IMAGEN_OUT_NCHW_F32) {
            float* y = reinterpret_cast<float*>(p.y);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (co + e < p.Cout) y[((size_t)b * p.Cout + co + e) * (p.OH * p.OW) + op[mi]] = v[e];
            continue;
          }
          if (addend) {
            const f16x4 ad = *reinterpret_cast<const f16x4*>(addend + (size_t)b * p.bs_add + (size_t)op[mi] * p.ld_add + co);
            v[0] += (float)ad[0] * g.x; v[1] += (float)ad[1] * g.y; v[2] += (float)ad[2] * g.z; v[3] += (float)ad[3] * g.w;
          } else if (res) {
            const f16x4 rr = *reinterpret_cast<const f16x4*>(res + (size_t)b * p.bs_res + (size_t)op[mi] * p.ld_res + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
          }
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (f16)v[e];
            const float r = (float)o[e];
            ssq_px[mi] += r * r;
          }
          f16* y = reinterpret_cast<f16*>(p.y);
          if (p.out_mode == IMAGEN_OUT_PIXEL_SHUFFLE) {
            const int Cq = p.Cout >> 2;
            const int sub = co / Cq, cc = co - sub * Cq;
            const int oy = tc.oy0 + pix_y[mi], ox = tc.ox0 + pix_x[mi];
            const int yy = 2 * oy + (sub >> 1), xx = 2 * ox + (sub & 1);
            *reinterpret_cast<f16x4*>(y + (size_t)b * p.bsy + ((size_t)yy * (2 * p.OW) + xx) * p.ldy + cc) = o;
          } else {
            *reinterpret_cast<f16x4*>(y + (size_t)b * p.bsy + (size_t)op[mi] * p.ldy + co) = o;
          }
        }
      }
  }
  if (p.ssq_out) {   // launcher guarantees tilesN == 1
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) ssq_px[mi] += __shfl_xor(ssq_px[mi], 32);
    if (WN == 1) {
      if (half == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          if (op[mi] >= 0) p.ssq_out[(size_t)b * (p.OH * p.OW) + op[mi]] = ssq_px[mi];
      }
    } else {
      if (half == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ep_red[(wm * WN + wn) * PXW + mi * 32 + l31] = ssq_px[mi];
      }
      __syncthreads();
      if (wn == 0 && half == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          float tot = 0.0f;
#pragma unroll
          for (int w = 0; w < WN; ++w) tot += ep_red[(wm * WN + w) * PXW + mi * 32 + l31];
          if (op[mi] >= 0) p.ssq_out[(size_t)b * (p.OH * p.OW) + op[mi]] = tot;
        }
      }
    }
  }
}
