// conv_dma.hip — the all-DMA 3x3 convolution for gfx950 (third igemm family): the instruction-lean form of conv_lds.hip for inputs that
// need no prologue (block2 of a ResnetBlock behind block1's post_pa epilogue; block1 behind an ACT_PREP pass; ip.py:671-691).
//
// Why it exists (profiles/r02_conv_probe_b.txt, r02_pmc_sq_*.json): with the activation prologue computed in the conv kernel a wave
// issues ~11.5 instructions per MFMA — 45 % of them the load -> fp32 transform -> ds_write staging, 25 % address arithmetic for the
// weight DMA — against a budget of 8 issue slots per 32-cycle v_mfma_f32_32x32x16_f16 (one wave per SIMD).  Here
//   * BOTH operands are copied global -> LDS by global_load_lds_dwordx4: no VGPR round trip, no VALU, no ds_write;
//   * the halo tile is DENSE in LDS ([halo row][halo pixel][4 x 16 B]); the bank-conflict-free image is obtained by swizzling the
//     SOURCE: lane l of a DMA instruction fills LDS slot l, and fetches the 8-channel group kg = slot ^ ((x >> 1) & 3) of its pixel;
//     the MFMA B-fragment read applies the same involution (verified exhaustively for 8- and 16-pixel-wide tiles: every
//     ds_read_b128 lane group hits 16 distinct 16-byte bank slots for all 9 taps);  out-of-image pixels fetch from a zero page;
//   * every LDS address in the k loop is a per-lane register + an IMMEDIATE: tile width, ring depth and the chunk parity are
//     compile-time, so the loop body is ds_read / s_waitcnt / v_mfma plus 3 instructions per DMA;
//   * waves split the output channels (WN-major) and own private weight rings (counted vmcnt, no barrier); the activation double
//     buffer is handed over by ONE barrier per 32-channel chunk (18 K steps).
// Contract, packed weight layout and epilogue are those of the other families (ImagenIgemmParams; conv_epilogue.h).
#include <algorithm>
#include <type_traits>
#include <utility>
#include "common.h"
#include "conv_epilogue.h"

namespace {

template <class F, int... I>
__device__ __forceinline__ void cd_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cd_static_for(F&& f) {
  cd_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// direct-to-LDS copy (1 KiB per wave instruction), LDS base and the two barriers: lds_dma.h
#define cd_dma16 IMAGEN_DMA16
#define CD_LDS_BASE IMAGEN_LDS_BASE
#define CD_VM0_BARRIER IMAGEN_VM0_BARRIER
#define CD_LGKM0_BARRIER IMAGEN_LGKM0_BARRIER
// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]) + a compiler-level fence
#define CD_WAIT_VM(n)                                                                         \
  do {                                                                                        \
    __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14)); \
    asm volatile("" ::: "memory");                                                            \
  } while (0)

// PF: fragment prefetch distance in K steps (PF + 1 register sets).  With PF = 1 a wave that finds its ds_reads slower than one K step
// (LDS latency under 20 reads + 5 KiB of DMA writes per K step and CU: 250-300 cycles against 128 cycles of MFMA work) stalls every
// step: r02_pmc_sq_dma.json shows 46 % of the wave cycles in s_waitcnt at 37 % MFMA-pipe occupancy.
template <int MI, int NI, int WM, int WN, int TW, int RW, int PF, bool GEN>
__global__ __launch_bounds__(256, (MI * NI <= 2 ? 3 : 2)) void conv_dma_kernel(const ImagenIgemmParams p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(PF >= 1 && PF <= 3, "fragment prefetch distance");
  constexpr int NS = PF + 1;
  static_assert(RW == 3 || RW == 6 || RW == 9, "ring depths with compile-time slot indices (9 taps per chunk, two chunk parities)");
  constexpr int TP = 32 * MI * WM, TH = TP / TW, BN = 32 * NI * WN;
  constexpr int ITW = TW + 2, ITH = TH + 2, PITCH = ITW * 64;
  constexpr int NSLOT = ITH * ITW * 4;                  // 16-byte slots of one (dense) halo tile
  constexpr int NDMA = (NSLOT + 63) / 64, NJ = (NDMA + 3) / 4;   // DMA instructions per tile, per wave (all waves issue NJ: uniform vmcnt)
  constexpr int ABUF = NJ * 4 * 1024;
  constexpr int SLOTW = NI * 2048;                      // one wave-private weight stage: [NI][4 groups][32 couts][8 halves]
  constexpr int KD = NI * 2;                            // weight DMA instructions per wave and stage
  constexpr int PXW = 32 * MI;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  const int tilesX = (p.OW + TW - 1) / TW;
  const int tilesY = (p.OH + TH - 1) / TH;
  const int tilesN = (p.Cout + BN - 1) / BN;
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
    tc.oy0 = ty * TH;
    tc.ox0 = tx * TW;
    tc.n0 = nt * BN;
  }
  const int NC = p.Cin_pad >> 5;
  const unsigned lds0 = CD_LDS_BASE(smem);
  const unsigned ring0 = lds0 + 2 * ABUF + wave * (RW * SLOTW);
  float* const ep_red = reinterpret_cast<float*>(smem + 2 * ABUF + 4 * RW * SLOTW);

  // ---- weight stream: per-lane source pointers of the KD pieces of a stage (lanes 0-31: even 8-channel group, 32-63: odd), advanced
  //      by one stage (4 packed group rows) after every issue
  const size_t wrow = (size_t)p.Cout_pad * 16;
  const char* wsrc[NI][2];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      wsrc[ni][j] = reinterpret_cast<const char*>(p.w) + ((size_t)(half + 2 * j) * p.Cout_pad + tc.n0 + wn * (NI * 32) + ni * 32 + l31) * 16;
  const size_t wstage = 4 * wrow;
  auto dma_weight_piece = [&](int slot, int ni, int j) __attribute__((always_inline)) {   // slot / ni / j: compile-time at every call site
    cd_dma16(wsrc[ni][j], __builtin_amdgcn_readfirstlane(ring0 + slot * SLOTW + ni * 2048 + j * 1024));
    wsrc[ni][j] += wstage;
  };
  auto dma_weights = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 2; ++j) dma_weight_piece(slot, ni, j);
  };

  // ---- activation stream: slot S = (wave + 4 j) * 64 + lane of the dense halo tile = (halo pixel S >> 2, position S & 3); the lane
  //      fetches channel group (S & 3) ^ ((hx >> 1) & 3) of that pixel, or 16 zero bytes outside the image / the tile
  const char* zero_src = reinterpret_cast<const char*>(p.w) + (size_t)(NC * 36) * wrow;   // the packed buffer's zero tail
  const char* asrc[NJ];
  unsigned ainc[NJ];
  {
    const f16* xb = reinterpret_cast<const f16*>(p.x1) + (size_t)tc.b * p.bs1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int S = (wave + 4 * j) * 64 + lane;
      const int hp = S >> 2, pos = S & 3;
      const int r = hp / ITW, hx = hp - r * ITW;
      const int gy = tc.oy0 - 1 + r, gx = tc.ox0 - 1 + hx;
      const bool ok = hp < ITH * ITW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const int kg = pos ^ ((hx >> 1) & 3);
      asrc[j] = ok ? reinterpret_cast<const char*>(xb + (size_t)(gy * p.W + gx) * p.ld1 + kg * 8) : zero_src;
      ainc[j] = ok ? 64u : 0u;
    }
  }
  auto dma_act_piece = [&](int buf, bool last, int j) __attribute__((always_inline)) {   // last: nothing to stage any more (uniform), copy zeros
    cd_dma16(last ? zero_src : asrc[j], __builtin_amdgcn_readfirstlane(lds0 + buf * ABUF + (wave + 4 * j) * 1024));
    asrc[j] += ainc[j];
  };
  auto dma_acts = [&](int buf, bool last) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) dma_act_piece(buf, last, j);
  };

  // ---- MFMA side: B-fragment address of (pixel fragment mi, tap column dx, K step ks) relative to a halo buffer; the tap row and
  //      the buffer are immediates
  int pix_y[MI], pix_x[MI];
  int bP[MI][3];   // K step 0; K step 1 reads channel groups 2 + half: the same address with bit 5 flipped
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int tp = (wm * MI + mi) * 32 + l31;
    const int py = tp / TW, px = tp - py * TW;
    pix_y[mi] = py;
    pix_x[mi] = px;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = px + dx;
      bP[mi][dx] = py * PITCH + hx * 64 + ((half ^ ((hx >> 1) & 3)) << 4);
    }
  }
  const int aL = 2 * ABUF + wave * (RW * SLOTW) + lane * 16;   // A fragment: + slot * SLOTW + ni * 2048 + ks * 1024

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.0f;

  struct Frags { f16x8 a[NI], b[MI]; };
  Frags F[NS];   // (indexed by compile-time constants only)
  // fragment f of a K step: f < NI: A fragment f, else B fragment f - NI (all arguments compile-time)
  auto read_frag = [&](Frags& Fr, int f, int buf, int t, int slot, int ks) __attribute__((always_inline)) {
    const int dy = t / 3, dx = t - 3 * dy;
    if (f < NI) {
      Fr.a[f] = *reinterpret_cast<const f16x8*>(smem + aL + (slot * SLOTW + f * 2048 + ks * 1024));
    } else {
      const int mi = f - NI;
      Fr.b[mi] = *reinterpret_cast<const f16x8*>(smem + (ks ? bP[mi][dx] ^ 32 : bP[mi][dx]) + (buf * ABUF + dy * PITCH));
    }
  };
  auto read_frags = [&](Frags& Fr, int buf, int t, int slot, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < NI + MI; ++f) read_frag(Fr, f, buf, t, slot, ks);
  };
  auto mfma_step = [&](const Frags& Fr) __attribute__((always_inline)) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fr.a[ni], Fr.b[mi], acc[ni][mi], 0, 0, 0);
  };

  // ================================================================================================ pipeline
  // L2 warm-up: the weights of a layer were last touched a whole denoiser step ago — every stage of every workgroup would start with
  // a miss to the Infinity Cache that a 2-5 stage look-ahead cannot cover.  The workgroups of an XCD (blockIdx % 8 by observation;
  // only speed depends on it) each touch their share of the packed weights once, one dword per 128-byte line, before anything
  // else is issued; nothing waits for it except the first vmcnt(0) below, where the first stages' DMAs wait anyway.
  {
    const size_t wbytes = (size_t)(NC * 36) * wrow;
    const unsigned nloc = (gridDim.x + 7) >> 3, lw = blockIdx.x >> 3;
    const size_t per = ((wbytes + nloc - 1) / nloc + 127) & ~(size_t)127;
    const char* base = reinterpret_cast<const char*>(p.w) + (size_t)lw * per;
    const size_t lim = (size_t)lw * per < wbytes ? min(per, wbytes - (size_t)lw * per) : 0;
    // (a direct-to-LDS load into the epilogue scratch, 4 bytes per lane: an asm load with a VGPR destination would land in a register
    // the compiler has long since reused)
    const unsigned sink = __builtin_amdgcn_readfirstlane(lds0 + 2 * ABUF + 4 * RW * SLOTW);   // 256 bytes, every wave the same (never read)
    for (size_t off = (size_t)tid * 128; off < lim; off += 256 * 128) IMAGEN_WARM_DMA4(base + off, sink);
  }
  dma_acts(0, false);
#pragma unroll
  for (int j = 0; j < RW - 1; ++j) dma_weights(j);
  CD_VM0_BARRIER();
  cd_static_for<NS>([&](auto kc) __attribute__((always_inline)) {   // (set PF is overwritten by the first loop step)
    constexpr int k = decltype(kc)::value;
    read_frags(F[k % NS], 0, k / 2, (k / 2) % RW, k % 2);
  });

  // one 32-channel chunk = 9 taps x 2 K steps; PAR = chunk parity (halo buffer, fragment-set phase, and the ring phase when 9 % RW != 0).
  //
  // A wave is IN-ORDER: while an MFMA waits for the matrix pipe (32 cycles per 32x32x16) nothing behind it issues, so a step written
  // as "all loads, then all MFMAs" overlaps its loads only with the LAST MFMA (round-2 probe dma_probe.py: the MFMAs alone 17 us, everything
  // else alone 12 us, together 25 us of a 31 us launch).  Each step is therefore issued as MFMA, a few fillers, MFMA, a few fillers
  // ...: the fragment reads of the step PF ahead first, then (first K step of a stage) the DMA pieces that refill the weight ring
  // and, at tap 0, the next chunk's halo tile — 2-3 single-issue instructions per 32-cycle MFMA slot, pinned with sched_barrier.
  //
  // vmcnt budget when stage u (chunk-relative, 9.. = the next chunk) is awaited in K step (t, ks): the weight DMAs of the stages
  // u+1 .. L are younger (L = t+RW-1 once this stage's refill has been issued, i.e. at ks = 1, else t+RW-2), and so are this chunk's
  // activation DMAs (issued in step (0, 0), ahead of that step's refill) if u <= RW-2 and the wait sits behind step (0, 0).
  auto chunk = [&](auto parc, int c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value;
    constexpr int S0 = (9 * PAR) % RW;   // ring slot of tap 0 of this chunk
    const bool last = c + 1 >= NC;
    cd_static_for<18>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      constexpr int t = k / 2, ks = k % 2;
      constexpr int kk = k + PF;                                   // the K step whose fragments are read now
      constexpr int u = kk / 2;                                    // its stage
      constexpr int L = ks == 1 ? t + RW - 1 : t + RW - 2;
      constexpr int N = (L - u) * KD + ((u <= RW - 2 && k > 0) ? NJ : 0);
      static_assert(L - u >= 0, "stage not issued yet");
      constexpr bool new_stage = kk % 2 == 0;                      // first read of stage u: this wave's own weight DMA must have landed
      constexpr int M = MI * NI, R = NI + MI;
      constexpr int DA = (ks == 0 && t == 0) ? NJ : 0;
      constexpr int DW = ks == 0 ? KD : 0;
      constexpr int NF = R + DA + DW, per = (NF + M - 1) / M;
      Frags& cur = F[(18 * PAR + k) % NS];
      Frags& nx = F[(18 * PAR + kk) % NS];
      cd_static_for<M>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int ni = g / MI, mi = g % MI;
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[ni], cur.b[mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g == 0) {
          if constexpr (new_stage) CD_WAIT_VM(N);
          // chunk boundary: every wave's part of the next halo tile has landed, everybody is done reading this one
          if constexpr (kk == 18) CD_LGKM0_BARRIER();
        }
        cd_static_for<per>([&](auto fc) __attribute__((always_inline)) {
          constexpr int f = g * per + decltype(fc)::value;
          if constexpr (f < R) {
            if constexpr (kk < 18) read_frag(nx, f, PAR, u, (S0 + u) % RW, kk % 2);
            else read_frag(nx, f, PAR ^ 1, u - 9, (S0 + u) % RW, kk % 2);
          } else if constexpr (f < R + DA) {
            dma_act_piece(PAR ^ 1, last, f - R);                   // the next chunk's halo tile (its buffer was released by the last barrier)
          } else if constexpr (f < NF) {
            dma_weight_piece((S0 + t + RW - 1) % RW, (f - R - DA) / 2, (f - R - DA) % 2);   // refill the slot of the stage that just finished
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };
  for (int c = 0; c < NC; c += 2) {
    chunk(std::integral_constant<int, 0>{}, c);
    if (c + 1 >= NC) break;
    chunk(std::integral_constant<int, 1>{}, c + 1);
  }
  CD_WAIT_VM(0);   // stray look-ahead DMAs must not outlive the workgroup's LDS allocation
  __syncthreads();

  cl_epilogue<MI, NI, WM, WN, GEN>(p, tc, acc, pix_y, pix_x, ep_red, reinterpret_cast<float*>(smem), wm, wn, half, l31);
}

template <int MI, int NI, int WM, int WN, int TW, int RW, int PF, bool GEN>
int cd_launch_gen(const ImagenIgemmParams& p, hipStream_t s) {
  constexpr int TP = 32 * MI * WM, TH = TP / TW, BN = 32 * NI * WN;
  constexpr int NJ = (((TH + 2) * (TW + 2) * 4 + 63) / 64 + 3) / 4;
  IMAGEN_CHECK(p.TH == TH && p.TW == TW, "conv_dma: cfg %d has %dx%d tiles (got %dx%d)", p.cfg, TH, TW, p.TH, p.TW);
  IMAGEN_CHECK(p.stride == 1 && p.KH == 3 && p.KW == 3 && p.pad == 1, "conv_dma: 3x3 stride-1 convolutions only");
  IMAGEN_CHECK(!p.x2 && p.C2 == 0 && !p.mu && !p.rs && !p.pa && !p.ps && !p.ssq_a && p.act_in == IMAGEN_ACT_NONE,
               "conv_dma: single input without prologue only (run ACT_PREP first)");
  IMAGEN_CHECK(p.Cin_pad == p.C1 && p.C1 % 32 == 0 && p.ld1 % 8 == 0, "conv_dma: C1 %d must be a multiple of 32 (ld1 %d of 8)", p.C1, p.ld1);
  IMAGEN_CHECK(p.Cout_pad % BN == 0, "conv_dma: Cout_pad %d not a multiple of %d", p.Cout_pad, BN);
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NCHW_F32 || p.Cout % 4 == 0, "conv_dma: Cout %d must be a multiple of 4", p.Cout);
  IMAGEN_CHECK(p.out_mode != IMAGEN_OUT_PIXEL_SHUFFLE || p.Cout % 16 == 0, "conv_dma: pixel-shuffle needs Cout %% 16 == 0");
  IMAGEN_CHECK(!p.post_pa || (p.post_ps && p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN && !p.addend && !p.res && !p.ssq_out &&
                              p.act_out == IMAGEN_ACT_NONE && p.Cout % 4 == 0),
               "conv_dma: post_pa needs post_ps, a plain NHWC output and one workgroup covering all %d output channels (tile has %d)", p.Cout, BN);
  IMAGEN_CHECK(!(p.addend && p.res), "conv_dma: addend and residual are mutually exclusive");
  IMAGEN_CHECK(!p.gca_part || (p.gca_wk && !GEN && !p.post_pa && p.Cout <= BN), "conv_dma: gca_part needs gca_wk, a plain NHWC output and one tile covering all %d couts", p.Cout);
  IMAGEN_CHECK(!p.ssq_out || (p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN),
               "conv_dma: ssq_out needs NHWC output and one workgroup covering all %d output channels (tile has %d)", p.Cout, BN);
  const size_t lds = (size_t)2 * NJ * 4096 + (size_t)4 * RW * (NI * 2048) + (size_t)(4 * 32 * MI) * sizeof(float) + 16;
  IMAGEN_CHECK(lds <= 160 * 1024, "conv_dma: LDS tile %zu bytes too large", lds);
  auto kern = conv_dma_kernel<MI, NI, WM, WN, TW, RW, PF, GEN>;
  static bool attr_done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("conv_dma: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev] = true;
  }
  const int tilesX = (p.OW + TW - 1) / TW, tilesY = (p.OH + TH - 1) / TH;
  const int total = p.B * tilesX * tilesY * ((p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(total), dim3(256), lds, s, p);
  return imagen_hip_status("conv_dma launch");
}

template <int MI, int NI, int WM, int WN, int TW, int RW, int PF = 1>
int cd_launch(const ImagenIgemmParams& p, hipStream_t s) {
  const bool plain = p.act_out == IMAGEN_ACT_NONE && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res;
  return plain ? cd_launch_gen<MI, NI, WM, WN, TW, RW, PF, false>(p, s) : cd_launch_gen<MI, NI, WM, WN, TW, RW, PF, true>(p, s);
}


struct CdCfg { int MI, NI, WM, WN, TW, RW, PF; };
constexpr CdCfg kCdCfgs[] = {
    {4, 1, 1, 4, 16, 3, 1},   //  0: 128 px (8x16) x 128 co
    {4, 1, 1, 4, 16, 6, 1},   //  1: ... weight ring 6 stages deep
    {2, 1, 1, 4, 8, 3, 1},    //  2:  64 px (8x8)  x 128 co
    {2, 1, 1, 4, 8, 6, 1},    //  3
    {2, 2, 1, 4, 8, 3, 1},    //  4:  64 px (8x8)  x 256 co
    {2, 2, 1, 4, 8, 6, 1},    //  5
    {4, 1, 2, 2, 16, 3, 1},   //  6: 256 px (16x16) x  64 co
    {2, 1, 2, 2, 16, 3, 1},   //  7: 128 px (8x16) x  64 co
    {2, 1, 2, 2, 16, 6, 1},   //  8
    {1, 1, 2, 2, 8, 6, 1},    //  9:  64 px (8x8)  x  64 co
    {2, 1, 4, 1, 16, 3, 1},   // 10: 256 px (16x16) x  32 co
    {1, 1, 4, 1, 16, 3, 1},   // 11: 128 px (8x16) x  32 co
    {2, 1, 1, 4, 8, 9, 1},    // 12:  64 px (8x8)  x 128 co, ring = a whole chunk
    {4, 1, 1, 4, 16, 6, 2},   // 13: 128 px x 128 co, fragments prefetched 2 K steps ahead
    {4, 1, 1, 4, 16, 6, 3},   // 14: ... 3 K steps ahead
    {2, 2, 1, 4, 8, 6, 2},    // 15:  64 px x 256 co, 2 K steps ahead
    {2, 1, 1, 4, 8, 6, 2},    // 16:  64 px x 128 co, 2 K steps ahead
    {2, 1, 1, 4, 8, 6, 3},    // 17:  64 px x 128 co, 3 K steps ahead
    {2, 1, 2, 2, 16, 6, 2},   // 18: 128 px x  64 co, 2 K steps ahead
    {4, 1, 2, 2, 16, 6, 2},   // 19: 256 px x  64 co, 2 K steps ahead
};
constexpr int kNumCdCfgs = sizeof(kCdCfgs) / sizeof(kCdCfgs[0]);

}  // namespace

int imagen_conv_dma_num_configs() { return kNumCdCfgs; }

int imagen_conv_dma_ring(int idx) { return (idx < 0 || idx >= kNumCdCfgs) ? -1 : kCdCfgs[idx].RW; }

int imagen_conv_dma_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (idx < 0 || idx >= kNumCdCfgs) return -1;
  const CdCfg& c = kCdCfgs[idx];
  if (tile_pixels) *tile_pixels = 32 * c.MI * c.WM;
  if (tile_cout) *tile_cout = 32 * c.NI * c.WN;
  if (kgroups) *kgroups = 4;
  return 0;
}

long imagen_conv_dma_lds_bytes(int idx, int KH, int KW, int TH, int TW) {
  if (idx < 0 || idx >= kNumCdCfgs || KH != 3 || KW != 3) return -1;
  const CdCfg& c = kCdCfgs[idx];
  if (TW != c.TW || TH * TW != 32 * c.MI * c.WM) return -1;
  const int NJ = (((TH + 2) * (TW + 2) * 4 + 63) / 64 + 3) / 4;
  return 2L * NJ * 4096 + 4L * c.RW * (c.NI * 2048) + 4L * 32 * c.MI * 4 + 16;
}

int launch_conv_dma(const ImagenIgemmParams* pp, int idx, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  switch (idx) {
    case 0: return cd_launch<4, 1, 1, 4, 16, 3>(p, s);
    case 1: return cd_launch<4, 1, 1, 4, 16, 6>(p, s);
    case 2: return cd_launch<2, 1, 1, 4, 8, 3>(p, s);
    case 3: return cd_launch<2, 1, 1, 4, 8, 6>(p, s);
    case 4: return cd_launch<2, 2, 1, 4, 8, 3>(p, s);
    case 5: return cd_launch<2, 2, 1, 4, 8, 6>(p, s);
    case 6: return cd_launch<4, 1, 2, 2, 16, 3>(p, s);
    case 7: return cd_launch<2, 1, 2, 2, 16, 3>(p, s);
    case 8: return cd_launch<2, 1, 2, 2, 16, 6>(p, s);
    case 9: return cd_launch<1, 1, 2, 2, 8, 6>(p, s);
    case 10: return cd_launch<2, 1, 4, 1, 16, 3>(p, s);
    case 11: return cd_launch<1, 1, 4, 1, 16, 3>(p, s);
    case 12: return cd_launch<2, 1, 1, 4, 8, 9>(p, s);
    case 13: return cd_launch<4, 1, 1, 4, 16, 6, 2>(p, s);
    case 14: return cd_launch<4, 1, 1, 4, 16, 6, 3>(p, s);
    case 15: return cd_launch<2, 2, 1, 4, 8, 6, 2>(p, s);
    case 16: return cd_launch<2, 1, 1, 4, 8, 6, 2>(p, s);
    case 17: return cd_launch<2, 1, 1, 4, 8, 6, 3>(p, s);
    case 18: return cd_launch<2, 1, 2, 2, 16, 6, 2>(p, s);
    case 19: return cd_launch<4, 1, 2, 2, 16, 6, 2>(p, s);
  }
  imagen_set_error("conv_dma: bad cfg index %d", idx);
  return -1;
}
