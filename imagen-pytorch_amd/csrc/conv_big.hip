// conv_big.hip — the big-tile all-DMA 3x3 convolution for gfx950 (kernel family 5): prologue-free 3x3 stride-1 convs with 128-channel
// output tiles (Block = ... -> Conv3x3, ip.py:671-691, behind an activated input: ACT_PREP, a GCA_TAIL or a post_pa epilogue).
//
// Why it exists: the two C >= 128 kernels (igemm.hip cfg 3, conv_dma.hip) give a wave a 64 x 32 output tile, i.e. 1.25-1.5 KB
// of LDS operand reads per 32-cycle MFMA, and re-stage the weights of a 128-cout block for every 64-128 pixels; both sit at the operand
// delivery bound (LDS reads + direct-to-LDS writes ~1.2x the matrix-pipe time, the CU's 64 B/clk vector memory path half full of
// weights).  Here
//   * one workgroup of 8 waves owns a 256-pixel (16x16) x 128-cout tile — or 128 pixels x 128 couts with the two K=16 halves of every
//     32-channel chunk split over two wave groups (KS = 2; summed through LDS before the epilogue) where a 256-pixel tile would leave
//     half the chip idle (the 32^2 maps) — one workgroup per CU, all 160 KB of LDS;
//   * every wave computes 64 pixels x 64 couts (2 x 2 MFMA fragments): 1 KB of ds_read_b128 per MFMA, and each weight byte staged
//     once per workgroup serves 128-256 pixels;
//   * both operands arrive by global_load_lds_dwordx4 (conv_dma.hip's dense swizzled halo image; weights in SHARED ring stages of one
//     tap row = 3 taps x 32 channels x 128 couts = 24 KB);
//   * ONE workgroup barrier per tap row (72 / 36 MFMAs per wave), placed two K steps before the row ends: the wave first waits
//     (counted vmcnt) for its own pieces of the NEXT row's stage, the barrier publishes that stage and retires the previous one, whose
//     ring slot is refilled at once (stage s + WR - 1), and the fragment reads of the next row's first K steps follow in the same
//     row's last two steps — no wave ever reads a stage in the step that waited for it, and the matrix pipe holds two K steps of work
//     while the barrier resolves;
//   * fragments are prefetched two K steps ahead into three register sets; ring and halo slots are runtime offsets added per read
//     (two waves per SIMD: the issue slots are there), so ring depth and chunk count need no unrolling.
// Contract, packed weight layout and epilogue are those of the other families (ImagenIgemmParams; conv_epilogue.h).
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <utility>
#include "common.h"
#include "conv_epilogue.h"

namespace {

template <class F, int... I>
__device__ __forceinline__ void cb_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cb_static_for(F&& f) {
  cb_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// direct-to-LDS copies (1 KiB / 256 B per wave instruction), LDS base, barrier and counted vmcnt wait: lds_dma.h
#define CB_DMA16 IMAGEN_DMA16
#define CB_DMA4 IMAGEN_DMA4
#define CB_LDS_BASE IMAGEN_LDS_BASE
#define CB_BARRIER IMAGEN_BARRIER
#define CB_WAIT_VM IMAGEN_WAIT_VM

// -DCB_TRACE (tools/conv_bench.py --trace, a throw-away variant library: never in the product build): thread 0 of every workgroup keeps s_memtime
// at the phase boundaries in registers (a store inside the pipeline would shift the counted vmcnt waits) and writes them, the constant-clock time
// of entry and exit and the hardware ids of its CU behind the epilogue, into a buffer handed over by imagen_debug_conv_big_trace() (eight
// slots of 4096 workgroups: the launcher numbers the launches, so that back-to-back launches can be told apart).
#ifdef CB_TRACE
__device__ unsigned long long* g_cb_trace = nullptr;
#define CB_TRACE_DECL()                                  \
  unsigned long long cb_t[8] = {};                       \
  const unsigned cb_slot = (p.launcher_word >> 16) & 7;  \
  const unsigned long long cb_r0 = __builtin_amdgcn_s_memrealtime()
#define CB_STAMP(i) (cb_t[i] = __builtin_amdgcn_s_memtime())
#define CB_TRACE_FLUSH()                                                                       \
  do {                                                                                         \
    if (threadIdx.x == 0 && g_cb_trace) {                                                      \
      unsigned long long* o = g_cb_trace + ((size_t)cb_slot * 4096 + blockIdx.x) * 16;        \
      for (int i = 0; i < 8; ++i) o[i] = cb_t[i];                                              \
      o[8] = cb_r0;                                                                            \
      o[9] = __builtin_amdgcn_s_memrealtime();                                                 \
      o[10] = __builtin_amdgcn_s_getreg((3 << 11) | 20);  /* XCC_ID */                         \
      o[11] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  /* HW_ID */                          \
    }                                                                                          \
  } while (0)
#else
#define CB_TRACE_DECL() ((void)0)
#define CB_STAMP(i) ((void)0)
#define CB_TRACE_FLUSH() ((void)0)
#endif

constexpr int cb_halo_pieces(int TH, int TW) { return (((TH + 2) * (TW + 2) * 4 + 63) / 64 + 7) / 8; }   // DMA instructions per wave and halo tile
constexpr int cb_scratch_floats(int WM, int WN) { return 4 * 64 * WN + WM * WN * 64 + 8 + WM * (64 * WN + 4) + WM * WN * 64; }   // ep_par + ep_red
// LDS image: [pipeline: HR halo buffers | WR weight stages] [epilogue operands + scratch (ep_par, ep_red)] [warm-up sink]; after the loop
// the K-group reduction slab (KS = 2, 64 KB) and the output staging tile (TP x (2 BN + 16) bytes) alias the dead pipeline memory
constexpr long cb_pipe_bytes(int WM, int WN, int TW, int WR, int HR) {
  return (long)HR * cb_halo_pieces(64 * WM / TW, TW) * 8192 + (long)WR * (3 * 4 * 64 * WN * 16);
}
constexpr long cb_lds_bytes(int WM, int WN, int KS, int TW, int WR, int HR) {
  const long pipe = cb_pipe_bytes(WM, WN, TW, WR, HR);
  const long tail = (KS == 2 ? 65536L : 0L) + (long)(64 * WM) * (2 * 64 * WN + 16);
  return (pipe > tail ? pipe : tail) + (long)cb_scratch_floats(WM, WN) * 4 + 256;
}

// WM x WN waves of 64 px x 64 couts, KS wave groups over the K=16 halves of a chunk; WR weight ring stages (tap rows), HR halo buffers
template <int WM, int WN, int KS, int TW, int WR, int HR, bool GEN>
__global__ __launch_bounds__(512, 1) void conv_big_kernel(const ImagenIgemmParams p) {
  constexpr int MI = 2, NI = 2;
  static_assert(WM * WN * KS == 8, "8 waves per workgroup");
  static_assert(WN == 2, "128-cout tiles (the weight stage pieces are dealt 3 per wave)");
  static_assert(KS == 1 || KS == 2, "K split over one or two wave groups");
  static_assert(WR >= 3 && HR >= 2, "ring depths");
  constexpr int NQ = WM * WN;                  // waves of one K group = accumulator tiles of the workgroup
  constexpr int KSW = 2 / KS;                  // K=16 steps per tap and wave
  constexpr int SPP = 3 * KSW;                 // K steps per phase (= tap row)
  constexpr int NS = 3;                        // fragment register sets: prefetch distance 2
  constexpr int SYNC_K = SPP - 2;              // the K step of a phase that holds the wait + barrier + refill
  constexpr int TP = 64 * WM, TH = TP / TW, BN = 64 * WN;
  constexpr int ITW = TW + 2, ITH = TH + 2, PITCH = ITW * 64;
  constexpr int NSLOT = ITH * ITW * 4;         // 16-byte slots of one (dense) halo tile
  constexpr int NJ = cb_halo_pieces(TH, TW);
  constexpr int ABUF = NJ * 8 * 1024;
  constexpr int GSTR = BN * 16;                // one 8-channel group of a tap: [BN couts][8 halves]
  constexpr int WSTAGE = 3 * 4 * GSTR;         // one tap row
  constexpr int KD = WSTAGE / 1024 / 8;        // weight DMA instructions per wave and stage
  static_assert(KD == 3, "weight pieces per wave");
  constexpr int RING0 = HR * ABUF;
  constexpr int PIPE = RING0 + WR * WSTAGE;
  static_assert(PIPE == cb_pipe_bytes(WM, WN, TW, WR, HR), "LDS layout");
  constexpr int STG0 = KS == 2 ? 65536 : 0;    // output staging tile (behind the K-group reduction slab), aliases the dead pipeline
  constexpr int EPP0 = (int)cb_lds_bytes(WM, WN, KS, TW, WR, HR) - 256 - cb_scratch_floats(WM, WN) * 4;   // ep_par | ep_red: never aliased
  static_assert(EPP0 >= PIPE && EPP0 >= STG0 + TP * (2 * BN + 16), "LDS layout");
  constexpr int PXW = 32 * MI;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  CB_TRACE_DECL();
  CB_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int kg = wave / NQ, wq = wave % NQ;
  const int wm = wq / WN, wn = wq % WN;

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
  const int NSTG = 3 * NC;
  const unsigned lds0 = CB_LDS_BASE(smem);
  const size_t wrow = (size_t)p.Cout_pad * 16;

  // ---- weight stream: stage s = packed rows [12 s, 12 s + 12) (row = (chunk, tap, 8-channel group)); piece q = wave * 3 + i of a stage is
  //      the 64-cout half (q & 1) of row q >> 1 and lands at ring slot + q KiB, i.e. the stage image is [tap][group][128 couts][8 halves]
  const char* wsrc[KD];
#pragma unroll
  for (int i = 0; i < KD; ++i) {
    const int q = wave * KD + i;
    wsrc[i] = reinterpret_cast<const char*>(p.w) + ((size_t)(q >> 1) * p.Cout_pad + tc.n0 + (q & 1) * 64 + lane) * 16;
  }
  const size_t wstage_bytes = 12 * wrow;
  int w_issued = 0;                // stages issued so far (the stream repeats its last stage past the end: same count for every phase)
  unsigned w_islot = 0;            // ring byte offset of the next stage to issue
  auto dma_weight_piece = [&](int i) __attribute__((always_inline)) {
    CB_DMA16(wsrc[i], __builtin_amdgcn_readfirstlane(lds0 + RING0 + w_islot + (wave * KD + i) * 1024));
  };
  auto weight_stage_issued = [&]() __attribute__((always_inline)) {
    const size_t inc = (w_issued + 1 < NSTG) ? wstage_bytes : 0;
#pragma unroll
    for (int i = 0; i < KD; ++i) wsrc[i] += inc;
    ++w_issued;
    w_islot = (w_islot + WSTAGE == (unsigned)(WR * WSTAGE)) ? 0u : w_islot + WSTAGE;
  };

  // ---- activation stream (conv_dma.hip's image): slot S = (wave + 8 j) * 64 + lane of the dense halo tile = (halo pixel S >> 2, position
  //      S & 3); the lane fetches channel group (S & 3) ^ ((hx >> 1) & 3) of that pixel, or 16 zero bytes outside the image / the tile
  const char* zero_src = reinterpret_cast<const char*>(p.w) + (size_t)(NC * 36) * wrow;   // the packed buffer's zero tail
  const char* asrc[NJ];
  unsigned ainc[NJ];
  {
    const f16* xb = reinterpret_cast<const f16*>(p.x1) + (size_t)tc.b * p.bs1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int S = (wave + 8 * j) * 64 + lane;
      const int hp = S >> 2, pos = S & 3;
      const int r = hp / ITW, hx = hp - r * ITW;
      const int gy = tc.oy0 - 1 + r, gx = tc.ox0 - 1 + hx;
      const bool ok = hp < ITH * ITW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const int kgp = pos ^ ((hx >> 1) & 3);
      asrc[j] = ok ? reinterpret_cast<const char*>(xb + (size_t)(gy * p.W + gx) * p.ld1 + kgp * 8) : zero_src;
      ainc[j] = ok ? 64u : 0u;
    }
  }
  int h_issued = 0;                // halo chunks issued so far
  unsigned h_islot = 0;
  auto dma_act_piece = [&](int j) __attribute__((always_inline)) {   // past the last chunk: zeros (uniform)
    CB_DMA16(h_issued < NC ? asrc[j] : zero_src, __builtin_amdgcn_readfirstlane(lds0 + h_islot + (wave + 8 * j) * 1024));
  };
  auto act_chunk_issued = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) asrc[j] += ainc[j];
    ++h_issued;
    h_islot = (h_islot + ABUF == (unsigned)(HR * ABUF)) ? 0u : h_islot + ABUF;
  };

  // ---- MFMA side: B-fragment address of (pixel fragment mi, tap column dx, K step ks) relative to a halo buffer (the tap row is an
  //      immediate, the buffer a runtime offset); A-fragment address of the lane relative to a ring stage
  int pix_y[MI], pix_x[MI];
  int bP[KSW][MI][3];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int tp = (wm * MI + mi) * 32 + l31;
    const int py = tp / TW, px = tp - py * TW;
    pix_y[mi] = py;
    pix_x[mi] = px;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = px + dx;
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) {
        const int grp = (KS == 2 ? kg : ks) * 2 + half;
        bP[ks][mi][dx] = py * PITCH + hx * 64 + ((grp ^ ((hx >> 1) & 3)) << 4);
      }
    }
  }
  const int aL = RING0 + (KS == 2 ? kg * 2 * GSTR : 0) + half * GSTR + (wn * 64 + l31) * 16;   // + stage + tap * 4 GSTR + ks * 2 GSTR + ni * 512

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.0f;

  struct Frags { f16x8 a[NI], b[MI]; };
  Frags F[NS];   // (indexed by compile-time constants only)
  // fragment f of K step (tap t = 3 dy + dx, ks): f < NI: A fragment f, else B fragment f - NI; wsl / hsl: ring / halo byte offsets
  auto read_frag = [&](Frags& Fr, int f, int dy, int dx, int ks, unsigned wsl, unsigned hsl) __attribute__((always_inline)) {
    if (f < NI) {
      Fr.a[f] = *reinterpret_cast<const f16x8*>(smem + (aL + wsl) + (dx * 4 * GSTR + ks * 2 * GSTR + f * 512));
    } else {
      const int mi = f - NI;
      Fr.b[mi] = *reinterpret_cast<const f16x8*>(smem + (bP[ks][mi][dx] + hsl) + dy * PITCH);
    }
  };

  // ================================================================================================ pipeline
  // Instruction warm-up (common.h) and L2 warm-up of the weights (conv_dma.hip: the workgroups of an XCD — blockIdx % 8 by observation; only
  // speed depends on it — each touch their share of the packed weights once, one dword per 128-byte line, into a sink nobody reads)
  CB_STAMP(1);
  // (round 6, call M: the warm-up as two non-blocking loads per thread BEHIND the first stage's copies — one round trip instead of two — moved
  // ~1000 cycles from this phase into the wait for the first stage: 4650 against 5040 cycles for both, the step unchanged.  The blocking loop stays.)
  const unsigned warm = imagen_code_warm(((unsigned)p.launcher_word & 0xffffu) << 8, tid, 512);
  {
    const size_t wbytes = (size_t)(NC * 36) * wrow;
    const unsigned nloc = (gridDim.x + 7) >> 3, lw = blockIdx.x >> 3;
    const size_t per = ((wbytes + nloc - 1) / nloc + 127) & ~(size_t)127;
    const char* base = reinterpret_cast<const char*>(p.w) + (size_t)lw * per;
    const size_t lim = (size_t)lw * per < wbytes ? min(per, wbytes - (size_t)lw * per) : 0;
    const unsigned sink = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(cb_lds_bytes(WM, WN, KS, TW, WR, HR) - 256));
    for (size_t off = (size_t)tid * 128; off < lim; off += 512 * 128) IMAGEN_WARM_DMA4(base + off, sink);
  }
  // the per-channel epilogue operands (bias | post_pa | post_ps | gca_wk of this tile's 128 couts: ep_par[v * BN + i]) ride at the head of
  // the copy queue, one 64-float piece per wave: the epilogue then holds no global load (conv_epilogue.h PRELOADED)
  {
    const int v = wave >> 1, co = tc.n0 + (wave & 1) * 64 + lane;
    const float* src = nullptr;
    if (v == 0) src = p.bias ? p.bias + co : nullptr;   // (padded to Cout_pad by the host)
    else if (v == 1) src = (p.post_pa && co < p.Cout) ? p.post_pa + (size_t)tc.b * p.post_pstride + co : nullptr;
    else if (v == 2) src = (p.post_pa && co < p.Cout) ? p.post_ps + (size_t)tc.b * p.post_pstride + co : nullptr;
    else src = (p.gca_part && co < p.Cout) ? p.gca_wk + co : nullptr;
    CB_DMA4(src ? reinterpret_cast<const char*>(src) : zero_src, __builtin_amdgcn_readfirstlane(lds0 + EPP0 + (v * BN + (wave & 1) * 64) * 4));
  }
  // first the halo tile and the stage of phase 0, then the rest of the look-ahead: only the former is waited for here (all CUs fill at
  // once: ~11 B / clk / CU, MI355X_MICROARCH.md), the wait of phase 0 covers stage 1 like any other
#pragma unroll
  for (int j = 0; j < NJ; ++j) dma_act_piece(j);
  act_chunk_issued();
#pragma unroll
  for (int i = 0; i < KD; ++i) dma_weight_piece(i);
  weight_stage_issued();
#pragma unroll
  for (int h = 1; h < HR - 1; ++h) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) dma_act_piece(j);
    act_chunk_issued();
  }
#pragma unroll
  for (int s = 1; s < WR - 1; ++s) {
#pragma unroll
    for (int i = 0; i < KD; ++i) dma_weight_piece(i);
    weight_stage_issued();
  }
  CB_STAMP(2);
  CB_WAIT_VM((HR - 2) * NJ + (WR - 2) * KD);
  CB_BARRIER();
  CB_STAMP(3);
  unsigned wcur = 0, hcur = 0;   // ring / halo byte offsets of the phase being multiplied
  cb_static_for<2>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    cb_static_for<NI + MI>([&](auto fc) __attribute__((always_inline)) {
      read_frag(F[k], decltype(fc)::value, 0, k / KSW, k % KSW, wcur, hcur);
    });
  });

  // A wave is in order: each K step is issued as MFMA, a few fillers, MFMA, ... (pinned with sched_barrier): the fragment reads of the
  // step two ahead, and in steps SYNC_K / SYNC_K + 1 of a phase the wait + barrier and the refill pieces.
  //
  // vmcnt budget of the wait in phase s = (chunk, dy) for stage s + 1 (issued in phase s + 2 - WR): younger are the pieces of the
  // phases s + 3 - WR .. s - 1 — KD weight pieces each, and NJ halo pieces ahead of them in the phases with dy = 0.  The halo of
  // chunk c + 1 (first read behind the barrier of phase (c, 2), whose wait is for stage 3 c + 3) needs no count of its own: it is issued
  // in phase 3 (c + 2 - HR), ahead of that phase's weight pieces, stage 3 c + 3 in phase 3 c + 4 - WR — not earlier iff WR <= 3 HR - 2.
  static_assert(3 * (HR - 1) + 1 >= WR, "the next chunk's halo must be issued no later than the stage that follows it");
  for (int c = 0; c < NC; ++c) {
    cb_static_for<3>([&](auto dyc) __attribute__((always_inline)) {
      constexpr int dy = decltype(dyc)::value;
      const unsigned wnext = (wcur + WSTAGE == (unsigned)(WR * WSTAGE)) ? 0u : wcur + WSTAGE;
      const unsigned hnext = dy == 2 ? ((hcur + ABUF == (unsigned)(HR * ABUF)) ? 0u : hcur + ABUF) : hcur;
      constexpr int N_WAIT = [] {
        int n = 0;
        for (int j = 1; j <= WR - 3; ++j) n += KD + ((((dy - j) % 3 + 3) % 3) == 0 ? NJ : 0);
        return n;
      }();
      constexpr int NPIECE = KD + (dy == 0 ? NJ : 0);   // refill pieces of this phase: the halo tile first (dy = 0), then the weights
      static_assert(NPIECE <= 7, "refill pieces fit the filler slots of two K steps");
      cb_static_for<SPP>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr int gstep = dy * SPP + k;             // K step of the chunk (SPP * 3 steps: a multiple of NS)
        constexpr int kk = k + 2;                       // the step whose fragments are read now
        constexpr bool cross = kk >= SPP;               // ... belongs to the next phase
        constexpr int k2 = cross ? kk - SPP : kk;
        constexpr int dx2 = k2 / KSW, ks2 = k2 % KSW;
        constexpr int dy2 = cross ? (dy + 1) % 3 : dy;
        Frags& cur = F[gstep % NS];
        Frags& nx = F[(gstep + 2) % NS];
        cb_static_for<4>([&](auto gc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value;
          constexpr int ni = g / MI, mi = g % MI;
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[ni], cur.b[mi], acc[ni][mi], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (k == SYNC_K && g == 0) {
            {
              CB_WAIT_VM(N_WAIT);   // this wave's pieces of the next stage (and of the next chunk's halo tile) have landed
              CB_BARRIER();         // ... everybody's have; nobody reads the previous stage (or, dy = 0, the previous chunk's halo) any more
            }
          }
          read_frag(nx, g, dy2, dx2, ks2, cross ? wnext : wcur, (cross && dy == 2) ? hnext : hcur);
          // refill: piece slots are (SYNC_K, g = 1..3) and (SYNC_K + 1, g = 0..3)
          constexpr int slot = k == SYNC_K ? g - 1 : (k == SYNC_K + 1 ? 3 + g : -1);
          if constexpr (slot >= 0 && slot < NPIECE) {
            {
              if constexpr (dy == 0 && slot < NJ) dma_act_piece(slot);
              else dma_weight_piece(slot - (dy == 0 ? NJ : 0));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (k == SYNC_K + 1) {
          if constexpr (dy == 0) act_chunk_issued();
          weight_stage_issued();
        }
      });
      wcur = wnext;
      hcur = hnext;
    });
  }
  CB_STAMP(4);
  CB_WAIT_VM(0);   // the look-ahead copies must not outlive the pipeline's LDS image (the epilogue scratch aliases it)
  __syncthreads();

  if constexpr (KS == 2) {   // sum the two K groups: group 1 parks its accumulators in LDS ([tile][register][lane]) and leaves
    float* red = reinterpret_cast<float*>(smem);
    if (kg == 1) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((wq * 4 + ni * MI + mi) * 16 + r) * 64 + lane] = acc[ni][mi][r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][mi][r] += red[((wq * 4 + ni * MI + mi) * 16 + r) * 64 + lane];
  }
  imagen_code_warm_sink(warm);
  CB_STAMP(5);
  float* const ep_par = reinterpret_cast<float*>(smem + EPP0);
  float* const ep_red = ep_par + (4 * BN + NQ * PXW + 8 + WM * (BN + 4));
  cl_epilogue<MI, NI, WM, WN, GEN, true, !GEN>(p, tc, acc, pix_y, pix_x, ep_red, ep_par, wm, wn, half, l31, smem + STG0);
  CB_STAMP(6);
  CB_TRACE_FLUSH();
}

struct CbCfg { int WM, WN, KS, TW, WR, HR; };
constexpr CbCfg kCbCfgs[] = {
    {4, 2, 1, 16, 4, 2},   // 0: 256 px (16x16) x 128 co, 8 waves of 64 x 64
    {2, 2, 2, 16, 4, 2},   // 1: 128 px (8x16)  x 128 co, 2 K groups x 4 waves of 64 x 64
    {2, 2, 2, 16, 4, 3},   // 2: ... three halo buffers (the next chunk's tile is requested a chunk earlier)
    {4, 2, 1, 16, 3, 2},   // 3: 256 px, weight ring 3 stages deep
};
constexpr int kNumCbCfgs = sizeof(kCbCfgs) / sizeof(kCbCfgs[0]);

template <int WM, int WN, int KS, int TW, int WR, int HR, bool GEN>
int cb_launch_gen(const ImagenIgemmParams& p, hipStream_t s) {
  constexpr int TP = 64 * WM, TH = TP / TW, BN = 64 * WN;
  IMAGEN_CHECK(p.TH == TH && p.TW == TW, "conv_big: cfg %d has %dx%d tiles (got %dx%d)", p.cfg, TH, TW, p.TH, p.TW);
  IMAGEN_CHECK(p.stride == 1 && p.KH == 3 && p.KW == 3 && p.pad == 1, "conv_big: 3x3 stride-1 convolutions only");
  IMAGEN_CHECK(!p.x2 && p.C2 == 0 && !p.mu && !p.rs && !p.pa && !p.ps && !p.ssq_a && p.act_in == IMAGEN_ACT_NONE,
               "conv_big: single input without prologue only (run ACT_PREP first)");
  IMAGEN_CHECK(p.Cin_pad == p.C1 && p.C1 % 32 == 0 && p.ld1 % 8 == 0, "conv_big: C1 %d must be a multiple of 32 (ld1 %d of 8)", p.C1, p.ld1);
  IMAGEN_CHECK(p.Cout_pad % BN == 0, "conv_big: Cout_pad %d not a multiple of %d", p.Cout_pad, BN);
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NCHW_F32 || p.Cout % 4 == 0, "conv_big: Cout %d must be a multiple of 4", p.Cout);
  IMAGEN_CHECK(p.out_mode != IMAGEN_OUT_PIXEL_SHUFFLE || p.Cout % 16 == 0, "conv_big: pixel-shuffle needs Cout %% 16 == 0");
  IMAGEN_CHECK(!p.post_pa || (p.post_ps && p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN && !p.addend && !p.res && !p.ssq_out &&
                              p.act_out == IMAGEN_ACT_NONE && p.Cout % 4 == 0),
               "conv_big: post_pa needs post_ps, a plain NHWC output and one workgroup covering all %d output channels (tile has %d)", p.Cout, BN);
  IMAGEN_CHECK(!(p.addend && p.res), "conv_big: addend and residual are mutually exclusive");
  IMAGEN_CHECK(!p.gca_part || (p.gca_wk && !GEN && !p.post_pa && p.Cout <= BN), "conv_big: gca_part needs gca_wk, a plain NHWC output and one tile covering all %d couts", p.Cout);
  IMAGEN_CHECK(!p.ssq_out || (p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN),
               "conv_big: ssq_out needs NHWC output and one workgroup covering all %d output channels (tile has %d)", p.Cout, BN);
  constexpr size_t lds = (size_t)cb_lds_bytes(WM, WN, KS, TW, WR, HR);
  static_assert(lds <= 160 * 1024, "conv_big: LDS image too large");
  auto kern = conv_big_kernel<WM, WN, KS, TW, WR, HR, GEN>;
  static bool attr_done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("conv_big: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev] = true;
  }
  const int tilesX = (p.OW + TW - 1) / TW, tilesY = (p.OH + TH - 1) / TH;
  const int total = p.B * tilesX * tilesY * ((p.Cout + BN - 1) / BN);
  // code size of this instantiation (for the kernel's instruction warm-up), looked up once by its mangled name
  static const unsigned code_q = [] {
    char name[160];
    snprintf(name, sizeof(name), "_ZN12_GLOBAL__N_115conv_big_kernelILi%dELi%dELi%dELi%dELi%dELi%dELb%dEEEv17ImagenIgemmParams", WM, WN, KS, TW, WR, HR,
             GEN ? 1 : 0);
    return std::min(imagen_kernel_code_bytes(name) >> 8, 0xffffu);
  }();
  ImagenIgemmParams q = p;
  q.launcher_word = (int)code_q;
#ifdef CB_TRACE
  static unsigned launch_no = 0;
  q.launcher_word |= (int)((launch_no++ & 7u) << 16);
#endif
  hipLaunchKernelGGL(kern, dim3(total), dim3(512), lds, s, q);
  return imagen_hip_status("conv_big launch");
}

template <int WM, int WN, int KS, int TW, int WR, int HR>
int cb_launch(const ImagenIgemmParams& p, hipStream_t s) {
  const bool plain = p.act_out == IMAGEN_ACT_NONE && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res;
  return plain ? cb_launch_gen<WM, WN, KS, TW, WR, HR, false>(p, s) : cb_launch_gen<WM, WN, KS, TW, WR, HR, true>(p, s);
}

}  // namespace

#ifdef CB_TRACE
extern "C" int imagen_debug_conv_big_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cb_trace), &buf, sizeof(buf)); }
#endif

int imagen_conv_big_num_configs() { return kNumCbCfgs; }

int imagen_conv_big_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (idx < 0 || idx >= kNumCbCfgs) return -1;
  const CbCfg& c = kCbCfgs[idx];
  if (tile_pixels) *tile_pixels = 64 * c.WM;
  if (tile_cout) *tile_cout = 64 * c.WN;
  if (kgroups) *kgroups = 4;
  return 0;
}

long imagen_conv_big_lds_bytes(int idx, int KH, int KW, int TH, int TW) {
  if (idx < 0 || idx >= kNumCbCfgs || KH != 3 || KW != 3) return -1;
  const CbCfg& c = kCbCfgs[idx];
  if (TW != c.TW || TH * TW != 64 * c.WM) return -1;
  return cb_lds_bytes(c.WM, c.WN, c.KS, c.TW, c.WR, c.HR);
}

int launch_conv_big(const ImagenIgemmParams* pp, int idx, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  switch (idx) {
    case 0: return cb_launch<4, 2, 1, 16, 4, 2>(p, s);
    case 1: return cb_launch<2, 2, 2, 16, 4, 2>(p, s);
    case 2: return cb_launch<2, 2, 2, 16, 4, 3>(p, s);
    case 3: return cb_launch<4, 2, 1, 16, 3, 2>(p, s);
  }
  imagen_set_error("conv_big: bad cfg index %d", idx);
  return -1;
}
