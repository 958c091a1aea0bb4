// conv_lds.hip — the LDS-staged implicit-GEMM convolution / linear layer for gfx950 (second igemm family, tile cfg ids >= 16).
// Same contract as igemm.hip (ImagenIgemmParams in include/imagen_hip.h: Block = ChanRMSNorm -> scale/shift -> SiLU -> Conv3x3,
// ip.py:671-691; res_conv ip.py:732; nn.Linear), same packed weight layout, same accumulator orientation (D[cout][pixel], lane =
// pixel) — a different machine underneath:
//
//   * BOTH MFMA operands come out of LDS.  Weights are pre-packed in fragment order [(chunk, tap, 8-channel group)][Cout_pad][8],
//     so the slice a tile needs for one (chunk, tap) "stage" is G contiguous runs of BN*16 bytes: it is copied HBM/L2 -> LDS by
//     direct-to-LDS loads (global_load_lds_dwordx4: no VGPRs, no VALU, asynchronous) into a ring of RW stage slots, RW-1 stages
//     ahead of its use.  One fetch feeds every pixel sub-tile of the workgroup (the wave-specialised family streams fragments from
//     L2 per wave: 64 B/clk/CU at full MFMA rate, the L2->CU limit).
//   * ALL 4 waves of the 256-thread workgroup issue MFMAs (wave tile MI x NI fragments of 32 px x 32 cout); 2-4 workgroups are
//     co-resident per CU (64 KB LDS, <= 128-256 VGPRs), so one workgroup's staging stalls are another one's MFMA time.
//   * The waves of a workgroup split the OUTPUT CHANNELS first (WN-major), and every wave owns a PRIVATE weight ring: it waits for
//     its own DMA with a counted vmcnt and needs no barrier for the weight stream at all.  Only the shared activation tile is
//     handed over by a barrier, once per 8*G-channel chunk (9 taps = 18 K steps of MI*NI MFMAs for a 3x3 conv) — between two
//     barriers the waves drift apart freely, so one wave's wait is covered by the others' MFMAs.
//   * Activations: halo tile of one 8*G-channel chunk, global -> registers -> prologue ((x - mu) * rs * a + s -> SiLU, fp32) -> LDS,
//     double-buffered per chunk; the loads of chunk c+1 are issued at the first tap of chunk c, transformed and written mid-chunk.
//     LDS image [halo row][pixel][8*G ch + 16 B pad] with the ROW PITCH chosen per tile width so that every ds_read_b128 lane group
//     of a B fragment hits 16 distinct 16-byte bank slots (pitch = 0 mod 256 B for 16-wide tiles, 128 mod 256 for 8-wide ones).
//   * The K loop is software-pipelined by hand: the A/B fragments of K step k+1 are read from LDS before the MFMAs of step k are
//     issued (two register sets, pinned with sched_barrier), across taps and across the chunk boundary.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.h"
#include "conv_epilogue.h"

namespace {

template <class F, int... I>
__device__ __forceinline__ void cl_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cl_static_for(F&& f) {
  cl_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __attribute__((aligned(16))) float clOnes8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
__device__ __attribute__((aligned(16))) float clZeros8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

// 16-byte staging items per thread and chunk
constexpr int cl_stage_slots(int TP, int G, int TAPS) {
  if (TAPS == 9) return TP == 64 ? 2 : TP == 128 ? 3 : 6;   // 10x10 | 10x18 | 18x18 (10x34) halo tiles of 32-channel chunks
  return TP * G / 256;                                       // 1x1: no halo
}

// s_waitcnt vmcnt(n) only (expcnt / lgkmcnt fields at their maximum = no wait); gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14].
// The builtin (not inline asm: the waitcnt pass forgets its scoreboard behind an asm statement and drains lgkmcnt(0) at the next LDS
// use) plus an empty asm as the compiler-level fence that keeps the LDS reads of the awaited slot below it.
#define CL_WAIT_VM(n)                                                                             \
  do {                                                                                            \
    __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14));     \
    asm volatile("" ::: "memory");                                                                \
  } while (0)

// -DCL_PROBE (tools/build_probe_lib.sh builds a separate library; tools/conv_probe.py drives it): ImagenIgemmParams.dbg bits ablate
// parts of the kernel to attribute time — 1: no waits for the weight DMA, 2: no MFMAs, 4: no activation staging inside the loop, 8: no
// stores, 16: no weight DMA inside the loop, 32: no chunk barrier, 64: no B-fragment reads, 128: no A-fragment reads, 256: no chunk
// rotation.  Results of ablated runs are numerically meaningless; the product library compiles all of it out.
template <int MI, int NI, int WM, int WN, int G, int TAPS, int RW, bool GEN>
__global__ __launch_bounds__(256, (MI * NI <= 2 ? 3 : 2)) void conv_lds_kernel(const ImagenIgemmParams p, const int RP) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(TAPS == 1 || TAPS == 9, "1x1 or 3x3");
  constexpr int BN = 32 * NI * WN;
  constexpr int TP = 32 * MI * WM;
  constexpr int KC = 8 * G;
  constexpr int PS = G * 16 + 16;          // LDS bytes per staged pixel
  constexpr int LOG2G = (G == 4) ? 2 : (G == 8) ? 3 : 4;
  constexpr int KW = TAPS == 9 ? 3 : 1;
  constexpr int KSTEPS = G / 2;            // K=16 MFMA steps per stage
  constexpr int SLOTW = NI * G * 512;      // bytes of one wave-private weight stage slot: [NI][G groups][32 couts][8 halves]
  constexpr int kItems = cl_stage_slots(TP, G, TAPS);
  constexpr int PXW = 32 * MI;
  constexpr int KD = NI * KSTEPS;          // weight DMA instructions per wave and stage (1 KiB each: two 8-channel groups x 32 couts)

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  // ---- this workgroup's tile (XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs; each XCD gets a contiguous tile range)
  const int tilesX = (p.OW + p.TW - 1) / p.TW;
  const int tilesY = (p.OH + p.TH - 1) / p.TH;
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
    tc.oy0 = ty * p.TH;
    tc.ox0 = tx * p.TW;
    tc.n0 = nt * BN;
  }

  const int ITW = p.TW - 1 + KW, ITH = p.TH - 1 + KW;
  const int abuf_bytes = ITH * RP;
  const int NC = p.Cin_pad / KC;
  char* const abuf0 = smem;
  char* const ring = smem + 2 * abuf_bytes + wave * (RW * SLOTW);      // this wave's private weight ring
  float* const ep_red = reinterpret_cast<float*>(smem + 2 * abuf_bytes + 4 * RW * SLOTW);   // [4 waves][PXW]
  char* const lds_dummy = reinterpret_cast<char*>(ep_red + 4 * PXW);  // 16 bytes

  // ================================================================================================ weight DMA (wave-private)
  // stage s = chunk * TAPS + tap covers packed group rows [(chunk*KGP + tap*G) .. +G); lanes 0-31 fetch the even group of a pair, lanes
  // 32-63 the odd one, 32 couts each: the LDS image [group][cout][8 halves] is exactly the MFMA A-fragment order (lane * 16 bytes)
  constexpr int KGP = ((TAPS * G + 1) / 2) * 2;
  const char* const wsrc_lane = reinterpret_cast<const char*>(p.w) + ((size_t)half * p.Cout_pad + tc.n0 + wn * (NI * 32) + l31) * 16;
  const size_t wrow = (size_t)p.Cout_pad * 16;   // bytes per packed group row
  // (inline asm, not __builtin_amdgcn_global_load_lds: hipcc's waitcnt pass books an LDS-DMA as a pending FLAT access and then
  // drains lgkmcnt(0) at every MFMA that consumes a ds_read — the fragment pipeline below would stall once per stage.  An asm load
  // is invisible to the pass: its completion is counted by hand (CL_WAIT_VM), and the compiler's own vmcnt waits for the ordinary
  // activation loads only become stricter, never weaker: vmcnt retires in issue order.)
  auto dma_stage = [&](int chunk, int tap, int slot) __attribute__((always_inline)) {
    const char* src = wsrc_lane + (size_t)(chunk * KGP + tap * G) * wrow;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(ring + slot * SLOTW);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < KSTEPS; ++j) {
        const char* g = src + (size_t)(2 * j) * wrow + ni * 512;
        const unsigned d = __builtin_amdgcn_readfirstlane(dst + ni * (G * 512) + j * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(d) : "memory");
      }
  };

  // ================================================================================================ activation staging
  const float inv_itw = 1.0f / (float)ITW;
  const f16* x1 = reinterpret_cast<const f16*>(p.x1);
  const f16* x2 = reinterpret_cast<const f16*>(p.x2);
  const float* dummy_f = reinterpret_cast<const float*>(p.w);
  const int ld1_s = __builtin_amdgcn_readfirstlane(p.ld1), ld2_s = __builtin_amdgcn_readfirstlane(p.ld2);
  const int my_cg = tid & (G - 1);
  const int pix0 = tid >> LOG2G;
  const int iy_first = (int)(((float)pix0 + 0.5f) * inv_itw);
  const int ix_first = pix0 - iy_first * ITW;
  const int step_y = (256 >> LOG2G) / ITW, step_x = (256 >> LOG2G) % ITW;
  const float* q1_base = p.rs ? p.rs : (p.ssq_a ? p.ssq_a : dummy_f);
  const int q1_on = (p.rs || p.ssq_a) ? 1 : 0;
  const float* q2_base = p.mu ? p.mu : ((!p.rs && p.ssq_b) ? p.ssq_b : dummy_f);
  const int q2_on = (p.mu || (!p.rs && p.ssq_b)) ? 1 : 0;
  const float* pa_base = p.pa ? p.pa : clOnes8;
  const float* ps_base = p.ps ? p.ps : clZeros8;
  const int pa_on = p.pa ? 1 : 0, ps_on = p.ps ? 1 : 0;
  const bool use_rs = p.rs != nullptr, use_ssq = !use_rs && p.ssq_a != nullptr, use_ssqb = use_ssq && p.ssq_b != nullptr;
  const bool use_mu = p.mu != nullptr, use_silu = p.act_in == IMAGEN_ACT_SILU;

  uint4 st_raw[kItems];
  float st_q1[kItems], st_q2[kItems];
  unsigned st_mask = 0;
  float4 st_a0, st_a1, st_s0, st_s1;

  const int iy0 = tc.oy0 - p.pad, ix0 = tc.ox0 - p.pad;   // stride 1
  const int sp0 = tc.b * (p.H * p.W);
  auto load_set = [&](int chunk) __attribute__((always_inline)) {
    st_mask = 0;
    const int cc = chunk * KC + my_cg * 8;
    const bool from1 = cc < p.C1;
    const bool chan_ok = from1 || (cc - p.C1 < p.C2);
    const f16* base = from1 ? x1 + (size_t)tc.b * p.bs1 + cc : x2 + (size_t)tc.b * p.bs2 + (cc - p.C1);
    const int ld = from1 ? ld1_s : ld2_s;
    int iy = iy_first, ix = ix_first;
    cl_static_for<kItems>([&](auto ic) __attribute__((always_inline)) {
      constexpr int it = decltype(ic)::value;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool ok = chan_ok && iy < ITH && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      ix += step_x;
      iy += step_y;
      if (ix >= ITW) { ix -= ITW; ++iy; }
      const int gp = ok ? gy * p.W + gx : 0;
      const f16* src = ok ? base + (size_t)gp * ld : x1;
      st_raw[it] = *reinterpret_cast<const uint4*>(src);
      if (ok) st_mask |= 1u << it;
      const int sp = sp0 + gp;
      st_q1[it] = q1_base[sp * q1_on];
      st_q2[it] = q2_base[sp * q2_on];
    });
    const int o = tc.b * p.pstride + chunk * KC + my_cg * 8;
    const float4* qa = reinterpret_cast<const float4*>(pa_base + o * pa_on);
    const float4* qs = reinterpret_cast<const float4*>(ps_base + o * ps_on);
    st_a0 = qa[0];
    st_a1 = qa[1];
    st_s0 = qs[0];
    st_s1 = qs[1];
  };
  // transform + LDS write, without control flow (mode choices are uniform selects; with no prologue configured the arithmetic
  // degenerates to (x - 0) * 1 * 1 + 0, exact in fp16 -> fp32 -> fp16)
  auto write_set = [&](char* buf) __attribute__((always_inline)) {
    const float a[8] = {st_a0.x, st_a0.y, st_a0.z, st_a0.w, st_a1.x, st_a1.y, st_a1.z, st_a1.w};
    const float s[8] = {st_s0.x, st_s0.y, st_s0.z, st_s0.w, st_s1.x, st_s1.y, st_s1.z, st_s1.w};
    int iy = iy_first, ix = ix_first;
    cl_static_for<kItems>([&](auto ic) __attribute__((always_inline)) {
      constexpr int it = decltype(ic)::value;
      char* dst = iy < ITH ? buf + iy * RP + ix * PS + my_cg * 16 : lds_dummy;
      ix += step_x;
      iy += step_y;
      if (ix >= ITW) { ix -= ITW; ++iy; }
      const bool ok = (st_mask & (1u << it)) != 0;
      const f16x8 in = *reinterpret_cast<const f16x8*>(&st_raw[it]);
      const float q = st_q1[it] + (use_ssqb ? p.ssq_wb * st_q2[it] : 0.0f);
      const float rq = __builtin_amdgcn_rsqf(fmaxf(q, 1e-24f));
      const float rs = use_rs ? st_q1[it] : (use_ssq ? rq : 1.0f);
      const float mu = use_mu ? st_q2[it] : 0.0f;
      float v[8], e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ((float)in[j] - mu) * rs * a[j] + s[j];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(-1.4426950408889634f * v[j]);
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_rcpf(1.0f + e[j]);
      f16x8 out;
#pragma unroll
      for (int j = 0; j < 8; ++j) out[j] = (f16)(use_silu ? v[j] * e[j] : v[j]);
      uint4 ow = *reinterpret_cast<const uint4*>(&out);
      ow = ok ? ow : make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(dst) = ow;
    });
  };

  // ================================================================================================ MFMA side
  int a_base[MI];
  int pix_y[MI], pix_x[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int tp = (wm * MI + mi) * 32 + l31;
    const int py = tp / p.TW, px = tp - py * p.TW;
    pix_y[mi] = py;
    pix_x[mi] = px;
    a_base[mi] = py * RP + px * PS + half * 16;
  }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.0f;

  // fragments of one K=16 step; two sets alternate (the reads of step k+1 are issued before the MFMAs of step k)
  struct Frags { f16x8 a[NI], b[MI]; };
  Frags F0, F1;
  auto read_frags = [&](Frags& F, const char* abuf, int tap_off, const char* wslot, int ks) __attribute__((always_inline)) {
    if (!CL_DBG(128)) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) F.a[ni] = *reinterpret_cast<const f16x8*>(wslot + ni * (G * 512) + ks * 1024 + lane * 16);
    }
    if (!CL_DBG(64)) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) F.b[mi] = *reinterpret_cast<const f16x8*>(abuf + a_base[mi] + tap_off + ks * 32);
    }
  };
  auto mfma_step = [&](const Frags& F) __attribute__((always_inline)) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[ni], F.b[mi], acc[ni][mi], 0, 0, 0);
  };

  // ================================================================================================ pipeline
  // (Starting every tile at its own chunk, so that the CUs of an XCD do not pull the same weights through the same L2 channels
  // at the same time, was measured and changes nothing: tools/conv_probe.py, profiles/r02_conv_probe_b.txt.)
  auto rot = [&](int c) __attribute__((always_inline)) -> int { return c; };
  // prologue: this wave's weight stages 0..RW-2 in flight, chunk 0 staged, fragments of the first K step read
#pragma unroll
  for (int j = 0; j < RW - 1; ++j) {
    const int cj = j / TAPS;
    dma_stage(cj < NC ? rot(cj) : NC + (cj - NC), j % TAPS, j);   // (past the end: the zero tail behind the last chunk)
  }
  load_set(rot(0));
  write_set(abuf0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  read_frags(F0, abuf0, 0, ring, 0);

  constexpr int L = kItems * 3 + 4;   // ordinary loads of one load_set (activations, two statistics, affine)
  constexpr int WRITE_TAP = TAPS == 9 ? 5 : 0;
  static_assert((TAPS * KSTEPS) % 2 == 0, "fragment set parity must repeat per chunk");
  static_assert(TAPS == 1 || RW - 1 <= TAPS, "the weight look-ahead may cross one chunk boundary only");
  int s = 0;   // global stage index of tap 0 of the current chunk
  for (int c = 0; c < NC; ++c, s += TAPS) {
    const char* abuf = abuf0 + (c & 1) * abuf_bytes;
    char* abuf_next = abuf0 + ((c + 1) & 1) * abuf_bytes;
    const int c_next = c + 1 < NC ? rot(c + 1) : rot(c);
    cl_static_for<TAPS * KSTEPS>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      constexpr int t = k / KSTEPS, ks = k % KSTEPS;
      Frags& cur = (k & 1) ? F1 : F0;
      Frags& nxt = (k & 1) ? F0 : F1;
      if constexpr (ks == 0) {
        // 3x3: transform + write the next chunk's halo tile BEFORE this stage's DMA is issued — the compiler's wait for the
        // activation loads is a full vmcnt(0) (it cannot see the asm DMAs), and the youngest DMA is then a whole stage old
        if constexpr (TAPS == 9 && t == WRITE_TAP) {
          if (!CL_DBG(4)) write_set(abuf_next);
        }
        // refill the slot of the stage that just finished with stage s + t + RW - 1 (past the end: the packed buffer's zero
        // tail, into a slot this wave never reads again)
        int cn = c, tn = t + RW - 1;
        if (TAPS == 1) { cn = c + RW - 1; tn = 0; }
        else if (tn >= TAPS) { tn -= TAPS; ++cn; }
        if (!CL_DBG(16)) dma_stage(cn < NC ? rot(cn) : NC + (cn - NC), tn, (s + t + RW - 1) % RW);
        if constexpr (t == 0) {
          if (!CL_DBG(4)) load_set(c_next);
        }
      }
      if constexpr (TAPS == 1 && ks == KSTEPS - 1) {
        if (!CL_DBG(4)) write_set(abuf_next);
      }
      // ---- fragments of the next K step
      if constexpr (ks + 1 < KSTEPS) {
        read_frags(nxt, abuf, (t / KW) * RP + (t % KW) * PS, ring + ((s + t) % RW) * SLOTW, ks + 1);
      } else {
        // first step of the next stage: its weight slot must have landed (this wave's own DMA: no barrier); everything issued
        // after that DMA may stay in flight
        if (!CL_DBG(1 | 16)) {
          if (TAPS == 1) CL_WAIT_VM(KD);                                    // (load_set's loads were consumed by write_set above)
          else if (t <= RW - 2 && t < WRITE_TAP && !CL_DBG(4)) CL_WAIT_VM((RW - 2) * KD + L);   // load_set's loads were issued after that DMA
          else CL_WAIT_VM((RW - 2) * KD);
        }
        if constexpr (t + 1 < TAPS) {
          read_frags(nxt, abuf, ((t + 1) / KW) * RP + ((t + 1) % KW) * PS, ring + ((s + t + 1) % RW) * SLOTW, 0);
        } else {
          // chunk boundary: every wave has written its part of the next activation buffer and finished reading this one
          if (!CL_DBG(32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          read_frags(nxt, abuf_next, 0, ring + ((s + TAPS) % RW) * SLOTW, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!CL_DBG(2)) mfma_step(cur);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  CL_WAIT_VM(0);   // stray look-ahead DMAs must not outlive the workgroup's LDS allocation
  __syncthreads(); // (ep_red below is disjoint from the rings, but the allocation is released when the LAST wave ends)

  cl_epilogue<MI, NI, WM, WN, GEN>(p, tc, acc, pix_y, pix_x, ep_red, reinterpret_cast<float*>(smem), wm, wn, half, l31);
}

// row pitch of the staged halo tile (bytes): conflict-free B-fragment reads (see the file header)
inline int cl_row_pitch(int TW, int ITW, int PS) {
  const int nat = ITW * PS;
  if (TW % 32 == 0) return nat;                 // one tile row per 32-pixel fragment
  const int want = (TW == 16) ? 0 : 128;        // TW == 8: rows y..y+3 of a fragment shifted by 8 bank slots each
  int rp = (nat + 255) / 256 * 256 + want;
  if (rp - 256 >= nat) rp -= 256;
  return rp;
}

template <int MI, int NI, int WM, int WN, int G, int TAPS, int RW, bool GEN>
int cl_launch_gen(const ImagenIgemmParams& p, hipStream_t s) {
  constexpr int TP = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int PS = G * 16 + 16;
  constexpr int KW = TAPS == 9 ? 3 : 1;
  const int ITW = p.TW - 1 + KW, ITH = p.TH - 1 + KW;
  const int IT = ITH * ITW;
  IMAGEN_CHECK(p.TH * p.TW == TP, "conv_lds: tile %dx%d does not match cfg %d (%d pixels)", p.TH, p.TW, p.cfg, TP);
  IMAGEN_CHECK(p.TW == 8 || p.TW == 16 || p.TW % 32 == 0, "conv_lds: tile width %d (8, 16 or a multiple of 32)", p.TW);
  IMAGEN_CHECK(p.stride == 1 && p.KH == KW && p.KW == KW && p.pad == (KW - 1) / 2, "conv_lds: %dx%d stride %d pad %d unsupported by cfg %d",
               p.KH, p.KW, p.stride, p.pad, p.cfg);
  IMAGEN_CHECK(IT * G <= cl_stage_slots(TP, G, TAPS) * 256, "conv_lds: halo tile too large (%d px x %d groups > %d staging slots)", IT, G,
               cl_stage_slots(TP, G, TAPS));
  IMAGEN_CHECK(p.Cout_pad % BN == 0, "conv_lds: Cout_pad %d not a multiple of %d", p.Cout_pad, BN);
  IMAGEN_CHECK(p.Cin_pad % (8 * G) == 0, "conv_lds: Cin_pad %d not a multiple of %d", p.Cin_pad, 8 * G);
  IMAGEN_CHECK(p.C1 % 8 == 0 && p.C2 % 8 == 0 && p.ld1 % 8 == 0 && (p.x2 == nullptr || p.ld2 % 8 == 0),
               "conv_lds: channel counts / strides must be multiples of 8 (C1=%d C2=%d ld1=%d ld2=%d)", p.C1, p.C2, p.ld1, p.ld2);
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NCHW_F32 || p.Cout % 4 == 0, "conv_lds: Cout %d must be a multiple of 4", p.Cout);
  IMAGEN_CHECK(p.out_mode != IMAGEN_OUT_PIXEL_SHUFFLE || p.Cout % 16 == 0, "conv_lds: pixel-shuffle needs Cout %% 16 == 0");
  IMAGEN_CHECK(!p.post_pa || (p.post_ps && p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN && !p.addend && !p.res && !p.ssq_out &&
                              p.act_out == IMAGEN_ACT_NONE && p.Cout % 4 == 0),
               "conv_lds: post_pa needs post_ps, a plain NHWC output and one workgroup covering all %d output channels (tile has %d)", p.Cout, BN);
  IMAGEN_CHECK(!(p.addend && p.res), "conv_lds: addend and residual are mutually exclusive");
  IMAGEN_CHECK(!p.gca_part || (p.gca_wk && !GEN && !p.post_pa && p.Cout <= BN), "conv_lds: gca_part needs gca_wk, a plain NHWC output and one tile covering all %d couts", p.Cout);
  IMAGEN_CHECK(!p.ssq_out || (p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN),
               "conv_lds: ssq_out needs NHWC output and one workgroup covering all %d output channels (tile has %d)", p.Cout, BN);
  const int RP = cl_row_pitch(p.TW, ITW, PS);
  const size_t lds = (size_t)2 * ITH * RP + (size_t)4 * RW * (NI * G * 512) + (size_t)(4 * 32 * MI) * sizeof(float) + 16;
  IMAGEN_CHECK(lds <= 160 * 1024, "conv_lds: LDS tile %zu bytes too large", lds);
  auto kern = conv_lds_kernel<MI, NI, WM, WN, G, TAPS, RW, GEN>;
  static bool attr_done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("conv_lds: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev] = true;
  }
  const int tilesX = (p.OW + p.TW - 1) / p.TW, tilesY = (p.OH + p.TH - 1) / p.TH;
  const int total = p.B * tilesX * tilesY * ((p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(total), dim3(256), lds, s, p, RP);
  return imagen_hip_status("conv_lds launch");
}

template <int MI, int NI, int WM, int WN, int G, int TAPS, int RW>
int cl_launch(const ImagenIgemmParams& p, hipStream_t s) {
  const bool plain = p.act_out == IMAGEN_ACT_NONE && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res;
  return plain ? cl_launch_gen<MI, NI, WM, WN, G, TAPS, RW, false>(p, s) : cl_launch_gen<MI, NI, WM, WN, G, TAPS, RW, true>(p, s);
}

struct ClCfg { int MI, NI, WM, WN, G, RW3; };   // RW3: weight ring depth (stages) of the 3x3 instantiation; 1x1: 3
constexpr ClCfg kClCfgs[] = {
    {4, 1, 1, 4, 4, 4},   // 16: 128 px x 128 co   (C_out >= 128)
    {2, 1, 1, 4, 4, 4},   // 17:  64 px x 128 co   (small maps)
    {2, 2, 1, 4, 4, 4},   // 18:  64 px x 256 co   (C_out = 256 on 32^2 maps: the activation tile is staged once for all couts)
    {4, 1, 2, 2, 4, 4},   // 19: 256 px x  64 co   (C_out = 64, big maps)
    {2, 1, 2, 2, 4, 4},   // 20: 128 px x  64 co
    {1, 1, 2, 2, 4, 4},   // 21:  64 px x  64 co
    {2, 1, 4, 1, 4, 4},   // 22: 256 px x  32 co   (C_out = 32)
    {1, 1, 4, 1, 4, 4},   // 23: 128 px x  32 co
    {4, 1, 1, 4, 8, 0},   // 24: 128 px x 128 co, 64-channel chunks (1x1 only)
    {2, 1, 1, 4, 8, 0},   // 25:  64 px x 128 co, 64-channel chunks (1x1 only)
    {4, 1, 2, 2, 8, 0},   // 26: 256 px x  64 co, 64-channel chunks (1x1 only)
    {2, 1, 4, 1, 8, 0},   // 27: 256 px x  32 co, 64-channel chunks (1x1 only)
    {4, 1, 1, 4, 4, 8},   // 28: 128 px x 128 co, weight ring 8 stages deep (one workgroup per CU: the 32^2 maps have no more anyway)
    {2, 1, 1, 4, 4, 8},   // 29:  64 px x 128 co, weight ring 8 stages deep
};
constexpr int kNumClCfgs = sizeof(kClCfgs) / sizeof(kClCfgs[0]);

template <int MI, int NI, int WM, int WN, int G, int RW3 = 4>
int cl_launch_taps(const ImagenIgemmParams& p, hipStream_t s) {
  if (p.KH == 1 && p.KW == 1) return cl_launch<MI, NI, WM, WN, G, 1, 3>(p, s);
  if constexpr (G == 4) {
    if (p.KH == 3 && p.KW == 3) return cl_launch<MI, NI, WM, WN, G, 9, RW3>(p, s);
  }
  imagen_set_error("conv_lds: cfg %d supports 1x1%s kernels only (got %dx%d)", p.cfg, G == 4 ? " / 3x3" : "", p.KH, p.KW);
  return -1;
}

}  // namespace

int imagen_conv_lds_num_configs() { return kNumClCfgs; }

int imagen_conv_lds_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (idx < 0 || idx >= kNumClCfgs) return -1;
  const ClCfg& c = kClCfgs[idx];
  if (tile_pixels) *tile_pixels = 32 * c.MI * c.WM;
  if (tile_cout) *tile_cout = 32 * c.NI * c.WN;
  if (kgroups) *kgroups = c.G;
  return 0;
}

int imagen_conv_lds_stage_slots(int idx, int KH, int KW) {
  if (idx < 0 || idx >= kNumClCfgs) return -1;
  const ClCfg& c = kClCfgs[idx];
  if (KH == 3 && KW == 3) return c.G == 4 ? cl_stage_slots(32 * c.MI * c.WM, c.G, 9) : 0;
  if (KH == 1 && KW == 1) return cl_stage_slots(32 * c.MI * c.WM, c.G, 1);
  return 0;
}

long imagen_conv_lds_lds_bytes(int idx, int KH, int KW, int TH, int TW) {
  if (idx < 0 || idx >= kNumClCfgs || TH < 1 || TW < 1) return -1;
  const ClCfg& c = kClCfgs[idx];
  const int slots = imagen_conv_lds_stage_slots(idx, KH, KW);
  if (slots <= 0 || TH * TW != 32 * c.MI * c.WM) return -1;
  if (!(TW == 8 || TW == 16 || TW % 32 == 0)) return -1;
  const int ITW = TW - 1 + KW, ITH = TH - 1 + KH;
  if (ITW * ITH * c.G > slots * 256) return -1;
  const int PS = c.G * 16 + 16;
  const long lds = 2L * ITH * cl_row_pitch(TW, ITW, PS) + 4L * (KH * KW == 9 ? c.RW3 : 3) * (c.NI * c.G * 512) + 4L * 32 * c.MI * 4 + 16;
  return lds <= 160 * 1024 ? lds : -1;
}

int launch_conv_lds(const ImagenIgemmParams* pp, int idx, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  switch (idx) {
    case 0: return cl_launch_taps<4, 1, 1, 4, 4>(p, s);
    case 1: return cl_launch_taps<2, 1, 1, 4, 4>(p, s);
    case 2: return cl_launch_taps<2, 2, 1, 4, 4>(p, s);
    case 3: return cl_launch_taps<4, 1, 2, 2, 4>(p, s);
    case 4: return cl_launch_taps<2, 1, 2, 2, 4>(p, s);
    case 5: return cl_launch_taps<1, 1, 2, 2, 4>(p, s);
    case 6: return cl_launch_taps<2, 1, 4, 1, 4>(p, s);
    case 7: return cl_launch_taps<1, 1, 4, 1, 4>(p, s);
    case 8: return cl_launch_taps<4, 1, 1, 4, 8>(p, s);
    case 9: return cl_launch_taps<2, 1, 1, 4, 8>(p, s);
    case 10: return cl_launch_taps<4, 1, 2, 2, 8>(p, s);
    case 11: return cl_launch_taps<2, 1, 4, 1, 8>(p, s);
    case 12: return cl_launch_taps<4, 1, 1, 4, 4, 8>(p, s);
    case 13: return cl_launch_taps<2, 1, 1, 4, 4, 8>(p, s);
  }
  imagen_set_error("conv_lds: bad cfg index %d", idx);
  return -1;
}
