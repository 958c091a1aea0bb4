// conv_small.hip — the 3x3 convolutions of the SMALL maps (ninth igemm family, gfx950): the 8^2 / 16^2 (/ 32^2) levels, 64 - 1024 input
// channels, i.e. a few thousand output pixels against a reduction of K = 576 .. 9216.
//
// Why it exists (round 5, profiles/r05_graph_profile.txt): on the wave-specialised kernel (and on the all-DMA family's 64-pixel tiles)
// these launches take 11 - 19 us for 0.3 - 1.8 GFLOP — [128 -> 128 @8^2 x 16 images] is 16 workgroups walking 36 barrier-synchronised K
// chunks one after the other, [384 -> 256 @8^2] 64 workgroups walking 108.  The time is the LENGTH of the serial K loop, not its work, and
// every tile shape of those kernels that adds workgroups shortens nothing.  Splitting K over workgroups would add a seam (partials through
// HBM + a second pass: MI355X_MICROARCH.md prices it at 5 - 13 us, the whole launch).  This kernel splits K INSIDE the workgroup:
//   * one workgroup of 8 waves per (32 output pixels of one image) x (32 NT output channels) tile, NT = 1 | 2 | 4;  wave w owns cout
//     fragment w % NT and the K slice w / NT of 8 / NT slices — each wave runs a private, barrier-free K loop of 1/8 .. 1/2 of the
//     reduction on ONE 32 x 32 accumulator fragment (v_mfma_f32_32x32x16_f16), then the 8 / NT partial fragments of a cout fragment are
//     summed through LDS (4 KB each);
//   * A operand (weights): 16-byte loads straight from the packed buffer (imagen_pack_igemm_weights order: a lane's fragment is
//     contiguous, a K = 16 step is two consecutive group rows) into a register ring 8 steps deep, scalar base + one per-lane offset
//     (rowchain.hip's weight stream): every workgroup streams its 32 NT x K slice out of L2, 18 - 221 KB;
//   * B operand (pixels): the (TH + 2) x (TW + 2) halo tile of the image is staged ONCE into LDS as [position][Cin] fp16, with the Block
//     prologue (ChanRMSNorm statistics from the producers' sums of squares | explicit mu / rs, per-channel affine, SiLU) applied on the
//     way and zeros outside the image, one barrier; a tap is then an address offset.  Pitch 2 Cin + 16 bytes (16 x odd), and the lane ->
//     pixel assignment follows the hardware's ds_read_b128 service groups ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} per half-wave:
//     MI355X_MICROARCH.md, LDS) so that each group's 16 positions are distinct mod 16 for every tap: conflict-free fragment reads;
//   * epilogue: the waves of K slice 0 own the summed fragments and run conv_epilogue.h as a (1 x NT)-wave workgroup — every epilogue of
//     the contract, the all-cout ones (ssq_out / post_pa / GlobalContext partials) where 32 NT covers Cout; the other waves have left
//     (s_barrier counts live waves).
// Measured (round 5, calls I - L, tools/small_bench.py: cold operands, rocprofv3 durations + an s_memtime phase timeline): a launch is two cold
// round trips (kernel arguments, then the halo tile: ~2300 cycles each behind a flushed L2 / TLB) + the K loop + ~4k cycles of K-split sum and
// epilogue.  The K loop runs at ~285 cycles per K = 16 step per wave whatever its instruction count: 8 KB in flight per wave against that
// latency.  Touching the rest of the slab into L2 first (one dword per line, direct-to-LDS sink) halved the K loop and cost the staging phase
// the same 5k cycles — the address path takes a line per clock, a touch is as expensive there as the load it prepares — so it is not kept.
// [384 -> 256 @8^2 x 16]: 20.0 -> 12.8 us, [256 -> 256 @8^2]: 15.6 -> 12.0, [192 -> 128 @16^2]: 15.1 -> 13.5; the 32^2 maps and the 512- / 1024-
// channel layers of C2 lose (32-pixel tiles stream every weight byte through each pixel tile's CU: 75 - 94 us against 33 - 48), the planner keeps them away.
// Contract: ImagenIgemmParams with KH = KW = 3, stride 1, pad 1, G = 4 packing — or KH = KW = 1, pad 0, any G >= 2 (one tap, no halo: the 1x1
// res_conv / upsample GEMMs of the same maps; the packed rows of a 1x1 layer are consecutive 8-channel groups for every G); C1 % 8 == 0, C1 + C2 == Cin_pad (a multiple of 32);
// output tiles of 32 pixels as 4 x 8, 2 x 16 or 1 x 32 (partial tiles at the map's edge are masked).
#include <cstdio>
#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int CS_THREADS = 512;
#ifndef CS_RING
#define CS_RING 8                   // weight fragments in flight per wave (call P: 16, with one workgroup per CU for the registers, is no faster)
#endif
#ifndef CS_MINW
#define CS_MINW 4                   // minimum waves per SIMD the register allocation leaves room for (4: two workgroups per CU — one stages while the other multiplies)
#endif
#ifndef CS_PHASES
#define CS_PHASES 4                 // the pixel tiles that share a weight slab start their K walk at 0, 1/4, 1/2, 3/4 of the slice (wrapping): behind the
                                    // first quarter a wave meets lines a neighbour pulled into L2 a quarter earlier (1: every tile walks from the start)
#endif
#ifndef CS_BATCH
#define CS_BATCH 6                  // staged 16-byte pieces in flight per thread and round (8 spills under CS_MINW 4)
#endif

__host__ __device__ inline int cs_row_positions(int TW) { return TW == 8 ? 12 : TW + 2; }   // staged positions per halo row (8-wide tiles: 2 unused, so that rows 0 / 2 and 1 / 3 complement each other mod 16)
__host__ __device__ inline int cs_pitch(int Cin_pad) { return 2 * Cin_pad + 16; }
__host__ __device__ inline size_t cs_tile_bytes(int TH, int TW, int Cin_pad) { return (size_t)(TH + 2) * cs_row_positions(TW) * cs_pitch(Cin_pad); }

// -DCS_TRACE (tools/small_bench.py --trace, a throw-away variant library: never in the product build): thread 0 of every workgroup stamps
// s_memtime at the phase boundaries into a buffer handed over by imagen_debug_conv_small_trace()
#ifdef CS_TRACE
__device__ unsigned long long* g_cs_trace = nullptr;
#define CS_STAMP(i)                                                                                                        \
  do {                                                                                                                     \
    if (threadIdx.x == 0 && g_cs_trace) g_cs_trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define CS_STAMP(i) ((void)0)
#endif

template <int NT>
struct CsEp {   // conv_epilogue.h scratch of a (1 x NT)-wave workgroup, floats
  static constexpr int BN = 32 * NT;
  static constexpr int PAR = 4 * BN + NT * 32 + 8 + (BN + 4);
  static constexpr int RED = NT * 32;
};

template <int NT>
__host__ __device__ inline size_t cs_lds_bytes(int TH, int TW, int Cin_pad) {
  size_t body = cs_tile_bytes(TH, TW, Cin_pad);
  const size_t partials = (size_t)(8 / NT - 1) * NT * 4096;   // the K-split partial fragments alias the (dead) halo tile
  if (partials > body) body = partials;
  return ((body + 15) & ~(size_t)15) + (size_t)(CsEp<NT>::PAR + CsEp<NT>::RED) * sizeof(float);
}

// NT: 32-cout fragments per workgroup tile.  PRO: the input-side prologue (statistics / affine / SiLU).  GEN: generic epilogue (conv_epilogue.h).
template <int NT, bool PRO, bool GEN>
__global__ __launch_bounds__(CS_THREADS, CS_MINW) void conv_small_kernel(const ImagenIgemmParams p, unsigned code_bytes) {
  constexpr int KS = 8 / NT;        // K slices
  constexpr int BN = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CS_STAMP(0);
  const unsigned warm = imagen_code_warm(code_bytes, threadIdx.x, CS_THREADS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wn = wave % NT, ks = wave / NT;

  const int TH = p.TH, TW = p.TW;
  const int P = cs_row_positions(TW), pitch = cs_pitch(p.Cin_pad);
  const int pad = p.pad, ntaps = p.KH * p.KW;          // 3x3 pad 1 | 1x1 pad 0 (the same kernel: one tap, no halo)
  const int HW_ = TW + 2 * pad, HH = TH + 2 * pad;     // halo tile (used positions)

  // ---- tile of this workgroup: contiguous ranges of the tile list per XCD (blockIdx goes round-robin over the 8 XCDs, each with its own L2).
  //      The list is ordered so that a range shares what is larger: cout slab fastest (the slabs of a pixel tile stage the same halo tile,
  //      every XCD streams all weights) where the map outweighs the weights, pixel tile fastest (an XCD streams ONE slab's weights and
  //      stages the whole map) where the weights outweigh the map — [384 -> 256 @8^2 x 16]: 1.8 MB of weights against 0.8 MB of pixels
  const int tilesX = (p.OW + TW - 1) / TW, tilesY = (p.OH + TH - 1) / TH, tilesN = (p.Cout + BN - 1) / BN;
  ClTile tc;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int npix = p.B * tilesY * tilesX;
    const bool slab_major = p.Cout * ntaps > p.B * p.OH * p.OW;   // weight bytes (2 taps Cin Cout) > map bytes (2 Cin B OH OW)
    int nt;
    if (slab_major) { nt = t / npix; t -= nt * npix; }
    else { nt = t % tilesN; t /= tilesN; }
    const int tx = t % tilesX;
    t /= tilesX;
    const int ty = t % tilesY;
    tc.b = t / tilesY;
    tc.oy0 = ty * TH;
    tc.ox0 = tx * TW;
    tc.n0 = nt * BN;
  }

  // ---- the epilogue's per-channel operands (bias, post_pa / post_ps, gca_wk): requested first, parked in LDS before the staging barrier
  //      (conv_epilogue.h then runs PRELOADED: no dependent global round trip at the end of the kernel)
  float* const ep_par = reinterpret_cast<float*>(smem + cs_lds_bytes<NT>(TH, TW, p.Cin_pad) - (size_t)(CsEp<NT>::PAR + CsEp<NT>::RED) * sizeof(float));
  float* const ep_red = ep_par + CsEp<NT>::PAR;
  // (every load of the staging phase is unconditional and global: absent operands read the head of the weight buffer and are replaced by their
  // neutral constant afterwards — a conditional or generic-address load makes the compiler's vmcnt bookkeeping fall back to full drains,
  // which would wait for the weight ring, the coldest request of the phase)
  const float* const fw = reinterpret_cast<const float*>(p.w);
  float epv[4];
  {
    const int co = min(tc.n0 + tid, p.Cout_pad - 1);   // (the bias is padded to Cout_pad by the host; post_pa / post_ps / gca_wk are not)
    const int cc = min(co, p.Cout - 1);
    epv[0] = (p.bias ? p.bias : fw)[p.bias ? co : 0];
    epv[1] = (p.post_pa ? p.post_pa + (size_t)tc.b * p.post_pstride : fw)[p.post_pa ? cc : 0];
    epv[2] = (p.post_pa ? p.post_ps + (size_t)tc.b * p.post_pstride : fw)[p.post_pa ? cc : 0];
    epv[3] = (p.gca_part ? p.gca_wk : fw)[p.gca_part ? cc : 0];
    if (!p.bias) epv[0] = 0.0f;
    if (co >= p.Cout) epv[1] = epv[2] = epv[3] = 0.0f;
  }

  // ---- the generic epilogue's per-pixel operands (gate * addend | residual: 64 bytes per pixel and cout fragment) are touched NOW by the lanes that
  //      will read them, one dword each, so that the reads at the end of the kernel are L2 hits instead of a third cold round trip
  unsigned ep_touch = 0;
  if constexpr (GEN) {
    const f16* eop = p.addend ? reinterpret_cast<const f16*>(p.addend) + (size_t)tc.b * p.bs_add : (p.res ? reinterpret_cast<const f16*>(p.res) + (size_t)tc.b * p.bs_res : nullptr);
    if (eop && ks == 0) {
      const int eld = p.addend ? p.ld_add : p.ld_res;
      const int l = lane & 31;                             // (any pixel of the tile per lane: the whole tile x this wave's cout fragment gets covered)
      const int ty = l / TW, tx = l - ty * TW;
      const int oy = min(tc.oy0 + ty, p.OH - 1), ox = min(tc.ox0 + tx, p.OW - 1);
      ep_touch = *reinterpret_cast<const unsigned*>(eop + (size_t)(oy * p.OW + ox) * eld + min(tc.n0 + wn * 32 + 16 * half, p.Cout - 2));
    }
  }

  // ---- stage the halo tile.  A thread owns ONE 8-channel group (its affine stays in registers, its address arithmetic is a multiply) and
  //      walks the positions slot, slot + nslots, ...: 512 / (Cin / 8) positions per round of the workgroup, CS_BATCH rounds in flight
  const int ppr = p.Cin_pad >> 3;                       // 8-channel groups per position
  const int npos = HH * HW_;
  const float inv_w = 1.0f / (float)HW_;
  const int nslots = CS_THREADS / ppr;
  const int slot = (int)(((float)tid + 0.5f) * (1.0f / (float)ppr));
  const int cg = tid - slot * ppr;
  const bool active = slot < nslots;
  const int c = cg * 8;
  const bool from1 = c < p.C1;
  const f16* bp = from1 ? reinterpret_cast<const f16*>(p.x1) + (size_t)tc.b * p.bs1 + c : reinterpret_cast<const f16*>(p.x2) + (size_t)tc.b * p.bs2 + (c - p.C1);
  const int ld = from1 ? p.ld1 : p.ld2;
  float a[8], sh[8];
  if constexpr (PRO) {
    const float4* qa = reinterpret_cast<const float4*>(p.pa ? p.pa + (size_t)tc.b * p.pstride + c : fw);
    const float4* qs = reinterpret_cast<const float4*>(p.ps ? p.ps + (size_t)tc.b * p.pstride + c : fw);
    const float4 a0 = qa[0], a1 = qa[1], b0 = qs[0], b1 = qs[1];
    a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
    sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // absent factors: neutral constants
      a[j] = p.pa ? a[j] : 1.0f;
      sh[j] = p.ps ? sh[j] : 0.0f;
    }
  }
  const bool use_rs = p.rs != nullptr, use_ssq = !use_rs && p.ssq_a != nullptr, use_ssqb = use_ssq && p.ssq_b != nullptr;
  const bool use_mu = p.mu != nullptr, use_silu = p.act_in == IMAGEN_ACT_SILU;
  const float* q1_base = use_rs ? p.rs : (use_ssq ? p.ssq_a : fw);          // rs | ssq_a | (loaded, unused)
  const float* q2_base = use_mu ? p.mu : (use_ssqb ? p.ssq_b : fw);        // mu | ssq_b | (loaded, unused)
  const int q1_on = (use_rs || use_ssq) ? 1 : 0, q2_on = (use_mu || use_ssqb) ? 1 : 0;
  const int sp0 = tc.b * (p.H * p.W);

  // the weight stream of this wave: cout fragment tc.n0 / 32 + wn, the (channel chunk, tap) units [u0, u1) of NU = 9 per 32-channel chunk —
  // a unit is two K = 16 steps (channels 0-15 and 16-31 of the chunk at one tap), four consecutive group rows of the packed buffer
  const int NU = (p.Cin_pad >> 5) * ntaps;
  const int u0 = __builtin_amdgcn_readfirstlane((NU * ks) / KS), u1 = __builtin_amdgcn_readfirstlane((NU * (ks + 1)) / KS);
  const int nun = u1 - u0;
  // the walk of this wave: units u0 + (i + rot) mod nun, i = 0 .. nun - 1.  All pixel tiles of an image batch read the same slab, in lockstep when
  // they all start at its head: every ring refill is then a cold miss for every one of them (call J: ~2300 cycles against ~500 out of L2)
  const int rot = __builtin_amdgcn_readfirstlane(nun >= 2 * CS_PHASES ? (((tc.b * tilesY + tc.oy0 / TH) * tilesX + tc.ox0 / TW) % CS_PHASES) * nun / CS_PHASES : 0);
  auto walk = [&](int i) __attribute__((always_inline)) -> int {   // i < nun
    const int x = i + rot;
    return u0 + (x >= nun ? x - nun : x);
  };
  const char* wbase = reinterpret_cast<const char*>(p.w) + ((size_t)(tc.n0 >> 5) + wn) * 512;
  const unsigned w_lane = ((unsigned)half * (unsigned)p.Cout_pad + (unsigned)l31) * 16u;
  const size_t w_step = (size_t)p.Cout_pad * 32;       // two group rows per K = 16 step
  auto weight_frag = [&](int s) __attribute__((always_inline)) -> f16x8 {
    return *reinterpret_cast<const f16x8*>(wbase + (size_t)s * w_step + w_lane);
  };
  f16x8 ring[CS_RING];

  bool ring_filled = false;
  for (int base = 0; base < npos; base += CS_BATCH * nslots) {
    uint4 raw[CS_BATCH];
    float q1[CS_BATCH], q2[CS_BATCH];
    int dst[CS_BATCH];
    unsigned okmask = 0;
#pragma unroll
    for (int k = 0; k < CS_BATCH; ++k) {
      const int pos = base + slot + k * nslots;
      const bool in = active & (pos < npos);
      const int pp = in ? pos : 0;
      const int hy = (int)(((float)pp + 0.5f) * inv_w);
      const int hx = pp - hy * HW_;
      const int gy = tc.oy0 - pad + hy, gx = tc.ox0 - pad + hx;
      // (bitwise, not short-circuit: one straight-line address computation per piece; Cin_pad == C1 + C2, so every staged channel exists)
      const bool ok = in & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
      const int gp = ok ? gy * p.W + gx : 0;
      raw[k] = *reinterpret_cast<const uint4*>(bp + gp * ld);
      if constexpr (PRO) {
        q1[k] = q1_base[(sp0 + gp) * q1_on];
        q2[k] = q2_base[(sp0 + gp) * q2_on];
      }
      okmask |= (ok ? 1u : 0u) << k;
      dst[k] = in ? (hy * P + hx) * pitch + cg * 16 : -1;
    }
    if (!ring_filled) {   // the first ring fill, behind the first round's requests (loads return in order: in front, it would hold them back — call L)
      ring_filled = true;
#pragma unroll
      for (int i = 0; i < CS_RING / 2; ++i) {
        const int u = walk(i < nun ? i : (nun > 0 ? nun - 1 : 0));   // (a 1x1 layer with fewer chunks than K slices leaves waves without a unit: they request unit u0 — in range — and multiply nothing)
        ring[2 * i] = weight_frag(2 * u);
        ring[2 * i + 1] = weight_frag(2 * u + 1);
      }
      imagen_code_warm_sink(warm);
      IMAGEN_SINK(ep_touch);
      if (tid < BN) {   // (the next-oldest requests, exact count)
        ep_par[tid] = epv[0];
        ep_par[BN + tid] = epv[1];
        ep_par[2 * BN + tid] = epv[2];
        ep_par[3 * BN + tid] = epv[3];
      }
    }
#pragma unroll
    for (int k = 0; k < CS_BATCH; ++k) {
      uint4 ow = raw[k];
      if constexpr (PRO) {
        const f16x8 in = __builtin_bit_cast(f16x8, raw[k]);
        // ChanRMSNorm statistics straight from the producers' per-pixel sums of squares: 1 / max(sqrt(q), 1e-12) (ip.py:328)
        const float q = q1[k] + (use_ssqb ? p.ssq_wb * q2[k] : 0.0f);
        const float rs = use_rs ? q1[k] : (use_ssq ? __builtin_amdgcn_rsqf(fmaxf(q, 1e-24f)) : 1.0f);
        const float mu = use_mu ? q2[k] : 0.0f;
        float v[8], e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((float)in[j] - mu) * rs * a[j] + sh[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(-1.4426950408889634f * v[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_rcpf(1.0f + e[j]);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)(use_silu ? v[j] * e[j] : v[j]);
        ow = __builtin_bit_cast(uint4, o);
      }
      if (!(okmask & (1u << k))) ow = make_uint4(0, 0, 0, 0);   // outside the image: zero padding
      if (dst[k] >= 0) *reinterpret_cast<uint4*>(smem + dst[k]) = ow;
    }
  }
  CS_STAMP(1);
  __syncthreads();
  CS_STAMP(2);

  // ---- lane -> pixel of the tile: hardware service group g (0 | 1) and rank r (0 .. 15) of the lane inside its half-wave
  int g, r;
  if (l31 < 4) { g = 0; r = l31; }
  else if (l31 < 12) { g = 1; r = l31 - 4; }
  else if (l31 < 16) { g = 0; r = l31 - 8; }
  else if (l31 < 20) { g = 1; r = l31 - 8; }
  else if (l31 < 28) { g = 0; r = l31 - 12; }
  else { g = 1; r = l31 - 16; }
  int pix_y[1], pix_x[1];
  if (TW == 8) { pix_y[0] = 2 * (r >> 3) + g; pix_x[0] = r & 7; }        // 4 x 8: rows 0, 2 | rows 1, 3
  else if (TW == 16) { pix_y[0] = g; pix_x[0] = r; }                      // 2 x 16: row 0 | row 1
  else { pix_y[0] = 0; pix_x[0] = 16 * g + r; }                           // 1 x 32: left | right half
  const char* xl = smem + (pix_y[0] * P + pix_x[0]) * pitch + half * 16;

  // ---- the K loop of this wave, a unit (two MFMAs) per turn: the unit's tap is a position offset, its chunk a channel offset; the B fragments
  //      of a unit are requested one unit ahead (their LDS latency sits behind the previous unit's MFMAs), its weights four units ahead
  f32x16 acc[1][1];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[0][0][i] = 0.f;
  int uabs = walk(0);
  int chunk = uabs / ntaps, tap = uabs - chunk * ntaps;
  auto unit_off = [&]() __attribute__((always_inline)) -> int {
    const int dy = (tap * 11) >> 5;                     // tap / 3 for tap < 9
    const int dx = tap - 3 * dy;
    const int off = (dy * P + dx) * pitch + chunk * 64;
    if (++uabs == u1) {                                 // the walk wraps to the head of the slice
      uabs = u0;
      chunk = u0 / ntaps;
      tap = u0 - chunk * ntaps;
    } else if (++tap == ntaps) {
      tap = 0;
      ++chunk;
    }
    return off;
  };
  f16x8 b0, b1;
  {
    const int off = unit_off();
    b0 = *reinterpret_cast<const f16x8*>(xl + off);
    b1 = *reinterpret_cast<const f16x8*>(xl + off + 32);
  }
  constexpr int RU = CS_RING / 2;   // units in the ring
  int ub = 0;
  for (; ub + RU < nun; ub += RU) {   // straight-line body: RU units, each followed by the request of the unit RU ahead (clamped to the last one)
#pragma unroll
    for (int i = 0; i < RU; ++i) {
      const int off = unit_off();
      const f16x8 n0 = *reinterpret_cast<const f16x8*>(xl + off), n1 = *reinterpret_cast<const f16x8*>(xl + off + 32);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[2 * i], b0, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[2 * i + 1], b1, acc[0][0], 0, 0, 0);
      const int un = ub + i + RU;
      const int u = walk(un < nun ? un : nun - 1);
      ring[2 * i] = weight_frag(2 * u);
      ring[2 * i + 1] = weight_frag(2 * u + 1);
      b0 = n0;
      b1 = n1;
      __builtin_amdgcn_sched_barrier(0);   // pins the requests here (the scheduler otherwise sinks look-ahead loads to their use)
    }
  }
#pragma unroll
  for (int i = 0; i < RU; ++i) {   // the last (up to RU) units are in the ring
    if (ub + i < nun) {
      f16x8 n0 = b0, n1 = b1;
      if (ub + i + 1 < nun) {
        const int off = unit_off();
        n0 = *reinterpret_cast<const f16x8*>(xl + off);
        n1 = *reinterpret_cast<const f16x8*>(xl + off + 32);
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[2 * i], b0, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[2 * i + 1], b1, acc[0][0], 0, 0, 0);
      b0 = n0;
      b1 = n1;
    }
  }

  CS_STAMP(3);
  // ---- K-split: the partial fragments of slices 1 .. KS - 1 are added into slice 0's through LDS (the halo tile is dead behind the barrier)
  if constexpr (KS > 1) {
    __syncthreads();
    if (ks > 0) {
      f32x4* dstp = reinterpret_cast<f32x4*>(smem + (size_t)((ks - 1) * NT + wn) * 4096);
#pragma unroll
      for (int q = 0; q < 4; ++q) dstp[q * 64 + lane] = f32x4{acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll 2   // (not all 7: 28 partial quads in registers at once are the kernel's register peak)
    for (int k = 1; k < KS; ++k) {
      const f32x4* src = reinterpret_cast<const f32x4*>(smem + (size_t)((k - 1) * NT + wn) * 4096);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = src[q * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0][0][4 * q + e] += v[e];
      }
    }
  }

  CS_STAMP(4);
  // ---- epilogue: the NT live waves as a (1 x NT)-wave workgroup of conv_epilogue.h (its scratch sits behind everything else; operands preloaded)
  cl_epilogue<1, 1, 1, NT, GEN, true>(p, tc, acc, pix_y, pix_x, ep_red, ep_par, 0, wn, half, l31);
  CS_STAMP(5);
}

template <int NT, bool PRO, bool GEN>
int cs_launch(const ImagenIgemmParams& p, hipStream_t s) {
  auto kern = conv_small_kernel<NT, PRO, GEN>;
  const size_t lds = cs_lds_bytes<NT>(p.TH, p.TW, p.Cin_pad);
  IMAGEN_CHECK(lds <= 160 * 1024, "conv_small: %zu bytes of LDS (Cin_pad %d)", lds, p.Cin_pad);
  static bool attr_done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("conv_small: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  static const unsigned code_bytes = [] {
    char name[160];
    snprintf(name, sizeof(name), "_ZN12_GLOBAL__N_117conv_small_kernelILi%dELb%dELb%dEEEv17ImagenIgemmParamsj", NT, PRO ? 1 : 0, GEN ? 1 : 0);
    return imagen_kernel_code_bytes(name);
  }();
  const int total = p.B * ((p.OH + p.TH - 1) / p.TH) * ((p.OW + p.TW - 1) / p.TW) * ((p.Cout + 32 * NT - 1) / (32 * NT));
  hipLaunchKernelGGL(kern, dim3(total), dim3(CS_THREADS), lds, s, p, code_bytes);
  return imagen_hip_status("conv_small launch");
}

template <int NT>
int cs_dispatch(const ImagenIgemmParams& p, hipStream_t s) {
  const bool pro = p.mu || p.rs || p.pa || p.ps || p.ssq_a || p.act_in != IMAGEN_ACT_NONE;
  const bool plain = p.act_out == IMAGEN_ACT_NONE && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res;
  if (pro) return plain ? cs_launch<NT, true, false>(p, s) : cs_launch<NT, true, true>(p, s);
  return plain ? cs_launch<NT, false, false>(p, s) : cs_launch<NT, false, true>(p, s);
}

constexpr int kCsNT[3] = {1, 2, 4};

}  // namespace

#ifdef CS_TRACE
extern "C" int imagen_debug_conv_small_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cs_trace), &buf, sizeof(buf)); }
#endif

int imagen_conv_small_num_configs() { return 3; }

int imagen_conv_small_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (idx < 0 || idx >= 3) return -1;
  if (tile_pixels) *tile_pixels = 32;
  if (tile_cout) *tile_cout = 32 * kCsNT[idx];
  if (kgroups) *kgroups = 4;
  return 0;
}

long imagen_conv_small_lds_bytes(int idx, int KH, int KW, int TH, int TW) {
  if (idx < 0 || idx >= 3 || KH != KW || (KH != 3 && KH != 1) || TH * TW != 32 || (TW != 8 && TW != 16 && TW != 32)) return -1;
  return (long)cs_lds_bytes<1>(TH, TW, 256);   // (a typical layer: the real figure grows with Cin and is checked at launch)
}

int launch_conv_small(const ImagenIgemmParams* pp, int idx, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  IMAGEN_CHECK(idx >= 0 && idx < 3, "conv_small: bad cfg index %d", idx);
  IMAGEN_CHECK(((p.KH == 3 && p.KW == 3 && p.pad == 1) || (p.KH == 1 && p.KW == 1 && p.pad == 0)) && p.stride == 1 && p.OH == p.H && p.OW == p.W,
               "conv_small: 3x3 pad-1 | 1x1 pad-0 stride-1 convolutions only");
  IMAGEN_CHECK(p.TH * p.TW == 32 && (p.TW == 8 || p.TW == 16 || p.TW == 32), "conv_small: 4x8 / 2x16 / 1x32 tiles (got %dx%d)", p.TH, p.TW);
  IMAGEN_CHECK(p.C1 % 8 == 0 && p.C1 > 0 && p.C2 % 8 == 0 && p.Cin_pad == p.C1 + p.C2 && p.Cin_pad % 32 == 0 && (p.C2 == 0 || p.x2),
               "conv_small: inputs in 8-channel groups, Cin_pad = C1 + C2 in 32-channel chunks (got %d + %d, padded %d)", p.C1, p.C2, p.Cin_pad);
  IMAGEN_CHECK(p.ld1 % 8 == 0 && p.bs1 % 8 == 0 && (p.C2 == 0 || (p.ld2 % 8 == 0 && p.bs2 % 8 == 0)) && ((size_t)p.x1 & 15) == 0 && ((size_t)p.x2 & 15) == 0,
               "conv_small: input rows must be 16-byte aligned");
  const int NT = kCsNT[idx];
  IMAGEN_CHECK(p.Cout_pad % (32 * NT) == 0, "conv_small: Cout_pad %d not a multiple of %d", p.Cout_pad, 32 * NT);
  IMAGEN_CHECK(p.act_in == IMAGEN_ACT_NONE || p.act_in == IMAGEN_ACT_SILU, "conv_small: input activation none | SiLU");
  IMAGEN_CHECK(!p.mu || p.rs, "conv_small: mu needs rs");
  IMAGEN_CHECK(p.pstride == 0 || p.pstride >= p.Cin_pad || (!p.pa && !p.ps), "conv_small: per-batch affine rows shorter than Cin_pad");
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NCHW_F32 || p.Cout % 4 == 0, "conv_small: Cout %d must be a multiple of 4", p.Cout);
  IMAGEN_CHECK(p.out_mode != IMAGEN_OUT_PIXEL_SHUFFLE || p.Cout % 16 == 0, "conv_small: pixel-shuffle needs Cout %% 16 == 0");
  IMAGEN_CHECK(!(p.addend && p.res), "conv_small: addend and residual are mutually exclusive");
  IMAGEN_CHECK(!p.addend || p.gate, "conv_small: addend requires gate");
  const bool full = p.ssq_out || p.post_pa || p.gca_part;
  IMAGEN_CHECK(!full || p.Cout <= 32 * NT, "conv_small: ssq_out / post_pa / gca_part need one tile over all %d couts", p.Cout);
  IMAGEN_CHECK(!p.post_pa || (p.post_ps && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res && !p.ssq_out && p.act_out == IMAGEN_ACT_NONE),
               "conv_small: post_pa needs post_ps and a plain NHWC output");
  IMAGEN_CHECK(!p.gca_part || (p.gca_wk && !p.post_pa && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res && p.act_out == IMAGEN_ACT_NONE),
               "conv_small: gca_part needs gca_wk and a plain NHWC output");
  switch (NT) {
    case 1: return cs_dispatch<1>(p, s);
    case 2: return cs_dispatch<2>(p, s);
    default: return cs_dispatch<4>(p, s);
  }
}
