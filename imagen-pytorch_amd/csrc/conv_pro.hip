// conv_pro.hip — the streaming 3x3 convolution with the Block prologue IN FLIGHT (seventh igemm family, gfx950): 32 or 64 output channels
// from one to three 32-channel input chunks of one or two tensors (x, or the up path's concat cat(x, skip * 2^-1/2)), stride 1, pad 1, NHWC
// fp16 — Block (ip.py:671-691: ChanRMSNorm -> per-channel affine -> SiLU -> Conv2d 3x3) at the 256^2 / 128^2 levels of the README
// super-resolution unet, where the input must still go through the prologue (block1 of the up-path blocks: the norm runs over the concat;
// block1 behind a downsample): 64 -> 32 and 32 -> 32 @256^2, 96 -> 64 @128^2.
//
// Why it exists (round 3, profiles/r03_graph_profile.txt): conv_stream.hip runs the 64 -> 32 launches at 0.23 of the HBM rate (111 us for
// 201 MB), the wave-specialised kernel the 96 -> 64 ones at 0.17 of the MFMA peak (69 us).  The streaming kernel's eight waves move in lock
// step — wait for the tile's direct-to-LDS copies, barrier, transform the tile IN PLACE in LDS, barrier, multiply, store — one workgroup per
// CU with two inputs, so every one of those latencies is exposed, and both MFMA operands come out of LDS.  Here
//   * the raw rows of a tile are requested into REGISTERS (plain 16-byte global loads, one per lane and slot) TWO tiles ahead — two register
//     sets take turns, and a unit's registers are refilled the moment its prologue has consumed them — so a load has two tile periods to
//     land (round 4, calls A / B: with one period the kernel ran at latency per tile — 3.5 TB/s raw — with the prologue's VALU time on top);
//   * the prologue runs on those registers — one LDS write per element, no read-modify-write of a landed tile — in six parts per unit
//     (8 channels of one halo pixel) that sit BETWEEN the K steps of the tile being multiplied, a few VALU instructions behind every MFMA,
//     in one branch-free basic block: matrix pipe and VALU overlap inside the wave, and across the co-resident workgroups;
//   * the WEIGHTS of the first input chunks LIVE IN REGISTERS (18 K-steps x 4 VGPRs per chunk, loaded once per persistent workgroup), the
//     last chunk(s) in LDS where the register file is needed for the second row set: at most two thirds of the A fragments come out of LDS;
//   * a workgroup is FOUR waves per 32 output channels (wave = 32 pixels x 32 couts) on a 8 x 16-pixel tile; ONE workgroup barrier per
//     tile (two with 64 output channels and the output-side norm, whose sum of squares crosses the two cout waves of a pixel); every
//     workgroup owns a contiguous range of tiles (dealt XCD by XCD), so halo rows shared by neighbouring tiles meet in one L2 and the
//     per-image operands are refreshed once or twice per workgroup.
// Epilogue: plain NHWC fp16 (+ the per-pixel sum of squares, 32 couts), or the output-side Block prologue post_pa / post_ps.
// Everything else (addend / residual / other output modes / other channel counts) stays with the other families: the planner asks.
#include <algorithm>
#include <utility>
#include "common.h"

namespace {

template <class F, int... I>
__device__ __forceinline__ void cp_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cp_static_for(F&& f) {   // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): compile-time indices
  cp_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});   // (a 54-step `#pragma unroll` body is left rolled: arrays then go to scratch)
}

constexpr int CP_TW = 16, CP_TH = 8, CP_ITW = CP_TW + 2, CP_ITH = CP_TH + 2;
constexpr int CP_NPX = CP_ITH * CP_ITW;          // 180 halo pixels
constexpr int CP_NSLOT = CP_NPX * 4;             // 720 16-byte slots per 32-channel chunk (pixel-major, 4 channel groups each)
constexpr int CP_ABUF = CP_NSLOT * 16 + 64;      // bytes of one chunk's tile image + a 16-byte dump slot for the threads' surplus slots
constexpr int CP_PITCH = CP_ITW * 64;            // bytes per halo row
constexpr int CP_WCH = 18 * 1024;                // packed weights of one (chunk, 32-cout block): 18 K steps x 1 KiB

constexpr int cp_nj(int nco) { return (CP_NSLOT + 256 * nco - 1) / (256 * nco); }       // slots per thread and chunk: 3 (256 threads) | 2 (512)
constexpr int cp_pset(int nco) { return 2 * 128 + 3 * 32 * nco; }                       // floats: [pa 128 | ps 128 | bias | post_pa | post_ps]
constexpr size_t cp_lds_bytes(int nch, int nco, int wl) {
  return (size_t)2 * nch * CP_ABUF + (size_t)wl * nco * CP_WCH + (size_t)2 * cp_pset(nco) * sizeof(float) + (nco > 1 ? 2 * 2 * 128 * sizeof(float) : 0);
}

// channel group (0..3) stored at position `pos` of halo column hx: group ^ swizzle — conflict-free ds_read_b128 for every tap offset
// (checked exhaustively over the four 16-lane groups of the instruction for pitch 18 pixels)
__device__ __forceinline__ int cp_swz(int hx) { return (hx >> 1) & 3; }

// NCH: 32-channel input chunks; NCO: 32-cout blocks (4 waves each); WL: the LAST WL chunks' weights live in LDS; PRO: Block prologue;
// WPS: waves per SIMD the register budget is capped for (= resident workgroups per CU x NCO)
template <int NCH, int NCO, int WL, bool PRO, int WPS>
__global__ __launch_bounds__(256 * NCO, WPS) void conv_pro_kernel(const ImagenIgemmParams p, int tiles_per_wg) {
  constexpr int NT = 256 * NCO, NJ = cp_nj(NCO), ABUF = CP_ABUF, PSET = cp_pset(NCO);
  constexpr int NU = NJ * NCH;                 // prologue units per tile and thread
  constexpr int STEPS = 18 * NCH;              // K steps per tile and wave
  constexpr int SPU = STEPS / NU;              // K steps per unit: 6 (NJ = 3) | 9 (NJ = 2); the unit's six parts ride on the first six
  constexpr int NR = NCH - WL;                 // chunks whose weights live in registers
  static_assert(STEPS % NU == 0 && SPU >= 6 && NR >= 0, "steps per unit");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const acts = smem;                                                         // [2 tiles][NCH][ABUF]
  char* const wlds = smem + 2 * NCH * ABUF;                                        // [WL][NCO][18][1 KiB]
  float* const par = reinterpret_cast<float*>(wlds + WL * NCO * CP_WCH);           // [2 image parities][PSET]
  float* const red = par + 2 * PSET;                                               // NCO = 2: [2 tile parities][2 cout blocks][128 pixels] sums of squares

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int pb = wave & 3, cb = wave >> 2;     // pixel block (tile rows 2 pb, 2 pb + 1), 32-cout block of this wave

  const int tilesX = (p.OW + CP_TW - 1) / CP_TW, tilesY = (p.OH + CP_TH - 1) / CP_TH;
  const int per_img = tilesX * tilesY;
  const int total = p.B * per_img;
  // contiguous tile range of this workgroup; workgroup ids are dealt round-robin to the XCDs by the dispatcher, so the ranges of one XCD
  // are made neighbours (speed only: any placement is correct)
  int wg = blockIdx.x;
  if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int t_begin = wg * tiles_per_wg, t_end = min(t_begin + tiles_per_wg, total);
  if (t_begin >= t_end) return;

  // ---- weights: A fragment of K step s (tap s / 2, channel groups 2 (s & 1) + half) of chunk ch, cout block cb — registers | LDS
  f16x8 areg[NR > 0 ? NR : 1][18];
  {
    const f16x8* wl = reinterpret_cast<const f16x8*>(p.w) + (size_t)half * p.Cout_pad + cb * 32 + l31;
#pragma unroll
    for (int ch = 0; ch < NR; ++ch)
#pragma unroll
      for (int s = 0; s < 18; ++s) areg[ch][s] = wl[(size_t)(ch * 36 + 2 * s) * p.Cout_pad];
    if constexpr (WL > 0) {   // [chunk][cout block][K step][lane] images, 16 bytes per lane: the fragment order itself
      for (int i = tid; i < WL * NCO * 18 * 64; i += NT) {
        const int ln = i & 63, s = (i >> 6) % 18, blk = (i >> 6) / 18;
        const int ch = NR + blk / NCO, cbb = blk % NCO;
        *reinterpret_cast<f16x8*>(wlds + (size_t)i * 16) =
            reinterpret_cast<const f16x8*>(p.w)[(size_t)(ch * 36 + 2 * s + (ln >> 5)) * p.Cout_pad + cbb * 32 + (ln & 31)];
      }
    }
  }

  // ---- this thread's staging slots S = tid + NT j: halo pixel S >> 2, position S & 3 (= tid & 3), channel group pos ^ swz(hx)
  int s_yx[NJ];        // halo row << 8 | halo column << 2 | channel group
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int S = tid + NT * j;
    const int hp = min(S >> 2, CP_NPX - 1);          // (slots past the tile re-read the last pixel; they land in the dump slot)
    const int hy = hp / CP_ITW, hx = hp - hy * CP_ITW;
    s_yx[j] = (hy << 8) | (hx << 2) | ((tid & 3) ^ cp_swz(hx));
  }
  const int s_last = min(tid + NT * (NJ - 1), CP_NSLOT) * 16;   // LDS offset of the thread's last slot (the dump slot beyond the tile)

  const f16* const x1 = reinterpret_cast<const f16*>(p.x1);
  const f16* const x2 = reinterpret_cast<const f16*>(p.x2);
  const int n1 = p.C1 >> 5;                           // chunks that come from x1 (workgroup-uniform)
  const float* const ssq_b = p.ssq_b ? p.ssq_b : p.ssq_a;       // (no second tensor in the norm: its weight is zero, the load stays unconditional)
  const float wb = p.ssq_b ? p.ssq_wb : 0.0f;
  const bool has_ps = PRO && p.ps != nullptr;
  const int HWin = p.H * p.W;

  struct Tile { int b, oy0, ox0; };
  auto tile_at = [&](int t) __attribute__((always_inline)) -> Tile {
    Tile c;
    c.b = t / per_img;
    const int r = t - c.b * per_img;
    const int ty = r / tilesX;
    c.oy0 = ty * CP_TH;
    c.ox0 = (r - ty * tilesX) * CP_TW;
    return c;
  };

  // raw rows (and statistics) of one tile, as requested
  struct Raw {
    uint4 x[NCH][NJ];
    float qa[NJ], qb[NJ];   // PRO: the sums of squares of the slot's pixel in the two tensors, AS LOADED (combined where they are used: any
                            // arithmetic on them at the request site makes the compiler wait for the loads at the loop's back edge)
    unsigned ok;            // bit j: the slot's pixel lies inside the image
  };
  // request chunk `ch` of slot j of tile c; with the slot's LAST chunk also its statistics and its in-image bit (both are read by every
  // unit of the slot, so they may only change once the slot's last unit is through)
  auto request_unit = [&](Raw& R, const Tile& c, int j, int ch) __attribute__((always_inline)) {
    const int gy = c.oy0 - 1 + (s_yx[j] >> 8), gx = c.ox0 - 1 + ((s_yx[j] >> 2) & 63), kg8 = (s_yx[j] & 3) * 8;
    const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const int gp = ok ? gy * p.W + gx : 0;
    const bool from1 = ch < n1;                       // (workgroup-uniform)
    const f16* src = from1 ? x1 + (size_t)c.b * p.bs1 + (size_t)gp * p.ld1 + ch * 32 : x2 + (size_t)c.b * p.bs2 + (size_t)gp * p.ld2 + (ch - n1) * 32;
    R.x[ch][j] = *reinterpret_cast<const uint4*>(src + kg8);
    if (ch == NCH - 1) {
      if constexpr (PRO) {
        const size_t sp = (size_t)c.b * HWin + gp;
        R.qa[j] = p.ssq_a[sp];
        R.qb[j] = ssq_b[sp];
      }
      R.ok = (R.ok & ~(1u << j)) | ((ok ? 1u : 0u) << j);
    }
  };
  auto request = [&](Raw& R, const Tile& c) __attribute__((always_inline)) {
    cp_static_for<NU>([&](auto ic) __attribute__((always_inline)) { request_unit(R, c, decltype(ic)::value / NCH, decltype(ic)::value % NCH); });
  };

  // per-image operands -> parameter set b & 1 (two consecutive images can be live: tile t in b, tile t + 1 in b + 1)
  int par_b[2] = {-1, -1};
  auto refresh = [&](int b) __attribute__((always_inline)) {   // (workgroup-uniform)
    if (par_b[b & 1] == b) return;
    float* ps_ = par + (b & 1) * PSET;
    if (tid < 128) {
      float a = 0.f, s = 0.f;
      if (PRO && tid < 32 * NCH) {
        a = p.pa[(size_t)b * p.pstride + tid];
        if (has_ps) s = p.ps[(size_t)b * p.pstride + tid];
      }
      ps_[tid] = a;
      ps_[128 + tid] = s;
    } else if (tid < 128 + 32 * NCO) {
      const int c = tid - 128;
      ps_[256 + c] = p.bias ? p.bias[c] : 0.0f;
      ps_[256 + 32 * NCO + c] = p.post_pa ? p.post_pa[(size_t)b * p.post_pstride + c] : 0.0f;
      ps_[256 + 64 * NCO + c] = p.post_pa ? p.post_ps[(size_t)b * p.post_pstride + c] : 0.0f;
    }
    par_b[b & 1] = b;
    __syncthreads();
  };

  // The prologue of one UNIT (slot j of chunk ch: 8 channels of one halo pixel) in SIX PARTS:
  //   part 0: rs = 1 / ||pixel||;  parts 1-4: two channels each: silu(x * rs * pa + ps) -> one packed dword;  part 5: zero padding, ds_write_b128.
  // Branch-free (absent shifts read zeros from the parameter set; the launcher admits act_in = SiLU only): the K loop stays one basic block.
  struct Unit { float rs; unsigned o[4]; };
  auto transform_part = [&](const Raw& R, Unit& U, int u, int part, int buf, int b) __attribute__((always_inline)) {
    const int j = u / NCH, ch = u - j * NCH;
    if (part == 0) {
      U.rs = 1.0f;
      if constexpr (PRO) U.rs = __builtin_amdgcn_rsqf(fmaxf(R.qa[j] + wb * R.qb[j], 1e-24f));
    } else if (part <= 4) {
      const int e = 2 * (part - 1);
      const unsigned w = e == 0 ? R.x[ch][j].x : e == 2 ? R.x[ch][j].y : e == 4 ? R.x[ch][j].z : R.x[ch][j].w;
      if constexpr (PRO) {
        const float* pa_l = par + (b & 1) * PSET + (s_yx[j] & 3) * 8 + ch * 32 + e;
        const float2 av = *reinterpret_cast<const float2*>(pa_l), sv = *reinterpret_cast<const float2*>(pa_l + 128);
        const f16x2 in = __builtin_bit_cast(f16x2, w);
        f16x2 out;
        out[0] = (f16)silu_f((float)in[0] * U.rs * av.x + sv.x);
        out[1] = (f16)silu_f((float)in[1] * U.rs * av.y + sv.y);
        U.o[part - 1] = __builtin_bit_cast(unsigned, out);
      } else {
        U.o[part - 1] = w;
      }
    } else if (part == 5) {
      const bool ok = (R.ok >> j) & 1u;      // zero padding applies to the ACTIVATED tensor
      const uint4 ow = ok ? make_uint4(U.o[0], U.o[1], U.o[2], U.o[3]) : make_uint4(0u, 0u, 0u, 0u);
      const int so = j == NJ - 1 ? s_last : (tid + NT * j) * 16;
      *reinterpret_cast<uint4*>(acts + (buf * NCH + ch) * ABUF + so) = ow;
    }
  };

  // ---- MFMA side: lane = pixel (row 2 pb + (l31 >> 4), column l31 & 15) x the 32 output channels of cout block cb
  const int py = 2 * pb + (l31 >> 4), px = l31 & 15;
  int bA[6];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int hx = px + dx;
    const int a0 = py * CP_PITCH + hx * 64 + ((half ^ cp_swz(hx)) << 4);
    bA[2 * dx] = a0;            // K step 0 of a tap: channel groups 0 / 1
    bA[2 * dx + 1] = a0 ^ 32;   // K step 1: groups 2 / 3
  }
  const char* const wl_l = wlds + cb * CP_WCH + lane * 16;   // this lane's A fragments of the LDS-resident chunks

  // two register sets of raw rows: in iteration `it` (tile t = t_begin + it) set it & 1 holds tile t + 1 — consumed by this iteration's
  // prologue and refilled, unit by unit, with tile t + 3 — and set (it + 1) & 1 holds tile t + 2, in flight
  Raw RA, RB;
  RA.ok = RB.ok = 0;
  Tile cur = tile_at(t_begin);
  refresh(cur.b);
  request(RA, cur);
  {
    Unit U;
    cp_static_for<NU * 6>([&](auto ic) __attribute__((always_inline)) { transform_part(RA, U, decltype(ic)::value / 6, decltype(ic)::value % 6, 0, cur.b); });
  }
  Tile nxt = tile_at(min(t_begin + 1, t_end - 1));
  request(RA, nxt);
  request(RB, tile_at(min(t_begin + 2, t_end - 1)));
  int buf = 0;

  // one tile: K loop of tile t with the prologue of tile t + 1 between its steps (step s carries part s % SPU of unit s / SPU; as soon as a
  // unit has left its registers the same unit of tile t + 3 is requested into them), then the epilogue.  Past the end of the range the
  // last tile is requested / transformed again (into a buffer nobody reads any more): no branch in the loop.
  auto one_tile = [&](Raw& R, int t) __attribute__((always_inline)) {
    __syncthreads();            // tile t's image is complete; nobody reads the other buffer (tile t - 1) or the previous epilogue's operands any more
    if (t + 1 < t_end) refresh(nxt.b);   // (uniform; a second barrier only when tile t + 1 opens a new image)
    const char* ab = acts + buf * NCH * ABUF;
    const float* ep = par + (cur.b & 1) * PSET + 256;
    const Tile far = tile_at(min(t + 3, t_end - 1));

    f32x16 acc;                 // starts at the bias
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bq = *reinterpret_cast<const float4*>(ep + 32 * cb + 8 * q + 4 * half);
      acc[4 * q] = bq.x; acc[4 * q + 1] = bq.y; acc[4 * q + 2] = bq.z; acc[4 * q + 3] = bq.w;
    }
    auto bfrag = [&](int s) __attribute__((always_inline)) -> f16x8 {
      const int ch = s / 18, k = s - 18 * ch;
      const int tap = k >> 1, ks = k & 1;
      const int dy = tap / 3, dx = tap - 3 * dy;
      return *reinterpret_cast<const f16x8*>(ab + ch * ABUF + bA[2 * dx + ks] + dy * CP_PITCH);
    };
    auto afrag = [&](int s) __attribute__((always_inline)) -> f16x8 {     // (LDS-resident chunks only)
      const int ch = s / 18, k = s - 18 * ch;
      return *reinterpret_cast<const f16x8*>(wl_l + ((ch - NR) * NCO * 18 + k) * 1024);
    };
    f16x8 bfr[2], afr[2];
    bfr[0] = bfrag(0);
    if constexpr (NR == 0) afr[0] = afrag(0);
    Unit U;
    cp_static_for<STEPS>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s + 1 < STEPS) {      // the fragments of step s + 1 are requested before the MFMA of step s
        bfr[(s + 1) & 1] = bfrag(s + 1);
        if constexpr ((s + 1) / 18 >= NR) afr[(s + 1) & 1] = afrag(s + 1);
      }
      if constexpr (s / 18 < NR) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[s / 18][s % 18], bfr[s & 1], acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[s & 1], bfr[s & 1], acc, 0, 0, 0);
      constexpr int u = s / SPU, part = s % SPU;
      if constexpr (part < 6) transform_part(R, U, u, part, buf ^ 1, nxt.b);
      if constexpr (part == 5) request_unit(R, far, u / NCH, u % NCH);
      __builtin_amdgcn_sched_barrier(0);   // keeps each MFMA next to its share of the VALU work
    });

    // ---- epilogue of tile t: register quad q holds couts 32 cb + 8q + 4*half + {0..3} of the lane's pixel
    {
      const int oy = cur.oy0 + py, ox = cur.ox0 + px;
      const bool px_ok = oy < p.OH && ox < p.OW;
      const size_t opix = (size_t)(px_ok ? oy * p.OW + ox : 0);
      char* yb = reinterpret_cast<char*>(reinterpret_cast<f16*>(p.y) + (size_t)cur.b * p.bsy + opix * p.ldy) + 64 * cb + 32 * half;
      f16x4 oq[4];
      float ssq = 0.f;
      if (p.post_pa) {   // output-side Block prologue: v / ||v|| * post_pa + post_ps -> SiLU (norm over ALL couts of the pixel)
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) tot += acc[r] * acc[r];
        tot += __shfl_xor(tot, 32);
        if constexpr (NCO > 1) {   // the pixel's other 32 couts sit in wave (pb, cb ^ 1): one hop through LDS (the tile parity keeps the next
          float* rr = red + (buf & 1) * 256;                     // tile's sums away from a slower wave's read of this one's)
          if (half == 0) rr[cb * 128 + pb * 32 + l31] = tot;
          __syncthreads();
          tot += rr[(cb ^ 1) * 128 + pb * 32 + l31];
        }
        const float rsn = __builtin_amdgcn_rsqf(fmaxf(tot, 1e-24f));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 pa = *reinterpret_cast<const float4*>(ep + 32 * NCO + 32 * cb + 8 * q + 4 * half);
          const float4 ps = *reinterpret_cast<const float4*>(ep + 64 * NCO + 32 * cb + 8 * q + 4 * half);
          const float pav[4] = {pa.x, pa.y, pa.z, pa.w}, psv[4] = {ps.x, ps.y, ps.z, ps.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) oq[q][e] = (f16)silu_f(acc[4 * q + e] * rsn * pav[e] + psv[e]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            oq[q][e] = (f16)acc[4 * q + e];
            const float r = (float)oq[q][e];   // statistics of the value the consumer will read back
            ssq += r * r;
          }
      }
      // quads (q, q + 2) exchanged between the half-waves: the lower half-wave then owns channels 0-15 of the block, the upper one
      // 16-31, as two 16-byte pieces each
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const imagen_u32x4 piece = imagen_pair_quads(oq[q], oq[q + 2]);
        if (px_ok) *reinterpret_cast<imagen_u32x4*>(yb + 16 * q) = piece;
      }
      if constexpr (NCO == 1) {
        if (p.ssq_out && !p.post_pa) {
          ssq += __shfl_xor(ssq, 32);
          if (half == 0 && px_ok) p.ssq_out[(size_t)cur.b * (p.OH * p.OW) + opix] = ssq;
        }
      }
    }
    cur = nxt;
    nxt = tile_at(min(t + 2, t_end - 1));
    buf ^= 1;
  };

  // tiles in PAIRS, the second one unconditionally: a range of odd length computes its last tile twice (same values, same addresses).  With a
  // conditional second half the compiler sees a path from one first half straight into the next, along which only the first set's own
  // refills are younger than the rows it waits for — and waits with vmcnt(8) instead of vmcnt(20): the second set's loads drained every tile.
  for (int t = t_begin; t < t_end; t += 2) {
    one_tile(RA, t);
    one_tile(RB, t + 1);
  }
}

// ---- instantiations: <NCH, NCO, WL, PRO, WPS>
//   32 -> 32 : <1, 1, 0>  2 workgroups / CU   (72 weight registers, two row sets of 15; the cap of three workgroups spills)
//   64 -> 32 : <2, 1, 1>  2 workgroups / CU   (chunk 1 from LDS: its 72 registers hold the second row set; 60 KB of LDS)
//   64 -> 64 : <2, 2, 1>  1 workgroup of 8 waves / CU
//   96 -> 64 : <3, 2, 2>  1 workgroup of 8 waves / CU   (chunks 1, 2 from LDS: 72 KB, six tile images 68 KB)
template <int NCH, int NCO, int WL, bool PRO, int WPS>
int cp_launch(const ImagenIgemmParams& p, hipStream_t s) {
  auto kern = conv_pro_kernel<NCH, NCO, WL, PRO, WPS>;
  constexpr size_t lds = cp_lds_bytes(NCH, NCO, WL);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_done[16] = {};
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("conv_pro: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int tilesX = (p.OW + CP_TW - 1) / CP_TW, tilesY = (p.OH + CP_TH - 1) / CP_TH;
  const int total = p.B * tilesX * tilesY;
  const int per_cu = std::max(1, std::min(WPS / NCO, (int)((160 * 1024) / lds)));
  int resident = std::max(8, std::max(1, cus) * per_cu);
  resident -= resident % 8;                                     // (the XCD dealing of the tile ranges wants a multiple of 8)
  int tiles_per_wg = std::max(1, (total + resident - 1) / resident);
  tiles_per_wg += tiles_per_wg & 1;                             // (tiles go in pairs: an odd range would compute its last tile twice)
  int gx = (total + tiles_per_wg - 1) / tiles_per_wg;
  gx = (gx + 7) / 8 * 8;                                        // (surplus workgroups find an empty range and leave)
  hipLaunchKernelGGL(kern, dim3(gx), dim3(256 * NCO), lds, s, p, tiles_per_wg);
  return imagen_hip_status("conv_pro launch");
}

}  // namespace

// ---- family interface (igemm.hip lists this family behind the big-tile one): cfg 0 = 32 output channels, cfg 1 = 64
int imagen_conv_pro_num_configs() { return 2; }

int imagen_conv_pro_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (idx < 0 || idx > 1) return -1;
  if (tile_pixels) *tile_pixels = CP_TH * CP_TW;
  if (tile_cout) *tile_cout = 32 * (idx + 1);
  if (kgroups) *kgroups = 4;
  return 0;
}

long imagen_conv_pro_lds_bytes(int idx, int KH, int KW, int TH, int TW) {
  if (idx < 0 || idx > 1 || KH != 3 || KW != 3 || TH != CP_TH || TW != CP_TW) return -1;
  return idx == 0 ? (long)cp_lds_bytes(2, 1, 1) : (long)cp_lds_bytes(3, 2, 2);
}

int launch_conv_pro(const ImagenIgemmParams* pp, int idx, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  IMAGEN_CHECK(idx == 0 || idx == 1, "conv_pro: bad cfg");
  const int nco = idx + 1;
  IMAGEN_CHECK(p.TH == CP_TH && p.TW == CP_TW, "conv_pro: 8x16 tiles (got %dx%d)", p.TH, p.TW);
  IMAGEN_CHECK(p.stride == 1 && p.KH == 3 && p.KW == 3 && p.pad == 1 && p.OH == p.H && p.OW == p.W, "conv_pro: 3x3 stride-1 pad-1 convolutions only");
  IMAGEN_CHECK(p.C1 % 32 == 0 && p.C2 % 32 == 0 && p.C1 > 0 && (p.C2 == 0 || p.x2) && p.Cin_pad == p.C1 + p.C2, "conv_pro: inputs in 32-channel chunks (C1 %d C2 %d)",
               p.C1, p.C2);
  const int nch = (p.C1 + p.C2) / 32;
  IMAGEN_CHECK((nco == 1 && (nch == 1 || nch == 2)) || (nco == 2 && (nch == 2 || nch == 3)), "conv_pro: %d input channels -> %d couts is not instantiated",
               32 * nch, 32 * nco);
  IMAGEN_CHECK(p.ld1 % 8 == 0 && (p.C2 == 0 || p.ld2 % 8 == 0) && ((size_t)p.x1 & 15) == 0 && ((size_t)p.x2 & 15) == 0 && p.bs1 % 8 == 0 && p.bs2 % 8 == 0,
               "conv_pro: input rows must keep 16-byte alignment");
  IMAGEN_CHECK(p.Cout == 32 * nco && p.Cout_pad % 32 == 0 && p.Cout_pad >= p.Cout, "conv_pro: cfg %d takes exactly %d output channels (Cout %d)", idx, 32 * nco, p.Cout);
  IMAGEN_CHECK(!p.mu && !p.rs, "conv_pro: the prologue takes its statistics from ssq_a / ssq_b (no mu / rs)");
  const bool pro = p.ssq_a != nullptr || p.pa != nullptr || p.ps != nullptr || p.act_in != IMAGEN_ACT_NONE;
  IMAGEN_CHECK(!pro || (p.ssq_a && p.pa && p.act_in == IMAGEN_ACT_SILU && ((p.C2 == 0) == (p.ssq_b == nullptr))),
               "conv_pro: the prologue is ssq_a (+ ssq_b with a second tensor) statistics, pa (ps), SiLU");
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res && p.act_out == IMAGEN_ACT_NONE && !p.gca_part,
               "conv_pro: plain NHWC output only (no addend / residual / output activation / GlobalContext partials)");
  IMAGEN_CHECK(p.ldy % 8 == 0 && p.bsy % 8 == 0 && ((size_t)p.y & 15) == 0, "conv_pro: output rows must keep 16-byte alignment");
  IMAGEN_CHECK(!p.post_pa || (p.post_ps && !p.ssq_out), "conv_pro: post_pa needs post_ps and excludes ssq_out");
  IMAGEN_CHECK(!p.ssq_out || nco == 1, "conv_pro: ssq_out with 32 output channels only");
  switch (nco * 100 + nch * 10 + (pro ? 1 : 0)) {
    case 110: return cp_launch<1, 1, 0, false, 2>(p, s);
    case 111: return cp_launch<1, 1, 0, true, 2>(p, s);
    case 120: return cp_launch<2, 1, 1, false, 2>(p, s);
    case 121: return cp_launch<2, 1, 1, true, 2>(p, s);
    case 220: return cp_launch<2, 2, 1, false, 2>(p, s);
    case 221: return cp_launch<2, 2, 1, true, 2>(p, s);
    case 230: return cp_launch<3, 2, 2, false, 2>(p, s);
    default: return cp_launch<3, 2, 2, true, 2>(p, s);
  }
}
