// conv_pro.hip — the streaming 3x3 convolution with the Block prologue IN FLIGHT (seventh igemm family, gfx950): C_out = 32 from one or
// two 32-channel inputs (x, or the up path's concat cat(x, skip * 2^-1/2)), stride 1, pad 1, NHWC fp16 — Block (ip.py:671-691:
// ChanRMSNorm -> per-channel affine -> SiLU -> Conv2d 3x3) at the 256^2 / 128^2 levels of the README super-resolution unet, where the
// input must still go through the prologue (block1 of the up-path blocks: the norm runs over the concat; block1 behind a downsample).
//
// Why it exists (round 3, profiles/r03_graph_profile.txt): conv_stream.hip runs these launches at 0.23 of the HBM rate (64 -> 32 @256^2:
// 111 us for 201 MB; 32 -> 32: 62 us for 134 MB).  Its workgroup of eight waves moves in lock step — wait for the tile's direct-to-LDS
// copies, barrier, transform the tile IN PLACE in LDS (read, fp32 math, write back), barrier, multiply, store — one workgroup per CU with
// two inputs, so every one of those latencies is exposed, and both MFMA operands come out of LDS (the LDS is as busy as the matrix pipe).
// Here
//   * the raw rows of tile t+2 are requested into REGISTERS (plain 16-byte global loads, one per lane and slot) while tile t is being
//     multiplied; the prologue runs on those registers — one LDS write per element, no read-modify-write of a landed tile, no DMA wait;
//   * the WEIGHTS LIVE IN REGISTERS (18 K-steps x 4 VGPRs per 32-channel input, loaded once per persistent workgroup): the only LDS
//     reads are the B fragments — half the LDS traffic of the streaming kernel;
//   * a workgroup is FOUR waves (one per SIMD) on a 8 x 16-pixel tile, several workgroups per CU: the workgroups drift apart, so one's
//     VALU phase (the prologue: two transcendentals per element) runs beside another's MFMA phase on the same SIMD; inside a wave the
//     prologue of tile t+1 sits between the K-steps of tile t, slot by slot, in one branch-free basic block;
//   * ONE workgroup barrier per tile; every workgroup owns a contiguous range of tiles (dealt XCD by XCD), so halo rows shared by
//     neighbouring tiles meet in one L2 and the per-image operands are refreshed once or twice per workgroup.
// Epilogue: plain NHWC fp16 (+ the per-pixel sum of squares), or the output-side Block prologue post_pa / post_ps (conv_epilogue.h).
// Everything else (addend / residual / other output modes / C_out != 32) stays with the other families: the planner asks.
#include <algorithm>
#include "common.h"

namespace {

constexpr int CP_TW = 16, CP_TH = 8, CP_ITW = CP_TW + 2, CP_ITH = CP_TH + 2;
constexpr int CP_NPX = CP_ITH * CP_ITW;          // 180 halo pixels
constexpr int CP_NSLOT = CP_NPX * 4;             // 720 16-byte slots per 32-channel input (pixel-major, 4 channel groups each)
constexpr int CP_NT = 256;                       // threads: 4 waves, wave w owns tile rows 2w, 2w + 1
constexpr int CP_NJ = (CP_NSLOT + CP_NT - 1) / CP_NT;   // 3 slots per thread and input
constexpr int CP_ABUF = CP_NJ * CP_NT * 16;      // bytes of one input's tile image (768 slots: the last 48 are padding, written and never read)
constexpr int CP_PITCH = CP_ITW * 64;            // bytes per halo row
constexpr int CP_PAR = 2 * (2 * 64 + 3 * 32);    // floats: two parameter sets (image parity) of [pa 64 | ps 64 | bias 32 | post_pa 32 | post_ps 32]

constexpr size_t cp_lds_bytes(int nch) { return (size_t)2 * nch * CP_ABUF + (size_t)CP_PAR * sizeof(float); }

// channel group (0..3) stored at position `pos` of halo column hx: group ^ swizzle — conflict-free ds_read_b128 for every tap offset
// (checked exhaustively over the four 16-lane groups of the instruction for pitch 18 pixels)
__device__ __forceinline__ int cp_swz(int hx) { return (hx >> 1) & 3; }

template <int NCH, bool PRO>
__global__ __launch_bounds__(CP_NT, NCH == 1 ? 3 : 2) void conv_pro_kernel(const ImagenIgemmParams p, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const acts = smem;                                                    // [2 tiles][NCH][CP_ABUF]
  float* const par = reinterpret_cast<float*>(smem + 2 * NCH * CP_ABUF);      // [2][pa 64 | ps 64 | bias 32 | post_pa 32 | post_ps 32]
  constexpr int PSET = 2 * 64 + 3 * 32;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int tilesX = (p.OW + CP_TW - 1) / CP_TW, tilesY = (p.OH + CP_TH - 1) / CP_TH;
  const int per_img = tilesX * tilesY;
  const int total = p.B * per_img;
  // contiguous tile range of this workgroup; workgroup ids are dealt round-robin to the XCDs by the dispatcher, so the ranges of one XCD
  // are made neighbours (speed only: any placement is correct)
  int wg = blockIdx.x;
  if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int t_begin = wg * tiles_per_wg, t_end = min(t_begin + tiles_per_wg, total);
  if (t_begin >= t_end) return;

  // ---- weights -> registers: A fragment of K step s (tap s / 2, channel groups 2 (s & 1) + half) of input chunk ch
  f16x8 areg[NCH][18];
  {
    const f16x8* wl = reinterpret_cast<const f16x8*>(p.w) + (size_t)half * p.Cout_pad + l31;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int s = 0; s < 18; ++s) areg[ch][s] = wl[(size_t)(ch * 36 + 2 * s) * p.Cout_pad];
  }

  // ---- this thread's staging slots S = tid + 256 j: halo pixel S >> 2, position S & 3 (= tid & 3), channel group pos ^ swz(hx)
  int s_yx[CP_NJ];        // halo row << 8 | halo column << 2 | channel group
#pragma unroll
  for (int j = 0; j < CP_NJ; ++j) {
    const int S = tid + CP_NT * j;
    const int hp = min(S >> 2, CP_NPX - 1);          // (slots past the tile — j = 2, tid >= 208 — re-read the last pixel into the padding)
    const int hy = hp / CP_ITW, hx = hp - hy * CP_ITW;
    s_yx[j] = (hy << 8) | (hx << 2) | ((tid & 3) ^ cp_swz(hx));
  }

  const f16* const x1 = reinterpret_cast<const f16*>(p.x1);
  const f16* const x2 = reinterpret_cast<const f16*>(p.x2);
  constexpr bool has_b = NCH == 2;   // (launcher-checked: the prologue over two inputs has both statistics)
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
    uint4 x[NCH][CP_NJ];
    float qa[CP_NJ], qb[CP_NJ];   // PRO: the sums of squares of the slot's pixel in the two inputs, AS LOADED (combined where they are used: any
                                  // arithmetic on them at the request site makes the compiler wait for the loads at the loop's back edge)
    unsigned ok;           // bit j: the slot's pixel lies inside the image
  };
  // request input `ch` of slot j of tile c; with the slot's LAST input also its statistics and its in-image bit (both are read by every
  // unit of the slot, so they may only change once the slot's last unit is through)
  auto request_unit = [&](Raw& R, const Tile& c, int j, int ch) __attribute__((always_inline)) {
    const int gy = c.oy0 - 1 + (s_yx[j] >> 8), gx = c.ox0 - 1 + ((s_yx[j] >> 2) & 63), kg8 = (s_yx[j] & 3) * 8;
    const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const int gp = ok ? gy * p.W + gx : 0;
    if (ch == 0) R.x[0][j] = *reinterpret_cast<const uint4*>(x1 + (size_t)c.b * p.bs1 + (size_t)gp * p.ld1 + kg8);
    if constexpr (NCH == 2) {
      if (ch == 1) R.x[1][j] = *reinterpret_cast<const uint4*>(x2 + (size_t)c.b * p.bs2 + (size_t)gp * p.ld2 + kg8);
    }
    if (ch == NCH - 1) {
      if constexpr (PRO) {
        const size_t sp = (size_t)c.b * HWin + gp;
        R.qa[j] = p.ssq_a[sp];
        if constexpr (has_b) R.qb[j] = p.ssq_b[sp];
      }
      R.ok = (R.ok & ~(1u << j)) | ((ok ? 1u : 0u) << j);
    }
  };
  auto request = [&](Raw& R, const Tile& c) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < CP_NJ; ++j)
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) request_unit(R, c, j, ch);
  };

  // per-image operands -> parameter set b & 1 (two consecutive images can be live: tile t in b, tile t + 1 in b + 1)
  int par_b[2] = {-1, -1};
  auto refresh = [&](int b) __attribute__((always_inline)) {   // (workgroup-uniform)
    if (par_b[b & 1] == b) return;
    float* ps_ = par + (b & 1) * PSET;
    if (tid < 64) {
      float a = 0.f, s = 0.f;
      if (PRO && tid < 32 * NCH) {
        a = p.pa[(size_t)b * p.pstride + tid];
        if (has_ps) s = p.ps[(size_t)b * p.pstride + tid];
      }
      ps_[tid] = a;
      ps_[64 + tid] = s;
    } else if (tid < 96) {
      const int c = tid - 64;
      ps_[128 + c] = p.bias ? p.bias[c] : 0.0f;
      ps_[160 + c] = p.post_pa ? p.post_pa[(size_t)b * p.post_pstride + c] : 0.0f;
      ps_[192 + c] = p.post_pa ? p.post_ps[(size_t)b * p.post_pstride + c] : 0.0f;
    }
    par_b[b & 1] = b;
    __syncthreads();
  };

  // The prologue of one UNIT (slot j of input ch: 8 channels of one halo pixel) in SIX PARTS, so that it can sit between the K steps of the
  // tile being multiplied, a few VALU instructions behind every MFMA (a unit per six K steps: 18 NCH steps = 3 NCH units):
  //   part 0: rs = 1 / ||pixel||;  parts 1-4: two channels each: silu(x * rs * pa + ps) -> one packed dword;  part 5: zero padding, ds_write_b128.
  // Branch-free (absent shifts read zeros from the parameter set; the launcher admits act_in = SiLU only): the K loop stays one basic block.
  struct Unit { float rs; unsigned o[4]; };
  auto transform_part = [&](const Raw& R, Unit& U, int u, int part, int buf, int b) __attribute__((always_inline)) {
    const int j = u / NCH, ch = u - j * NCH;
    if (part == 0) {
      U.rs = 1.0f;
      if constexpr (PRO) {
        float q = R.qa[j];
        if constexpr (has_b) q += p.ssq_wb * R.qb[j];
        U.rs = __builtin_amdgcn_rsqf(fmaxf(q, 1e-24f));
      }
    } else if (part <= 4) {
      const int e = 2 * (part - 1);
      const unsigned w = e == 0 ? R.x[ch][j].x : e == 2 ? R.x[ch][j].y : e == 4 ? R.x[ch][j].z : R.x[ch][j].w;
      if constexpr (PRO) {
        const float* pa_l = par + (b & 1) * PSET + (s_yx[j] & 3) * 8 + ch * 32 + e;
        const float2 av = *reinterpret_cast<const float2*>(pa_l), sv = *reinterpret_cast<const float2*>(pa_l + 64);
        const f16x2 in = __builtin_bit_cast(f16x2, w);
        f16x2 out;
        out[0] = (f16)silu_f((float)in[0] * U.rs * av.x + sv.x);
        out[1] = (f16)silu_f((float)in[1] * U.rs * av.y + sv.y);
        U.o[part - 1] = __builtin_bit_cast(unsigned, out);
      } else {
        U.o[part - 1] = w;
      }
    } else {
      const bool ok = (R.ok >> j) & 1u;      // zero padding applies to the ACTIVATED tensor
      const uint4 ow = ok ? make_uint4(U.o[0], U.o[1], U.o[2], U.o[3]) : make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(acts + (buf * NCH + ch) * CP_ABUF + (tid + CP_NT * j) * 16) = ow;
    }
  };

  // ---- MFMA side: lane = pixel (row 2 wave + (l31 >> 4), column l31 & 15) x all 32 output channels
  const int py = 2 * wave + (l31 >> 4), px = l31 & 15;
  int bA[6];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int hx = px + dx;
    const int a0 = py * CP_PITCH + hx * 64 + ((half ^ cp_swz(hx)) << 4);
    bA[2 * dx] = a0;            // K step 0 of a tap: channel groups 0 / 1
    bA[2 * dx + 1] = a0 ^ 32;   // K step 1: groups 2 / 3
  }

  Tile cur = tile_at(t_begin);
  Raw R;
  R.ok = 0;
  refresh(cur.b);
  request(R, cur);
  {
    Unit U;
#pragma unroll
    for (int u = 0; u < CP_NJ * NCH; ++u)
#pragma unroll
      for (int part = 0; part < 6; ++part) transform_part(R, U, u, part, 0, cur.b);
  }
  Tile nxt = cur;
  bool more = t_begin + 1 < t_end;
  if (more) {
    nxt = tile_at(t_begin + 1);
    request(R, nxt);
  }
  int buf = 0;

  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();            // tile t's image is complete; nobody reads the other buffer (tile t - 1) or the previous epilogue's operands any more
    if (more) refresh(nxt.b);   // (uniform; a second barrier only when tile t + 1 opens a new image)
    const char* ab = acts + buf * NCH * CP_ABUF;
    const float* ep = par + (cur.b & 1) * PSET + 128;

    // accumulators start at the bias
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bq = *reinterpret_cast<const float4*>(ep + 8 * q + 4 * half);
      acc[4 * q] = bq.x; acc[4 * q + 1] = bq.y; acc[4 * q + 2] = bq.z; acc[4 * q + 3] = bq.w;
    }

    // K loop of tile t with the prologue of tile t + 1 between its steps: step s carries part s % 6 of unit s / 6.  Past the last tile the
    // units of a stale R go to the other buffer, which nobody reads any more: no branch in the loop.  The B fragment of step s + 1 is
    // requested before the MFMA of step s; the scheduling fence after every step keeps each MFMA next to its share of the VALU work.
    constexpr int STEPS = 18 * NCH;
    auto bfrag = [&](int s) __attribute__((always_inline)) -> f16x8 {
      const int ch = s / 18, k = s - 18 * ch;
      const int tap = k >> 1, ks = k & 1;
      const int dy = tap / 3, dx = tap - 3 * dy;
      return *reinterpret_cast<const f16x8*>(ab + ch * CP_ABUF + bA[2 * dx + ks] + dy * CP_PITCH);
    };
    // ... and as soon as a unit of tile t + 1 has left its registers (part 5), the same unit of tile t + 2 is requested into them: every
    // load has a whole tile period (K loop, epilogue, barrier) to land, with no second register set.  Past the end of the range the last
    // tile is requested again (never consumed): no branch.
    const bool more2 = t + 2 < t_end;
    const Tile nn = tile_at(min(t + 2, t_end - 1));
    f16x8 bfr[2];
    bfr[0] = bfrag(0);
    Unit U;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s + 1 < STEPS) bfr[(s + 1) & 1] = bfrag(s + 1);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[s / 18][s % 18], bfr[s & 1], acc, 0, 0, 0);
      transform_part(R, U, s / 6, s % 6, buf ^ 1, nxt.b);
      if (s % 6 == 5) request_unit(R, nn, (s / 6) / NCH, (s / 6) % NCH);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue of tile t: register quad q holds couts 8q + 4*half + {0..3} of the lane's pixel
    {
      const int oy = cur.oy0 + py, ox = cur.ox0 + px;
      const bool px_ok = oy < p.OH && ox < p.OW;
      const size_t opix = (size_t)(px_ok ? oy * p.OW + ox : 0);
      char* yb = reinterpret_cast<char*>(reinterpret_cast<f16*>(p.y) + (size_t)cur.b * p.bsy + opix * p.ldy) + 32 * half;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[r];
      f16x4 oq[4];
      float ssq = 0.f;
      if (p.post_pa) {   // output-side Block prologue: v / ||v|| * post_pa + post_ps -> SiLU (norm over the pixel's 32 couts)
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) tot += v[r] * v[r];
        tot += __shfl_xor(tot, 32);
        const float rsn = __builtin_amdgcn_rsqf(fmaxf(tot, 1e-24f));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 pa = *reinterpret_cast<const float4*>(ep + 32 + 8 * q + 4 * half);
          const float4 ps = *reinterpret_cast<const float4*>(ep + 64 + 8 * q + 4 * half);
          const float pav[4] = {pa.x, pa.y, pa.z, pa.w}, psv[4] = {ps.x, ps.y, ps.z, ps.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) oq[q][e] = (f16)silu_f(v[4 * q + e] * rsn * pav[e] + psv[e]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            oq[q][e] = (f16)v[4 * q + e];
            const float r = (float)oq[q][e];   // statistics of the value the consumer will read back
            ssq += r * r;
          }
      }
      // quads (q, q + 2) exchanged between the half-waves: the lower half-wave then owns channels 0-15, the upper one 16-31, as two
      // 16-byte pieces each
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const imagen_u32x4 piece = imagen_pair_quads(oq[q], oq[q + 2]);
        if (px_ok) *reinterpret_cast<imagen_u32x4*>(yb + 16 * q) = piece;
      }
      if (p.ssq_out && !p.post_pa) {
        ssq += __shfl_xor(ssq, 32);
        if (half == 0 && px_ok) p.ssq_out[(size_t)cur.b * (p.OH * p.OW) + opix] = ssq;
      }
    }

    cur = nxt;
    nxt = nn;
    more = more2;
    buf ^= 1;
  }
}

template <int NCH, bool PRO>
int cp_launch(const ImagenIgemmParams& p, hipStream_t s) {
  auto kern = conv_pro_kernel<NCH, PRO>;
  constexpr size_t lds = cp_lds_bytes(NCH);
  const int tilesX = (p.OW + CP_TW - 1) / CP_TW, tilesY = (p.OH + CP_TH - 1) / CP_TH;
  const int total = p.B * tilesX * tilesY;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int per_cu = NCH == 1 ? 3 : 2;
  int resident = std::max(8, std::max(1, cus) * per_cu);
  resident -= resident % 8;                                     // (the XCD dealing of the tile ranges wants a multiple of 8)
  const int tiles_per_wg = std::max(1, (total + resident - 1) / resident);
  int gx = (total + tiles_per_wg - 1) / tiles_per_wg;
  gx = (gx + 7) / 8 * 8;                                        // (surplus workgroups find an empty range and leave)
  hipLaunchKernelGGL(kern, dim3(gx), dim3(CP_NT), lds, s, p, tiles_per_wg);
  return imagen_hip_status("conv_pro launch");
}

}  // namespace

// ---- family interface (igemm.hip lists this family behind the big-tile one)
int imagen_conv_pro_num_configs() { return 1; }

int imagen_conv_pro_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (idx != 0) return -1;
  if (tile_pixels) *tile_pixels = CP_TH * CP_TW;
  if (tile_cout) *tile_cout = 32;
  if (kgroups) *kgroups = 4;
  return 0;
}

long imagen_conv_pro_lds_bytes(int idx, int KH, int KW, int TH, int TW) {
  if (idx != 0 || KH != 3 || KW != 3 || TH != CP_TH || TW != CP_TW) return -1;
  return (long)cp_lds_bytes(2);
}

int launch_conv_pro(const ImagenIgemmParams* pp, int idx, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  IMAGEN_CHECK(idx == 0, "conv_pro: bad cfg");
  IMAGEN_CHECK(p.TH == CP_TH && p.TW == CP_TW, "conv_pro: 8x16 tiles (got %dx%d)", p.TH, p.TW);
  IMAGEN_CHECK(p.stride == 1 && p.KH == 3 && p.KW == 3 && p.pad == 1 && p.OH == p.H && p.OW == p.W, "conv_pro: 3x3 stride-1 pad-1 convolutions only");
  IMAGEN_CHECK(p.C1 == 32 && (p.C2 == 0 || (p.C2 == 32 && p.x2)) && p.Cin_pad == p.C1 + p.C2, "conv_pro: inputs of 32 (+ 32) channels (C1 %d C2 %d)", p.C1, p.C2);
  IMAGEN_CHECK(p.ld1 % 8 == 0 && (p.C2 == 0 || p.ld2 % 8 == 0) && ((size_t)p.x1 & 15) == 0 && ((size_t)p.x2 & 15) == 0 && p.bs1 % 8 == 0 && p.bs2 % 8 == 0,
               "conv_pro: input rows must keep 16-byte alignment");
  IMAGEN_CHECK(p.Cout == 32 && p.Cout_pad % 32 == 0, "conv_pro: exactly 32 output channels (Cout %d)", p.Cout);
  IMAGEN_CHECK(!p.mu && !p.rs, "conv_pro: the prologue takes its statistics from ssq_a / ssq_b (no mu / rs)");
  const bool pro = p.ssq_a != nullptr || p.pa != nullptr || p.ps != nullptr || p.act_in != IMAGEN_ACT_NONE;
  IMAGEN_CHECK(!pro || (p.ssq_a && p.pa && p.act_in == IMAGEN_ACT_SILU && (p.C2 == 0 || p.ssq_b)), "conv_pro: the prologue is ssq_a (+ ssq_b) statistics, pa (ps), SiLU");
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res && p.act_out == IMAGEN_ACT_NONE && !p.gca_part,
               "conv_pro: plain NHWC output only (no addend / residual / output activation / GlobalContext partials)");
  IMAGEN_CHECK(p.ldy % 8 == 0 && p.bsy % 8 == 0 && ((size_t)p.y & 15) == 0, "conv_pro: output rows must keep 16-byte alignment");
  IMAGEN_CHECK(!p.post_pa || (p.post_ps && !p.ssq_out), "conv_pro: post_pa needs post_ps and excludes ssq_out");
  static bool attr_done[16][4] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int key = (p.C2 ? 2 : 0) | (pro ? 1 : 0);
  const void* fn = key == 0 ? reinterpret_cast<const void*>(conv_pro_kernel<1, false>)
                 : key == 1 ? reinterpret_cast<const void*>(conv_pro_kernel<1, true>)
                 : key == 2 ? reinterpret_cast<const void*>(conv_pro_kernel<2, false>)
                            : reinterpret_cast<const void*>(conv_pro_kernel<2, true>);
  if (dev < 0 || dev >= 16 || !attr_done[dev][key]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("conv_pro: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < 16) attr_done[dev][key] = true;
  }
  switch (key) {
    case 0: return cp_launch<1, false>(p, s);
    case 1: return cp_launch<1, true>(p, s);
    case 2: return cp_launch<2, false>(p, s);
    default: return cp_launch<2, true>(p, s);
  }
}
