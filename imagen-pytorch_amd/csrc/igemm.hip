// igemm.hip — implicit-GEMM convolution / linear layer on CDNA4 MFMA (v_mfma_f32_32x32x16_f16) with the
// Imagen block prologue (ChanRMSNorm / LayerNorm statistics + per-(batch,channel) affine + SiLU) fused into
// the activation staging and bias / activation / gate*addend / residual / pixel-shuffle fused into the
// epilogue.  Replaces the ATen sequences cited at ImagenIgemmParams in include/imagen_hip.h.
//
// Data flow per workgroup (512 threads = 8 wave64: 4 producer + 4 consumer waves, a persistent walk over output tiles of TP pixels
// x BN output channels — see the comment above igemm_kernel):
//   HBM (NHWC fp16) --16B/lane loads, two phases ahead--> producer VGPRs --prologue in fp32--> LDS halo tile [pixels][8*G ch]
//   LDS --ds_read_b128 per (tap, 8-channel group)--> consumer MFMA B operand (pixels are the N/lane dimension)
//   packed weights (L2-resident, fragment order) --16B/lane loads, 6-12 steps ahead--> MFMA A operand (output channels = rows)
//   accumulators D[cout][pixel]: lane = pixel, 4 consecutive couts per register quad -> 8B NHWC stores, all after the last load.
// The k dimension runs over channel chunks of 8*G channels; inside a chunk over (tap, 8-channel group) pairs; two consecutive
// groups (lane>>5 selects) feed one K=16 MFMA.  A "phase" = (tile, chunk); one s_barrier per phase hands an LDS buffer from the
// producers to the consumers.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

struct TileCfg { int MI, NI, WM, WN, G; };
// 16-byte staging items per producer thread and phase, by tile pixels / k-chunk depth / kernel footprint: the producers hold
// TWO such register sets, so the count is kept as small as the instantiation's use allows (the host asks
// imagen_igemm_stage_slots() and only picks tile shapes that fit)
constexpr int kBiasLds = 1024;   // output channels whose bias the consumers keep in LDS (larger layers load it per channel quad)

constexpr int stage_slots(int TP, int G, int KSC) {
  if (KSC == 18 && G == 4) return TP == 64 ? 2 : TP == 128 ? 3 : 6;   // 3x3: 10x10 | 10x18 | 18x18 halo tiles
  if (KSC == 2 && G == 4) return TP / 64;                             // 1x1
  if (KSC == 8 && G == 4) return TP == 64 ? 4 : 6;                    // 2x2 stride-2 downsample: 4x the output pixels
  if (KSC == 6 && G == 4) return TP == 64 ? 3 : TP == 128 ? 4 : 5;    // 3x1 (the causal temporal conv of Imagen-Video over rows = frames): 10x8 (one frame: 3x64) | 18x8 | 18x16 halo tiles
  if (KSC == 4 && G == 8) return TP / 32;                             // 1x1, 64-channel chunks
  if (KSC == 8 && G == 16) return TP / 16;                            // 1x1, 128-channel chunks
  return 4;                                                           // generic k-loop (8-channel chunks, 15x15 cross-embed conv, ...)
}

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
//
// WAVE-SPECIALISED, PERSISTENT workgroups of 8 waves.  vmcnt retires loads in issue order, so a wave that streams weight
// fragments (L2 latency) and also has activation loads (HBM latency) in flight stalls on the slower stream at every weight
// wait — staging and matrix work cannot overlap inside one wave.  Hence two roles with independent counters:
//   waves 4-7 (producers): global loads of phase q+2 in flight, prologue math + LDS writes of phase q+1
//   waves 0-3 (consumers): weight ring + MFMAs of phase q out of LDS, then the epilogue of a finished tile
// where a "phase" is (output tile, channel chunk).  One s_barrier per phase hands an LDS buffer over.  The grid is sized to
// what the chip holds and every workgroup walks a strided list of tiles, so the pipeline also runs ACROSS tiles (layers with
// one or two chunks per tile — C_in = 32 / 64, most of the 256^2 / 128^2 levels — have no other overlap to offer): the
// consumers' epilogue stores fly while the producers already stage the next tile, and the weight ring wraps to the next
// tile's first steps kLookAhead steps before a tile ends.  Workgroup ids are dealt round-robin to the 8 XCDs, so each XCD
// (own L2) gets a contiguous range of tiles and neighbouring halos meet in the same L2.
struct TileCoord { int b, oy0, ox0, n0; };

__device__ __attribute__((aligned(16))) float kOnes8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
__device__ __attribute__((aligned(16))) float kZeros8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

// LDS hand-over between the roles: LDS traffic of this wave retired, then the workgroup barrier.  Deliberately NOT
// __syncthreads(): global loads stay in flight across it.
__device__ __forceinline__ void lds_barrier() { IMAGEN_LGKM0_BARRIER(); }


// GEN: the generic epilogue (output activation, gate*addend / residual, pixel-shuffle and fp32-NCHW stores).  The launcher picks the
// GEN = false instantiation for plain NHWC outputs (optionally with ssq_out or the post_pa output-side prologue): its epilogue is
// one branch-free block — the generic one tests act_out / out_mode / addend / res per element and quad, ~100 scalar branches
// per tile that were measured (s_memtime stamps) at 7k of a tile's 13k cycles on the 32-channel 256^2 layers.
template <int MI, int NI, int WM, int WN, int G, int KSC, bool GEN>
__global__ __launch_bounds__(512, (MI * NI <= 2 ? 4 : 2)) void igemm_kernel(const ImagenIgemmParams p) {
  static_assert(WM * WN == 4, "4 consumer waves per workgroup");
  constexpr int BN = 32 * NI * WN;
  constexpr int KC = Geo<G>::KC;
  constexpr int PS = Geo<G>::PS;
  constexpr int LOG2G = (G == 1) ? 0 : (G == 2) ? 1 : (G == 4) ? 2 : (G == 8) ? 3 : 4;
  constexpr int PXW = 32 * MI;  // pixels per consumer wave
  constexpr int kMaxItems = stage_slots(32 * MI * WM, G, KSC);

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const bool producer = tid >= 256;   // wave-uniform role
  const int rtid = tid & 255;         // thread index inside the role
  const unsigned warm = imagen_code_warm((unsigned)p.launcher_word << 8, tid, 512);   // (code size / 256, set by the launcher)

  // ---- the tile list of this workgroup
  const int tilesX = (p.OW + p.TW - 1) / p.TW;
  const int tilesY = (p.OH + p.TH - 1) / p.TH;
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int total_tiles = p.B * tilesY * tilesX * tilesN;
  int t_cursor, t_end, t_step;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    const int per_xcd = (total_tiles + 7) >> 3;
    t_cursor = xcd * per_xcd + lw;
    t_end = min((xcd + 1) * per_xcd, total_tiles);
    t_step = gridDim.x >> 3;
  } else {
    t_cursor = blockIdx.x;
    t_end = total_tiles;
    t_step = gridDim.x;
  }
  if (t_cursor >= t_end) return;
  auto decode = [&](int t) __attribute__((always_inline)) -> TileCoord {   // cout tile fastest: the cout tiles of a pixel tile run side by side
    TileCoord c;
    const int nt = t % tilesN;
    t /= tilesN;
    const int tx = t % tilesX;
    t /= tilesX;
    const int ty = t % tilesY;
    c.b = t / tilesY;
    c.oy0 = ty * p.TH;
    c.ox0 = tx * p.TW;
    c.n0 = nt * BN;
    return c;
  };

  const int ITW = (p.TW - 1) * p.stride + p.KW;
  const int ITH = (p.TH - 1) * p.stride + p.KH;
  const int IT = ITH * ITW;
  const int buf_bytes = IT * PS;
  const int ntap = p.KH * p.KW;
  const int KG = ntap * G;            // 8-channel groups per chunk
  const int KS = KSC > 0 ? KSC : (KG + 1) >> 1;  // K=16 MFMA steps per chunk (odd KG: last half-step is zero weights)
  const int NC = p.Cin_pad / KC;      // chunks
  const bool ssq_sync = (p.ssq_out != nullptr || p.post_pa != nullptr) && WN > 1;   // the epilogue's cross-wave reduction needs one extra workgroup barrier per tile

  if (producer) {
    // =========================================================================================== producers (waves 4-7)
    // Two register sets (A, B) alternate between phases, so the global loads of phase q+2 are already in flight while phase
    // q+1 is transformed and written: the load latency overlaps the prologue math instead of preceding it.  Every load is
    // unconditional (out-of-range items, absent statistics / affine arrays and phases past the end read a valid dummy
    // address): the code is straight-line, so the compiler's vmcnt waits are exact — waiting for the older set leaves the
    // younger set's loads in flight.
    const int items = IT << LOG2G;
    const float inv_itw = 1.0f / (float)ITW;
    const f16* x1 = reinterpret_cast<const f16*>(p.x1);
    const f16* x2 = reinterpret_cast<const f16*>(p.x2);
    const float* dummy_f = reinterpret_cast<const float*>(p.w);   // >= 32 readable, 16-byte aligned bytes
    // scalar copies: a per-lane select between two kernel arguments is otherwise compiled into a per-lane LOAD from the argument
    // segment (select of addresses) with a full vmcnt(0) drain in the middle of the staging code
    const int ld1_s = __builtin_amdgcn_readfirstlane(p.ld1), ld2_s = __builtin_amdgcn_readfirstlane(p.ld2);
    // per-chunk prologue affine of THIS thread's 8-channel group: every item of a thread has the same group (256 % G == 0) and
    // the tile lies in one batch row, so the 8 + 8 floats are loaded once per phase
    const int my_cg = rtid & (G - 1);
    // tile-independent geometry of this thread's staging items: item `it` is halo pixel pix0 + it * (256 / G); its (iy, ix) is
    // walked incrementally from item 0's (two registers instead of one per item — the two staging sets need the rest)
    const int pix0 = rtid >> LOG2G;
    const int iy_first = (int)(((float)pix0 + 0.5f) * inv_itw);
    const int ix_first = pix0 - iy_first * ITW;
    const int step_y = (256 >> LOG2G) / ITW, step_x = (256 >> LOG2G) % ITW;
    const float* q1_base = p.rs ? p.rs : (p.ssq_a ? p.ssq_a : dummy_f);            // rs | ssq_a
    const int q1_on = (p.rs || p.ssq_a) ? 1 : 0;
    const float* q2_base = p.mu ? p.mu : ((!p.rs && p.ssq_b) ? p.ssq_b : dummy_f);  // mu | ssq_b
    const int q2_on = (p.mu || (!p.rs && p.ssq_b)) ? 1 : 0;
    const float* pa_base = p.pa ? p.pa : kOnes8;    // absent affine: neutral constants instead of per-element selects
    const float* ps_base = p.ps ? p.ps : kZeros8;
    const int pa_on = p.pa ? 1 : 0, ps_on = p.ps ? 1 : 0;
    const bool raw_copy = !p.pa && !p.ps && !p.rs && !p.ssq_a && !p.mu && p.act_in == IMAGEN_ACT_NONE;

    struct StageSet {
      uint4 raw[kMaxItems];
      float q1[kMaxItems], q2[kMaxItems];   // statistics as loaded; the arithmetic happens at write time
      unsigned mask;
      int b, chunk;                         // phase identity (for the affine load that follows one phase later)
    };
    // (Three sets — every load two phase periods in flight, the affine requested an iteration ahead in two register copies — were
    // measured in round 3's call H: +1.4 % per step pair.  The phase period is not set by the load latency.)
    StageSet A, B;
    float4 st_a0, st_a1, st_s0, st_s1;      // affine of the phase about to be written (shared by both sets)

    // phase cursor: (tl, chunk) is the phase whose loads are issued next; past the end it stays on the last phase (harmless re-loads)
    TileCoord tl = decode(t_cursor);
    int chunk = 0;
    const int tiles_mine = (t_end - t_cursor + t_step - 1) / t_step;
    const int n_phases = tiles_mine * NC;
    auto advance = [&]() __attribute__((always_inline)) {
      if (chunk + 1 < NC) ++chunk;
      else if (t_cursor + t_step < t_end) {
        chunk = 0;
        t_cursor += t_step;
        tl = decode(t_cursor);
      }
    };

    // (Round 3, calls J / K: with no producer work, no MFMAs and no stores a res_conv launch keeps 60-67 % of its time — the per-tile
    // skeleton of barriers + the generic epilogue.  Having the producers touch the next tile's gate * addend rows into L2 changed
    // nothing (9.397 vs 9.377 ms per step pair): those operands are L2 hits already; what is left is the epilogue's own instruction
    // stream and its one dependent round trip per tile.)
    char* lds_dummy = smem + 2 * buf_bytes + (4 * PXW + kBiasLds) * (int)sizeof(float);
    auto load_set = [&](StageSet& S) __attribute__((always_inline)) {
      S.mask = 0;
      S.b = tl.b;
      S.chunk = chunk;
      const int b = tl.b;
      const int iy0 = tl.oy0 * p.stride - p.pad, ix0 = tl.ox0 * p.stride - (p.pad_x1 ? p.pad_x1 - 1 : p.pad);
      const int cc = chunk * KC + my_cg * 8;
      const bool from1 = cc < p.C1;
      const bool chan_ok = from1 || (cc - p.C1 < p.C2);
      const f16* base = from1 ? x1 + (size_t)b * p.bs1 + cc : x2 + (size_t)b * p.bs2 + (cc - p.C1);
      const int ld = from1 ? ld1_s : ld2_s;
      const int sp0 = b * (p.H * p.W);
      int iy = iy_first, ix = ix_first;
      static_for<kMaxItems>([&](auto ic) __attribute__((always_inline)) {
        constexpr int it = decltype(ic)::value;
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool ok = chan_ok && iy < ITH && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;   // iy >= ITH: slot beyond the tile
        ix += step_x;
        iy += step_y;
        if (ix >= ITW) { ix -= ITW; ++iy; }
        const int gp = ok ? gy * p.W + gx : 0;
        const f16* src = ok ? base + (size_t)gp * ld : x1;
        S.raw[it] = *reinterpret_cast<const uint4*>(src);
        if (ok) S.mask |= 1u << it;
        const int sp = sp0 + gp;
        S.q1[it] = q1_base[sp * q1_on];
        S.q2[it] = q2_base[sp * q2_on];
      });
    };
    auto load_affine = [&](const StageSet& S) __attribute__((always_inline)) {
      const int o = S.b * p.pstride + S.chunk * KC + my_cg * 8;
      const float4* qa = reinterpret_cast<const float4*>(pa_base + o * pa_on);
      const float4* qs = reinterpret_cast<const float4*>(ps_base + o * ps_on);
      st_a0 = qa[0];
      st_a1 = qa[1];
      st_s0 = qs[0];
      st_s1 = qs[1];
    };

    // transform + LDS write of a staged set
    // transform + LDS write of a staged set — WITHOUT control flow: one producer wave is a chain of dependent VALU ops (measured
    // ~10 cycles per instruction), so the items of a set must be interleaved by the scheduler, which stops at every branch.
    // Mode choices (rs | ssq | none, mu, activation) are uniform selects; out-of-image items are zeroed by a select; slots beyond
    // the tile write to a 16-byte dummy behind the epilogue scratch.
    const bool use_rs = p.rs != nullptr, use_ssq = !use_rs && p.ssq_a != nullptr, use_ssqb = use_ssq && p.ssq_b != nullptr;
    const bool use_mu = p.mu != nullptr, use_silu = p.act_in == IMAGEN_ACT_SILU;
    auto write_set = [&](const StageSet& S, char* buf) __attribute__((always_inline)) {
      if (raw_copy) {   // input already activated by its producer (post_pa epilogue) or a plain GEMM operand: zero-fill only
        static_for<kMaxItems>([&](auto ic) __attribute__((always_inline)) {
          constexpr int it = decltype(ic)::value;
          const int idx = rtid + it * 256;
          const uint4 v = (S.mask & (1u << it)) ? S.raw[it] : make_uint4(0, 0, 0, 0);
          char* dst = idx < items ? buf + (idx >> LOG2G) * PS + my_cg * 16 : lds_dummy;
          *reinterpret_cast<uint4*>(dst) = v;
        });
        return;
      }
      const float a[8] = {st_a0.x, st_a0.y, st_a0.z, st_a0.w, st_a1.x, st_a1.y, st_a1.z, st_a1.w};
      const float s[8] = {st_s0.x, st_s0.y, st_s0.z, st_s0.w, st_s1.x, st_s1.y, st_s1.z, st_s1.w};
      static_for<kMaxItems>([&](auto ic) __attribute__((always_inline)) {
        constexpr int it = decltype(ic)::value;
        const int idx = rtid + it * 256;
        const f16x8 in = *reinterpret_cast<const f16x8*>(&S.raw[it]);
        // ChanRMSNorm statistics straight from the producers' per-pixel sums of squares: 1 / max(sqrt(q), 1e-12) (ip.py:328)
        const float q = S.q1[it] + (use_ssqb ? p.ssq_wb * S.q2[it] : 0.0f);
        const float rq = __builtin_amdgcn_rsqf(fmaxf(q, 1e-24f));
        const float rs = use_rs ? S.q1[it] : (use_ssq ? rq : 1.0f);
        const float mu = use_mu ? S.q2[it] : 0.0f;
        const bool ok = (S.mask & (1u << it)) != 0;
        // stage-wise over the 8 channels (all affines, then all exp2, then all rcp ...): a source order in which neighbouring
        // instructions are independent — the quarter-rate exp / rcp of one element otherwise sit back to back behind s_nops
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
        ow = ok ? ow : make_uint4(0, 0, 0, 0);   // out-of-image pixels / channels: zero padding (on the packed words)
        char* dst = idx < items ? buf + (idx >> LOG2G) * PS + my_cg * 16 : lds_dummy;
        *reinterpret_cast<uint4*>(dst) = ow;
      });
    };

    char* buf0 = smem;
    char* buf1 = smem + buf_bytes;
    int c_done = 0;   // chunk index of the phase the consumers are working on
    auto phase_end = [&]() __attribute__((always_inline)) {
      lds_barrier();
      if (++c_done == NC) {
        c_done = 0;
        if (ssq_sync) lds_barrier();   // pairs with the consumers' epilogue reduction
      }
    };
    // (the cursor is advanced AFTER the transform: the scalar control flow of advance()/decode() between a set's loads and the
    // other set's first use makes the compiler's vmcnt bookkeeping fall back to a full drain)
    load_set(A);            // phase 0
    load_affine(A);
    advance();
    load_set(B);            // phase 1
    write_set(A, buf0);     // waits for A and the affine; B stays in flight
    load_affine(B);
    advance();
    lds_barrier();          // phase 0 is in buffer 0
    imagen_code_warm_sink(warm);
    for (int q = 0; q < n_phases; q += 2) {
      // consumers: phase q out of buf0
      load_set(A);          // phase q+2
      write_set(B, buf1);   // phase q+1
      load_affine(A);
      advance();
      phase_end();
      if (q + 1 >= n_phases) break;
      // consumers: phase q+1 out of buf1
      load_set(B);          // phase q+3
      write_set(A, buf0);   // phase q+2
      load_affine(B);
      advance();
      phase_end();
    }
    return;
  }

  // ============================================================================================= consumers (waves 0-3)
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  float* ep_red = reinterpret_cast<float*>(smem + 2 * buf_bytes);   // [4 waves][PXW] epilogue scratch, disjoint from the staging buffers

  // ---- per-lane output pixel coordinates inside the tile (lane = pixel in the MFMA N dimension)
  int a_base[MI];
  int pix_y[MI], pix_x[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int tp = (wm * MI + mi) * 32 + l31;
    const int py = tp / p.TW, px = tp - py * p.TW;
    pix_y[mi] = py;
    pix_x[mi] = px;
    a_base[mi] = ((py * p.stride) * ITW + px * p.stride) * PS;
  }

  // ---- weights: packed [chunk*KGP + kg][Cout_pad] x (8 halves); this lane's rows start at wlane, a tile's at its n0
  const f16x8* wlane = reinterpret_cast<const f16x8*>(p.w) + (size_t)half * p.Cout_pad + wn * (NI * 32) + l31;
  const int wstep = 2 * p.Cout_pad;  // one K=16 step

  f32x16 acc[NI][MI];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.0f;
  };
  zero_acc();

  // ---- weight fragment pipeline, continuous over chunks AND tiles
  // wq[j] holds the fragments of K=16 step (gstep + j); w_ofs is the offset (in fragments) of step (gstep + kLookAhead)
  // look-ahead depth of the weight ring: a wave consumes one fragment per MI MFMAs, so the single-MFMA-per-step tilings (small
  // feature maps: few workgroups, each streaming its whole weight slice) need a deeper ring to cover the L2 round trip
  // ring depths measured in the model (A/B builds of round 2/3: 8/6 beat 12/10 and 16/12 by 1-2 %, 8/8 ties)
  constexpr int kWantAhead = (MI * NI == 1) ? 8 : 6;
  constexpr int kLookAhead = KSC == 0 ? 1 : (KSC >= kWantAhead ? kWantAhead : KSC);
  f16x8 wq[kLookAhead][NI];
  int w_ofs = 0;

  // (the ring runs kLookAhead steps past the tile's last step: the packed buffer carries a zero tail for that)
  auto compute = [&](const char* buf) __attribute__((always_inline)) {
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
      // prefetch the weight fragments kLookAhead steps ahead (L2 latency ~ 2-3 MFMA groups); uniform wrap at the tile end
      f16x8 wnew[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wnew[ni] = wlane[w_ofs + ni * 32];
      w_ofs += wstep;
      if (ks + 1 < KS) {
        const int a1 = next_aoff();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) anext[mi] = *reinterpret_cast<const f16x8*>(buf + a_base[mi] + a1);
      }
      // pin the software pipeline: under register pressure the scheduler otherwise sinks the look-ahead loads down to their
      // first use (load, wait, MFMA — every step then pays a full L2 round trip)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[0][ni], afrag[mi], acc[ni][mi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
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
      static_for<KSC>([&](auto ic) __attribute__((always_inline)) { do_step(decltype(ic)::value); });
    } else {
      for (int ks = 0; ks < KS; ++ks) do_step(ks);
    }
  };

  // ---- epilogue of one finished tile: lane = pixel; register quad q holds couts 8q + 4*half + {0..3} of each 32-cout fragment
  const f16* addend = reinterpret_cast<const f16*>(p.addend);
  const f16* res = reinterpret_cast<const f16*>(p.res);
  // Under load a dependent global load costs ~2k cycles (measured with s_memtime stamps: the 18-step k-loop of a tile took 5k
  // cycles, four bias-load -> use -> store rounds of the old epilogue 14k), so the bias is staged in LDS once per workgroup:
  // the epilogue of a plain conv then contains no global load at all.
  float* ep_bias = ep_red + 4 * PXW;   // [kBiasLds] floats
  const bool bias_lds = p.bias != nullptr && p.Cout_pad <= kBiasLds;
  if (bias_lds)
    for (int i = rtid; i < p.Cout_pad; i += 256) ep_bias[i] = p.bias[i];   // visible after the phase-0 barrier below
  // The weight ring is dead during the arithmetic of the epilogue (its registers go to the packed outputs) and is re-primed
  // with the NEXT tile's first steps right before the stores: those loads are older than the stores, so the next tile's first
  // weight wait does not cover a store either.
  auto prime_weights = [&](int n0) __attribute__((always_inline)) {
    w_ofs = n0;
#pragma unroll
    for (int j = 0; j < kLookAhead; ++j) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wq[j][ni] = wlane[w_ofs + ni * 32];
      w_ofs += wstep;
    }
  };
  auto epilogue = [&](const TileCoord& tc, int n0_next) __attribute__((always_inline)) {
    const int b = tc.b, n0 = tc.n0;
    // The per-lane constants of the epilogue are rematerialised HERE instead of being kept live across the k loop: at the 128-VGPR
    // budget the compiler otherwise hoists them out of the tile loop and spills them, and every reload in the epilogue is a scratch_load
    // followed by s_waitcnt vmcnt(0), i.e. a full drain of the re-primed weight ring (19 per tile in <2,1,1,4,4,18,false>, 46 in its
    // GEN twin — none this way; profiles/r02_static_spills_igemm_*.txt, measured -1 % per step pair in round 3's call A).  The empty
    // asm makes the thread id opaque inside the tile loop, so nothing derived from it can be hoisted; the names shadow the outer ones.
    int tid_e = threadIdx.x;
    IMAGEN_OPAQUE(tid_e);
    const int half = (tid_e >> 5) & 1, l31 = tid_e & 31;
    const int wave_e = __builtin_amdgcn_readfirstlane(tid_e >> 6);
    const int wm = wave_e / WN, wn = wave_e % WN;
    int pix_y[MI], pix_x[MI];
    {
      const int tw_sh = __builtin_ctz(p.TW);   // launcher-checked: TW is a power of two (every host-side tile shape is)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int tp = (wm * MI + mi) * 32 + l31;
        pix_y[mi] = tp >> tw_sh;
        pix_x[mi] = tp & (p.TW - 1);
      }
    }
    if (p.post_pa) {
      // ---- output-side Block prologue: two passes over the accumulators (norm over all Cout of the pixel, then activate + store)
      float tot[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) tot[mi] = 0.0f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
          if (co >= p.Cout) continue;
          float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias_lds) bq = *reinterpret_cast<const float4*>(ep_bias + co);
          else if (p.bias) bq = *reinterpret_cast<const float4*>(p.bias + co);
          const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = acc[ni][mi][4 * q + e] + bb[e];
              acc[ni][mi][4 * q + e] = v;   // keep h for the second pass
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
        lds_barrier();   // whole workgroup (the producers execute the matching barrier)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          float t = 0.0f;
#pragma unroll
          for (int w = 0; w < WN; ++w) t += ep_red[(wm * WN + w) * PXW + mi * 32 + l31];
          tot[mi] = t;
        }
      }
      float rsn[MI];
      int opx[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        rsn[mi] = __builtin_amdgcn_rsqf(fmaxf(tot[mi], 1e-24f));
        const int oy = tc.oy0 + pix_y[mi], ox = tc.ox0 + pix_x[mi];
        opx[mi] = (oy < p.OH && ox < p.OW) ? oy * p.OW + ox : -1;
      }
      f16* y = reinterpret_cast<f16*>(p.y);
      f16x4 pout[NI][4][MI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
          if (co >= p.Cout) continue;
          const float4 pa = *reinterpret_cast<const float4*>(p.post_pa + (size_t)b * p.post_pstride + co);
          const float4 ps = *reinterpret_cast<const float4*>(p.post_ps + (size_t)b * p.post_pstride + co);
          const float pav[4] = {pa.x, pa.y, pa.z, pa.w}, psv[4] = {ps.x, ps.y, ps.z, ps.w};
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            if (opx[mi] < 0) continue;
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16)silu_f(acc[ni][mi][4 * q + e] * rsn[mi] * pav[e] + psv[e]);
            pout[ni][q][mi] = o;
          }
        }
      // stores only after the last load (see the note at the store loop of the plain path below)
      prime_weights(n0_next);
      if ((p.Cout & 7) == 0) {   // 16-byte pieces (imagen_pair_quads: all lanes take part in the exchange)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 16 * half;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              const imagen_u32x4 v = imagen_pair_quads(pout[ni][q][mi], pout[ni][q + 2][mi]);
              if (co < p.Cout && opx[mi] >= 0) *reinterpret_cast<imagen_u32x4*>(y + (size_t)b * p.bsy + (size_t)opx[mi] * p.ldy + co) = v;
            }
          }
        return;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
          if (co >= p.Cout) continue;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            if (opx[mi] >= 0) *reinterpret_cast<f16x4*>(y + (size_t)b * p.bsy + (size_t)opx[mi] * p.ldy + co) = pout[ni][q][mi];
        }
      return;
    }
    float ssq_px[MI];  // per-pixel sum of squares of this wave's stored channels (for the consumer's ChanRMSNorm)
    int op[MI];        // output pixel index, -1: outside the image
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ssq_px[mi] = 0.0f;
      const int oy = tc.oy0 + pix_y[mi], ox = tc.ox0 + pix_x[mi];
      op[mi] = (oy < p.OH && ox < p.OW) ? oy * p.OW + ox : -1;
    }
    if constexpr (!GEN) {   // plain NHWC output, bias in LDS (launcher-checked): one branch-free block
      f16x4 pv[NI][4][MI];   // packed outputs (they take over the accumulators' registers as those die)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_barrier(0);   // one channel quad at a time (register footprint)
          const int co = (n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half) & (kBiasLds - 1);   // in range for the LDS read; padded couts are never stored
          const float4 bq = bias_lds ? *reinterpret_cast<const float4*>(ep_bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[e] = (f16)(acc[ni][mi][4 * q + e] + bb[e]);
              const float r = (float)o[e];
              ssq_px[mi] += r * r;
            }
            pv[ni][q][mi] = o;
          }
        }
      prime_weights(n0_next);
      f16* y = reinterpret_cast<f16*>(p.y) + (size_t)b * p.bsy;
      if ((p.Cout & 7) == 0) {   // 16-byte pieces (imagen_pair_quads: all lanes take part in the exchange)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 16 * half;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              const imagen_u32x4 v = imagen_pair_quads(pv[ni][q][mi], pv[ni][q + 2][mi]);
              if (co < p.Cout && op[mi] >= 0) *reinterpret_cast<imagen_u32x4*>(y + (size_t)op[mi] * p.ldy + co) = v;
            }
          }
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              if (co < p.Cout && op[mi] >= 0) *reinterpret_cast<f16x4*>(y + (size_t)op[mi] * p.ldy + co) = pv[ni][q][mi];
          }
      }
    } else {
    f16x4 outv[NI][4][MI];   // packed outputs (they take over the accumulators' registers as those die)
    // gate + (addend | residual) quads of the channel-quad groups
    struct EpiOps { float4 g; f16x4 ar[MI]; };
    auto load_group = [&](int g, EpiOps& o) __attribute__((always_inline)) {
      const int co = n0 + (wn * NI + (g >> 2)) * 32 + 8 * (g & 3) + 4 * half;
      if (co >= p.Cout) return;
      if (addend) {
        o.g = *reinterpret_cast<const float4*>(p.gate + (size_t)b * p.gate_stride + co);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          o.ar[mi] = *reinterpret_cast<const f16x4*>(addend + (size_t)b * p.bs_add + (size_t)max(op[mi], 0) * p.ld_add + co);
      } else if (res) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          o.ar[mi] = *reinterpret_cast<const f16x4*>(res + (size_t)b * p.bs_res + (size_t)max(op[mi], 0) * p.ld_res + co);
      }
    };
    // ALL groups' operands are requested before the first one is used (4 + 2 MI registers per group): one memory round trip per
    // tile instead of one per channel quad — in the graph the res_conv launches (a 1x1 GEMM of 1-3 k-chunks behind this epilogue)
    // cost 1.18 ms of a 10.5 ms step pair with the one-group-ahead form (round-2 probe ablate_step.sh)
    EpiOps E[4 * NI];
    // 16-byte pieces for the NHWC operands and outputs when the channel counts allow it (imagen_pair_quads: a lane's quads q and q + 2
    // against its half-wave partner's): half the memory instructions of the 8-byte form, each a full 16 bytes per lane
    const f16* eop = addend ? addend + (size_t)b * p.bs_add : (res ? res + (size_t)b * p.bs_res : nullptr);
    const int eld = addend ? p.ld_add : p.ld_res;
    const bool wide = p.out_mode == IMAGEN_OUT_NHWC && ((p.Cout | p.ldy | eld) & 7) == 0 && (p.bsy & 7) == 0 &&
                      (((size_t)p.y | (size_t)eop) & 15) == 0;
    // pixel-shuffle outputs in 16-byte pieces too: 8 consecutive packed couts (s1, s2, c) share their sub-pixel when Cout / 4 is a multiple of 8
    const bool wide_ps = p.out_mode == IMAGEN_OUT_PIXEL_SHUFFLE && (p.Cout & 31) == 0 && (p.ldy & 7) == 0 && (p.bsy & 7) == 0 &&
                         ((size_t)p.y & 15) == 0;
    if (wide && eop) {
      imagen_u32x4 raw[NI][2][MI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int cx = n0 + (wn * NI + ni) * 32 + 8 * qp + 16 * half;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            raw[ni][qp][mi] = imagen_u32x4{0u, 0u, 0u, 0u};
            if (cx < p.Cout) raw[ni][qp][mi] = *reinterpret_cast<const imagen_u32x4*>(eop + (size_t)max(op[mi], 0) * eld + cx);
          }
        }
      if (addend) {
#pragma unroll
        for (int g = 0; g < 4 * NI; ++g) {
          const int co = n0 + (wn * NI + (g >> 2)) * 32 + 8 * (g & 3) + 4 * half;
          if (co < p.Cout) E[g].g = *reinterpret_cast<const float4*>(p.gate + (size_t)b * p.gate_stride + co);
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) imagen_unpair_quads(raw[ni][qp][mi], E[ni * 4 + qp].ar[mi], E[ni * 4 + qp + 2].ar[mi]);
    } else {
      static_for<4 * NI>([&](auto gc) __attribute__((always_inline)) { load_group(decltype(gc)::value, E[decltype(gc)::value]); });
    }
    static_for<4 * NI>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      constexpr int ni = g >> 2, q = g & 3;
      EpiOps& cur = E[g];
      __builtin_amdgcn_sched_barrier(0);   // one group at a time: bounds the register footprint of the arithmetic
      const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
      if (co < p.Cout) {
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias_lds) bq = *reinterpret_cast<const float4*>(ep_bias + co);
        else if (p.bias) bq = *reinterpret_cast<const float4*>(p.bias + co);   // > kBiasLds couts: padded to Cout_pad by the host
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          outv[ni][q][mi] = f16x4{};   // defined for the lane exchange of the 16-byte stores even where nothing is stored
          if (op[mi] < 0) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * q + e] + bb[e];
          if (p.act_out == IMAGEN_ACT_SILU) {        // one scalar branch per quad, not per element
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
            const f16x4 ad = cur.ar[mi];
            v[0] += (float)ad[0] * cur.g.x; v[1] += (float)ad[1] * cur.g.y; v[2] += (float)ad[2] * cur.g.z; v[3] += (float)ad[3] * cur.g.w;
          } else if (res) {
            const f16x4 rr = cur.ar[mi];
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
          outv[ni][q][mi] = o;   // stored below, after the last load of this epilogue
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) outv[ni][q][mi] = f16x4{};
      }
    });
    // all stores together: vmcnt retires in issue order and counts stores, so a wait on a load issued AFTER a store also waits
    // for that store's acknowledgement (~1.5k cycles under load; a load -> use -> store loop per channel quad was measured at
    // 14k cycles per tile).  With every store behind the last load, no wait in the epilogue covers one.
    prime_weights(n0_next);
    if (wide) {
      f16* y = reinterpret_cast<f16*>(p.y) + (size_t)b * p.bsy;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int cx = n0 + (wn * NI + ni) * 32 + 8 * qp + 16 * half;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const imagen_u32x4 v = imagen_pair_quads(outv[ni][qp][mi], outv[ni][qp + 2][mi]);
            if (cx < p.Cout && op[mi] >= 0) *reinterpret_cast<imagen_u32x4*>(y + (size_t)op[mi] * p.ldy + cx) = v;
          }
        }
    } else if (wide_ps) {
      f16* y = reinterpret_cast<f16*>(p.y) + (size_t)b * p.bsy;
      const int Cq = p.Cout >> 2;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int cx = n0 + (wn * NI + ni) * 32 + 8 * qp + 16 * half;     // 8 consecutive packed couts of one sub-pixel
          const int sub = cx / Cq, c = cx - sub * Cq;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const imagen_u32x4 v = imagen_pair_quads(outv[ni][qp][mi], outv[ni][qp + 2][mi]);
            const int oy = tc.oy0 + pix_y[mi], ox = tc.ox0 + pix_x[mi];
            const int yy = 2 * oy + (sub >> 1), xx = 2 * ox + (sub & 1);
            if (cx < p.Cout && op[mi] >= 0) *reinterpret_cast<imagen_u32x4*>(y + ((size_t)yy * (2 * p.OW) + xx) * p.ldy + c) = v;
          }
        }
    } else if (p.out_mode != IMAGEN_OUT_NCHW_F32) {
      f16* y = reinterpret_cast<f16*>(p.y);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = n0 + (wn * NI + ni) * 32 + 8 * q + 4 * half;
          if (co >= p.Cout) continue;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            if (op[mi] < 0) continue;
            if (p.out_mode == IMAGEN_OUT_PIXEL_SHUFFLE) {
              // output channels are packed (s1, s2, c): cout = (2*s1 + s2) * Cq + c   (PixelShuffle(2), ip.py:616)
              const int Cq = p.Cout >> 2;
              const int sub = co / Cq, c = co - sub * Cq;
              const int oy = tc.oy0 + pix_y[mi], ox = tc.ox0 + pix_x[mi];
              const int yy = 2 * oy + (sub >> 1), xx = 2 * ox + (sub & 1);
              *reinterpret_cast<f16x4*>(y + (size_t)b * p.bsy + ((size_t)yy * (2 * p.OW) + xx) * p.ldy + c) = outv[ni][q][mi];
            } else {
              *reinterpret_cast<f16x4*>(y + (size_t)b * p.bsy + (size_t)op[mi] * p.ldy + co) = outv[ni][q][mi];
            }
          }
        }
    }
    }   // generic path
    // optional: emit the per-pixel sum of squares (launcher guarantees one workgroup covers all Cout: tilesN == 1)
    if (p.ssq_out) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) ssq_px[mi] += __shfl_xor(ssq_px[mi], 32);  // both lane halves hold disjoint channel quads
      if (WN == 1) {
        if (half == 0) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int oy = tc.oy0 + pix_y[mi], ox = tc.ox0 + pix_x[mi];
            if (oy < p.OH && ox < p.OW) p.ssq_out[(size_t)b * (p.OH * p.OW) + oy * p.OW + ox] = ssq_px[mi];
          }
        }
      } else {
        if (half == 0) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) ep_red[(wm * WN + wn) * PXW + mi * 32 + l31] = ssq_px[mi];
        }
        lds_barrier();   // whole workgroup (the producers execute the matching barrier); ep_red is next written a phase barrier later
        if (wn == 0 && half == 0) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < WN; ++w) tot += ep_red[(wm * WN + w) * PXW + mi * 32 + l31];
            const int oy = tc.oy0 + pix_y[mi], ox = tc.ox0 + pix_x[mi];
            if (oy < p.OH && ox < p.OW) p.ssq_out[(size_t)b * (p.OH * p.OW) + oy * p.OW + ox] = tot;
          }
        }
      }
    }
  };

  // ---- main loop: tiles x chunks, one hand-over barrier per phase
  TileCoord tc = decode(t_cursor);
  prime_weights(tc.n0);
  // Epilogue operands of the first tile (residual | gate * addend): touched NOW, one dword per pixel row of this wave's channel
  // fragment(s), so that the loads of the epilogue itself — dependent on nothing but issued after the last k step — find the lines
  // (and the page translations) in place: in the denoiser step those loads were measured at 7-12k cycles per tile
  // (round-2 probe insitu_trace.py: to_time_cond, ff.lin2, res_conv), most of a small GEMM's run time.
  unsigned warm_ep = 0;
  if constexpr (GEN) {
    const f16* eop = addend ? addend + (size_t)tc.b * p.bs_add : (res ? res + (size_t)tc.b * p.bs_res : nullptr);
    const int eld = addend ? p.ld_add : p.ld_res;
    if (eop) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int oy = tc.oy0 + pix_y[mi], ox = tc.ox0 + pix_x[mi];
        const int opx = (oy < p.OH && ox < p.OW) ? oy * p.OW + ox : 0;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int co = min(tc.n0 + (wn * NI + ni) * 32 + 4 * half, p.Cout - 4);
          warm_ep ^= *reinterpret_cast<const volatile unsigned*>(eop + (size_t)opx * eld + co);
        }
      }
      if (addend) warm_ep ^= *reinterpret_cast<const volatile unsigned*>(p.gate + (size_t)tc.b * p.gate_stride + min(tc.n0 + wn * NI * 32 + (lane & 31), p.Cout - 1));
    }
  }
  lds_barrier();   // phase 0 staged
  imagen_code_warm_sink(warm);
  imagen_code_warm_sink(warm_ep);
  int cur = 0;
  while (true) {
    const int t_next = t_cursor + t_step;
    const int n0_next = t_next < t_end ? decode(t_next).n0 : tc.n0;
    for (int chunk = 0; chunk < NC; ++chunk) {
      compute(smem + cur * buf_bytes);
      lds_barrier();   // done with buf[cur]; the producers have filled buf[cur^1]
      cur ^= 1;
    }
    epilogue(tc, n0_next);
    if (t_next >= t_end) break;
    zero_acc();
    t_cursor = t_next;
    tc = decode(t_cursor);
  }
}

inline int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

template <int MI, int NI, int WM, int WN, int G, int KSC, bool GEN>
int launch_gen(const ImagenIgemmParams& p, hipStream_t s);

template <int MI, int NI, int WM, int WN, int G, int KSC>
int launch_ksc(const ImagenIgemmParams& p, hipStream_t s) {
  const bool plain = p.act_out == IMAGEN_ACT_NONE && p.out_mode == IMAGEN_OUT_NHWC && !p.addend && !p.res &&
                     (p.bias == nullptr || p.Cout_pad <= kBiasLds);
  return plain ? launch_gen<MI, NI, WM, WN, G, KSC, false>(p, s) : launch_gen<MI, NI, WM, WN, G, KSC, true>(p, s);
}

template <int MI, int NI, int WM, int WN, int G, int KSC, bool GEN>
int launch_gen(const ImagenIgemmParams& p, hipStream_t s) {
  constexpr int TP = 32 * MI * WM, BN = 32 * NI * WN;
  const int ITW = (p.TW - 1) * p.stride + p.KW, ITH = (p.TH - 1) * p.stride + p.KH;
  const int IT = ITH * ITW;
  IMAGEN_CHECK(p.TH * p.TW == TP, "igemm: tile %dx%d does not match cfg %d (%d pixels)", p.TH, p.TW, p.cfg, TP);
  IMAGEN_CHECK(IT * G <= stage_slots(TP, G, KSC) * 256, "igemm: halo tile too large (%d px x %d groups > %d staging slots)", IT, G,
               stage_slots(TP, G, KSC));
  IMAGEN_CHECK(p.Cout_pad % BN == 0, "igemm: Cout_pad %d not a multiple of %d", p.Cout_pad, BN);
  IMAGEN_CHECK(p.Cin_pad % (8 * G) == 0, "igemm: Cin_pad %d not a multiple of %d", p.Cin_pad, 8 * G);
  IMAGEN_CHECK(p.C1 % 8 == 0 && p.C2 % 8 == 0 && p.ld1 % 8 == 0 && (p.x2 == nullptr || p.ld2 % 8 == 0),
               "igemm: channel counts / strides must be multiples of 8 (C1=%d C2=%d ld1=%d ld2=%d)", p.C1, p.C2, p.ld1, p.ld2);
  IMAGEN_CHECK(p.out_mode == IMAGEN_OUT_NCHW_F32 || p.Cout % 4 == 0, "igemm: Cout %d must be a multiple of 4", p.Cout);
  IMAGEN_CHECK(p.out_mode != IMAGEN_OUT_PIXEL_SHUFFLE || p.Cout % 16 == 0, "igemm: pixel-shuffle needs Cout %% 16 == 0");
  IMAGEN_CHECK(!p.post_pa || (p.post_ps && p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN && !p.addend && !p.res && !p.ssq_out &&
                              p.act_out == IMAGEN_ACT_NONE && p.Cout % 4 == 0),
               "igemm: post_pa needs post_ps, a plain NHWC output and one workgroup covering all %d output channels (tile has %d)", p.Cout, BN);
  IMAGEN_CHECK(!(p.addend && p.res), "igemm: addend and residual are mutually exclusive");
  IMAGEN_CHECK(!p.ssq_out || (p.out_mode == IMAGEN_OUT_NHWC && p.Cout <= BN),
               "igemm: ssq_out needs NHWC output and one workgroup covering all %d output channels (tile has %d)", p.Cout, BN);
  const size_t lds = (size_t)2 * IT * Geo<G>::PS + (size_t)(4 * 32 * MI + kBiasLds) * sizeof(float) + 16;   // staging double buffer + epilogue scratch + bias + dummy
  IMAGEN_CHECK(lds <= 160 * 1024, "igemm: LDS tile %zu bytes too large", lds);
  auto kern = igemm_kernel<MI, NI, WM, WN, G, KSC, GEN>;
  static bool attr_done[16] = {};   // the attribute is per DEVICE (a process may sample on several GPUs)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { imagen_set_error("igemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  // persistent grid: what the chip holds at once (register- and LDS-limited workgroups per CU), evened out over the rounds
  const int tilesX = (p.OW + p.TW - 1) / p.TW, tilesY = (p.OH + p.TH - 1) / p.TH;
  const int total = p.B * tilesX * tilesY * ((p.Cout + BN - 1) / BN);
  static size_t occ_lds = 0;
  static int occ_blocks = 0;
  if (occ_blocks == 0 || occ_lds != lds) {   // resident workgroups per CU of THIS instantiation at this LDS size
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 512, lds) != hipSuccess || nb < 1) nb = 1;
    occ_blocks = nb;
    occ_lds = lds;
  }
  // (persistent grids capped at ONE workgroup per CU, to leave registers / LDS for the other lanes' kernels, were measured in round 3's
  // call C: sequential +7 %, lanes throughput unchanged — not kept)
  const int resident = std::max(8, num_cus() * occ_blocks / 8 * 8);
  int gx;
  if (total <= resident) {
    gx = total;
  } else {
    const int rounds = (total + resident - 1) / resident;
    gx = (total + rounds - 1) / rounds;
    gx = std::min(resident, (gx + 7) / 8 * 8);     // multiple of 8: one contiguous tile range per XCD
  }
  // code size of this instantiation (for the kernel's instruction warm-up), looked up once by its mangled name
  static const unsigned code_q = [] {
    char name[160];
    snprintf(name, sizeof(name), "_ZN12_GLOBAL__N_112igemm_kernelILi%dELi%dELi%dELi%dELi%dELi%dELb%dEEEv17ImagenIgemmParams", MI, NI, WM, WN, G, KSC,
             GEN ? 1 : 0);
    return std::min(imagen_kernel_code_bytes(name) >> 8, 0xffffu);
  }();
  ImagenIgemmParams q = p;
  q.launcher_word = (int)code_q;
  hipLaunchKernelGGL(kern, dim3(gx), dim3(512), lds, s, q);
  return imagen_hip_status("igemm launch");
}

template <int MI, int NI, int WM, int WN, int G>
int launch_cfg(const ImagenIgemmParams& p, hipStream_t s) {
  const int ks = (p.KH * p.KW * G + 1) / 2;
  if (G == 4) {
    if (ks == 18) return launch_ksc<MI, NI, WM, WN, G, (G == 4 ? 18 : 0)>(p, s);
    if (ks == 2) return launch_ksc<MI, NI, WM, WN, G, (G == 4 ? 2 : 0)>(p, s);
    if (ks == 8) return launch_ksc<MI, NI, WM, WN, G, (G == 4 ? 8 : 0)>(p, s);
    if (ks == 6) return launch_ksc<MI, NI, WM, WN, G, (G == 4 ? 6 : 0)>(p, s);   // three taps (round 6: the generic loop looks one step ahead for its weight fragment)
  }
  if (G == 8 && ks == 4) return launch_ksc<MI, NI, WM, WN, G, (G == 8 ? 4 : 0)>(p, s);
  if (G == 16 && ks == 8) return launch_ksc<MI, NI, WM, WN, G, (G == 16 ? 8 : 0)>(p, s);
  // the 15x15 CrossEmbed window over the 8-channel packed image (init_conv, ip.py:1051-1076): 113 K=16 steps in ONE chunk.  The generic
  // loop looks a single step ahead for its weight fragment (an L2 round trip per step: 190 us at 256^2, 8x its MFMA time); unrolled, the
  // ring runs 6-8 steps ahead like every other layer's
  if (G == 1 && ks == 113) return launch_ksc<MI, NI, WM, WN, G, (G == 1 ? 113 : 0)>(p, s);
  return launch_ksc<MI, NI, WM, WN, G, 0>(p, s);
}

}  // namespace

// kernel family 2 (conv_dma.hip): tile cfg ids kNumCfgs ..   (family 1, the LDS-staged kernel with an in-kernel prologue, was retired in round 3)
int imagen_conv_dma_num_configs();
int imagen_conv_dma_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups);
long imagen_conv_dma_lds_bytes(int idx, int KH, int KW, int TH, int TW);
int imagen_conv_dma_ring(int idx);
int launch_conv_dma(const ImagenIgemmParams* p, int idx, hipStream_t s);
static inline int cfg_base_dma() { return kNumCfgs; }
// kernel family 3 (conv_stream.hip): tile cfg ids behind family 2's
int imagen_conv_stream_num_configs();
int imagen_conv_stream_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups);
long imagen_conv_stream_lds_bytes(int idx, int KH, int KW, int TH, int TW);
int launch_conv_stream(const ImagenIgemmParams* p, int idx, hipStream_t s);
static inline int cfg_base_stream() { return cfg_base_dma() + imagen_conv_dma_num_configs(); }
// kernel family 4 (conv_pw.hip): the streaming pointwise convolution, tile cfg ids behind family 3's
int imagen_conv_pw_num_configs();
int imagen_conv_pw_config_info(int idx, int* tile_pixels, int* tile_cout, int* kchunks);
long imagen_conv_pw_lds_bytes(int idx, int KH, int KW, int TH, int TW);
int launch_conv_pw(const ImagenIgemmParams* p, int idx, hipStream_t s);
static inline int cfg_base_pw() { return cfg_base_stream() + imagen_conv_stream_num_configs(); }
// kernel family 5 (conv_big.hip): the big-tile all-DMA 3x3 convolution (256 / 128 px x 128 couts, 64 x 64 per wave), tile cfg ids behind family 4's
int imagen_conv_big_num_configs();
int imagen_conv_big_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups);
long imagen_conv_big_lds_bytes(int idx, int KH, int KW, int TH, int TW);
int launch_conv_big(const ImagenIgemmParams* p, int idx, hipStream_t s);
static inline int cfg_base_big() { return cfg_base_pw() + imagen_conv_pw_num_configs(); }
// kernel family 6 (conv_pro.hip): the streaming 3x3 convolution with the Block prologue on register-staged rows (C_out = 32 from 32 | 32 + 32 channels)
int imagen_conv_pro_num_configs();
int imagen_conv_pro_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups);
long imagen_conv_pro_lds_bytes(int idx, int KH, int KW, int TH, int TW);
int launch_conv_pro(const ImagenIgemmParams* p, int idx, hipStream_t s);
static inline int cfg_base_pro() { return cfg_base_big() + imagen_conv_big_num_configs(); }
// kernel family 7 (conv_gemm.hip): the tiled pointwise GEMM of the token / small-map layers (1x1, 128-row x 128-cout workgroup tiles, K loop)
int imagen_conv_gemm_num_configs();
int imagen_conv_gemm_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups);
long imagen_conv_gemm_lds_bytes(int idx, int KH, int KW, int TH, int TW);
int launch_conv_gemm(const ImagenIgemmParams* p, int idx, hipStream_t s);
static inline int cfg_base_gemm() { return cfg_base_pro() + imagen_conv_pro_num_configs(); }
// kernel family 8 (conv_small.hip): the 3x3 convolutions of the small maps (32 pixels x 32 | 64 | 128 couts per workgroup, K split over its waves)
int imagen_conv_small_num_configs();
int imagen_conv_small_config_info(int idx, int* tile_pixels, int* tile_cout, int* kgroups);
long imagen_conv_small_lds_bytes(int idx, int KH, int KW, int TH, int TW);
int launch_conv_small(const ImagenIgemmParams* p, int idx, hipStream_t s);
static inline int cfg_base_small() { return cfg_base_gemm() + imagen_conv_gemm_num_configs(); }
static inline int cfg_end() { return cfg_base_small() + imagen_conv_small_num_configs(); }

int launch_igemm(const ImagenIgemmParams* pp, hipStream_t s) {
  const ImagenIgemmParams& p = *pp;
  IMAGEN_CHECK(p.cfg >= 0 && p.cfg < cfg_end(), "igemm: bad cfg %d", p.cfg);
  IMAGEN_CHECK(p.x1 && p.w && p.y, "igemm: null x1/w/y");
  IMAGEN_CHECK(!p.addend || p.gate, "igemm: addend requires gate");
  IMAGEN_CHECK(p.cfg >= kNumCfgs || (p.TW > 0 && (p.TW & (p.TW - 1)) == 0), "igemm: tile width %d is not a power of two", p.TW);
  IMAGEN_CHECK(!p.pad_x1 || p.cfg < kNumCfgs, "igemm: pad_x1 (an x padding of its own) is implemented by kernel family 0 only (cfg %d)", p.cfg);
  IMAGEN_CHECK(p.pad_x1 >= 0, "igemm: pad_x1 %d", p.pad_x1);
  IMAGEN_CHECK(!p.gca_part || p.cfg >= kNumCfgs, "igemm: gca_part is implemented by the kernel families 2, 5, 7 and 8 only (cfg %d)", p.cfg);
  if (p.cfg >= cfg_base_small()) return launch_conv_small(pp, p.cfg - cfg_base_small(), s);
  if (p.cfg >= cfg_base_gemm()) return launch_conv_gemm(pp, p.cfg - cfg_base_gemm(), s);
  if (p.cfg >= cfg_base_pro()) return launch_conv_pro(pp, p.cfg - cfg_base_pro(), s);
  if (p.cfg >= cfg_base_big()) return launch_conv_big(pp, p.cfg - cfg_base_big(), s);
  if (p.cfg >= cfg_base_pw()) return launch_conv_pw(pp, p.cfg - cfg_base_pw(), s);
  if (p.cfg >= cfg_base_stream()) return launch_conv_stream(pp, p.cfg - cfg_base_stream(), s);
  if (p.cfg >= cfg_base_dma()) return launch_conv_dma(pp, p.cfg - cfg_base_dma(), s);
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

extern "C" int imagen_igemm_num_configs(void) { return cfg_end(); }

extern "C" int imagen_igemm_config_family(int cfg) {   // 0: wave-specialised persistent kernel (this file), 2: all-DMA kernel (conv_dma.hip), 3: streaming kernel (conv_stream.hip)
  if (cfg < 0 || cfg >= imagen_igemm_num_configs()) return -1;
  if (cfg >= cfg_base_small()) return 8; // 8: small-map 3x3 convolution with the K split over the waves of a workgroup (conv_small.hip)
  if (cfg >= cfg_base_gemm()) return 7;  // 7: tiled pointwise GEMM (conv_gemm.hip)
  if (cfg >= cfg_base_pro()) return 6;   // 6: streaming kernel with the prologue on register-staged rows (conv_pro.hip)
  if (cfg >= cfg_base_big()) return 5;   // 5: big-tile all-DMA kernel (conv_big.hip)
  return cfg >= cfg_base_pw() ? 4 : cfg >= cfg_base_stream() ? 3 : cfg >= cfg_base_dma() ? 2 : 0;   // 4: streaming pointwise kernel (conv_pw.hip)
}

extern "C" int imagen_igemm_config_ring(int cfg) {   // weight look-ahead ring depth in stages (family 2; 0 elsewhere)
  return (cfg >= cfg_base_dma() && cfg < cfg_base_stream()) ? imagen_conv_dma_ring(cfg - cfg_base_dma()) : 0;
}

extern "C" int imagen_igemm_config_info(int cfg, int* tile_pixels, int* tile_cout, int* kgroups) {
  if (cfg >= cfg_base_small()) return imagen_conv_small_config_info(cfg - cfg_base_small(), tile_pixels, tile_cout, kgroups);
  if (cfg >= cfg_base_gemm()) return imagen_conv_gemm_config_info(cfg - cfg_base_gemm(), tile_pixels, tile_cout, kgroups);
  if (cfg >= cfg_base_pro()) return imagen_conv_pro_config_info(cfg - cfg_base_pro(), tile_pixels, tile_cout, kgroups);
  if (cfg >= cfg_base_big()) return imagen_conv_big_config_info(cfg - cfg_base_big(), tile_pixels, tile_cout, kgroups);
  if (cfg >= cfg_base_pw()) return imagen_conv_pw_config_info(cfg - cfg_base_pw(), tile_pixels, tile_cout, kgroups);   // (family 4: kgroups = 32-channel input chunks)
  if (cfg >= cfg_base_stream()) return imagen_conv_stream_config_info(cfg - cfg_base_stream(), tile_pixels, tile_cout, kgroups);
  if (cfg >= cfg_base_dma()) return imagen_conv_dma_config_info(cfg - cfg_base_dma(), tile_pixels, tile_cout, kgroups);
  if (cfg < 0 || cfg >= kNumCfgs) return -1;
  const TileCfg& c = kCfgs[cfg];
  if (tile_pixels) *tile_pixels = 32 * c.MI * c.WM;
  if (tile_cout) *tile_cout = 32 * c.NI * c.WN;
  if (kgroups) *kgroups = c.G;
  return 0;
}

static constexpr int ksc_of(int G, int ks) {   // the launch_cfg dispatch, as a function
  if (G == 4 && (ks == 18 || ks == 2 || ks == 8 || ks == 6)) return ks;
  if (G == 8 && ks == 4) return ks;
  if (G == 16 && ks == 8) return ks;
  if (G == 1 && ks == 113) return ks;
  return 0;
}

extern "C" int imagen_igemm_stage_slots(int cfg, int KH, int KW) {
  if (cfg >= cfg_base_small()) return (KH == 3 && KW == 3) ? 1 << 20 : 0;
  if (cfg >= cfg_base_gemm()) return (KH == 1 && KW == 1) ? 1 << 20 : 0;
  if (cfg >= cfg_base_pro()) return (KH == 3 && KW == 3) ? 1 << 20 : 0;
  if (cfg >= cfg_base_big()) return (KH == 3 && KW == 3) ? 1 << 20 : 0;
  if (cfg >= cfg_base_pw()) return (KH == 1 && KW == 1) ? 1 << 20 : 0;
  if (cfg >= cfg_base_dma()) return (KH == 3 && KW == 3) ? 1 << 20 : 0;   // (no register staging: the tile shape is fixed per cfg; families 2 and 3)
  if (cfg < 0 || cfg >= kNumCfgs || KH < 1 || KW < 1) return -1;
  const TileCfg& c = kCfgs[cfg];
  return stage_slots(32 * c.MI * c.WM, c.G, ksc_of(c.G, (KH * KW * c.G + 1) / 2));
}

// dynamic LDS bytes of a launch of `cfg` with a KH x KW kernel (at `stride`) and a TH x TW output tile; -1: the combination is not launchable
extern "C" long imagen_igemm_lds_bytes(int cfg, int KH, int KW, int stride, int TH, int TW) {
  if (cfg >= cfg_base_small()) return stride == 1 ? imagen_conv_small_lds_bytes(cfg - cfg_base_small(), KH, KW, TH, TW) : -1;
  if (cfg >= cfg_base_gemm()) return stride == 1 ? imagen_conv_gemm_lds_bytes(cfg - cfg_base_gemm(), KH, KW, TH, TW) : -1;
  if (cfg >= cfg_base_pro()) return stride == 1 ? imagen_conv_pro_lds_bytes(cfg - cfg_base_pro(), KH, KW, TH, TW) : -1;
  if (cfg >= cfg_base_big()) return stride == 1 ? imagen_conv_big_lds_bytes(cfg - cfg_base_big(), KH, KW, TH, TW) : -1;
  if (cfg >= cfg_base_pw()) return stride == 1 ? imagen_conv_pw_lds_bytes(cfg - cfg_base_pw(), KH, KW, TH, TW) : -1;
  if (cfg >= cfg_base_stream()) return stride == 1 ? imagen_conv_stream_lds_bytes(cfg - cfg_base_stream(), KH, KW, TH, TW) : -1;
  if (cfg >= cfg_base_dma()) return stride == 1 ? imagen_conv_dma_lds_bytes(cfg - cfg_base_dma(), KH, KW, TH, TW) : -1;
  if (cfg < 0 || KH < 1 || KW < 1 || TH < 1 || TW < 1) return -1;
  const TileCfg& c = kCfgs[cfg];
  if (TH * TW != 32 * c.MI * c.WM) return -1;
  const int IT = ((TH - 1) * stride + KH) * ((TW - 1) * stride + KW);
  if (IT * c.G > imagen_igemm_stage_slots(cfg, KH, KW) * 256) return -1;
  const long PS = c.G == 1 ? 16 : c.G * 16 + 16;
  const long lds = 2 * IT * PS + (long)(4 * 32 * c.MI + kBiasLds) * (long)sizeof(float) + 16;
  return lds <= 160 * 1024 ? lds : -1;
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
constexpr int kTailSteps = 16;  // >= the kernel's weight look-ahead (12 for the single-MFMA tilings)

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
