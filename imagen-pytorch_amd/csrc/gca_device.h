// GlobalContext finalisation shared by the stand-alone GCA kernels and the fused igemm epilogue
// (reference: GlobalContext.forward, ip.py:965-970).  Called by ALL 256 threads of one workgroup.
#pragma once
#include "common.h"

constexpr int kGcaScratchFloats = 1024;   // reduction scratch beyond ctx / hid / wgt (4 floats per thread)

// out[o] = emit(o, sum_i W[i * ldw + o] * in[i]) for o < n_out.  One workgroup; the whole job is a latency chain of
// n_out * n_in / 256 loads per thread, so a thread owns V consecutive outputs (one vector load) and keeps 8 loads in flight.
template <int V, class Emit>
__device__ __forceinline__ void gca_matvec(int n_out, int n_in, const float* wt, int ldw, const float* in, float* s_red, Emit emit) {
  const int tid = threadIdx.x;
  const int nv = n_out / V;
  const int opt = nv < 256 ? nv : 256;   // output vectors per pass
  const int slices = 256 / opt;          // input slices summed through LDS
  const int sl = tid / opt, oo = tid - sl * opt;
  constexpr int U = 8;
  for (int o0 = 0; o0 < nv; o0 += opt) {
    float a[V];
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = 0.f;
    if (sl < slices && o0 + oo < nv) {
      const float* w = wt + (size_t)(o0 + oo) * V;
      int i = sl;
      for (; i + (U - 1) * slices < n_in; i += U * slices) {
        float v[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float* q = w + (size_t)(i + u * slices) * ldw;
          if constexpr (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(q);
            v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
          } else if constexpr (V == 2) {
            const float2 t = *reinterpret_cast<const float2*>(q);
            v[u][0] = t.x; v[u][1] = t.y;
          } else {
            v[u][0] = q[0];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float x = in[i + u * slices];
#pragma unroll
          for (int e = 0; e < V; ++e) a[e] += v[u][e] * x;
        }
      }
      for (; i < n_in; i += slices) {
        const float x = in[i];
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += w[(size_t)i * ldw + e] * x;
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) s_red[e * 256 + tid] = a[e];
    __syncthreads();
    if (tid < opt && o0 + tid < nv) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float t = 0.f;
        for (int q = 0; q < slices; ++q) t += s_red[e * 256 + q * opt + tid];
        emit((o0 + tid) * V + e, t);
      }
    }
    __syncthreads();
  }
}

// squeeze MLP on a context vector already in LDS: gate = sigmoid(W2 silu(W1 ctx + b1) + b2).  ctx: [C], hid: [hidden] scratch,
// s_red: kGcaScratchFloats scratch (all LDS).  Called by all 256 threads; ctx must be visible (barrier before the call).
__device__ __forceinline__ void gca_mlp(const float* ctx, float* hid, float* s_red, int C, int hidden, const float* w1t, const float* b1,
                                        const float* w2t, const float* b2, float* gate) {
  if ((hidden & 3) == 0)
    gca_matvec<4>(hidden, C, w1t, hidden, ctx, s_red, [&](int o, float t) __attribute__((always_inline)) { hid[o] = silu_f(t + b1[o]); });
  else
    gca_matvec<1>(hidden, C, w1t, hidden, ctx, s_red, [&](int o, float t) __attribute__((always_inline)) { hid[o] = silu_f(t + b1[o]); });
  gca_matvec<4>(C, hidden, w2t, C, hid, s_red, [&](int o, float t) __attribute__((always_inline)) { gate[o] = sigmoid_f(t + b2[o]); });
}

// part: [chunks][C + 2] = (max logit, sum exp, sum exp * h[c]) per chunk of pixels of ONE image (8-byte aligned rows).
//   ctx[c] = sum_i part[i][2+c] * exp(m_i - M) / sum_i s_i * exp(m_i - M)
//   gate   = sigmoid(W2 silu(W1 ctx + b1) + b2)        (w1t: [C][hidden], w2t: [hidden][C], 16-byte aligned, C % 8 == 0)
// lds: scratch of at least C + hidden + chunks + kGcaScratchFloats floats.
// the merge alone: ctx[C] (LDS) from the chunk rows; wgt: [chunks], s_red: kGcaScratchFloats (LDS).  All 256 threads; ctx is visible on return.
__device__ __forceinline__ void gca_merge(const float* part, int chunks, int C, float* ctx, float* wgt, float* s_red) {
  const int tid = threadIdx.x;
  const int stride = C + 2;
  const int lane = tid & 63, wave = tid >> 6;
  float lm = -3.0e38f;
  for (int i = tid; i < chunks; i += 256) lm = fmaxf(lm, part[(size_t)i * stride]);
  for (int off = 32; off > 0; off >>= 1) lm = fmaxf(lm, __shfl_xor(lm, off));
  if (lane == 0) s_red[wave] = lm;
  __syncthreads();
  const float M = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  float ls = 0.f;
  for (int i = tid; i < chunks; i += 256) {
    const float w = __expf(part[(size_t)i * stride] - M);
    wgt[i] = w;
    ls += part[(size_t)i * stride + 1] * w;
  }
  for (int off = 32; off > 0; off >>= 1) ls += __shfl_xor(ls, off);
  if (lane == 0) s_red[4 + wave] = ls;
  __syncthreads();   // also publishes wgt[]
  const float inv_S = 1.0f / (s_red[4] + s_red[5] + s_red[6] + s_red[7]);
  __syncthreads();   // s_red is reused by the matvec
  gca_matvec<2>(C, chunks, part + 2, stride, wgt, s_red, [&](int o, float t) __attribute__((always_inline)) { ctx[o] = t * inv_S; });
}

__device__ __forceinline__ void gca_finalize(const float* part, int chunks, int C, int hidden, const float* w1t, const float* b1,
                                             const float* w2t, const float* b2, float* gate, float* lds) {
  float* ctx = lds;
  float* hid = ctx + C;
  float* wgt = hid + hidden;
  float* s_red = wgt + chunks;
  const int tid = threadIdx.x;
  const int stride = C + 2;
  // (max, sum-exp) of the chunks: ONE round trip (both values of a chunk sit in the same cache line), block reductions by
  // wave shuffles + one LDS hop each — the 256-thread trees this replaces cost 16 barriers and a second round trip
  const int lane = tid & 63, wave = tid >> 6;
  float lm = -3.0e38f;
  for (int i = tid; i < chunks; i += 256) lm = fmaxf(lm, part[(size_t)i * stride]);
  for (int off = 32; off > 0; off >>= 1) lm = fmaxf(lm, __shfl_xor(lm, off));
  if (lane == 0) s_red[wave] = lm;
  __syncthreads();
  const float M = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  float ls = 0.f;
  for (int i = tid; i < chunks; i += 256) {
    const float w = __expf(part[(size_t)i * stride] - M);
    wgt[i] = w;
    ls += part[(size_t)i * stride + 1] * w;
  }
  for (int off = 32; off > 0; off >>= 1) ls += __shfl_xor(ls, off);
  if (lane == 0) s_red[4 + wave] = ls;
  __syncthreads();   // also publishes wgt[]
  const float inv_S = 1.0f / (s_red[4] + s_red[5] + s_red[6] + s_red[7]);
  __syncthreads();   // s_red is reused by the matvecs
  gca_matvec<2>(C, chunks, part + 2, stride, wgt, s_red, [&](int o, float t) __attribute__((always_inline)) { ctx[o] = t * inv_S; });
  gca_mlp(ctx, hid, s_red, C, hidden, w1t, b1, w2t, b2, gate);
}


// ------------------------------------------------------------------------------------------------ the finalisation INSIDE the producing launch (round 6)
// The workgroup that writes the LAST chunk row of an image finalises that image's gate itself, so no GCA_FINAL launch follows the conv (or the
// GCA_PARTIAL pass) that emitted the rows: 31 launches of 8.5 - 13.4 us per DDPM step pair of the benchmark, each a 16-workgroup kernel between two
// chip-filling ones.  Protocol (cdna_hip_programming.md, Guideline 16, recipe R1): the chunk row is stored write-through (agent-scope stores), every
// storing wave drains its stores, the workgroup meets at a barrier, ONE lane takes a ticket on the image's device-scope counter; the workgroup that
// draws the last ticket resets the counter (so that a graph replay starts from zero), runs one agent-scope acquire and then reads all rows with plain
// loads.  The hand-off does not depend on dispatch order or placement: whichever workgroup arrives last has, by the counter, every other row behind a
// completed write-through store.
//
// gca_epilogue_final<NT>: the finalisation by the NT (64 .. 1024, a multiple of 64) threads of one workgroup, as gca_final_fast_body does it with 1024:
// every global load the gate depends on — chunk statistics, this thread's slice of the chunk rows, of both squeeze-MLP matrices, the biases — is
// requested before the first wait; everything else runs out of registers and LDS.  C, hidden: powers of two, 8 <= C <= NT, 4 <= hidden <= 4 NT,
// chunks <= 4 NT (the caller checks; ops.gca_epilogue_final_ok mirrors it).  lds: 4 NT + 2 C + 2 hidden + chunks + 64 floats of dead LDS, 16-byte aligned.
// t: the thread's index in [0, NT).  The gate goes to global memory (gate[C] of this image).
constexpr int kGcaEpiW = 8;    // prefetched float4 per thread and weight matrix
constexpr int kGcaEpiP = 4;    // prefetched chunk-row elements per thread
constexpr int kGcaEpiS = 4;    // chunk statistics per thread

__host__ __device__ constexpr int gca_epilogue_final_lds_floats(int NT, int C, int hidden, int chunks) { return 4 * NT + 2 * C + 2 * hidden + chunks + 64; }

template <int NT, class Emit>
__device__ __forceinline__ void gca_epi_matvec(const float4 (&w)[kGcaEpiW], const float* wt, int n_in, int n_out, const float* in, float4* red, int t, Emit emit) {
  const int nvec = n_out >> 2;                       // <= NT
  const int rpp = NT / nvec;                         // rows per pass
  const int cg = t & (nvec - 1), r = t / nvec;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < kGcaEpiW; ++k) {
    const int row = r + k * rpp;
    const float x = row < n_in ? in[row] : 0.f;
    a.x += w[k].x * x; a.y += w[k].y * x; a.z += w[k].z * x; a.w += w[k].w * x;
  }
  for (int row = r + kGcaEpiW * rpp; row < n_in; row += rpp) {
    const float4 q = *reinterpret_cast<const float4*>(wt + (size_t)row * n_out + cg * 4);
    const float x = in[row];
    a.x += q.x * x; a.y += q.y * x; a.z += q.z * x; a.w += q.w * x;
  }
  red[t] = a;
  __syncthreads();
  for (int o = t; o < n_out; o += NT) {
    const float* col = reinterpret_cast<const float*>(red + (o >> 2)) + (o & 3);
    float sum = 0.f;
    for (int rr = 0; rr < rpp; ++rr) sum += col[(size_t)rr * nvec * 4];
    emit(o, sum);
  }
  __syncthreads();
}

template <int NT>
__device__ __forceinline__ void gca_epi_prefetch(float4 (&w)[kGcaEpiW], const float* wt, int n_in, int n_out, int t) {
  const int nvec = n_out >> 2, rpp = NT / nvec;
  const int cg = t & (nvec - 1), r = t / nvec;
#pragma unroll
  for (int k = 0; k < kGcaEpiW; ++k) {
    const int row = r + k * rpp;
    w[k] = row < n_in ? *reinterpret_cast<const float4*>(wt + (size_t)row * n_out + cg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int NT>
__device__ __forceinline__ void gca_epilogue_final(const float* part, int chunks, int C, int hidden, const float* w1t, const float* b1, const float* w2t,
                                                   const float* b2, float* gate, float* lds, int t) {
  constexpr int NW = NT / 64;
  float4* s_red = reinterpret_cast<float4*>(lds);            // [NT]
  float* s_ctx = lds + 4 * NT;                               // [C]
  float* s_b2 = s_ctx + C;                                   // [C]
  float* s_hid = s_b2 + C;                                   // [hidden]
  float* s_b1 = s_hid + hidden;                              // [hidden]
  float* s_wgt = s_b1 + hidden;                              // [chunks]
  float* s_sc = s_wgt + chunks;                              // [64]
  const int lane = t & 63, wave = t >> 6;
  const int stride = C + 2;
  // ---- every global load, before the first wait — in the order of use (vmcnt retires in issue order)
  float2 ms[kGcaEpiS];
#pragma unroll
  for (int j = 0; j < kGcaEpiS; ++j) {
    const int i = t + j * NT;
    ms[j] = i < chunks ? *reinterpret_cast<const float2*>(part + (size_t)i * stride) : make_float2(-3.0e38f, 0.f);
  }
  const int c = t & (C - 1), sl = t / C, nsl = NT / C;       // C <= NT
  float pv[kGcaEpiP];
#pragma unroll
  for (int j = 0; j < kGcaEpiP; ++j) {
    const int i = sl + j * nsl;
    pv[j] = i < chunks ? part[(size_t)i * stride + 2 + c] : 0.f;
  }
  for (int o = t; o < hidden; o += NT) s_b1[o] = b1[o];
  if (t < C) s_b2[t] = b2[t];
  float4 w1[kGcaEpiW], w2[kGcaEpiW];
  gca_epi_prefetch<NT>(w1, w1t, C, hidden, t);
  gca_epi_prefetch<NT>(w2, w2t, hidden, C, t);
  // ---- softmax merge weights of the chunks
  float m = ms[0].x;
#pragma unroll
  for (int j = 1; j < kGcaEpiS; ++j) m = fmaxf(m, ms[j].x);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (lane == 0) s_sc[wave] = m;
  __syncthreads();
  float M = s_sc[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) M = fmaxf(M, s_sc[w]);
  float ssum = 0.f;
#pragma unroll
  for (int j = 0; j < kGcaEpiS; ++j) {
    const int i = t + j * NT;
    if (i < chunks) {
      const float wg = __expf(ms[j].x - M);
      s_wgt[i] = wg;
      ssum += ms[j].y * wg;
    }
  }
  for (int off = 32; off > 0; off >>= 1) ssum += __shfl_xor(ssum, off);
  if (lane == 0) s_sc[32 + wave] = ssum;
  __syncthreads();   // also publishes s_wgt, s_b1, s_b2
  float S = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) S += s_sc[32 + w];
  const float inv_S = 1.0f / S;
  // ---- ctx[c] = sum_i part[i][2 + c] * wgt[i] / S
  float a = 0.f;
#pragma unroll
  for (int j = 0; j < kGcaEpiP; ++j) {
    const int i = sl + j * nsl;
    a += pv[j] * (i < chunks ? s_wgt[i] : 0.f);
  }
  for (int i = sl + kGcaEpiP * nsl; i < chunks; i += nsl) a += part[(size_t)i * stride + 2 + c] * s_wgt[i];
  float* red_f = reinterpret_cast<float*>(s_red);
  red_f[t] = a;
  __syncthreads();
  if (t < C) {
    float v = 0.f;
    for (int q = 0; q < nsl; ++q) v += red_f[q * C + t];
    s_ctx[t] = v * inv_S;
  }
  __syncthreads();
  // ---- squeeze MLP out of the prefetched registers
  gca_epi_matvec<NT>(w1, w1t, C, hidden, s_ctx, s_red, t, [&](int o, float v) __attribute__((always_inline)) { s_hid[o] = silu_f(v + s_b1[o]); });
  gca_epi_matvec<NT>(w2, w2t, hidden, C, s_hid, s_red, t, [&](int o, float v) __attribute__((always_inline)) { gate[o] = sigmoid_f(v + s_b2[o]); });
}

// The ticket.  Called by ALL threads of the workgroup right after its chunk row was stored with imagen_st_wt_f32 / _f32x2 (below): drains this wave's
// stores, meets the workgroup, draws the ticket.  Returns true (uniformly) in the workgroup that completed the image; that workgroup has then also run
// the acquire and may read every row of the image with plain loads.  s_flag: one int of LDS.
__device__ __forceinline__ void imagen_st_wt_f32(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool gca_ticket_is_last(unsigned* ticket, unsigned rows_per_image, int* s_flag, bool leader) {
  IMAGEN_WAIT_VM(0);            // EVERY storing wave drains its write-through stores
  __syncthreads();
  if (leader) {
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = tk == rows_per_image - 1u;
    if (last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch (graph replay) counts from zero again
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                           // ONE acquire covers the workgroup behind the barrier below
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}
