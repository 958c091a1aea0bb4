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

