// GlobalContext finalisation shared by the stand-alone GCA_FINAL kernel and the fused igemm epilogue
// (reference: GlobalContext.forward, ip.py:965-970).  Called by ALL 256 threads of one workgroup.
#pragma once
#include "common.h"

// part: [chunks][C + 2] = (max logit, sum exp, sum exp * h[c]) per chunk of pixels of ONE image.
//   ctx[c] = sum_i part[i][2+c] * exp(m_i - M) / sum_i s_i * exp(m_i - M)
//   gate   = sigmoid(W2 silu(W1 ctx + b1) + b2)        (w1t: [C][hidden], w2t: [hidden][C])
// lds: scratch of at least C + hidden + chunks + 256 floats.
__device__ __forceinline__ void gca_finalize(const float* part, int chunks, int C, int hidden, const float* w1t, const float* b1,
                                             const float* w2t, const float* b2, float* gate, float* lds) {
  float* ctx = lds;
  float* hid = ctx + C;
  float* wgt = hid + hidden;
  float* s_red = wgt + chunks;
  const int tid = threadIdx.x;
  const int stride = C + 2;
  float lm = -3.0e38f;
  for (int i = tid; i < chunks; i += 256) lm = fmaxf(lm, part[(size_t)i * stride]);
  s_red[tid] = lm;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) s_red[tid] = fmaxf(s_red[tid], s_red[tid + off]);
    __syncthreads();
  }
  const float M = s_red[0];
  __syncthreads();
  float ls = 0.f;
  for (int i = tid; i < chunks; i += 256) {
    const float w = __expf(part[(size_t)i * stride] - M);
    wgt[i] = w;
    ls += part[(size_t)i * stride + 1] * w;
  }
  s_red[tid] = ls;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) s_red[tid] += s_red[tid + off];
    __syncthreads();
  }
  const float S = s_red[0];
  __syncthreads();
  // ctx: thread = (chunk slice, channel); slices merged through LDS
  {
    const int cpt = C < 256 ? C : 256;
    const int slices = 256 / cpt;
    const int sl = tid / cpt, cc = tid - sl * cpt;
    for (int c0 = 0; c0 < C; c0 += cpt) {
      float a = 0.f;
      if (sl < slices && c0 + cc < C)
        for (int i = sl; i < chunks; i += slices) a += part[(size_t)i * stride + 2 + c0 + cc] * wgt[i];
      s_red[tid] = a;
      __syncthreads();
      if (tid < cpt && c0 + tid < C) {
        float t = 0.f;
        for (int q = 0; q < slices; ++q) t += s_red[q * cpt + tid];
        ctx[c0 + tid] = t / S;
      }
      __syncthreads();
    }
  }
  // squeeze MLP: out[o] = act(bias[o] + sum_i Wt[i][o] * in[i]); thread = (input slice, output)
  auto matvec = [&](int n_out, int n_in, const float* wt, const float* bias, const float* in, float* out_lds, float* out_gate)
                    __attribute__((always_inline)) {
    const int opt = n_out < 256 ? n_out : 256;
    const int slices = 256 / opt;
    const int sl = tid / opt, oo = tid - sl * opt;
    for (int o0 = 0; o0 < n_out; o0 += opt) {
      float a = 0.f;
      if (sl < slices && o0 + oo < n_out)
        for (int i = sl; i < n_in; i += slices) a += wt[(size_t)i * n_out + o0 + oo] * in[i];
      s_red[tid] = a;
      __syncthreads();
      if (tid < opt && o0 + tid < n_out) {
        float t = bias[o0 + tid];
        for (int q = 0; q < slices; ++q) t += s_red[q * opt + tid];
        if (out_lds) out_lds[o0 + tid] = silu_f(t);
        else out_gate[o0 + tid] = sigmoid_f(t);
      }
      __syncthreads();
    }
  };
  matvec(hidden, C, w1t, b1, ctx, hid, nullptr);
  matvec(C, hidden, w2t, b2, hid, nullptr, gate);
}
