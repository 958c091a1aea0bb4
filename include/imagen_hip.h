/*
 * imagen_hip.h — C ABI of libimagen_hip.so, the gfx950 (MI355X / CDNA4) kernel library behind the
 * Imagen cascaded-DDPM sampling path.
 *
 * The reference (lucidrains/imagen-pytorch v2, pure Python) has no FFI of its own: the "operator API"
 * of this path is the Python class surface (Unet.forward / Imagen.sample).  This header is the inner
 * boundary our Python drop-in classes call through: flat extern "C" entry points,
 * raw device pointers + sizes + a hipStream_t, no C++/torch types.  Each op below names the reference
 * lines (imagen_pytorch/imagen_pytorch.py = "ip.py") whose ATen sequence it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors kept alive by Python);
 *   - nothing allocates, frees or synchronises; every launch goes to the passed stream, so a whole
 *     denoiser step is capturable into a hipGraph;
 *   - activations are fp16 ("h") NHWC: element (b, y, x, c) at  b*bstride + (y*W + x)*ld + c;
 *     token tensors (b, n, c) are the same layout with H = 1, W = n;
 *   - statistics, biases, affine vectors, sampler state are fp32 ("f");
 *   - return value: 0 on success, otherwise a hipError_t / negative library code;
 *     imagen_last_error() returns a thread-local message.
 *
 * Every params struct is plain-old-data laid out as: pointers, then int32, then float — mirrored 1:1 by
 * ctypes.Structure classes in imagen-pytorch_amd/_abi.py; imagen_sizeof(kind) lets the host verify it.
 */
#ifndef IMAGEN_HIP_H
#define IMAGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMAGEN_ABI_VERSION 11 /* 11: ImagenDdpmUpdateParams.row_keys (per-row Philox key + sample index: requests merged into one batch draw their own noise);  10: ImagenIgemmParams.pad_x1 (a KH x KW window with its own x padding: the causal temporal conv of Imagen-Video as ONE (3 x 1)-tap launch);  9: LINEAR_F32, SCALE_SHIFT ss_f32;  8: ROWCHAIN, the latency probes;  3: head_dim in the attention / QNORM / KV_PREP params; 4: DDPM_UPDATE x0_thr; 5: GCA_TAIL; 6: ACT_PREP self_stat, STEP_SLICE;
                               * 7: every launch carries sizeof(its params struct) (a stale mirror of a struct fails loudly), ImagenIgemmParams.dbg -> launcher_word, kernel families 6 and 7, ImagenAttentionParams.softmax_mode */

typedef void* imagen_stream_t; /* hipStream_t */

enum ImagenOpKind {
  IMAGEN_OP_IGEMM = 1,         /* implicit-GEMM conv / linear on MFMA with fused prologue + epilogue   */
  IMAGEN_OP_ROWSTAT = 2,       /* per-pixel RMS / LayerNorm statistics                                 */
  IMAGEN_OP_ATTENTION = 3,     /* flash-style cosine-sim attention                                     */
  IMAGEN_OP_KV_PREP = 4,       /* l2norm*scale of k rows + transposed v into the attention K / V^T buffers */
  IMAGEN_OP_QNORM = 5,         /* in-place l2norm*scale of q rows                                      */
  IMAGEN_OP_GCA_PARTIAL = 6,   /* GlobalContext: per-block softmax-pool partials                       */
  IMAGEN_OP_GCA_FINAL = 7,     /* GlobalContext: combine partials + squeeze MLP -> gate                */
  IMAGEN_OP_GATE_RESIDUAL = 8, /* out = h*gate + res                                                   */
  IMAGEN_OP_LN_RESIDUAL = 9,   /* out = LayerNorm(y)*g (+beta) (+ res)                                 */
  IMAGEN_OP_TIME_EMBED = 10,   /* learned sinusoidal embedding + Linear + SiLU                         */
  IMAGEN_OP_SCALE_SHIFT = 11,  /* time-MLP output -> per-(batch,channel) prologue affine              */
  IMAGEN_OP_PACK_IMAGE = 12,   /* fp32 NCHW image(s) -> fp16 NHWC, zero-padded channels                */
  IMAGEN_OP_CFG_X0 = 13,       /* CFG combine + x0-from-noise + |x0| keys for the quantile             */
  IMAGEN_OP_QUANTILE = 14,     /* exact per-sample quantile (radix select + lerp)                      */
  IMAGEN_OP_DDPM_UPDATE = 15,  /* dynamic threshold + posterior mean/var + Philox noise                */
  IMAGEN_OP_ROWS_COPY = 16,    /* strided fp16 row copy / broadcast (token assembly)                   */
  IMAGEN_OP_MEMSET32 = 17,     /* fill fp32/int32 words                                                */
  IMAGEN_OP_SELECT_ROWS = 18,  /* per-(row, token) select between text tokens and null_text_embed      */
  IMAGEN_OP_MEAN_ROWS = 19,    /* mean over the token axis                                             */
  IMAGEN_OP_RANDN = 20,        /* counter-based (Philox4x32-10) standard normal fill, keyed by global sample index */
  IMAGEN_OP_LOWRES_PREP = 21,  /* nearest resize + normalise + noise-augment the previous stage's image  */
  IMAGEN_OP_LINCOMB = 22,      /* per-step weighted sum of up to 4 fp32 images (+ Philox noise): the EDM sampler's state updates */
  IMAGEN_OP_KV_PREP_MULTI = 23, /* several KV_PREP jobs (one per attention site) in ONE launch                         */
  IMAGEN_OP_TEMPORAL_PEG = 24,  /* Imagen-Video: depthwise causal conv over 3 frames + residual                        */
  IMAGEN_OP_TEMPORAL_ATTENTION = 25, /* Imagen-Video: per-pixel causal attention over the frames, with a bias table    */
  IMAGEN_OP_ACT_PREP = 26,     /* the IGEMM prologue as its own pass: norm -> affine -> SiLU of a (two-tensor) input, written as fp16 */
  IMAGEN_OP_GCA_TAIL = 27,     /* ResnetBlock tail in ONE launch: GlobalContext finalisation + h*gate + res (+ statistics, + the next Block's activated input) */
  IMAGEN_OP_STEP_SLICE = 28,   /* copy the current step's rows of up to four per-step tables (the timestep-only conditioning, computed for all steps at once) into the buffers the step's kernels read */
  IMAGEN_OP_ROWCHAIN = 29,     /* a chain of row-local token layers (attention out-projection + LayerNorm + FeedForward | a whole cross-attention | LayerNorm + q/k/v projection + K^/V^T rows) in ONE launch */
  IMAGEN_OP_LINEAR_F32 = 30,   /* a Linear on per-sample vectors in fp32 end to end (the timestep-conditioning chain: to_time_cond, the ResnetBlocks' time MLPs) */
  IMAGEN_OP_KIND_COUNT = 31
};

/* ------------------------------------------------------------------------------------------------
 * IGEMM — replaces: Block (ChanRMSNorm -> scale/shift -> SiLU -> Conv2d 3x3) ip.py:671-691;
 * CrossEmbedLayer ip.py:1051-1076; Downsample ip.py:633-640; PixelShuffleUpsample ip.py:603-631;
 * res_conv ip.py:732; final_conv ip.py:1436,1725; every nn.Linear on the path (ip.py:521-532,
 * 738-741, 782-791, 972-980, 1213-1228, 1260, 1283-1287).
 *
 *   in(p, c)  = concat(x1[.., :C1], x2[.., :C2])                       (x2 optional)
 *   a(p, c)   = act_in( (in - mu[p]) * rs[p] * pa[b, c] + ps[b, c] )   (each factor optional), 0 outside the image
 *   acc(q, o) = sum_{ky,kx,c} a(q*stride - (pad, pad_x) + (ky,kx), c) * W[o][ky][kx][c]      (pad_x = pad unless pad_x1 != 0; OH / OW as given: rows / columns
 *               past the input are zero, so pad is the padding BEFORE the image and a causal window simply has none behind it)
 *   v         = act_out(acc + bias[o]);  v += addend(q,o) * gate[b,o]  |  v += res(q,o)      (addend and res exclude each other)
 *   y         = v   (NHWC fp16 | pixel-shuffle NHWC fp16 | NCHW fp32);   ssq_out[q] = sum_o fp16(v)^2   (optional)
 *   or, with post_pa:  y = silu(v / max(||v||_2 over o, 1e-12) * post_pa[b,o] + post_ps[b,o])        (the next Block's prologue)
 *
 * Weights are pre-packed by imagen_pack_igemm_weights() into MFMA-fragment order.
 */
enum { IMAGEN_ACT_NONE = 0, IMAGEN_ACT_SILU = 1, IMAGEN_ACT_GELU = 2 };
enum { IMAGEN_OUT_NHWC = 0, IMAGEN_OUT_PIXEL_SHUFFLE = 1, IMAGEN_OUT_NCHW_F32 = 2 };

typedef struct ImagenIgemmParams {
  const void* x1;      /* fp16 */
  const void* x2;      /* fp16 or NULL */
  const float* mu;     /* [B*H*W] or NULL */
  const float* rs;     /* [B*H*W] or NULL */
  const float* pa;     /* [B or 1][Cin] or NULL */
  const float* ps;     /* [B or 1][Cin] or NULL */
  const void* w;       /* packed fp16 */
  const float* bias;   /* [Cout] or NULL */
  const void* addend;  /* fp16 NHWC at output resolution or NULL */
  const float* gate;   /* [B][Cout] (required with addend) */
  const void* res;     /* fp16 NHWC at output resolution or NULL */
  void* y;             /* fp16 NHWC / fp32 NCHW */
  const float* ssq_a;  /* optional per-input-pixel sum of squares of x1 (emitted by its producer): when set (and rs == NULL) */
  const float* ssq_b;  /*   rs = 1/max(sqrt(ssq_a + ssq_wb*ssq_b), 1e-12) — ChanRMSNorm statistics without a separate pass   */
  float* ssq_out;      /* optional per-output-pixel sum of squares of the stored fp16 output (NHWC mode, Cout <= tile couts) */
  /* optional output-side Block prologue (NHWC mode, Cout <= tile couts, no addend / residual / act_out): the stored tensor is
   * silu(h / max(||h||, 1e-12) * post_pa[b, c] + post_ps[b, c]) with h = acc + bias and the norm over all Cout channels of the
   * pixel (ChanRMSNorm -> scale/shift -> SiLU of the NEXT Block, ip.py:671-691, applied by the producer so that the consuming
   * conv stages its input with no arithmetic at all) */
  const float* post_pa; const float* post_ps;
  /* optional GlobalContext partials of the OUTPUT (plain NHWC mode, Cout <= tile couts, the kernel families 2, 5, 7 and 8): with
   * logit[q] = y[q, :] . gca_wk + gca_bk (ip.py:965-966), every output tile writes (max logit, sum exp, sum exp * y[q, c]) over its
   * pixels to gca_part[b][tile][Cout + 2] (tile = ty * tilesX + tx): exactly the rows GCA_PARTIAL produces with chunks = tiles per
   * image, ready for GCA_FINAL — the separate pass over the tensor disappears. */
  const float* gca_wk; float* gca_part;
  int32_t B, H, W;     /* input batch / spatial dims */
  int32_t C1, ld1, bs1; /* channels, pixel stride, batch stride (elements) of x1 */
  int32_t C2, ld2, bs2;
  int32_t KH, KW, stride, pad;
  int32_t OH, OW;
  int32_t Cin_pad;     /* C1+C2 rounded up to the k-chunk (8*G) */
  int32_t Cout, Cout_pad; /* Cout_pad = multiple of 32*NI*WN of the chosen tile */
  int32_t pstride;     /* batch stride of pa/ps in floats (0 = shared) */
  int32_t post_pstride; /* batch stride of post_pa / post_ps in floats */
  int32_t act_in, act_out;
  int32_t ld_add, bs_add, ld_res, bs_res;
  int32_t gate_stride;
  int32_t ldy, bsy;    /* output pixel / batch strides (elements) */
  int32_t pad_x1;      /* 0: the x padding equals `pad`; else x padding + 1 (kernel family 0 only).  Imagen-Video's causal Conv1d over frames
                        * (iv.py:436-449) in the (clip, frame, pixel) view: KH = 3, KW = 1, pad = 2, pad_x1 = 1, OH = frames */
  int32_t out_mode;
  int32_t TH, TW;      /* output tile (TH*TW must equal the tile's pixel count) */
  int32_t cfg;         /* tile configuration id, see imagen_igemm_config_info */
  int32_t launcher_word; /* pass 0: private to the launcher (it stores the kernel's code size here for the in-kernel instruction warm-up) */
  float ssq_wb;        /* weight of ssq_b (skip_connect_scale^2 for the concatenated skip tensor) */
  float gca_bk;        /* bias of the GlobalContext logit (to_k.bias) */
} ImagenIgemmParams;

/* ACT_PREP — the IGEMM prologue (Block: ChanRMSNorm -> scale/shift -> SiLU, ip.py:683-690) materialised once:
 *   y[p, c] = fp16( act_in( (concat(x1, x2)[p, c] - mu[p]) * rs[p] * pa[b, c] + ps[b, c] ) )      rs from ssq_a (+ ssq_wb * ssq_b) when rs == NULL
 * Used in front of the all-DMA conv kernel families (csrc/conv_dma.hip, conv_big.hip), which copy their input global -> LDS without touching it, for
 * the MFMA-bound layers (C >= 128): the activation is then computed once per element instead of once per staging workgroup
 * (halo overlap x output-channel tiles = 2.8x on the 384 -> 256 convs) and leaves the conv's instruction stream. */
typedef struct ImagenActPrepParams {
  const void* x1; const void* x2; const float* mu; const float* rs; const float* pa; const float* ps;
  const float* ssq_a; const float* ssq_b; void* y;
  int32_t rows, rows_per_batch;   /* pixels in total / per batch element */
  int32_t C1, ld1, bs1, C2, ld2, bs2;
  int32_t ldy, bsy, pstride, act_in;
  float ssq_wb;
  int32_t self_stat;   /* 1: the per-pixel sum of squares of x1's channels is computed by this launch (rs, mu, ssq_a NULL): rs = 1 / sqrt(sum_c x1^2
                        * + ssq_wb * ssq_b) — no ROWSTAT pass in front of it where the producer of x1 could not emit its statistics */
} ImagenActPrepParams;

/* ROWSTAT — replaces the reductions inside ChanRMSNorm (ip.py:322-329) and LayerNorm (ip.py:331-349,
 * nn.LayerNorm).  mode 0: rs = 1/max(sqrt(ssq1 + w2*ssq2), 1e-12), mu untouched.
 * mode 1: mu = mean, rs = rsqrt(var + eps) (biased variance, two-pass).
 * mode 2: rs = ssq1 + w2*ssq2 (raw sum of squares, the form IGEMM's ssq_a/ssq_b consume). */
typedef struct ImagenRowstatParams {
  const void* x1; const void* x2; float* mu; float* rs;
  int32_t rows, C1, ld1, C2, ld2, mode;
  int32_t rows_per_batch, bs1, bs2; /* batch strides (elements); rows_per_batch = rows if contiguous */
  float w2, eps;
} ImagenRowstatParams;

/* ATTENTION — replaces ip.py:559-590 (self, shared k/v head) and ip.py:812-833 (cross, per-head k/v),
 * also PerceiverAttention ip.py:424-444.  q rows are pre-normalised/scaled (QNORM) and include the
 * similarity scale * log2(e); K rows pre-normalised (KV_PREP); VT is V transposed [d][key].
 *   o[r] = softmax_j(q[r] . k[j]) @ v   for r in rows, j in [0, J)
 * Addressing: q/o row r of (batch b, head h): base + b*q_bs + h*q_hs + r*q_rs ; k: b*k_bs + h*k_hs + j*k_rs ;
 * vt: b*vt_bs + h*vt_hs + d*vt_ds + j.   head_dim: 64 (0 means 64) or 32 — the reference's UnetConfig default is 32 x 16 heads
 * (configs.py:48-49), every README config uses 64. */
typedef struct ImagenAttentionParams {
  const void* q; const void* k; const void* vt; void* o;
  int32_t B, heads, rows, J;
  int32_t q_bs, q_hs, q_rs;
  int32_t k_bs, k_hs, k_rs;
  int32_t vt_bs, vt_hs, vt_ds;
  int32_t o_bs, o_hs, o_rs;
  /* optional fused QNORM: q rows arrive raw and are l2-normalised * q_scale[head_dim] * q_mult while they are loaded */
  const float* q_scale; float q_mult;
  int32_t head_dim;
  /* softmax_mode 0: online softmax (running row maximum).  1: the CALLER guarantees |logit| <= logit_bound (log2 units, <= 14) for every
   * (query, key) pair — q and k rows are unit vectors times fixed parameter vectors, so q_mult * max_d |q_scale_d k_scale_d| is known when
   * the plan is built: the weights exp2(logit) of all keys then lie in fp16's normal range together and no maximum is tracked.  Same result
   * up to rounding; honoured by the 64-dim / >= 256-row tiling, ignored elsewhere. */
  int32_t softmax_mode; float logit_bound;
} ImagenAttentionParams;

/* KV_PREP — k/v rows -> attention operand buffers (null_kv, context kv, self kv; ip.py:545-561, 805-814).
 *   khat[b, h, r0 + r, :] = l2norm(k_src row) * k_scale ;  vt[b, h, :, r0 + r] = v_src row
 * src row r of batch b, head h at  b*src_bs + r*src_rs + h*src_hs (+ k_off / v_off);  src_bs = 0 broadcasts. */
typedef struct ImagenKvPrepParams {
  const void* k_src; const void* v_src; const float* k_scale; void* khat; void* vt;
  int32_t B, heads, rows, r0;
  int32_t src_bs, src_rs, src_hs;
  int32_t k_bs, k_hs, k_rs;
  int32_t vt_bs, vt_hs, vt_ds;
  int32_t src_is_f32; /* null_kv parameters are fp32 */
  int32_t head_dim;   /* 64 (0 means 64) or 32 */
} ImagenKvPrepParams;

/* KV_PREP_MULTI — the per-step context K/V rows of ALL attention sites (2 time tokens each) in one launch instead of one
 * tiny launch per site: `jobs` is a DEVICE array of n ImagenKvPrepParams, grid.z = job. */
typedef struct ImagenKvPrepMultiParams {
  const ImagenKvPrepParams* jobs; int32_t n, max_rows, max_bh; /* max over the jobs of rows and B*heads (grid extents) */
} ImagenKvPrepMultiParams;

/* ---- Imagen-Video (imagen_pytorch/imagen_video.py = "iv.py") ----------------------------------------------------------------
 * Video activations are fp16 [B, F, P, C]: the F frames of a clip are consecutive NHWC images of P = H*W pixels, so every per-frame
 * op of the image path applies unchanged with batch B*F, and every per-clip op (GlobalContext, space-time attention) with H*W := F*P.
 *
 * TEMPORAL_PEG — iv.py:1413-1414, Residual(Pad + depthwise Conv3d (3,1,1)):
 *   out[b,f,p,c] = x[b,f,p,c] + bias[c] + sum_{k<3} w[c][k] * x[b, f + k - (causal ? 2 : 1), p, c]     (zero outside [0, F))
 * TEMPORAL_ATTENTION — iv.py:499-570 applied along the frame axis (RearrangeTimeCentric, iv.py:257-270): for every clip b, pixel p
 * and head h the F frames of that pixel attend to each other (one shared key/value head) and to the learned null key:
 *   qh = l2norm(q[b,i,p,h,:]) * q_scale * scale ;  kh[0] = l2norm(null_k) * k_scale, kh[1+j] = l2norm(k[b,j,p,:]) * k_scale
 *   sim[i][j'] = qh . kh[j'] + bias[h][i][j']   (bias: [heads][F][F+1] fp32, column 0 = null-key bias, the rest = the generated
 *   relative position bias);  causal: keys with frame index > i are masked;  o[b,i,p,h,:] = softmax_j'(sim) @ [null_v, v[b,:,p,:]]
 * qkv rows (b, f, p) hold q (heads*64) | k (64) | v (64) at row stride ld (fp16); o rows have stride ld_o.  F <= 32. */
typedef struct ImagenTemporalPegParams {
  const void* x; const float* w; const float* bias; void* out;
  int32_t B, F, P, C, causal;
} ImagenTemporalPegParams;
typedef struct ImagenTemporalAttentionParams {
  const void* qkv; const float* null_kv; const float* q_scale; const float* k_scale; const float* bias; void* o;
  int32_t B, F, P, heads, ld, ld_o, causal; float scale;
} ImagenTemporalAttentionParams;

/* QNORM — q[r, h, :] = l2norm(q[r, h, :]) * q_scale * mult   in place (ip.py:559-560, 812-813). */
typedef struct ImagenQnormParams {
  void* q; const float* q_scale;
  int32_t rows, heads, ld; /* row stride in elements; head h at column h*head_dim */
  float mult;
  int32_t head_dim;        /* 64 (0 means 64) or 32 */
} ImagenQnormParams;

/* GCA_PARTIAL / GCA_FINAL — GlobalContext ip.py:945-970 on h (NHWC fp16):
 *   logit[p] = h[p,:].wk + bk ; per-chunk (max, sum exp, sum exp*h[p,:]) -> part[b][chunk][C+2]
 *   final: ctx = softmax-pooled mean; gate = sigmoid(W2 silu(W1 ctx + b1) + b2)   -> gate[b][C] */
typedef struct ImagenGcaPartialParams {
  const void* h; const float* wk; float* part;
  /* optional in-kernel finalisation (replaces the GCA_FINAL launch) when ONE chunk covers the image (small feature maps): the
   * workgroup of image b pools into LDS, runs the squeeze MLP and writes gate[b][C] */
  const float* w1t; const float* b1; const float* w2t; const float* b2; float* gate;
  int32_t B, HW, C, ld, chunks, hidden; float bk;
} ImagenGcaPartialParams;
typedef struct ImagenGcaFinalParams {
  const float* part; const float* w1t; const float* b1; const float* w2t; const float* b2; float* gate;
  int32_t B, C, hidden, chunks; /* w1t: [C][hidden] (= net.0.weight transposed), w2t: [hidden][C] (= net.2.weight transposed) */
  /* phase 0: the whole finalisation, one workgroup per image.  Wide blocks (C * hidden >= 128 Ki: 1-4 MB of squeeze-MLP weights that one
   * workgroup would stream alone) run it as TWO launches over many workgroups: phase 1 merges the chunks and writes
   * hid[b, :] = silu(W1 ctx + b1) (a workgroup per 32 hidden units and image), phase 2 writes gate[b, :] = sigmoid(W2 hid + b2)
   * (a workgroup per 64 channels and image).  hid: fp32 [B][hidden] scratch, required for phases 1 and 2. */
  float* hid; int32_t phase;
} ImagenGcaFinalParams;

/* GATE_RESIDUAL — ResnetBlock tail ip.py:755-757 with identity residual: out = h*gate[b,c] + res
 * (gate NULL -> 1).  Optionally emits rs_out = 1/max(||out||,1e-12) per pixel for the next ChanRMSNorm. */
typedef struct ImagenGateResidualParams {
  const void* h; const float* gate; const void* res; void* out; float* rs_out;
  int32_t rows, rows_per_batch, C, ld_h, ld_res, ld_out;
  int32_t raw_ssq; /* 1: rs_out receives the raw per-pixel sum of squares instead of 1/max(||out||, 1e-12) */
} ImagenGateResidualParams;

/* GCA_TAIL — the tail of an identity ResnetBlock (ip.py:753-757) as ONE launch instead of GCA_FINAL + GATE_RESIDUAL:
 *   gate[b, :] = sigmoid(W2 silu(W1 ctx[b] + b1) + b2),  ctx[b] = the softmax-pooled mean merged from part[b][chunks][C + 2]  (GCA_FINAL's contract;
 *                part == NULL: gate = gate_in[b, :] if gate_in != NULL, else 1)
 *   out[b, p, c] = fp16(h[b, p, c] * gate[b, c] + res[b, p, c])
 * Every workgroup (slab, b) recomputes the gate of its image in LDS (the partial rows and the squeeze MLP are a few KB out of L2) and then
 * streams its slab of pixel rows; the first loads of the slab are issued before the gate arithmetic, so the dependent round trips of the
 * finalisation overlap them.  Optional per-row outputs, all from the STORED fp16 values (what a consumer reads back):
 *   ssq_out[r]  = sum_c out^2                                   (raw sum of squares: ChanRMSNorm statistics of the next Block)
 *   mu_out[r], rs_out[r] = mean_c out, rsqrt(var_c out + eps)   (LayerNorm statistics of a following attention block)
 *   act_out[r, c] = fp16(silu(out * rsqrt(max(ssq, 1e-24)) * act_pa[c]))   — the NEXT Block's block1 input already through its
 *                 ChanRMSNorm -> SiLU (ip.py:683-690; block1 has no scale / shift), so that conv stages its input with no arithmetic.
 * C: a power of two in [8, 512]; hidden: a power of two (launcher-checked); rows are dense per image (row r of image b at (b*HW + r)*ld). */
typedef struct ImagenGcaTailParams {
  const void* h; const void* res; void* out;
  const float* part; const float* w1t; const float* b1; const float* w2t; const float* b2;
  const float* gate_in; float* gate;     /* gate: optional copy of the computed gate [B][C] (written by slab 0 of every image) */
  float* ssq_out; float* mu_out; float* rs_out;
  void* act_out; const float* act_pa;    /* act_pa: fp32 [C] = gamma * sqrt(C) of the next Block's ChanRMSNorm */
  int32_t B, HW, C, hidden, chunks, slabs;
  int32_t ld_h, ld_res, ld_out, ld_act;
  float eps;
} ImagenGcaTailParams;

/* LN_RESIDUAL — to_out LayerNorm + residual ip.py:529-532,1017 / nn.LayerNorm ip.py:1252:
 *   out = (y - mean)*rsqrt(var+eps)*g (+ beta) (+ res) */
typedef struct ImagenLnResidualParams {
  const void* y; const float* g; const float* beta; const void* res; void* out;
  float* ssq_out; /* optional raw per-row sum of squares of the stored output (feeds a following ChanRMSNorm) */
  float* mu_out; float* rs_out; /* optional LayerNorm statistics of the stored output rows: mean, rsqrt(var + eps_out) (feed a following LayerNorm -> GEMM) */
  int32_t rows, C, ld_y, ld_res, ld_out;
  int32_t rows_per_batch, bs_y, bs_res, bs_out; /* row r = (b, rr): address b*bs + rr*ld (rows_per_batch = rows if flat) */
  float eps, eps_out;
} ImagenLnResidualParams;

/* TIME_EMBED — LearnedSinusoidalPosEmb + Linear + SiLU ip.py:654-669, 1213-1217:
 *   hid[b,:] = silu(W [x, sin(2 pi x w), cos(2 pi x w)] + bias),  x = log-SNR condition of batch element b:
 *   x = times[b], or (step_ptr != NULL) the current step's log-SNR  coef[*step_ptr * 8 + 6]  (graph replay). */
typedef struct ImagenTimeEmbedParams {
  const float* times; const float* coef; const int32_t* step_ptr;
  const float* freqs; const float* w; const float* bias; void* hid;
  int32_t B, half_dim, out_dim, ld_hid;
  int32_t steps;   /* rows of coef: *step_ptr is clamped to [0, steps - 1] (0: not clamped) */
} ImagenTimeEmbedParams;

/* SCALE_SHIFT — ResnetBlock time_mlp tail ip.py:738-741 folded with block2's ChanRMSNorm gain, for every
 * ResnetBlock of the network at once (their time-MLPs are batched into one GEMM):
 *   pa[b,i] = gamma_s[i] * (ss[b, idx_scale[i]] + 1) ;  ps[b,i] = ss[b, idx_shift[i]]      i in [0, total_c)
 * gamma_s = block2.norm.gamma * sqrt(C) laid out block after block. */
typedef struct ImagenScaleShiftParams {
  const void* ss; const float* gamma_s; const int32_t* idx_scale; const int32_t* idx_shift;
  float* pa; float* ps;
  int32_t B, total_c, ld_ss;
  int32_t ss_f32;   /* 1: ss holds fp32 rows (LINEAR_F32's output), 0: fp16 rows (an IGEMM's) */
} ImagenScaleShiftParams;

/* LINEAR_F32 — nn.Linear on PER-SAMPLE vectors, fp32 end to end: to_time_cond (ip.py:1207-1210, 1575-1576: t = Linear(time_hiddens) + the text /
 * low-res hiddens) and the time MLPs of all ResnetBlocks (ip.py:715-718, 738-741: SiLU -> Linear, batched into one matrix).  These layers see
 * ONE row per sample and what they produce — the (scale, shift) of every Block — multiplies every pixel of that sample: an fp16 rounding of
 * the row (of t, of SiLU(t) as an MFMA operand, of the scale / shift rows) is not noise that averages over pixels but one error vector for
 * the whole map.  Plan interpreter, README unet1, null rows of three draws: 0.957 / 1.012 / 0.91e-3 with fp16 rows -> 0.890 / 0.951 / 0.893e-3
 * with fp32 rows; the FLOPs are nothing (K, Cout <= a few thousand, R rows; once per request for all steps in table mode).
 *   y[r, o] = bias[o] + sum_k f(x[r, k]) * wt[k, o]  (+ res[r, o]);   f = identity | SiLU (act_in);  fp32 accumulation, k ascending
 * x: fp16 rows (x_f32 = 0) or fp32 rows (1), ld_x elements apart;  wt: the module's weight TRANSPOSED, fp32 [K][Cout] (coalesced over o);
 * bias: fp32 [Cout] or null;  res: fp16 rows (ld_res) or null;  y: fp32 rows (ld_y). */
typedef struct ImagenLinearF32Params {
  const void* x; const float* wt; const float* bias; const void* res; float* y;
  int32_t rows, K, Cout, ld_x, ld_res, ld_y, x_f32, act_in;
} ImagenLinearF32Params;

/* PACK_IMAGE — x (+ lowres_cond_img) fp32 NCHW -> fp16 NHWC [B,H,W,Cpad] (ip.py:1550-1551 concat). */
typedef struct ImagenPackImageParams {
  const float* a; const float* b; void* out;
  int32_t B, Brep, H, W, Ca, Cb, Cpad;
} ImagenPackImageParams;

/* CFG_X0 — ip.py:1522 + 2085-2092: out = null + (cond-null)*s ; x0 = (x - sigma*out)/max(alpha,1e-8) (objective 0 = noise,
 * ip.py:314-318) | out (1 = x_start) | alpha*x - sigma*out (2 = v, ip.py:308-312);
 * writes x0 and |x0| (for the quantile).  pred is [2B,...] (cond first) when cfg != 0, else [B,...]. */
typedef struct ImagenCfgX0Params {
  const float* x; const float* pred; const float* coef; const int32_t* step_ptr; float* x0; float* absx0;
  int32_t B, n_per_sample, cfg, objective; float cond_scale;
} ImagenCfgX0Params;

/* QUANTILE — torch.quantile(|x0| per sample, q) ip.py:2097-2101 (linear interpolation), exact. */
typedef struct ImagenQuantileParams {
  const float* absx0; float* out; uint32_t* scratch; /* scratch: B * IMAGEN_QUANTILE_SCRATCH_WORDS uint32; must arrive
                                                      * cleared (all 0, word 1025 of each sample = 0xFFFFFFFF); the op leaves it cleared again */
  int32_t B, n; float q;
} ImagenQuantileParams;
#define IMAGEN_QUANTILE_SCRATCH_WORDS (4 * 256 + 8)

/* DDPM_UPDATE — ip.py:2103-2105, 252-270, 2160-2164 and final clamp/unnormalise ip.py:2281-2289.
 * coef table row (per step): [alpha, sigma, alpha_next, sigma_next, c, nonzero, log_snr, pad]. */
typedef struct ImagenDdpmUpdateParams {
  float* x; const float* x0; const float* quant; const float* coef; const float* noise; float* final_out;
  int32_t* step_ptr; /* device step counter: read by every sampler kernel of the step, incremented at the end */
  const uint32_t* seed_ptr; /* optional device [2] Philox key (overrides seed_lo/hi): lets one captured graph serve every sample() call */
  int32_t B, n_per_sample, dynamic_threshold, total_steps;
  int32_t sample_offset; /* global index of local sample 0 (batch sharding: noise is keyed by the global sample index) */
  uint32_t seed_lo, seed_hi, stream_id;
  int32_t no_advance; /* != 0: leave *step_ptr alone (inpainting: a LINCOMB re-noising step of the same table row follows, ip.py:2268-2275) */
  float* x0_thr;      /* optional: the thresholded x0 of this step (ip.py:2094-2107), same layout as x0 — the next step's self-conditioning
                       * input of a Unet(self_cond=True) (ip.py:2249, 1541-1543) */
  const uint32_t* row_keys; /* optional device [B][4]: {Philox key lo, key hi, global sample index, 0} of every row; overrides seed_ptr / seed_lo /
                             * seed_hi and sample_offset + row.  Several sample() requests merged into one batch (Imagen.sample_requests) then draw,
                             * row by row, exactly the noise each would draw alone (its own seed, its own sample indices) */
} ImagenDdpmUpdateParams;

/* RANDN — out[b, i] ~ N(0,1), Philox4x32-10 counter (i/4, tag, stream_id, sample_offset + b), key = seed.
 * Replaces torch.randn at ip.py:2195 (initial image) and ip.py:2449 (low-res augmentation noise). */
typedef struct ImagenRandnParams {
  float* out;
  int32_t B, n_per_sample, sample_offset;
  uint32_t seed_lo, seed_hi, stream_id, tag;
} ImagenRandnParams;

/* LOWRES_PREP — ip.py:2446-2449: nearest-neighbour resize (F.interpolate 'nearest') of the previous stage's [0,1] image,
 * normalise to [-1,1], noise-augment: out = alpha * (2*img - 1) + sigma * noise   (fp32 NCHW). */
typedef struct ImagenLowresPrepParams {
  const float* img; const float* noise; float* out;
  int32_t B, C, Hin, Win, Hout, Wout; float alpha, sigma;
} ImagenLowresPrepParams;

/* LINCOMB — the elementwise state updates of the ElucidatedImagen sampler (elucidated_imagen.py:481-540), one launch each:
 *   out = w0*t0 + w1*thr(t1, q1) + w2*t2 + w3*thr(t3, q3) + w4*z,   out2 = w5*out,   z ~ N(0,1) (Philox, keyed by the step)
 * with the six weights read from row *step_ptr of a device table coef[rows][8] (so one captured graph serves every step):
 *   x_hat = x + sqrt(sigma_hat^2 - sigma^2)*S_noise*z, c_in*x_hat         (:489-494, :360)
 *   Euler:  x_next = x_hat + (sigma_next - sigma_hat) * (x_hat - x0)/sigma_hat                      (:509-511)
 *   Heun :  x = x_hat + 0.5*(sigma_next - sigma_hat)*((x_hat - x0)/sigma_hat + (x_next - x0')/sigma_next)   (:528-529)
 * thr(t, q) = threshold_x_start (:309-321): thr_mode 1: clamp(t, -s, s)/s with s = max(q[b], 1); 2: clamp(t, -1, 1); 0: t.
 * final != 0: final_out = (clamp(out, -1, 1) + 1)/2 (:540, unnormalize_img).  advance != 0: *step_ptr += 1 afterwards.
 * mask != NULL: out = mask[i] != 0 ? (the sum above) : mask_else[i] — the inpainting blend `img * ~mask + q_sample(inpaint) * mask`
 * (ip.py:2244-2246) with t0 = the known image, w0 = alpha_t, w4 = sigma_t; the same op with t0 = x re-noises x_{t_next} -> x_t
 * (q_sample_from_to, ip.py:286-307).  In-place use (out == mask_else or out == t0) is allowed. */
typedef struct ImagenLincombParams {
  const float* t0; const float* t1; const float* t2; const float* t3;  /* fp32 [B, n_per_sample]; t1..t3 may be NULL */
  const float* q1; const float* q3;   /* [B] quantiles for thr_mode 1 */
  float* out; float* out2; float* final_out;  /* out2 / final_out may be NULL */
  const float* coef; int32_t* step_ptr;
  const uint32_t* seed_ptr;  /* optional device [2] Philox key (overrides seed_lo/hi) */
  int32_t B, n_per_sample, thr_mode, final, advance, sample_offset;
  uint32_t seed_lo, seed_hi, stream_id;
  const float* mask; const float* mask_else;  /* optional fp32 [B, n_per_sample] 0/1 mask and the image kept where it is 0 */
} ImagenLincombParams;

/* STEP_SLICE — the part of the denoiser that depends on the timestep and the request's conditioning but NOT on x_t: time embedding ->
 * time conditioning / time tokens (ip.py:1573-1578, 1588, 1660) -> every ResnetBlock's time-MLP scale / shift (ip.py:738-741) and the
 * time tokens' key / value projections of every attention site (ip.py:527, 783).  The sampler evaluates it for ALL steps of a stage in
 * one batched pass per request (tables [steps][words_k x 16 bytes], rows in the order of the sampler's coefficient table) and each
 * step copies its row into the buffers the step's kernels read:   dst_k[0 .. 16 words_k) = src_k[*step_ptr * 16 words_k ..),  k < 4. */
typedef struct ImagenStepSliceParams {
  const void* src0; const void* src1; const void* src2; const void* src3;
  void* dst0; void* dst1; void* dst2; void* dst3;
  const int32_t* step_ptr;
  int32_t words0, words1, words2, words3;   /* 16-byte words per step of each segment; 0 = segment unused */
  int32_t steps;                            /* rows of the tables: *step_ptr is clamped to [0, steps - 1] (0: not clamped) — a replay past the
                                             * last step (a warm-up launch, a counter left over from a longer schedule) reads the last row, not beyond */
} ImagenStepSliceParams;

/* ROWS_COPY — dst[b, r0 + r, :C] = src[b (or 0), r, :C]  (fp16). */
typedef struct ImagenRowsCopyParams {
  const void* src; void* dst;
  int32_t B, rows, C, src_bs, src_rs, dst_bs, dst_rs;
} ImagenRowsCopyParams;

/* SELECT_ROWS — text keep-mask select ip.py:1599-1632:
 *   dst[r, l, :] = (keep[r] && (mask == NULL || mask[src[r], l])) ? a[src[r], l, :] : nul[l, :]      (fp16 rows of C) */
typedef struct ImagenSelectRowsParams {
  const void* a; const void* nul; const uint8_t* mask; const int32_t* src; const uint8_t* keep; void* dst;
  int32_t R, L, C;
} ImagenSelectRowsParams;

/* MEAN_ROWS — out[b, :] = mean_r x[b, r, :]  (masked_mean with an all-true mask ip.py:490; text_tokens.mean ip.py:1640). */
typedef struct ImagenMeanRowsParams {
  const void* x; void* out;
  int32_t B, rows, C, bs_x, ld_x, ld_out;
} ImagenMeanRowsParams;

typedef struct ImagenMemset32Params { void* dst; uint32_t value; int32_t count; } ImagenMemset32Params;

/* ROWCHAIN — the small-map execution unit of the token path (round 5).  On maps of <= 32^2 pixels the transformer / cross-attention layers
 * of the denoiser are chains of launches that each take ~10 us for a few MFLOP per row: latency, not work.  Every layer below maps a token
 * row to a token row (the one exception, the keys / values of a cross-attention, are constants of the image), so a tile of 32 | 64 rows of ONE
 * image flows through the whole chain inside one workgroup — GEMM stages on the matrix pipe (weights straight from L2 into A fragments,
 * rows from LDS), the LayerNorms / residuals as row passes over the LDS tile between them — with the SAME rounding points as the launches
 * it replaces (fp16 wherever those stored fp16).  Rows are dense: row r at base + r * ld; rows_per_batch % tile == 0.
 *   mode 1, FF   (ip.py:529-532, 1017, 972-980, 1018; replaces to_out IGEMM -> LN_RESIDUAL -> lin1 IGEMM -> ROWSTAT -> lin2 IGEMM):
 *        x1  = fp16(LN(fp16(o W_out^T)) * g0 + res)                                    o = x [rows][inner] (the attention output)
 *        hid = fp16(gelu(fp16((x1 - mean x1) * rstd x1 * g1) W1^T))                                        [rows][hidden]
 *        out = fp16(fp16((hid - mean hid) * rstd hid * g2) W2^T + x1) ;  ssq_out[r] = sum_c out^2
 *   mode 2, XATTN (ip.py:759-834 inside a ResnetBlock, 745-751; replaces ROWSTAT -> to_q IGEMM -> ATTENTION -> to_out IGEMM -> LN_RESIDUAL):
 *        q   = fp16(fp16((x - mean x) * rstd x * g0) Wq^T)           [rows][heads * 64];  (mu, rs): the caller's statistics or computed here
 *        o_h = fp16(softmax_j(q^_h . K^[b, h, j]) @ V[b, h])          q^_h = fp16(q_h / max(|q_h|, 1e-12) * q_scale * q_mult), j < J (ATTENTION's contract)
 *        out = fp16(LN(fp16(o W_out^T)) * g1 + x) ;  ssq_out
 *   mode 3, QKV  (ip.py:521-561; replaces (ROWSTAT ->) qkv IGEMM -> KV_PREP of the self-attention rows):
 *        y   = fp16(fp16((x - mean x) * rstd x * g0) [Wq | Wkv]^T)    [rows][heads * 64 + 128];  out[r, : heads * 64] = q
 *        K^[b, r0 + n, :] = fp16(k / max(|k|, 1e-12) * k_scale),  V^T[b, :, r0 + n] = v      (KV_PREP's contract, one shared k / v head)
 *   mode 4, RESPREP (ip.py:741, 753-757 + the NEXT block's 683-690; replaces the up path's res_conv IGEMM -> the next Block's ACT_PREP):
 *        out = fp16(concat(x, x2) Wres^T + bias + addend * gate[b, :]) ;   (gate NULL: + addend)  ssq_out[r] = sum_c out^2            the ResnetBlock's `h * gate + res_conv(x)`
 *        prep_out[r, :] = fp16(silu(concat(out, prep_x2)[r, :] * rsqrt(max(ssq_out[r] + prep_ssq_wb * prep_ssq_b[r], 1e-24)) * prep_pa[:]))   (optional:
 *        ACT_PREP's contract with pstride 0 — the next Block's ChanRMSNorm over ITS concatenated input -> SiLU, written by the producer of `out`)
 * Weights: imagen_pack_igemm_weights() buffers of 1x1 layers with Cin % 32 == 0 (w_cout_pad* = their Cout_pad), bias-free but RESPREP's.
 * Launcher limits: modes 1-3: heads * 64 == 512; C (the layer width / RESPREP's Cout) a power of two in 32 .. 256, hidden a power of two <= 512;
 * RESPREP: C1 + C2 <= 512 in 32-channel chunks, prep_C2 % 8 == 0; tile = 64 rows when `tile64`. */
enum { IMAGEN_CHAIN_FF = 1, IMAGEN_CHAIN_XATTN = 2, IMAGEN_CHAIN_QKV = 3, IMAGEN_CHAIN_RESPREP = 4 };
typedef struct ImagenRowchainParams {
  const void* x; const void* res; void* out;
  const void* w0; const void* w1; const void* w2;      /* FF: W_out, W1, W2;  XATTN: Wq, W_out;  QKV: [Wq | Wkv] */
  const float* g0; const float* g1; const float* g2;   /* FF: to_out LN gain, FeedForward LN gains;  XATTN: norm gain, to_out LN gain;  QKV: norm gain */
  const float* mu; const float* rs;                     /* XATTN / QKV: optional LayerNorm statistics of the x rows from their producer (both or neither) */
  void* khat; void* vt;                                 /* XATTN: the site's K^ / V^T operand buffers (read);  QKV: the buffers to fill */
  const float* q_scale; const float* k_scale;
  float* ssq_out;
  /* RESPREP: x2 = the second (skip) input of the GEMM, bias [C], addend rows [rows][C] with gate [B][C] (gate NULL: 1); prep_*: the next Block's
   * second input rows, its per-row sum of squares, the per-channel gain [C + prep_C2] and the activated output rows [rows][C + prep_C2] */
  const void* x2; const float* bias; const void* addend; const float* gate;
  const void* prep_x2; const float* prep_ssq_b; const float* prep_pa; void* prep_out;
  int32_t mode, rows, rows_per_batch, C, inner, hidden, heads, J;
  int32_t ld_x, ld_res, ld_out;
  int32_t k_bs, k_hs, k_rs, vt_bs, vt_hs, vt_ds;        /* operand buffer strides (elements), as ImagenAttentionParams / ImagenKvPrepParams */
  int32_t r0;                                           /* QKV: first key row of the tile rows' keys (behind the context and null rows) */
  int32_t w_cout_pad0, w_cout_pad1, w_cout_pad2;
  int32_t tile64;                                       /* 1: 64-row tiles (rows_per_batch % 64 == 0): every weight fragment feeds two MFMAs (XATTN: always 32) */
  int32_t C2, ld_x2, ld_add, gate_stride, prep_C2, ld_prep_x2, ld_prep;   /* RESPREP (inner = C1, the channels of x) */
  float eps, q_mult, prep_ssq_wb;
} ImagenRowchainParams;

/* one entry of a plan: params_bytes = the caller's sizeof(params struct of `kind`) — checked against imagen_sizeof(kind) on every run, so a binding
 * built against another version of this header is refused instead of being read past its end */
typedef struct ImagenOpRef { int32_t kind; int32_t params_bytes; const void* params; } ImagenOpRef;

/* ---- entry points ------------------------------------------------------------------------------ */
int imagen_abi_version(void);
const char* imagen_last_error(void);
size_t imagen_sizeof(int kind);                       /* sizeof the params struct of an op kind */
/* launch one op: params_bytes must equal imagen_sizeof(kind) (the caller's sizeof of its mirror of the struct), else -1 and nothing is launched */
int imagen_launch(int kind, const void* params, size_t params_bytes, imagen_stream_t stream);
int imagen_plan_run(const ImagenOpRef* ops, int n, imagen_stream_t stream);

/* igemm tile selection + packing.  cfg ids are stable; *_tile_* describe a cfg. */
int imagen_igemm_num_configs(void);
int imagen_igemm_config_info(int cfg, int* tile_pixels, int* tile_cout, int* kgroups /* G: 8-channel groups per k-chunk */);
/* 16-byte staging slots per producer thread the instantiation (cfg, KHxKW kernel) holds: a tile shape is launchable iff
 * halo_pixels * G <= slots * 256 */
int imagen_igemm_stage_slots(int cfg, int KH, int KW);
/* kernel family of a tile cfg: 0 = wave-specialised persistent kernel (weights streamed from L2 per wave; every kernel size and
 * stride), (1 = the LDS-staged kernel with an in-kernel prologue, retired in round 3,)
 * 2 = all-DMA kernel (3x3 stride 1, one input tensor, NO prologue: both operands by direct-to-LDS loads; fixed tile shape per cfg),
 * 3 = streaming kernel (3x3 stride 1 to <= 32 channels from one or two 32-channel inputs, persistent, in-LDS prologue),
 * 4 = streaming pointwise kernel (1x1, raw inputs, weights in registers; `kgroups` of its config info = 32-channel input chunks),
 * 5 = big-tile all-DMA kernel (as 2, 128-cout tiles of 256 / 128 pixels, 64 x 64 per wave, one workgroup per CU),
 * 6 = streaming kernel with the Block prologue on register-staged rows (3x3 stride 1 to exactly 32 channels from one or two 32-channel
 *     inputs, raw or with the ssq-statistics prologue, plain / post_pa / ssq_out epilogue; weights in registers, 8 x 16 tiles),
 * 7 = tiled pointwise GEMM (1x1 stride 1, inputs in 32-channel chunks, raw or with the (x - mu) * rs * pa + ps prologue, every epilogue;
 *     128-pixel x 128-cout workgroup tiles, K loop with both operands register-staged two chunks ahead),
 * 8 = small-map 3x3 convolution (stride 1, pad 1, G = 4 packing, C1 + C2 == Cin_pad; 4x8 / 2x16 / 1x32 tiles of 32 pixels x 32 | 64 | 128 couts,
 *     the K loop split over the waves of the workgroup; every prologue and epilogue of the contract). */
int imagen_igemm_config_family(int cfg);
int imagen_igemm_config_ring(int cfg);   /* weight look-ahead ring depth in stages (family 2; 0 for the others) */
/* dynamic LDS bytes of a launch of `cfg` with a KHxKW kernel at `stride` and a THxTW output tile; -1 = not launchable */
long imagen_igemm_lds_bytes(int cfg, int KH, int KW, int stride, int TH, int TW);
/* Host-side pack: w_in fp32 [Cout][Cin][KH][KW] (Conv2d / Linear layout, HOST memory) -> packed fp16 (HOST memory)
 * in MFMA fragment order.  G = 8-channel groups per k-chunk (1, 2 or 4; must match the tile cfg used at launch),
 * Cout_pad = multiple of 128 (any tile cfg can then consume it).  in_scale (optional, [Cin]) is folded into W. */
size_t imagen_igemm_packed_elems(int G, int Cin, int Cout_pad, int KH, int KW);
int imagen_pack_igemm_weights(int G, const float* w_in, const float* in_scale, int Cin, int Cout, int Cout_pad, int KH, int KW,
                              uint16_t* w_out /* fp16 bits */);

/* hipGraph helpers (per-timestep capture). */
int imagen_graph_begin(imagen_stream_t stream);
int imagen_graph_end(imagen_stream_t stream, void** graph_exec_out);
int imagen_graph_launch(void* graph_exec, imagen_stream_t stream);
int imagen_graph_destroy(void* graph_exec);

/* HIP-event timing on an explicit stream (bench.py roofline leg). */
int imagen_event_create(void** ev);
int imagen_event_record(void* ev, imagen_stream_t stream);
int imagen_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int imagen_event_destroy(void* ev);

/* Calibration probes (bench.py `calibration`; not on the sampling path): a device copy of `bytes` (multiple of 16) repeated `reps`
 * times -> GB/s of read + written bytes; a dense v_mfma_f32_32x32x16_f16 loop on every CU -> TFLOP/s.  Both time themselves with HIP
 * events on `stream` and synchronise it. */
int imagen_probe_copy(void* dst, const void* src, size_t bytes, int reps, imagen_stream_t stream, float* gbs_out);
int imagen_probe_mfma(int iters, int reps, float* sink, imagen_stream_t stream, float* tflops_out);
/* Latency-side probes (round 5: the sampling path is dependent-latency-bound, and boxes that agree on the two throughput probes above differ
 * by 20 % on it).  imagen_probe_latency: one lane follows `hops` links of a chain of 128-byte nodes (word 0 of node i = index of the next node; the
 * caller builds ONE random cycle over all nodes) -> ns per dependent load; the working set picks L2 / Infinity Cache / HBM.
 * imagen_probe_launch_chain: `n` dependent one-wave launches (three kernel symbols in rotation) captured into a hipGraph, replayed `reps`
 * times -> us per dependent launch inside a graph.  Both time themselves with HIP events on `stream` and synchronise it. */
int imagen_probe_latency(const void* nodes, int hops, void* out_word, imagen_stream_t stream, float* ns_per_hop);
int imagen_probe_launch_chain(int n, int reps, void* counter_word, imagen_stream_t stream, float* us_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* IMAGEN_HIP_H */
