// capi.hip — the extern "C" surface of libimagen_hip.so (include/imagen_hip.h): op dispatch, plan
// execution, hipGraph capture helpers, HIP-event timing, error reporting.
#include <cstdarg>
#include <cstdio>
#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void imagen_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int imagen_abi_version(void) { return IMAGEN_ABI_VERSION; }
extern "C" const char* imagen_last_error(void) { return g_err; }

extern "C" size_t imagen_sizeof(int kind) {
  switch (kind) {
    case IMAGEN_OP_IGEMM: return sizeof(ImagenIgemmParams);
    case IMAGEN_OP_ROWSTAT: return sizeof(ImagenRowstatParams);
    case IMAGEN_OP_ATTENTION: return sizeof(ImagenAttentionParams);
    case IMAGEN_OP_KV_PREP: return sizeof(ImagenKvPrepParams);
    case IMAGEN_OP_QNORM: return sizeof(ImagenQnormParams);
    case IMAGEN_OP_GCA_PARTIAL: return sizeof(ImagenGcaPartialParams);
    case IMAGEN_OP_GCA_FINAL: return sizeof(ImagenGcaFinalParams);
    case IMAGEN_OP_GATE_RESIDUAL: return sizeof(ImagenGateResidualParams);
    case IMAGEN_OP_LN_RESIDUAL: return sizeof(ImagenLnResidualParams);
    case IMAGEN_OP_TIME_EMBED: return sizeof(ImagenTimeEmbedParams);
    case IMAGEN_OP_SCALE_SHIFT: return sizeof(ImagenScaleShiftParams);
    case IMAGEN_OP_PACK_IMAGE: return sizeof(ImagenPackImageParams);
    case IMAGEN_OP_CFG_X0: return sizeof(ImagenCfgX0Params);
    case IMAGEN_OP_QUANTILE: return sizeof(ImagenQuantileParams);
    case IMAGEN_OP_DDPM_UPDATE: return sizeof(ImagenDdpmUpdateParams);
    case IMAGEN_OP_ROWS_COPY: return sizeof(ImagenRowsCopyParams);
    case IMAGEN_OP_MEMSET32: return sizeof(ImagenMemset32Params);
    case IMAGEN_OP_SELECT_ROWS: return sizeof(ImagenSelectRowsParams);
    case IMAGEN_OP_MEAN_ROWS: return sizeof(ImagenMeanRowsParams);
    case IMAGEN_OP_RANDN: return sizeof(ImagenRandnParams);
    case IMAGEN_OP_LOWRES_PREP: return sizeof(ImagenLowresPrepParams);
    case IMAGEN_OP_LINCOMB: return sizeof(ImagenLincombParams);
    case IMAGEN_OP_KV_PREP_MULTI: return sizeof(ImagenKvPrepMultiParams);
    case IMAGEN_OP_TEMPORAL_PEG: return sizeof(ImagenTemporalPegParams);
    case IMAGEN_OP_TEMPORAL_ATTENTION: return sizeof(ImagenTemporalAttentionParams);
    case IMAGEN_OP_ACT_PREP: return sizeof(ImagenActPrepParams);
    case IMAGEN_OP_GCA_TAIL: return sizeof(ImagenGcaTailParams);
    case IMAGEN_OP_STEP_SLICE: return sizeof(ImagenStepSliceParams);
    case IMAGEN_OP_ROWCHAIN: return sizeof(ImagenRowchainParams);
    case IMAGEN_OP_LINEAR_F32: return sizeof(ImagenLinearF32Params);
    default: return 0;
  }
}

extern "C" int imagen_launch(int kind, const void* params, size_t params_bytes, imagen_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (!params) { imagen_set_error("imagen_launch: null params (kind %d)", kind); return -1; }
  if (imagen_sizeof(kind) != 0 && params_bytes != imagen_sizeof(kind)) {
    imagen_set_error("imagen_launch: op kind %d takes a %zu-byte params struct, the caller passed %zu bytes (a binding built against another include/imagen_hip.h?)",
                     kind, imagen_sizeof(kind), params_bytes);
    return -1;
  }
  switch (kind) {
    case IMAGEN_OP_IGEMM: return launch_igemm(static_cast<const ImagenIgemmParams*>(params), s);
    case IMAGEN_OP_ROWSTAT: return launch_rowstat(static_cast<const ImagenRowstatParams*>(params), s);
    case IMAGEN_OP_ATTENTION: return launch_attention(static_cast<const ImagenAttentionParams*>(params), s);
    case IMAGEN_OP_KV_PREP: return launch_kv_prep(static_cast<const ImagenKvPrepParams*>(params), s);
    case IMAGEN_OP_QNORM: return launch_qnorm(static_cast<const ImagenQnormParams*>(params), s);
    case IMAGEN_OP_GCA_PARTIAL: return launch_gca_partial(static_cast<const ImagenGcaPartialParams*>(params), s);
    case IMAGEN_OP_GCA_FINAL: return launch_gca_final(static_cast<const ImagenGcaFinalParams*>(params), s);
    case IMAGEN_OP_GATE_RESIDUAL: return launch_gate_residual(static_cast<const ImagenGateResidualParams*>(params), s);
    case IMAGEN_OP_LN_RESIDUAL: return launch_ln_residual(static_cast<const ImagenLnResidualParams*>(params), s);
    case IMAGEN_OP_TIME_EMBED: return launch_time_embed(static_cast<const ImagenTimeEmbedParams*>(params), s);
    case IMAGEN_OP_SCALE_SHIFT: return launch_scale_shift(static_cast<const ImagenScaleShiftParams*>(params), s);
    case IMAGEN_OP_PACK_IMAGE: return launch_pack_image(static_cast<const ImagenPackImageParams*>(params), s);
    case IMAGEN_OP_CFG_X0: return launch_cfg_x0(static_cast<const ImagenCfgX0Params*>(params), s);
    case IMAGEN_OP_QUANTILE: return launch_quantile(static_cast<const ImagenQuantileParams*>(params), s);
    case IMAGEN_OP_DDPM_UPDATE: return launch_ddpm_update(static_cast<const ImagenDdpmUpdateParams*>(params), s);
    case IMAGEN_OP_ROWS_COPY: return launch_rows_copy(static_cast<const ImagenRowsCopyParams*>(params), s);
    case IMAGEN_OP_MEMSET32: return launch_memset32(static_cast<const ImagenMemset32Params*>(params), s);
    case IMAGEN_OP_SELECT_ROWS: return launch_select_rows(static_cast<const ImagenSelectRowsParams*>(params), s);
    case IMAGEN_OP_MEAN_ROWS: return launch_mean_rows(static_cast<const ImagenMeanRowsParams*>(params), s);
    case IMAGEN_OP_RANDN: return launch_randn(static_cast<const ImagenRandnParams*>(params), s);
    case IMAGEN_OP_LOWRES_PREP: return launch_lowres_prep(static_cast<const ImagenLowresPrepParams*>(params), s);
    case IMAGEN_OP_LINCOMB: return launch_lincomb(static_cast<const ImagenLincombParams*>(params), s);
    case IMAGEN_OP_KV_PREP_MULTI: return launch_kv_prep_multi(static_cast<const ImagenKvPrepMultiParams*>(params), s);
    case IMAGEN_OP_TEMPORAL_PEG: return launch_temporal_peg(static_cast<const ImagenTemporalPegParams*>(params), s);
    case IMAGEN_OP_TEMPORAL_ATTENTION: return launch_temporal_attention(static_cast<const ImagenTemporalAttentionParams*>(params), s);
    case IMAGEN_OP_ACT_PREP: return launch_act_prep(static_cast<const ImagenActPrepParams*>(params), s);
    case IMAGEN_OP_GCA_TAIL: return launch_gca_tail(static_cast<const ImagenGcaTailParams*>(params), s);
    case IMAGEN_OP_STEP_SLICE: return launch_step_slice(static_cast<const ImagenStepSliceParams*>(params), s);
    case IMAGEN_OP_ROWCHAIN: return launch_rowchain(static_cast<const ImagenRowchainParams*>(params), s);
    case IMAGEN_OP_LINEAR_F32: return launch_linear_f32(static_cast<const ImagenLinearF32Params*>(params), s);
    default: imagen_set_error("imagen_launch: unknown op kind %d", kind); return -1;
  }
}

extern "C" int imagen_plan_run(const ImagenOpRef* ops, int n, imagen_stream_t stream) {
  for (int i = 0; i < n; ++i) {
    const int rc = imagen_launch(ops[i].kind, ops[i].params, (size_t)ops[i].params_bytes, stream);
    if (rc != 0) {
      char msg[400];
      snprintf(msg, sizeof(msg), "%s", g_err);
      imagen_set_error("plan op %d (kind %d): %s", i, ops[i].kind, msg);
      return rc;
    }
  }
  return 0;
}

#define HIP_TRY(expr, what)                                                   \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      imagen_set_error("%s: %s", what, hipGetErrorString(e_));                \
      return (int)e_;                                                         \
    }                                                                         \
  } while (0)

extern "C" int imagen_graph_begin(imagen_stream_t stream) {
  HIP_TRY(hipStreamBeginCapture(reinterpret_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
  return 0;
}

extern "C" int imagen_graph_end(imagen_stream_t stream, void** graph_exec_out) {
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamEndCapture(reinterpret_cast<hipStream_t>(stream), &graph), "hipStreamEndCapture");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { imagen_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return (int)e; }
  *graph_exec_out = exec;
  return 0;
}

extern "C" int imagen_graph_launch(void* graph_exec, imagen_stream_t stream) {
  HIP_TRY(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), reinterpret_cast<hipStream_t>(stream)), "hipGraphLaunch");
  return 0;
}

extern "C" int imagen_graph_destroy(void* graph_exec) {
  HIP_TRY(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)), "hipGraphExecDestroy");
  return 0;
}

extern "C" int imagen_event_create(void** ev) {
  hipEvent_t e;
  HIP_TRY(hipEventCreate(&e), "hipEventCreate");
  *ev = e;
  return 0;
}
extern "C" int imagen_event_record(void* ev, imagen_stream_t stream) {
  HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), reinterpret_cast<hipStream_t>(stream)), "hipEventRecord");
  return 0;
}
extern "C" int imagen_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
  HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(ev_stop)), "hipEventSynchronize");
  HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(ev_start), reinterpret_cast<hipEvent_t>(ev_stop)), "hipEventElapsedTime");
  return 0;
}
extern "C" int imagen_event_destroy(void* ev) {
  HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)), "hipEventDestroy");
  return 0;
}
