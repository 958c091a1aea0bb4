"""ctypes binding of libimagen_hip.so (C ABI: include/imagen_hip.h).

The params structs are mirrored by parsing the header itself, so the Python side can
never drift from the C side; `imagen_sizeof(kind)` is checked against every mirror at
load time.  There is deliberately NO fallback: if the shared library is missing or
does not load, importing the product path raises (the HIP path is the product).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "imagen_hip.h")
LIB_PATH = os.environ.get("IMAGEN_LIB_PATH") or os.path.join(HERE, "libimagen_hip.so")   # override: A/B builds of the kernel library

# Every IMAGEN_* variable anything in this repository still reads.  The A/B switches of earlier rounds became module constants (ops.CONV_DMA,
# engine.BIG_PREP, ... — a test or tool monkeypatches them); a script of those rounds that still exports one would silently compare two identical
# configurations, so an unknown IMAGEN_* variable is an error, not a no-op.
KNOWN_ENV = {"IMAGEN_LIB_PATH", "IMAGEN_TIMING", "IMAGEN_TIME_TABLE", "IMAGEN_TIME_TABLE_MAX_GB", "IMAGEN_CONV_PRO", "IMAGEN_CONV_GEMM",
             "IMAGEN_CONV_SMALL", "IMAGEN_ROWCHAIN", "IMAGEN_EMUL_TESTS", "IMAGEN_BENCH_LANES", "IMAGEN_BENCH_MODE", "IMAGEN_VIDEO_GPU_TESTS"}
_stale = sorted(k for k in os.environ if k.startswith("IMAGEN_") and k not in KNOWN_ENV)
if _stale:
    raise RuntimeError(f"{', '.join(_stale)}: not read by this version (switches of earlier rounds are module constants of imagen_pytorch_amd.ops / "
                       f".engine now: monkeypatch them); known variables: {', '.join(sorted(KNOWN_ENV))}")

_CTYPE = {
    "int32_t": ctypes.c_int32,
    "uint32_t": ctypes.c_uint32,
    "uint8_t": ctypes.c_uint8,
    "float": ctypes.c_float,
}


def _parse_header(path: str):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    enums: Dict[str, int] = {}
    for m in re.finditer(r"enum\s*\w*\s*\{(.*?)\}", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = [s.strip() for s in item.split("=")]
                nxt = int(val, 0)
            else:
                name = item
            enums[name] = nxt
            nxt += 1
    for m in re.finditer(r"#define\s+(IMAGEN_\w+)\s+\(?([0-9x*+ ]+)\)?\s*$", text, flags=re.M):
        try:
            enums[m.group(1)] = int(eval(m.group(2), {}, {}))  # noqa: S307 - numeric literals only
        except Exception:
            pass
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            if "*" in decl:
                # one pointer per declaration: "<qualifiers> type* name"
                fname = decl.split("*")[-1].strip()
                fields.append((fname, ctypes.c_void_p))
            else:
                parts = decl.split(" ", 1)
                ctype = _CTYPE[parts[0]]
                for fname in parts[1].split(","):
                    fields.append((fname.strip(), ctype))
        structs[name] = fields
    return enums, structs


ENUMS, _STRUCT_FIELDS = _parse_header(HEADER)


def _make_struct(name, fields):
    return type(name, (ctypes.Structure,), {"_fields_": fields})


STRUCTS = {name: _make_struct(name, fields) for name, fields in _STRUCT_FIELDS.items()}
globals().update(STRUCTS)

OP_STRUCT = {
    ENUMS["IMAGEN_OP_IGEMM"]: STRUCTS["ImagenIgemmParams"],
    ENUMS["IMAGEN_OP_ROWSTAT"]: STRUCTS["ImagenRowstatParams"],
    ENUMS["IMAGEN_OP_ATTENTION"]: STRUCTS["ImagenAttentionParams"],
    ENUMS["IMAGEN_OP_KV_PREP"]: STRUCTS["ImagenKvPrepParams"],
    ENUMS["IMAGEN_OP_QNORM"]: STRUCTS["ImagenQnormParams"],
    ENUMS["IMAGEN_OP_GCA_PARTIAL"]: STRUCTS["ImagenGcaPartialParams"],
    ENUMS["IMAGEN_OP_GCA_FINAL"]: STRUCTS["ImagenGcaFinalParams"],
    ENUMS["IMAGEN_OP_GATE_RESIDUAL"]: STRUCTS["ImagenGateResidualParams"],
    ENUMS["IMAGEN_OP_LN_RESIDUAL"]: STRUCTS["ImagenLnResidualParams"],
    ENUMS["IMAGEN_OP_TIME_EMBED"]: STRUCTS["ImagenTimeEmbedParams"],
    ENUMS["IMAGEN_OP_SCALE_SHIFT"]: STRUCTS["ImagenScaleShiftParams"],
    ENUMS["IMAGEN_OP_PACK_IMAGE"]: STRUCTS["ImagenPackImageParams"],
    ENUMS["IMAGEN_OP_CFG_X0"]: STRUCTS["ImagenCfgX0Params"],
    ENUMS["IMAGEN_OP_QUANTILE"]: STRUCTS["ImagenQuantileParams"],
    ENUMS["IMAGEN_OP_DDPM_UPDATE"]: STRUCTS["ImagenDdpmUpdateParams"],
    ENUMS["IMAGEN_OP_ROWS_COPY"]: STRUCTS["ImagenRowsCopyParams"],
    ENUMS["IMAGEN_OP_MEMSET32"]: STRUCTS["ImagenMemset32Params"],
    ENUMS["IMAGEN_OP_SELECT_ROWS"]: STRUCTS["ImagenSelectRowsParams"],
    ENUMS["IMAGEN_OP_MEAN_ROWS"]: STRUCTS["ImagenMeanRowsParams"],
    ENUMS["IMAGEN_OP_RANDN"]: STRUCTS["ImagenRandnParams"],
    ENUMS["IMAGEN_OP_LOWRES_PREP"]: STRUCTS["ImagenLowresPrepParams"],
    ENUMS["IMAGEN_OP_LINCOMB"]: STRUCTS["ImagenLincombParams"],
    ENUMS["IMAGEN_OP_KV_PREP_MULTI"]: STRUCTS["ImagenKvPrepMultiParams"],
    ENUMS["IMAGEN_OP_TEMPORAL_PEG"]: STRUCTS["ImagenTemporalPegParams"],
    ENUMS["IMAGEN_OP_TEMPORAL_ATTENTION"]: STRUCTS["ImagenTemporalAttentionParams"],
    ENUMS["IMAGEN_OP_ACT_PREP"]: STRUCTS["ImagenActPrepParams"],
    ENUMS["IMAGEN_OP_GCA_TAIL"]: STRUCTS["ImagenGcaTailParams"],
    ENUMS["IMAGEN_OP_STEP_SLICE"]: STRUCTS["ImagenStepSliceParams"],
    ENUMS["IMAGEN_OP_ROWCHAIN"]: STRUCTS["ImagenRowchainParams"],
    ENUMS["IMAGEN_OP_LINEAR_F32"]: STRUCTS["ImagenLinearF32Params"],
}
STRUCT_KIND = {v: k for k, v in OP_STRUCT.items()}

EXPORTED_SYMBOLS = [
    "imagen_abi_version", "imagen_last_error", "imagen_sizeof", "imagen_launch", "imagen_plan_run",
    "imagen_igemm_num_configs", "imagen_igemm_config_info", "imagen_igemm_stage_slots", "imagen_igemm_config_family", "imagen_igemm_config_ring", "imagen_igemm_lds_bytes", "imagen_igemm_packed_elems", "imagen_pack_igemm_weights",
    "imagen_graph_begin", "imagen_graph_end", "imagen_graph_launch", "imagen_graph_destroy",
    "imagen_event_create", "imagen_event_record", "imagen_event_elapsed_ms", "imagen_event_destroy",
    "imagen_probe_copy", "imagen_probe_mfma", "imagen_probe_latency", "imagen_probe_launch_chain",
]


class ImagenHipError(RuntimeError):
    pass


_lib = None


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """Load libimagen_hip.so (once).  torch is imported first so that the process-wide HIP runtime is torch's."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise ImagenHipError(
            f"{path} not found: the HIP extension is the product path and has no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)."
        )
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so, soname libamdhip64.so.7, before our DT_NEEDED resolves)

    lib = ctypes.CDLL(path)
    for sym in EXPORTED_SYMBOLS:
        if not hasattr(lib, sym):
            raise ImagenHipError(f"{path} does not export {sym}")
    lib.imagen_last_error.restype = ctypes.c_char_p
    lib.imagen_sizeof.restype = ctypes.c_size_t
    lib.imagen_sizeof.argtypes = [ctypes.c_int]
    lib.imagen_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.imagen_plan_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.imagen_igemm_config_info.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3
    lib.imagen_igemm_stage_slots.argtypes = [ctypes.c_int] * 3
    lib.imagen_igemm_config_family.argtypes = [ctypes.c_int]
    lib.imagen_igemm_config_ring.argtypes = [ctypes.c_int]
    lib.imagen_igemm_lds_bytes.restype = ctypes.c_long
    lib.imagen_igemm_lds_bytes.argtypes = [ctypes.c_int] * 6
    lib.imagen_igemm_packed_elems.restype = ctypes.c_size_t
    lib.imagen_igemm_packed_elems.argtypes = [ctypes.c_int] * 5
    lib.imagen_pack_igemm_weights.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    lib.imagen_graph_begin.argtypes = [ctypes.c_void_p]
    lib.imagen_graph_end.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.imagen_graph_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.imagen_graph_destroy.argtypes = [ctypes.c_void_p]
    lib.imagen_event_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    lib.imagen_event_record.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.imagen_event_elapsed_ms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.imagen_event_destroy.argtypes = [ctypes.c_void_p]
    lib.imagen_probe_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.imagen_probe_mfma.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.imagen_probe_latency.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.imagen_probe_launch_chain.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    if lib.imagen_abi_version() != ENUMS["IMAGEN_ABI_VERSION"]:
        raise ImagenHipError("libimagen_hip.so ABI version does not match include/imagen_hip.h")
    for kind, st in OP_STRUCT.items():
        if lib.imagen_sizeof(kind) != ctypes.sizeof(st):
            raise ImagenHipError(f"struct size mismatch for op kind {kind}: C {lib.imagen_sizeof(kind)} vs ctypes {ctypes.sizeof(st)}")
    _lib = lib
    return lib


def hip_runtime_copies() -> list:
    """Paths of every libamdhip64 mapped into this process (must be exactly one once torch + our .so are loaded)."""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    if path not in seen:
                        seen.append(path)
    except OSError:
        pass
    return seen


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load_library().imagen_last_error().decode(errors="replace")
        raise ImagenHipError(f"{what or 'libimagen_hip'} failed (rc={rc}): {msg}")


class OpRef(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("params_bytes", ctypes.c_int32), ("params", ctypes.c_void_p)]
