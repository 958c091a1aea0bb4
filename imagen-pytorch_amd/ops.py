"""Host-side op builders: torch tensors (device memory only) -> C-ABI params structs -> launch / plan.

Nothing here computes: every function fills an `Imagen*Params` struct (include/imagen_hip.h) with raw
device pointers and sizes and appends it to a `Plan`.  A plan is executed by ONE call into
`imagen_plan_run` (native loop over the ops) on a given HIP stream, and can be captured into a hipGraph.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _abi
from ._abi import ENUMS, STRUCTS, OpRef, check, load_library

ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
OUT_NHWC, OUT_PIXEL_SHUFFLE, OUT_NCHW_F32 = 0, 1, 2
LOG2E = 1.4426950408889634

# tile configurations of the igemm kernel: cfg id -> (tile pixels, tile couts, G)
_CFG_TABLE = None


def cfg_table():
    """[(tile pixels, tile couts, G, family)] per tile cfg id; family 0 = wave-specialised persistent kernel (csrc/igemm.hip),
    2 = all-DMA kernel (csrc/conv_dma.hip: prologue-free 3x3, stride 1), 3 = streaming kernel (csrc/conv_stream.hip), 4 = streaming pointwise
    kernel (conv_pw.hip), 5 = big-tile all-DMA kernel (conv_big.hip), 6 = streaming kernel with the prologue on register-staged rows (conv_pro.hip)."""
    global _CFG_TABLE
    if _CFG_TABLE is None:
        lib = load_library()
        tab = []
        for i in range(lib.imagen_igemm_num_configs()):
            tp, bn, g = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            lib.imagen_igemm_config_info(i, ctypes.byref(tp), ctypes.byref(bn), ctypes.byref(g))
            tab.append((tp.value, bn.value, g.value, lib.imagen_igemm_config_family(i)))
        _CFG_TABLE = tab
    return _CFG_TABLE


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def current_stream_handle() -> int:
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------------ plan

class Plan:
    """Ordered list of kernel launches with their params structs (kept alive here)."""

    def __init__(self, name: str = ""):
        self.name = name
        self.ops: List[tuple] = []   # (kind, struct, label)
        self.keep: List[object] = []  # tensors referenced by raw pointer
        self.twice: dict = {}         # doubled per-channel affines of this plan's split-precision launches (_twice)
        self._arr = None

    def add(self, struct, label: str = "", keep: Sequence = ()):
        kind = _abi.STRUCT_KIND[type(struct)]
        self.ops.append((kind, struct, label))
        self.keep.extend(k for k in keep if k is not None)
        self._arr = None
        return struct

    def extend(self, other: "Plan"):
        self.ops.extend(other.ops)
        self.keep.extend(other.keep)
        self._arr = None

    def __len__(self):
        return len(self.ops)

    def _array(self):
        if self._arr is None:
            arr = (OpRef * len(self.ops))()
            for i, (kind, st, _) in enumerate(self.ops):
                arr[i].kind = kind
                arr[i].params_bytes = ctypes.sizeof(st)
                arr[i].params = ctypes.addressof(st)
            self._arr = arr
        return self._arr

    def run(self, stream: Optional[int] = None):
        if not self.ops:
            return
        lib = load_library()
        s = current_stream_handle() if stream is None else stream
        check(lib.imagen_plan_run(ctypes.cast(self._array(), ctypes.c_void_p), len(self.ops), s), f"plan '{self.name}'")

    def run_stepwise(self, stream: Optional[int] = None, sync_each: bool = False):
        """Debug: launch op by op (optionally synchronising) so a faulting kernel is attributable."""
        lib = load_library()
        s = current_stream_handle() if stream is None else stream
        for kind, st, label in self.ops:
            check(lib.imagen_launch(kind, ctypes.addressof(st), ctypes.sizeof(st), s), f"op {label or kind}")
            if sync_each:
                torch.cuda.synchronize()


class Graph:
    """hipGraph capture of a plan (per-timestep graph)."""

    def __init__(self, plan: Plan, stream: torch.cuda.Stream):
        self.plan, self.stream = plan, stream
        lib = load_library()
        self._exec = ctypes.c_void_p()
        h = stream.cuda_stream
        check(lib.imagen_graph_begin(h), "graph begin")
        try:
            plan.run(h)
        finally:
            rc = lib.imagen_graph_end(h, ctypes.byref(self._exec))
        check(rc, "graph end")

    def launch(self):
        check(load_library().imagen_graph_launch(self._exec, self.stream.cuda_stream), "graph launch")

    def __del__(self):
        try:
            if self._exec:
                load_library().imagen_graph_destroy(self._exec)
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------ tensors

@dataclass
class Act:
    """fp16 NHWC activation view: element (b, y, x, c) at b*bs + (y*W + x)*ld + c (elements)."""
    t: torch.Tensor          # owning storage (kept alive)
    B: int
    H: int
    W: int
    C: int
    ld: int
    bs: int
    off: int = 0             # element offset into t
    ssq: Optional[torch.Tensor] = None   # fp32 [rows]: per-pixel sum of squares emitted by the producer (ChanRMSNorm statistics)
    tail_op: Optional[object] = None     # the GCA_TAIL params that produce this tensor: a consumer may ask it for more outputs (request_act)
    res_op: Optional[object] = None      # the ROWCHAIN RESPREP params that produce this tensor: a consumer may ask it for its activated input (request_prep)

    @property
    def ptr(self) -> int:
        return self.t.data_ptr() + 2 * self.off

    @property
    def rows(self) -> int:
        return self.B * self.H * self.W

    def tokens(self) -> "Act":
        """Same memory seen as (B, 1, H*W, C)."""
        return Act(self.t, self.B, 1, self.H * self.W, self.C, self.ld, self.bs, self.off)


def new_act(B, H, W, C, device, zero: bool = False) -> Act:
    t = (torch.zeros if zero else torch.empty)((B, H, W, C), dtype=torch.float16, device=device)
    return Act(t, B, H, W, C, C, H * W * C)


def act_from_nchw(x: torch.Tensor) -> Act:
    """fp32/fp16 NCHW torch tensor -> fp16 NHWC Act (test helper; plumbing, not on the hot path)."""
    B, C, H, W = x.shape
    t = x.permute(0, 2, 3, 1).contiguous().to(torch.float16)
    return Act(t, B, H, W, C, C, H * W * C)


def act_to_nchw(a: Act) -> torch.Tensor:
    assert a.ld == a.C and a.off == 0
    return a.t.reshape(a.B, a.H, a.W, a.C).permute(0, 3, 1, 2).float()


# ------------------------------------------------------------------------------------------------ weights

def _round_up(v, m):
    return (v + m - 1) // m * m


@dataclass
class PackedWeight:
    w: torch.Tensor           # packed fp16, device
    bias: Optional[torch.Tensor]  # fp32 [Cout_pad], device
    Cin: int
    Cout: int
    KH: int
    KW: int
    G: int
    Cin_pad: int
    Cout_pad: int
    split: bool = False       # split-precision weight (pack_weight(split=True)): Cin counts the input channels TWICE — the launch reads the same
                              # tensor as x1 and x2 against [fp16(W) | fp16(W - fp16(W))], i.e. the fp32 weight to ~2^-22 through two fp16 MFMA operands


# Host-logic tests (tests/plan_interp.py) execute plans on the CPU from the documented op contracts; the packed MFMA-fragment
# order is kernel-private, so with this switch on pack_weight() also remembers the plain (fp16-rounded) weight of every packed buffer.
KEEP_REFERENCE_WEIGHTS = False
REFERENCE_WEIGHTS: dict = {}   # packed.data_ptr() -> (w [Cout, Cin, KH, KW] fp32 holding fp16-rounded values, bias or None)


def choose_G(Cin: int, taps: int = 9) -> int:
    """8-channel groups per k-chunk.  1x1 convs / linears have no halo, so their chunks go 64-128 channels deep."""
    if taps == 1:
        if Cin % 128 == 0:
            return 16
        if Cin % 64 == 0:
            return 8
    if Cin % 32 == 0:
        return 4
    return 1


def pack_weight(w: torch.Tensor, bias: Optional[torch.Tensor], device, in_scale: Optional[torch.Tensor] = None,
                G: Optional[int] = None, split: bool = False) -> PackedWeight:
    """w: fp32 [Cout, Cin, KH, KW] or [Cout, Cin] (Linear).  Packs on the host through the C packer, uploads once.
    split: pack [hi | lo] along the input channels with hi = fp16(w), lo = fp16(w - hi) (Cin doubles; Cin % 8 == 0): igemm() then feeds the
    input tensor twice (x2 = x1), and the product is that of the fp32 weight to ~2^-22 instead of 2^-11 — for the launches whose time does
    not depend on their K (the once-per-request conditioning, the latency-bound small maps)."""
    lib = load_library()
    w = w.detach().float().cpu()
    if w.ndim == 2:
        w = w[:, :, None, None]
    w = w.contiguous()
    if split:
        assert w.shape[1] % 8 == 0, "split-precision weights need Cin % 8 == 0 (the second copy starts at an 8-channel group)"
        if in_scale is not None:
            w = w * in_scale.detach().float().cpu()[None, : w.shape[1], None, None]
            in_scale = None
        hi = w.half().float()
        w = torch.cat((hi, (w - hi).half().float()), dim=1).contiguous()
    Cout, Cin, KH, KW = w.shape
    G = G or choose_G(Cin, KH * KW)
    KC = 8 * G
    Cin_pad = _round_up(Cin, KC)
    Cout_pad = _round_up(Cout, 128 if Cout <= 128 else 256)   # every tile width (32 .. 256 couts) divides it
    n = lib.imagen_igemm_packed_elems(G, Cin, Cout_pad, KH, KW)
    out = torch.empty(n, dtype=torch.float16)
    sc = None if in_scale is None else in_scale.detach().float().cpu().contiguous()
    check(lib.imagen_pack_igemm_weights(G, w.data_ptr(), None if sc is None else sc.data_ptr(), Cin, Cout, Cout_pad, KH, KW,
                                        out.data_ptr()), "pack weights")
    b = None
    if bias is not None:
        b = torch.zeros(Cout_pad, dtype=torch.float32)
        b[:Cout] = bias.detach().float().cpu()
        b = b.to(device)
    packed = out.to(device)
    if KEEP_REFERENCE_WEIGHTS:
        REFERENCE_WEIGHTS[packed.data_ptr()] = ((w if sc is None else w * sc[None, :, None, None]).half().float(), bias)
    return PackedWeight(packed, b, Cin, Cout, KH, KW, G, Cin_pad, Cout_pad, split)


# ------------------------------------------------------------------------------------------------ igemm

import os as _os

MAX_LDS_BYTES = 160 * 1024


def _tile_shapes(tp: int, OH: int, OW: int):
    if OH == 1:
        return [(1, tp)]
    return [(tp // tw, tw) for tw in (8, 16, 32, 64) if tp % tw == 0 and tp // tw >= 1]


def launchable_shapes(cfg: int, OH: int, OW: int, KH: int, KW: int, stride: int):
    """Tile shapes (tiles, staged halo pixels, th, tw) of `cfg` the launcher accepts for this layer, best first: least padded pixels,
    then the fewest staged halo pixels."""
    lib = load_library()
    tp = cfg_table()[cfg][0]
    out = []
    for th, tw in _tile_shapes(tp, OH, OW):
        if lib.imagen_igemm_lds_bytes(cfg, KH, KW, stride, th, tw) <= 0:
            continue
        it = ((th - 1) * stride + KH) * ((tw - 1) * stride + KW)
        tiles = math.ceil(OH / th) * math.ceil(OW / tw)
        out.append((tiles, tiles * it, th, tw))
    return sorted(out)


# GlobalContext partials from the producing conv's epilogue ... only for launches of at most this many tiles: the partials cost every TILE a fixed ~2 us of reductions and barriers, a stand-alone
# pass over the tensor costs ~9 us per LAUNCH + its read (measured in the model: a loss on the 4096-tile 256^2 layers, a gain below)
GCA_EPILOGUE_MAX_TILES = 1024
CONV_DMA = 1       # (module constant since round 5; tests monkeypatch it) the all-DMA kernel family for prologue-free single-input 3x3 convs
CONV_STREAM = 1    # (module constant) the streaming kernel family (conv_stream.hip) for the 32-channel 3x3 convs
STREAM_MIN_TILES = 512   # ... of launches with at least this many 16x16 tiles (persistent workgroups need a few tiles each)


CONV_PRO = int(_os.environ.get("IMAGEN_CONV_PRO", "1"))             # A/B switch: conv_pro.hip for the 32-channel 3x3 convs that need the Block prologue (2: the raw ones too)
PRO_MIN_TILES = 1024     # ... of launches with at least this many 8x16 tiles (two per resident workgroup)


CONV_PW = 1        # (module constant) the streaming pointwise family (conv_pw.hip) for the large res_conv launches
PW_MIN_TILES = 512     # ... of at least this many tiles, i.e. two per CU (the big maps; below, the launch is latency-bound either way)


CONV_GEMM = int(_os.environ.get("IMAGEN_CONV_GEMM", "1"))             # A/B switch: the tiled pointwise GEMM (conv_gemm.hip) for the deep 1x1 layers
GEMM_MIN_K = 128         # ... with at least this many input channels (below: family 4 / the wave-specialised kernel)
GEMM_MAX_K = 2048        # ... and at most this many (the launcher's limit, conv_gemm.hip: the prologue affine table in LDS; a split-precision
                         # weight counts its input channels twice) — wider layers stay on the wave-specialised kernel
GEMM_MIN_TILES = 128     # ... and at least this many 128-row x 128-cout workgroup tiles
GEMM_MIN_COUT = 256      # ... that fill at least two output-channel slabs
CONV_BIG = 1       # (module constant) the big-tile all-DMA family (conv_big.hip) for the C >= 128 3x3 convs
BIG_MIN_WGS = 192      # ... of launches that give it at least this many workgroups (one per CU: below, the smaller tiles of family 2 fill the chip better)
BIG_PICKS = (3, 2)   # family-5 configuration of the 256- / 128-pixel tile (call R: the 3-stage weight ring and the third halo buffer are 2-3 % ahead)


def big_cfg(Cout: int, OH: int, OW: int, B: int) -> Optional[tuple]:
    """(cfg, th, tw) of the big-tile all-DMA family (family 5) for a prologue-free 3x3 stride-1 conv with 32-channel chunks, None where it
    does not apply: 128-cout tiles only; the 256-pixel tile where that still gives one workgroup per CU, else the 128-pixel tile with the
    K split, else nothing."""
    if not CONV_BIG or Cout % 128 != 0:
        return None
    fam5 = [i for i, c in enumerate(cfg_table()) if c[3] == 5]
    if len(fam5) <= max(BIG_PICKS):
        return None
    for i in (fam5[BIG_PICKS[0]], fam5[BIG_PICKS[1]]):
        sh = launchable_shapes(i, OH, OW, 3, 3, 1)
        if sh and B * sh[0][0] * (Cout // 128) >= BIG_MIN_WGS:
            return i, sh[0][2], sh[0][3]
    return None


def pw_cfg(kchunks: int, Cout: int) -> Optional[int]:
    """Tile cfg id of the streaming pointwise family (family 4) built for `kchunks` 32-channel input chunks that covers Cout, else None."""
    best = None
    for i, (tp, bn, kch, fam) in enumerate(cfg_table()):
        if fam == 4 and kch == kchunks and bn >= Cout and (best is None or bn < cfg_table()[best][1]):
            best = i
    return best


CONV_SMALL = int(_os.environ.get("IMAGEN_CONV_SMALL", "2"))   # A/B switch: conv_small.hip (family 8) for the 3x3 convs of the small maps (2: their 1x1 res_conv / upsample GEMMs too)
SMALL_MAX_ROWS = 4096   # ... of at most this many output pixels per launch (16 images of 8^2 / 16^2; call H: with the 32^2 maps, 16384, unet2 loses 1.3 ms per step)
SMALL_MAX_STREAM_MB = 64    # ... whose pixel tiles together stream at most this much weight data out of L2 (every 32-pixel tile reads all of its slab: README unet1's
                            # layers 38 - 57 MB; C2's 512 -> 512 @16^2 and 1024 -> 1024 @8^2 604 MB — 75 / 86 us against 33 / 48 on the wave-specialised kernel, round 5 call J;
                            # C2's 1x1 res_conv / upsample GEMMs of 100 - 134 MB — 60 us each here: 5.36 -> 5.11 ms per C2 step with them back on families 0 / 7, round 6 call D)


def small_cfg(Cout: int, full_cout: bool) -> Optional[int]:
    """Tile cfg id of the small-map family (family 8: 32 pixels x 32 | 64 | 128 couts per workgroup, K split over its waves): the 32-cout
    tile (most workgroups), or the narrowest tile over all Cout where the epilogue needs every channel of a pixel; None if there is none."""
    fam8 = sorted((c[1], i) for i, c in enumerate(cfg_table()) if c[3] == 8)
    if not fam8:
        return None
    if not full_cout:
        return fam8[0][1]
    return next((i for bn, i in fam8 if bn >= Cout), None)


def small_tile(OH: int, OW: int) -> Optional[tuple]:
    """The 32-pixel output tile of family 8 for an OH x OW map: 4 x 8, 2 x 16 or 1 x 32, dividing the map."""
    if OW == 8 and OH % 4 == 0:
        return 4, 8
    if OW == 16 and OH % 2 == 0:
        return 2, 16
    if OW % 32 == 0:
        return 1, 32
    return None


def small_lds_bytes(th: int, tw: int, Cin_pad: int, bn: int) -> int:
    """conv_small.hip's cs_lds_bytes (halo tile + affine table | K-split partials, + epilogue scratch)."""
    body = (th + 2) * (12 if tw == 8 else tw + 2) * (2 * Cin_pad + 16)
    nt = bn // 32
    return max(body, (8 // nt - 1) * nt * 4096) + 16 + 4 * (4 * bn + nt * 32 + 8 + bn + 4 + nt * 32)


def gemm_cfg() -> Optional[int]:
    """Tile cfg id of the tiled pointwise GEMM (family 7), None if the library has none."""
    return next((i for i, c in enumerate(cfg_table()) if c[3] == 7), None)


def pro_cfg(Cout: int = 32) -> Optional[int]:
    """Tile cfg id of the streaming family with the prologue on register-staged rows (family 6) built for exactly `Cout` (32 | 64) output
    channels, None if the library has none."""
    return next((i for i, c in enumerate(cfg_table()) if c[3] == 6 and c[1] == Cout), None)


def stream_cfg() -> Optional[int]:
    """Tile cfg id of the streaming family (family 3), None if the library has none."""
    return next((i for i, c in enumerate(cfg_table()) if c[3] == 3), None)


def _pick_dma(Cout: int, OH: int, OW: int, B: int, full_cout: bool):
    """All-DMA family (conv_dma.hip): fixed tile shape per cfg.  Preference by (tile pixels, tile couts, ring depth), first launchable
    entry wins; the deep rings go to the layers with at most ~2 workgroups per CU (nothing else hides the weight latency there)."""
    tab = cfg_table()
    lib = load_library()
    cand = {}
    for i, (tp, bn, g, fam) in enumerate(tab):
        if fam != 2:
            continue
        sh = launchable_shapes(i, OH, OW, 3, 3, 1)
        if sh:
            cand[(tp, bn, lib.imagen_igemm_config_ring(i))] = (i, sh[0])

    def wgs(tp, bn):
        k = next((k for k in cand if k[0] == tp and k[1] == bn), None)
        return B * cand[k][1][0] * math.ceil(Cout / bn) if k else 0

    if Cout > 128:   # (one tile over all 256 couts only where the epilogue needs them: post_pa / ssq_out — and the map is large enough)
        order = [(64, 256, 3), (64, 256, 6)] if full_cout and wgs(64, 256) >= 128 else []
        order += [(128, 128, 3), (128, 128, 6)] if wgs(128, 128) >= 256 else []
        order += [(64, 128, 6), (64, 128, 3)] if wgs(64, 128) >= 128 else []
        order += [(64, 64, 6), (64, 128, 6), (64, 256, 3)]
    elif Cout > 64:
        if wgs(128, 128) >= 512:
            order = [(128, 128, 3), (128, 128, 6)]
        elif wgs(128, 128) >= 256:
            order = [(128, 128, 6), (128, 128, 3)]
        else:
            order = []
        order += [(64, 128, 6), (64, 128, 3)] if wgs(64, 128) >= 128 or full_cout else []
        order += [(64, 64, 6), (64, 128, 6)]
    elif Cout > 32:
        order = [(256, 64, 3)] if wgs(256, 64) >= 1024 else []
        order += [(128, 64, 3 if wgs(128, 64) >= 1024 else 6)] if wgs(128, 64) >= 256 else []
        order += [(64, 64, 6), (128, 64, 6), (128, 64, 3)]
    else:
        order = [(256, 32, 3)] if wgs(256, 32) >= 1024 else []
        order += [(128, 32, 3), (256, 32, 3)]
    small = full_cout and all(key[1] < Cout for key in order[:1])   # the wide tile was skipped on a small map: the caller falls back
    for key in order:
        if key in cand and (not full_cout or small or key[1] >= Cout):
            i, (_, _, th, tw) = cand[key]
            return i, th, tw
    return None


def pick_cfg(G: int, Cout: int, OH: int, OW: int, B: int, KH: int = 1, KW: int = 1, stride: int = 1, full_cout: bool = False,
             family: Optional[int] = None, raw: bool = False):
    """Choose (cfg, TH, TW).

    Family 2 (all-DMA kernel, conv_dma.hip) takes the prologue-free single-input stride-1 3x3 convs with 32-channel chunks (`raw`).
    Family 0 (wave-specialised persistent kernel, igemm.hip) takes everything else, measured on MI355X (round-2 probe igemm_probe.py
    --sweep): the 64-pixel-per-wave tilings (MI <= 2) win everywhere, and when a layer has fewer workgroups than the chip has CUs
    the narrower output-channel tile (twice the workgroups) wins.  Preference order of (tile pixels, tile couts):
      Cout <= 32 : 256x32 when that still gives >= 1024 workgroups, else 128x32
      Cout <= 64 : 256x64 for k > 1 kernels with >= 1024 workgroups (profiles/r01_igemm_tile_sweep.txt), else 64x64
      Cout  > 64 : 128x128 when that gives >= 256 workgroups, 64x128 when >= 192, else 64x64
    full_cout: the caller wants the per-pixel sum of squares from the epilogue, which needs one tile to cover all Cout —
    tiles narrower than Cout are then only used when nothing wider exists.  Among the tile shapes of the chosen
    configuration: least padded pixels, then the fewest staged halo pixels."""
    tab = cfg_table()
    if raw and stride == 1 and KH == 3 and KW == 3 and G == 4 and family in (None, 5):
        got = big_cfg(Cout, OH, OW, B)
        if got is not None:
            return got
    assert family != 5, "no big-tile configuration for this layer"
    if raw and CONV_DMA and stride == 1 and KH == 3 and KW == 3 and G == 4 and family in (None, 2):
        got = _pick_dma(Cout, OH, OW, B, full_cout)
        if got is not None:
            return got
    assert family != 2, "no all-DMA tile configuration for this layer"
    for fam in ([family] if family is not None else [0]):
        avail = {}
        for i, (tp, bn, g, f) in enumerate(tab):
            if g == G and f == fam and (tp, bn) not in avail:
                sh = launchable_shapes(i, OH, OW, KH, KW, stride)
                if sh:
                    avail[(tp, bn)] = (i, sh[0])

        def wgs(key):
            return B * avail[key][1][0] * math.ceil(Cout / key[1]) if key in avail else 0

        if Cout <= 32:
            order = [(256, 32)] if wgs((256, 32)) >= 1024 else []
            order += [(128, 32), (256, 32), (64, 64), (64, 128)]
        elif Cout <= 64:
            order = [(256, 64)] if wgs((256, 64)) >= 1024 and KH * KW > 1 else []
            order += [(64, 64), (64, 128), (128, 32), (256, 32)]
        else:
            order = [(64, 128)] if wgs((64, 128)) >= 192 or (full_cout and Cout <= 128) else []
            order += [(64, 64), (64, 128), (128, 32), (256, 32)]
        if fam == 0:
            order += [(128, 128), (256, 64)]
        for key in order:
            if key in avail:
                i, (_, _, th, tw) = avail[key]
                return i, th, tw
    raise ValueError(f"no igemm tile configuration for G={G} Cout={Cout} {OH}x{OW} k{KH}x{KW} s{stride}")


def _twice(plan: "Plan", v: Optional[torch.Tensor], C: int, n: int) -> Optional[torch.Tensor]:
    """[v[:C] | v[:C] | 0 ..] of length n: the per-channel affine of a split-precision launch (the input is read twice).  The doubled copy
    belongs to the PLAN that uses it (it dies with the plan; a snapshot of `v` at plan build, like every folded / packed parameter)."""
    if v is None:
        return None
    key = (v.data_ptr(), C, n)
    hit = plan.twice.get(key)
    if hit is None or hit[0] is not v:
        out = torch.zeros(n, dtype=torch.float32, device=v.device)
        out[:C] = v.reshape(-1)[:C]
        out[C:2 * C] = v.reshape(-1)[:C]
        hit = plan.twice[key] = (v, out)
    return hit[1]


def igemm(plan: Plan, x1: Act, pw: PackedWeight, y, *, x2: Optional[Act] = None, mu=None, rs=None, pa=None, ps=None,
          pstride: int = 0, act_in: int = ACT_NONE, act_out: int = ACT_NONE, addend: Optional[Act] = None, gate=None,
          res: Optional[Act] = None, out_mode: int = OUT_NHWC, stride: int = 1, pad: Optional[int] = None,
          cfg: Optional[tuple] = None, ssq_a=None, ssq_b=None, ssq_wb: float = 1.0, ssq_out=None, post: Optional[dict] = None,
          gca: Optional[dict] = None, causal_rows: bool = False, label: str = ""):
    """... ssq_a / ssq_b: producers' per-pixel sums of squares of x1 / x2 (ChanRMSNorm statistics without a separate pass);
    ssq_out: emit the per-pixel sum of squares of the output — honoured only when the chosen tile covers all Cout
    (`p.ssq_emitted` tells the caller, who otherwise falls back to a ROWSTAT op)."""
    """Append one implicit-GEMM launch.  y: Act (NHWC / pixel-shuffle target) or fp32 NCHW tensor (OUT_NCHW_F32)."""
    KH, KW = pw.KH, pw.KW
    if pad is None:
        pad = (KH - 1) // 2 if stride == 1 else 0
    if pw.split:   # split-precision weight: the input is read twice, against the hi and the lo half of the weight
        assert x2 is None and ssq_b is None and pstride == 0, f"{label}: a split-precision weight takes one input tensor and batch-shared affines"
        x2 = x1
        pa, ps = _twice(plan, pa, x1.C, pw.Cin_pad), _twice(plan, ps, x1.C, pw.Cin_pad)
    H, W = x1.H, x1.W
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    pad_x1 = 0
    if causal_rows:   # a KH x 1 window that ends at its own row (Imagen-Video's causal Conv1d over frames in the (clip, frame, pixel) view): KH - 1 zero
        # rows before the image, none behind it, no x padding — one launch of kernel family 0 (ImagenIgemmParams.pad_x1)
        assert KW == 1 and stride == 1 and cfg is None, f"{label}: causal_rows is a KH x 1 stride-1 window"
        pad, pad_x1, OH, OW = KH - 1, 1, H, W
    C2 = x2.C if x2 is not None else 0
    assert x1.C + C2 == pw.Cin, f"{label}: input channels {x1.C}+{C2} != weight Cin {pw.Cin}"
    want_gca = gca is not None and out_mode == OUT_NHWC and act_out == ACT_NONE and addend is None and res is None and post is None
    if causal_rows:
        cfg = pick_cfg(pw.G, pw.Cout, OH, OW, x1.B, KH, KW, stride, full_cout=ssq_out is not None and out_mode == OUT_NHWC, family=0)
    small_3x3 = KH == 3 and KW == 3 and pad == 1 and pw.G == 4
    small_1x1 = KH == 1 and KW == 1 and pad == 0 and pw.G >= 2 and OH > 1 and CONV_SMALL >= 2   # (spatial maps only: the token linears keep their kernels)
    if cfg is None and CONV_SMALL and stride == 1 and (small_3x3 or small_1x1) and x1.B * OH * OW <= SMALL_MAX_ROWS:
        # family 8: the 3x3 convs (and, CONV_SMALL >= 2, the 1x1 res_conv / upsample GEMMs) of the small maps, any prologue of the contract
        # (statistics / affine / SiLU), any epilogue; the all-cout epilogues (ssq_out / post / GlobalContext partials) where a 32 | 64 | 128-cout
        # tile covers Cout
        full = (ssq_out is not None or post is not None or want_gca) and out_mode == OUT_NHWC
        tile = small_tile(OH, OW)
        sc = small_cfg(pw.Cout, full and pw.Cout <= 128)   # (wider layers: the 32-cout tile, statistics / post left to the caller's fallback as on family 0)
        if (tile is not None and sc is not None and x1.C % 8 == 0 and C2 % 8 == 0 and x1.C + C2 == pw.Cin_pad
                and x1.ld % 8 == 0 and x1.bs % 8 == 0 and (x2 is None or (x2.ld % 8 == 0 and x2.bs % 8 == 0))
                and act_in in (ACT_NONE, ACT_SILU) and (mu is None or rs is not None) and (pstride == 0 or pstride >= pw.Cin_pad)
                and (out_mode == OUT_NCHW_F32 or pw.Cout % 4 == 0) and (out_mode != OUT_PIXEL_SHUFFLE or pw.Cout % 16 == 0)
                and pw.Cin_pad % 32 == 0 and not (addend is not None and (res is not None or gate is None))
                and x1.ptr % 16 == 0 and (x2 is None or x2.ptr % 16 == 0)     # (launch_conv_small's own predicates: a shape it refuses falls through to the older families)
                and small_lds_bytes(tile[0], tile[1], pw.Cin_pad, cfg_table()[sc][1]) <= MAX_LDS_BYTES
                and (x1.B * OH * OW // 32) * pw.Cout_pad * pw.Cin_pad * 2 * KH * KW <= SMALL_MAX_STREAM_MB << 20):
            cfg = (sc, tile[0], tile[1])
    if cfg is None and CONV_PRO and KH == 3 and KW == 3 and stride == 1 and pad == 1 and pw.G == 4 and pro_cfg(pw.Cout) is not None:
        # family 6: exactly 32 output channels from 32 | 32 + 32 input channels, or 64 from two or three 32-channel chunks (64 | 64 + 32 | 32 + 32):
        # the ssq-statistics SiLU prologue on register-staged rows (CONV_PRO = 2: raw inputs too), plain / post (/ ssq_out, 32 couts) epilogue
        no_pro = mu is None and rs is None and pa is None and ps is None and ssq_a is None and act_in == ACT_NONE
        ssq_pro = mu is None and rs is None and pa is not None and ssq_a is not None and act_in == ACT_SILU and ((x2 is None) == (ssq_b is None))
        nch = (x1.C + C2) // 32
        chunks_ok = x1.C % 32 == 0 and C2 % 32 == 0 and pw.Cin_pad == x1.C + C2 and ((pw.Cout == 32 and nch in (1, 2) and x1.C == 32)
                                                                                      or (pw.Cout == 64 and nch in (2, 3)))
        tiles = x1.B * math.ceil(OH / 8) * math.ceil(OW / 16)
        gca_here = want_gca and x1.B * math.ceil(OH / 16) * math.ceil(OW / 16) <= GCA_EPILOGUE_MAX_TILES
        if (chunks_ok and x1.ld % 8 == 0 and x1.bs % 8 == 0 and (x2 is None or (x2.ld % 8 == 0 and x2.bs % 8 == 0))
                and (ssq_pro or (no_pro and CONV_PRO >= 2)) and not gca_here and (pw.Cout == 32 or post is not None or ssq_out is None)
                and tiles >= PRO_MIN_TILES and out_mode == OUT_NHWC and addend is None and res is None and act_out == ACT_NONE
                and isinstance(y, Act) and y.ld % 8 == 0 and y.bs % 8 == 0 and not pw.split):
            cfg = (pro_cfg(pw.Cout), 8, 16)
    if cfg is None and CONV_STREAM and KH == 3 and KW == 3 and stride == 1 and pad == 1 and pw.G == 4:
        # the streaming family: C_out <= 32 from one or two 32-channel inputs, raw or with the ssq-statistics Block prologue
        no_pro = mu is None and rs is None and pa is None and ps is None and ssq_a is None and act_in == ACT_NONE
        ssq_pro = mu is None and rs is None and pa is not None and ssq_a is not None and act_in in (ACT_NONE, ACT_SILU)
        tiles16 = x1.B * math.ceil(OH / 16) * math.ceil(OW / 16)
        gca_here = want_gca and tiles16 <= GCA_EPILOGUE_MAX_TILES   # (a layer whose epilogue emits the GlobalContext partials goes to family 2: the persistent streaming kernel measured slower with them, round 4 call F)
        if (pw.Cout <= 32 and x1.C == 32 and C2 in (0, 32) and pw.Cin_pad == x1.C + C2 and x1.ld % 8 == 0 and (x2 is None or x2.ld % 8 == 0)
                and (no_pro or ssq_pro) and not gca_here and tiles16 >= STREAM_MIN_TILES
                and stream_cfg() is not None):
            cfg = (stream_cfg(), 16, 16)
    if cfg is None and CONV_PW and KH == 1 and KW == 1 and stride == 1 and pad == 0 and out_mode == OUT_NHWC:
        # the streaming pointwise family: the res_conv GEMMs of the large maps (raw inputs in 32-channel chunks, <= 64 output channels,
        # bias + gate * addend | residual epilogue)
        no_pro = mu is None and rs is None and pa is None and ps is None and ssq_a is None and act_in == ACT_NONE
        if (no_pro and act_out == ACT_NONE and post is None and gca is None and x1.C % 32 == 0 and C2 % 32 == 0 and pw.Cin_pad == x1.C + C2
                and pw.Cout % 8 == 0 and x1.ld % 8 == 0 and (x2 is None or x2.ld % 8 == 0)
                and (addend is None or (addend.ld % 8 == 0 and addend.bs % 8 == 0)) and (res is None or (res.ld % 8 == 0 and res.bs % 8 == 0))
                and isinstance(y, Act) and y.ld % 8 == 0 and y.bs % 8 == 0):
            pc = pw_cfg((x1.C + C2) // 32, pw.Cout)
            if pc is not None and x1.B * math.ceil(OH * OW / cfg_table()[pc][0]) >= PW_MIN_TILES:
                tp = cfg_table()[pc][0]
                cfg = (pc, tp // min(OW, tp), min(OW, tp))
    if cfg is None and CONV_GEMM and KH == 1 and KW == 1 and stride == 1 and pad == 0 and gemm_cfg() is not None:
        # family 7, the tiled pointwise GEMM: the token / small-map 1x1 layers too deep for family 4's register-resident weights — raw rows
        # or the LayerNorm prologue (mu / rs statistics, per-channel affine), every epilogue; 128-row x 128-cout workgroup tiles
        ln_pro = ssq_a is None and ssq_b is None and act_in == ACT_NONE and (mu is None or rs is not None)
        tiles = x1.B * math.ceil(OH * OW / 128) * math.ceil(pw.Cout / 128)
        # (measured, call O: it wins where the epilogue is the plain store and the slabs are full — qkv 29.6 -> 18.6 us, to_q 23.6 -> 16.0 —
        # and loses with the generic epilogue, whose operand round trips a one-tile workgroup cannot hide: res_conv 17.0 -> 22.4;
        # IMAGEN_CONV_GEMM=2 routes those too)
        plain_ep = act_out == ACT_NONE and out_mode == OUT_NHWC and addend is None and res is None
        if ((CONV_GEMM >= 2 or (plain_ep and pw.Cout >= GEMM_MIN_COUT))
                and ln_pro and x1.C % 32 == 0 and C2 % 32 == 0 and pw.Cin_pad == x1.C + C2 and GEMM_MIN_K <= x1.C + C2 <= GEMM_MAX_K and tiles >= GEMM_MIN_TILES
                and x1.ld % 8 == 0 and x1.bs % 8 == 0 and (x2 is None or (x2.ld % 8 == 0 and x2.bs % 8 == 0))
                and (out_mode == OUT_NCHW_F32 or pw.Cout % 4 == 0)):   # (ssq_out / post / gca wider than the 128-cout tile: not emitted, as in family 0)
            tw = 128 if OW >= 128 else 1 << (OW.bit_length() - 1)   # (the largest power of two inside the row, 128 pixels per tile)
            cfg = (gemm_cfg(), 128 // tw, tw)
    if cfg is None:
        raw = (x2 is None and mu is None and rs is None and pa is None and ps is None and ssq_a is None and act_in == ACT_NONE
               and x1.C % 32 == 0 and pw.Cin_pad == x1.C and x1.ld % 8 == 0)
    if cfg is None:
        cfg = pick_cfg(pw.G, pw.Cout, OH, OW, x1.B, KH, KW, stride,
                       full_cout=(ssq_out is not None or post is not None or want_gca) and out_mode == OUT_NHWC, raw=raw)
    cid, th, tw = cfg
    p = STRUCTS["ImagenIgemmParams"]()
    p.x1, p.C1, p.ld1, p.bs1 = x1.ptr, x1.C, x1.ld, x1.bs
    if x2 is not None:
        assert (x2.B, x2.H, x2.W) == (x1.B, x1.H, x1.W)
        p.x2, p.C2, p.ld2, p.bs2 = x2.ptr, x2.C, x2.ld, x2.bs
    p.mu, p.rs, p.pa, p.ps = ptr(mu), ptr(rs), ptr(pa), ptr(ps)
    p.w, p.bias = pw.w.data_ptr(), ptr(pw.bias)
    p.B, p.H, p.W = x1.B, H, W
    p.KH, p.KW, p.stride, p.pad = KH, KW, stride, pad
    p.pad_x1 = pad_x1
    p.OH, p.OW = OH, OW
    p.Cin_pad, p.Cout, p.Cout_pad = pw.Cin_pad, pw.Cout, pw.Cout_pad
    p.pstride = pstride
    p.act_in, p.act_out = act_in, act_out
    if addend is not None:
        assert gate is not None and (addend.H, addend.W) == (OH, OW)
        p.addend, p.ld_add, p.bs_add = addend.ptr, addend.ld, addend.bs
        p.gate, p.gate_stride = gate.data_ptr(), pw.Cout
    if res is not None:
        assert (res.H, res.W, res.C) == (OH, OW, pw.Cout), f"{label}: residual shape"
        p.res, p.ld_res, p.bs_res = res.ptr, res.ld, res.bs
    p.out_mode = out_mode
    keep = [x1.t, x2.t if x2 is not None else None, mu, rs, pa, ps, pw.w, pw.bias, gate,
            addend.t if addend is not None else None, res.t if res is not None else None]
    if out_mode == OUT_NCHW_F32:
        assert isinstance(y, torch.Tensor) and y.dtype == torch.float32 and tuple(y.shape) == (x1.B, pw.Cout, OH, OW)
        p.y = y.data_ptr()
        keep.append(y)
    else:
        assert isinstance(y, Act)
        if out_mode == OUT_PIXEL_SHUFFLE:
            assert (y.H, y.W, y.C) == (2 * OH, 2 * OW, pw.Cout // 4), f"{label}: pixel-shuffle target shape"
        else:
            assert (y.H, y.W) == (OH, OW) and y.C >= pw.Cout or (y.H * y.W == OH * OW and y.C >= pw.Cout), f"{label}: output shape"
        p.y, p.ldy, p.bsy = y.ptr, y.ld, y.bs
        keep.append(y.t)
    p.TH, p.TW, p.cfg = th, tw, cid
    if ssq_a is not None:
        p.ssq_a, p.ssq_b, p.ssq_wb = ssq_a.data_ptr(), ptr(ssq_b), ssq_wb
        keep += [ssq_a, ssq_b]
    # post: dict(pa, ps, pstride) — the NEXT Block's ChanRMSNorm -> scale/shift -> SiLU applied to this conv's output in the epilogue
    # (`p.post_applied` tells the caller, who otherwise keeps the prologue on the consuming conv); excludes ssq_out
    posted = False
    if (post is not None and out_mode == OUT_NHWC and pw.Cout <= cfg_table()[cid][1] and addend is None and res is None
            and act_out == ACT_NONE and pw.Cout % 4 == 0):
        p.post_pa, p.post_ps, p.post_pstride = post["pa"].data_ptr(), post["ps"].data_ptr(), post["pstride"]
        keep += [post["pa"], post["ps"]]
        posted = True
        ssq_out = None
    emitted = False
    if ssq_out is not None and out_mode == OUT_NHWC and pw.Cout <= cfg_table()[cid][1]:
        p.ssq_out = ssq_out.data_ptr()
        keep.append(ssq_out)
        emitted = True
    # gca: dict(wk=fp32 [Cout], bk=float): GlobalContext partials of the output from the epilogue (kernel family 2, one tile over
    # all Cout); `p.gca_part_t` ([B, chunks, Cout + 2], chunks = tiles per image = `p.gca_chunks`) then feeds GCA_FINAL directly
    p.gca_part_t, p.gca_chunks = None, 0
    chunks = math.ceil(OH / th) * math.ceil(OW / tw)
    fam = cfg_table()[cid][3]
    if want_gca and pw.Cout <= cfg_table()[cid][1] and fam in (2, 5, 7, 8) and x1.B * chunks <= GCA_EPILOGUE_MAX_TILES:
        part = torch.empty(x1.B, chunks, pw.Cout + 2, dtype=torch.float32, device=x1.t.device)
        p.gca_wk, p.gca_part, p.gca_bk = gca["wk"].data_ptr(), part.data_ptr(), gca["bk"]
        keep += [gca["wk"], part]
        p.gca_part_t, p.gca_chunks = part, chunks
    plan.add(p, label or "igemm", keep)
    p.ssq_emitted = emitted
    p.post_applied = posted
    return p


# ------------------------------------------------------------------------------------------------ small ops

def act_prep(plan: Plan, x1: Act, y: Act, *, x2: Optional[Act] = None, mu=None, rs=None, pa=None, ps=None, pstride: int = 0,
             act_in: int = ACT_NONE, ssq_a=None, ssq_b=None, ssq_wb: float = 1.0, self_stat: bool = False, label: str = ""):
    """The IGEMM prologue as its own pass: y = fp16(act_in((concat(x1, x2) - mu) * rs * pa + ps)) (ImagenActPrepParams).  self_stat: the launch
    computes the sum of squares of x1's channels itself (rs = 1 / sqrt(that + ssq_wb * ssq_b))."""
    p = STRUCTS["ImagenActPrepParams"]()
    C2 = x2.C if x2 is not None else 0
    assert y.C == x1.C + C2 and (y.B, y.H * y.W) == (x1.B, x1.H * x1.W)
    p.x1, p.C1, p.ld1, p.bs1 = x1.ptr, x1.C, x1.ld, x1.bs
    if x2 is not None:
        assert (x2.B, x2.H, x2.W) == (x1.B, x1.H, x1.W)
        p.x2, p.C2, p.ld2, p.bs2 = x2.ptr, x2.C, x2.ld, x2.bs
    p.mu, p.rs, p.pa, p.ps = ptr(mu), ptr(rs), ptr(pa), ptr(ps)
    p.ssq_a, p.ssq_b, p.ssq_wb = ptr(ssq_a), ptr(ssq_b), ssq_wb
    p.y, p.ldy, p.bsy = y.ptr, y.ld, y.bs
    p.rows, p.rows_per_batch = x1.rows, x1.H * x1.W
    p.pstride, p.act_in = pstride, act_in
    if self_stat:
        assert mu is None and rs is None and ssq_a is None and (x1.C + C2) <= 512
        p.self_stat = 1
    plan.add(p, label or "act_prep", [x1.t, x2.t if x2 is not None else None, mu, rs, pa, ps, ssq_a, ssq_b, y.t])
    return p


def rowstat(plan: Plan, x1: Act, *, mode: int, rs: torch.Tensor, mu: Optional[torch.Tensor] = None, x2: Optional[Act] = None,
            w2: float = 1.0, eps: float = 1e-5, label: str = ""):
    p = STRUCTS["ImagenRowstatParams"]()
    p.x1, p.C1, p.ld1, p.bs1 = x1.ptr, x1.C, x1.ld, x1.bs
    if x2 is not None:
        p.x2, p.C2, p.ld2, p.bs2 = x2.ptr, x2.C, x2.ld, x2.bs
    p.mu, p.rs = ptr(mu), rs.data_ptr()
    p.rows, p.rows_per_batch, p.mode = x1.rows, x1.H * x1.W, mode
    p.w2, p.eps = w2, eps
    plan.add(p, label or "rowstat", [x1.t, x2.t if x2 is not None else None, mu, rs])
    return p


ATTN_BOUNDED = 1   # (module constant; 0: always the online softmax)
ATTN_BOUND_MAX = 14.0    # log2 units: exp2(s) of every key stays inside fp16's normal range (2^-14 .. 2^14) while |s| <= 14


def attention_logit_bound(q_scale: torch.Tensor, k_scale: torch.Tensor, q_mult: float) -> float:
    """|sum_d q^_d qs_d k^_d ks_d| * q_mult <= q_mult * max_d |qs_d ks_d| for unit vectors q^, k^ (Cauchy-Schwarz); + 1 % and 0.05 for the
    fp16 rounding of the normalised rows.  The parameters are fixed when a plan is built, so the bound is a plan constant."""
    m = float((q_scale.detach().float().cpu() * k_scale.detach().float().cpu()).abs().max())
    return q_mult * m * 1.01 + 0.05


def attention(plan: Plan, q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, o: torch.Tensor, *, B, heads, rows, J,
              q_strides, k_strides, vt_strides, o_strides, q_scale: Optional[torch.Tensor] = None, q_mult: float = 0.0, label: str = "",
              head_dim: int = 64, logit_bound: Optional[float] = None):
    """q_scale / q_mult: fuse QNORM into the q load (q rows raw); None: q was normalised by a QNORM op.  head_dim: 64 or 32.
    logit_bound: an upper bound of |q . k| in the kernel's log2 units (`attention_logit_bound`); bounds up to ATTN_BOUND_MAX select the
    bounded-logit softmax (no running maximum; softmax_mode 1 of include/imagen_hip.h), larger ones or None the online softmax."""
    assert head_dim in (32, 64), f"attention head dim {head_dim}: the kernels are built for 64 and 32"
    p = STRUCTS["ImagenAttentionParams"]()
    p.head_dim = head_dim
    if ATTN_BOUNDED and logit_bound is not None and logit_bound <= ATTN_BOUND_MAX:
        p.softmax_mode, p.logit_bound = 1, float(logit_bound)
    p.q, p.k, p.vt, p.o = q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr()
    p.q_scale, p.q_mult = ptr(q_scale), q_mult
    p.B, p.heads, p.rows, p.J = B, heads, rows, J
    p.q_bs, p.q_hs, p.q_rs = q_strides
    p.k_bs, p.k_hs, p.k_rs = k_strides
    p.vt_bs, p.vt_hs, p.vt_ds = vt_strides
    p.o_bs, p.o_hs, p.o_rs = o_strides
    plan.add(p, label or "attention", [q, k, vt, o, q_scale])
    return p


def kv_prep(plan: Plan, k_src: torch.Tensor, v_src: torch.Tensor, k_scale: torch.Tensor, khat: torch.Tensor, vt: torch.Tensor, *,
            B, heads, rows, r0, src_strides, k_strides, vt_strides, k_off: int = 0, v_off: int = 0, label: str = "",
            batch: Optional[list] = None, head_dim: int = 64):
    """k_off / v_off: element offsets of the k and v columns inside the source rows.  With `batch` the job is appended to that
    list instead of the plan; kv_prep_multi(plan, batch) then runs all of them in one launch."""
    assert head_dim in (32, 64)
    p = STRUCTS["ImagenKvPrepParams"]()
    p.head_dim = head_dim
    es = k_src.element_size()
    p.k_src, p.v_src = k_src.data_ptr() + k_off * es, v_src.data_ptr() + v_off * es
    p.k_scale, p.khat, p.vt = k_scale.data_ptr(), khat.data_ptr(), vt.data_ptr()
    p.B, p.heads, p.rows, p.r0 = B, heads, rows, r0
    p.src_bs, p.src_rs, p.src_hs = src_strides
    p.k_bs, p.k_hs, p.k_rs = k_strides
    p.vt_bs, p.vt_hs, p.vt_ds = vt_strides
    p.src_is_f32 = 1 if k_src.dtype == torch.float32 else 0
    if batch is not None:
        batch.append((p, [k_src, v_src, k_scale, khat, vt]))
    else:
        plan.add(p, label or "kv_prep", [k_src, v_src, k_scale, khat, vt])
    return p


def kv_prep_multi(plan: Plan, batch: list, device, label: str = ""):
    """One launch for the jobs collected with kv_prep(batch=...): their parameter blocks are uploaded once, at plan build."""
    if not batch:
        return None
    if len(batch) == 1:
        plan.add(batch[0][0], label or "kv_prep", batch[0][1])
        return batch[0][0]
    blob = b"".join(bytes(j) for j, _ in batch)
    jobs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    p = STRUCTS["ImagenKvPrepMultiParams"]()
    p.jobs, p.n = jobs.data_ptr(), len(batch)
    p.max_rows = max(j.rows for j, _ in batch)
    p.max_bh = max(j.B * j.heads for j, _ in batch)
    plan.add(p, label or "kv_prep_multi", [jobs] + [t for _, keep in batch for t in keep])
    return p


def qnorm(plan: Plan, q: torch.Tensor, q_scale: torch.Tensor, *, rows, heads, ld, mult, label: str = "", head_dim: int = 64):
    assert head_dim in (32, 64)
    p = STRUCTS["ImagenQnormParams"]()
    p.head_dim = head_dim
    p.q, p.q_scale, p.rows, p.heads, p.ld, p.mult = q.data_ptr(), q_scale.data_ptr(), rows, heads, ld, mult
    plan.add(p, label or "qnorm", [q, q_scale])
    return p


def gca_final(plan: Plan, part: torch.Tensor, w1t, b1, w2t, b2, gate: torch.Tensor, *, B: int, C: int, chunks: int, label: str = ""):
    """GCA_FINAL alone: merge `chunks` partial rows per image (from GCA_PARTIAL or from a conv epilogue) and run the squeeze MLP."""
    hidden = w1t.shape[1]
    split = GCA_FINAL_SPLIT and gca_final_is_wide(C, hidden) and chunks <= 1024
    hid = torch.empty(B, hidden, dtype=torch.float32, device=gate.device) if split else None
    f = None
    for phase in ((1, 2) if split else (0,)):
        f = STRUCTS["ImagenGcaFinalParams"]()
        f.part, f.w1t, f.b1, f.w2t, f.b2, f.gate = part.data_ptr(), w1t.data_ptr(), b1.data_ptr(), w2t.data_ptr(), b2.data_ptr(), gate.data_ptr()
        f.B, f.C, f.hidden, f.chunks = B, C, hidden, chunks
        f.hid, f.phase = ptr(hid), phase
        plan.add(f, (label or "gca") + (".final" if phase == 0 else f".final{phase}"), [part, w1t, b1, w2t, b2, gate, hid])
    return f


GCA_FINAL_SPLIT = 1   # (module constant; round 4, call V) the finalisation of wide blocks as two many-workgroup launches


def gca_final_is_wide(C: int, hidden: int) -> bool:
    """The squeeze MLP of a block this wide (>= 128 Ki weights per matrix: 1 MB and up) is streamed by many workgroups in two launches, not by
    one workgroup per image (GCA_FINAL phase 1 / 2) — and not redundantly by every workgroup of a fused tail."""
    return C * hidden >= 128 * 1024 and C <= 1024 and hidden <= 1024 and C % 2 == 0   # (the two-phase kernels' LDS tables hold 1024 entries)


def gca(plan: Plan, h: Act, wk, bk: float, w1t, b1, w2t, b2, part: torch.Tensor, gate: torch.Tensor, chunks: int, label: str = "",
        final: bool = True) -> bool:
    """GlobalContext gate of h.  w1t: [C, hidden] (net.0.weight transposed), w2t: [hidden, C] (net.2.weight transposed), fp32.
    One launch when the in-kernel finalisation applies (power-of-two C/8, scratch fits), else GCA_PARTIAL + GCA_FINAL.
    final=False: the caller finalises (GCA_TAIL merges `part` itself) — only the partial pass is emitted, unless one workgroup covers the
    image and finalises in place.  Returns True when `gate` has been written by the ops emitted here."""
    C = h.C
    hidden = w1t.shape[1]
    assert tuple(w1t.shape) == (C, w2t.shape[0]) and w2t.shape[1] == C
    p = STRUCTS["ImagenGcaPartialParams"]()
    p.h, p.wk, p.part = h.ptr, wk.data_ptr(), part.data_ptr()
    p.B, p.HW, p.C, p.ld, p.chunks, p.bk = h.B, h.H * h.W, C, h.ld, chunks, bk
    groups = C // 8
    single = (chunks == 1 and (groups & (groups - 1)) == 0 and groups <= 64
              and C + hidden + chunks + GCA_SCRATCH <= 2048
              and not (GCA_FINAL_SPLIT and gca_final_is_wide(C, hidden)))   # (a wide block's MLP: many workgroups, GCA_FINAL phases 1 / 2)
    keep = [h.t, wk, part]
    if single:
        p.w1t, p.b1, p.w2t, p.b2, p.gate, p.hidden = w1t.data_ptr(), b1.data_ptr(), w2t.data_ptr(), b2.data_ptr(), gate.data_ptr(), hidden
        keep += [w1t, b1, w2t, b2, gate]
    plan.add(p, (label or "gca") + (".fused" if single else ".partial"), keep)
    if single:
        return True
    if not final:
        return False
    gca_final(plan, part, w1t, b1, w2t, b2, gate, B=h.B, C=C, chunks=chunks, label=label)
    return True


def _pow2(v: int) -> bool:
    return v > 0 and (v & (v - 1)) == 0


def gca_tail_ok(C: int, hidden: Optional[int], chunks: int = 1) -> bool:
    """Shapes the fused tail kernel takes (launcher checks): power-of-two C in [8, 512], power-of-two squeeze width, <= 1024 chunks."""
    return _pow2(C) and 8 <= C <= 512 and (hidden is None or (_pow2(hidden) and 4 <= hidden <= 1024 and 1 <= chunks <= 1024))


def gca_tail(plan: Plan, h: Act, res: Act, out: Act, *, part: Optional[torch.Tensor] = None, chunks: int = 0, w1t=None, b1=None, w2t=None, b2=None,
             gate_in: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None, ssq_out: Optional[torch.Tensor] = None, label: str = ""):
    """Identity-ResnetBlock tail in one launch (ImagenGcaTailParams): GlobalContext finalisation from `part` (or a ready `gate_in`, or no
    gate) + out = h * gate + res (+ ssq_out).  The returned params are kept on `out.tail_op`: request_act() / request_ln_stats() let the
    consumer of `out` add the activated tensor / LayerNorm statistics to this launch."""
    for a in (h, res, out):
        assert a.ld * a.H * a.W == a.bs and (a.B, a.H * a.W, a.C) == (h.B, h.H * h.W, h.C), "gca_tail: dense, equally shaped tensors"
    p = STRUCTS["ImagenGcaTailParams"]()
    p.h, p.res, p.out = h.ptr, res.ptr, out.ptr
    p.B, p.HW, p.C = h.B, h.H * h.W, h.C
    p.ld_h, p.ld_res, p.ld_out = h.ld, res.ld, out.ld
    keep = [h.t, res.t, out.t, part, w1t, b1, w2t, b2, gate_in, gate, ssq_out]
    if part is not None:
        p.part, p.chunks, p.hidden = part.data_ptr(), chunks, w1t.shape[1]
        p.w1t, p.b1, p.w2t, p.b2 = w1t.data_ptr(), b1.data_ptr(), w2t.data_ptr(), b2.data_ptr()
        assert tuple(w1t.shape) == (h.C, w2t.shape[0]) and w2t.shape[1] == h.C
    p.gate_in, p.gate, p.ssq_out = ptr(gate_in), ptr(gate), ptr(ssq_out)
    p.eps = 1e-5
    # slabs per image: >= 32 KiB of output per workgroup, and at most ONE workgroup per CU over the batch — the kernel holds one
    # 1024-thread workgroup per CU, so a second round would pay the ~5 us gate derivation again with nothing resident to hide it
    # (measured, round 3 call B: 512 workgroups cost +10 us per launch at 256^2)
    p.slabs = max(1, min(math.ceil(p.HW * p.C * 2 / 32768), max(1, 256 // max(h.B, 1))))
    plan.add(p, label or "gca_tail", keep)
    out.tail_op = (p, plan)
    return p


def request_act(x: Act, pa: torch.Tensor) -> Optional[Act]:
    """Ask the GCA_TAIL launch that produces `x` to also write silu(ChanRMSNorm(x) * pa) — the next Block's activated input
    (ip.py:683-690) — and return that tensor; None if x has no such producer or it already serves another consumer."""
    if x.tail_op is None:
        return None
    p, plan = x.tail_op
    if p.act_out or x.ld != x.C or x.C % 8:
        return None
    xa = Act(torch.empty_like(x.t), x.B, x.H, x.W, x.C, x.C, x.bs)
    p.act_out, p.act_pa, p.ld_act = xa.ptr, pa.data_ptr(), xa.ld
    plan.keep += [xa.t, pa]
    return xa


def request_ln_stats(x: Act, eps: float = 1e-5):
    """Ask the GCA_TAIL launch that produces `x` for the LayerNorm statistics (mean, rstd) of its rows; None if unavailable."""
    if x.tail_op is None:
        return None
    p, plan = x.tail_op
    if p.mu_out and abs(p.eps - eps) > 1e-12:
        return None
    if not p.mu_out:
        mu = torch.empty(x.rows, dtype=torch.float32, device=x.t.device)
        rs = torch.empty(x.rows, dtype=torch.float32, device=x.t.device)
        p.mu_out, p.rs_out, p.eps = mu.data_ptr(), rs.data_ptr(), eps
        plan.keep += [mu, rs]
        p._ln = (mu, rs)
    return p._ln


GCA_SCRATCH = 1024   # csrc/gca_device.h kGcaScratchFloats
GCA_ONE_WG_ELEMS = 65536
GCA_TARGET_WGS = 1024


def gca_chunks(HW: int, B: int = 16, C: int = 0) -> int:
    """Pixel chunks per image for the stand-alone GlobalContext kernel: about 1024 workgroups over the batch (the 256 CUs
    stay busy even on the 32x32 maps), chunks of at least 64 pixels (the merge cost grows with the chunk count).  Small maps
    (HW * C <= 64 Ki elements, i.e. <= 128 KiB per image; measured: beyond that one workgroup per image streams too slowly) take ONE chunk: the workgroup then finalises the gate itself and
    the second launch disappears."""
    if 0 < HW * C <= GCA_ONE_WG_ELEMS:
        return 1
    target = max(1, GCA_TARGET_WGS // max(B, 1))
    chunk_px = max(64, math.ceil(HW / target))
    return max(1, math.ceil(HW / chunk_px))


def gate_residual(plan: Plan, h: Act, gate: Optional[torch.Tensor], res: Act, out: Act, rs_out: Optional[torch.Tensor] = None,
                  raw_ssq: bool = False, label: str = ""):
    assert h.ld * h.H * h.W == h.bs and res.ld * res.H * res.W == res.bs and out.ld * out.H * out.W == out.bs
    p = STRUCTS["ImagenGateResidualParams"]()
    p.h, p.gate, p.res, p.out, p.rs_out = h.ptr, ptr(gate), res.ptr, out.ptr, ptr(rs_out)
    p.rows, p.rows_per_batch, p.C = h.rows, h.H * h.W, h.C
    p.ld_h, p.ld_res, p.ld_out = h.ld, res.ld, out.ld
    p.raw_ssq = int(raw_ssq)
    plan.add(p, label or "gate_residual", [h.t, gate, res.t, out.t, rs_out])
    return p


def ln_residual(plan: Plan, y: Act, g: torch.Tensor, out: Act, *, beta=None, res: Optional[Act] = None, eps: float = 1e-5,
                ssq_out: Optional[torch.Tensor] = None, ln_stats_out: Optional[tuple] = None, eps_out: float = 1e-5, label: str = ""):
    """ln_stats_out = (mu, rs) fp32 [rows]: also emit the LayerNorm statistics of the stored output rows (for a LayerNorm -> GEMM that follows)."""
    p = STRUCTS["ImagenLnResidualParams"]()
    p.y, p.g, p.beta, p.res, p.out = y.ptr, g.data_ptr(), ptr(beta), (res.ptr if res is not None else None), out.ptr
    p.rows, p.C, p.ld_y, p.ld_res, p.ld_out, p.eps = y.rows, y.C, y.ld, (res.ld if res is not None else 0), out.ld, eps
    p.rows_per_batch = y.H * y.W
    p.bs_y, p.bs_res, p.bs_out = y.bs, (res.bs if res is not None else 0), out.bs
    assert out.H * out.W == y.H * y.W and out.B == y.B
    p.ssq_out = ptr(ssq_out)
    keep_stats = []
    if ln_stats_out is not None:
        p.mu_out, p.rs_out, p.eps_out = ln_stats_out[0].data_ptr(), ln_stats_out[1].data_ptr(), eps_out
        keep_stats = list(ln_stats_out)
    plan.add(p, label or "ln_residual", [y.t, g, beta, res.t if res is not None else None, out.t, ssq_out] + keep_stats)
    return p


ROWCHAIN = int(_os.environ.get("IMAGEN_ROWCHAIN", "2"))   # A/B switch: 0 = the launch-per-op plan; 1 = the token chains of the <= 32^2 levels as one ROWCHAIN
                                                           # launch each; 2 (default) = also the res_conv + gate tails of the big-tile levels (RESPREP)
CHAIN_TILE64_MIN_ROWS = 16384    # 64-row tiles (every weight fragment feeds two MFMAs) once that still gives one workgroup per CU


def rowchain_ok(C: int, N: int, heads: int, dh: int, *weights, hidden: Optional[int] = None) -> bool:
    """Shapes the ROWCHAIN launcher takes (csrc/rowchain.hip): 8 heads x 64, C (and the FeedForward width) a power of two in 32 .. 256 (512),
    32-row tiles inside one image, plain (unsplit, bias-free) 1x1 weights in 32-channel chunks."""
    if not ROWCHAIN or dh != 64 or heads * dh != 512 or not (_pow2(C) and 32 <= C <= 256) or N % 32 != 0:
        return False
    if hidden is not None and not (_pow2(hidden) and 32 <= hidden <= 512):
        return False
    return all(w is not None and not w.split and w.bias is None and w.KH == 1 and w.KW == 1 and w.Cin % 32 == 0 and w.Cin_pad == w.Cin for w in weights)


def _rowchain(plan: Plan, mode: int, x: Act, out: Act, rows_per_batch: int, label: str, keep: list, **f):
    p = STRUCTS["ImagenRowchainParams"]()
    p.mode = mode
    assert x.ld * x.H * x.W == x.bs and out.ld * out.H * out.W == out.bs, f"{label}: dense rows"
    p.x, p.ld_x, p.out, p.ld_out = x.ptr, x.ld, out.ptr, out.ld
    p.rows, p.rows_per_batch = x.rows, rows_per_batch
    p.eps = 1e-5
    p.tile64 = int(x.rows >= CHAIN_TILE64_MIN_ROWS and rows_per_batch % 64 == 0)
    for k, v in f.items():
        setattr(p, k, v)
    plan.add(p, label, [x.t, out.t] + keep)
    return p


def rowchain_ff(plan: Plan, o: Act, res: Act, out: Act, w_out: PackedWeight, g_out, w1: PackedWeight, g_ln0, w2: PackedWeight, g_ln1, *,
                rows_per_batch: int, ssq_out: Optional[torch.Tensor] = None, label: str = ""):
    """Attention out-projection -> LayerNorm + residual -> FeedForward (+ residual) of a TransformerBlock in one launch (ROWCHAIN mode FF):
    replaces to_out IGEMM, LN_RESIDUAL, lin1 IGEMM (+ GELU), ROWSTAT, lin2 IGEMM."""
    C, inner, hidden = out.C, o.C, w1.Cout
    assert (w_out.Cin, w_out.Cout, w1.Cin, w2.Cin, w2.Cout) == (inner, C, C, hidden, C) and res.C == C and res.ld * res.H * res.W == res.bs
    return _rowchain(plan, ENUMS["IMAGEN_CHAIN_FF"], o, out, rows_per_batch, label or "rowchain.ff",
                     [res.t, w_out.w, w1.w, w2.w, g_out, g_ln0, g_ln1, ssq_out],
                     res=res.ptr, ld_res=res.ld, w0=w_out.w.data_ptr(), w1=w1.w.data_ptr(), w2=w2.w.data_ptr(), g0=g_out.data_ptr(),
                     g1=g_ln0.data_ptr(), g2=g_ln1.data_ptr(), ssq_out=ptr(ssq_out), C=C, inner=inner, hidden=hidden, heads=inner // 64,
                     w_cout_pad0=w_out.Cout_pad, w_cout_pad1=w1.Cout_pad, w_cout_pad2=w2.Cout_pad)


def rowchain_xattn(plan: Plan, x: Act, out: Act, wq: PackedWeight, g_norm, w_out: PackedWeight, g_out, khat: torch.Tensor, vt: torch.Tensor, *,
                   heads: int, J: int, k_strides, vt_strides, q_scale: torch.Tensor, q_mult: float, rows_per_batch: int, ln_stats: Optional[tuple] = None,
                   ssq_out: Optional[torch.Tensor] = None, label: str = ""):
    """A whole cross-attention of a ResnetBlock in one launch (ROWCHAIN mode XATTN): LayerNorm -> to_q -> cosine-sim attention over the site's
    K^ / V^T operand buffers -> to_out -> LayerNorm + residual; replaces ROWSTAT, to_q IGEMM, ATTENTION, to_out IGEMM, LN_RESIDUAL."""
    C, inner = x.C, heads * 64
    assert (wq.Cin, wq.Cout, w_out.Cin, w_out.Cout, out.C) == (C, inner, inner, C, C)
    mu, rs = ln_stats if ln_stats is not None else (None, None)
    return _rowchain(plan, ENUMS["IMAGEN_CHAIN_XATTN"], x, out, rows_per_batch, label or "rowchain.xattn",
                     [wq.w, w_out.w, g_norm, g_out, khat, vt, q_scale, mu, rs, ssq_out],
                     w0=wq.w.data_ptr(), w1=w_out.w.data_ptr(), g0=g_norm.data_ptr(), g1=g_out.data_ptr(), mu=ptr(mu), rs=ptr(rs),
                     khat=khat.data_ptr(), vt=vt.data_ptr(), q_scale=q_scale.data_ptr(), q_mult=q_mult, ssq_out=ptr(ssq_out), C=C, inner=inner,
                     heads=heads, J=J, k_bs=k_strides[0], k_hs=k_strides[1], k_rs=k_strides[2], vt_bs=vt_strides[0], vt_hs=vt_strides[1],
                     vt_ds=vt_strides[2], w_cout_pad0=wq.Cout_pad, w_cout_pad1=w_out.Cout_pad)


def rowchain_qkv(plan: Plan, x: Act, qkv: Act, wqkv: PackedWeight, g_norm, khat: torch.Tensor, vt: torch.Tensor, k_scale: torch.Tensor, *,
                 heads: int, r0: int, k_strides, vt_strides, rows_per_batch: int, ln_stats: Optional[tuple] = None, label: str = ""):
    """The front of a self-attention in one launch (ROWCHAIN mode QKV): LayerNorm -> q | k | v projection -> q rows, K^ rows and V^T columns
    behind the site's conditioning rows; replaces (ROWSTAT,) qkv IGEMM, KV_PREP."""
    C, inner = x.C, heads * 64
    assert (wqkv.Cin, wqkv.Cout) == (C, inner + 128) and qkv.C >= inner
    mu, rs = ln_stats if ln_stats is not None else (None, None)
    return _rowchain(plan, ENUMS["IMAGEN_CHAIN_QKV"], x, qkv, rows_per_batch, label or "rowchain.qkv",
                     [wqkv.w, g_norm, khat, vt, k_scale, mu, rs],
                     w0=wqkv.w.data_ptr(), g0=g_norm.data_ptr(), mu=ptr(mu), rs=ptr(rs), khat=khat.data_ptr(), vt=vt.data_ptr(),
                     k_scale=k_scale.data_ptr(), C=C, inner=inner, heads=heads, r0=r0, k_bs=k_strides[0], k_hs=k_strides[1], k_rs=k_strides[2],
                     vt_bs=vt_strides[0], vt_hs=vt_strides[1], vt_ds=vt_strides[2], w_cout_pad0=wqkv.Cout_pad)


RESPREP_MAX_ROWS = 16384   # (call C: at 65536 rows — unet2's 64^2 level, 92 MB a launch — the one-tile-per-workgroup chain runs at 2 TB/s, 45 us against
                           # 34 for the streaming res_conv + ACT_PREP pair; at 16384 rows it is 22 against 33)


def resprep_ok(x: Act, skip: Optional[Act], w: PackedWeight, N: int) -> bool:
    """Shapes ROWCHAIN mode RESPREP takes: a 1x1 res_conv of 32-channel chunks (<= 512 input channels) to a power-of-two 128 | 256 output channels."""
    C2 = skip.C if skip is not None else 0
    if x.rows > RESPREP_MAX_ROWS:
        return False
    return bool(ROWCHAIN >= 2 and w.KH == 1 and w.KW == 1 and not w.split and x.C % 32 == 0 and C2 % 32 == 0 and w.Cin_pad == x.C + C2 <= 512
                and w.Cout in (128, 256) and N % 32 == 0 and x.ld * x.H * x.W == x.bs and (skip is None or skip.ld * skip.H * skip.W == skip.bs))


def rowchain_resprep(plan: Plan, x: Act, skip: Optional[Act], addend: Act, gate: Optional[torch.Tensor], out: Act, w: PackedWeight, *, rows_per_batch: int,
                     ssq_out: torch.Tensor, label: str = ""):
    """The tail of a ResnetBlock with a res_conv — out = h * gate + res_conv(concat(x, skip)) (ip.py:741, 753-757) — as a ROWCHAIN launch (mode
    RESPREP) that can also write the NEXT Block's activated input (request_prep): the consumer then needs no ACT_PREP pass."""
    assert (addend.H * addend.W, addend.C) == (x.H * x.W, w.Cout) and addend.ld * addend.H * addend.W == addend.bs
    f = dict(w0=w.w.data_ptr(), bias=ptr(w.bias), addend=addend.ptr, gate=ptr(gate), ld_add=addend.ld, gate_stride=w.Cout,
             ssq_out=ssq_out.data_ptr(), C=w.Cout, inner=x.C, w_cout_pad0=w.Cout_pad)
    keep = [w.w, w.bias, addend.t, gate, ssq_out]
    if skip is not None:
        f.update(x2=skip.ptr, C2=skip.C, ld_x2=skip.ld)
        keep.append(skip.t)
    p = _rowchain(plan, ENUMS["IMAGEN_CHAIN_RESPREP"], x, out, rows_per_batch, label or "rowchain.resprep", keep, **f)
    out.res_op = (p, plan)
    return p


def request_prep(x: Act, skip: Optional[Act], ssq_skip: Optional[torch.Tensor], ssq_wb: float, pa: torch.Tensor) -> Optional[Act]:
    """Ask the RESPREP launch that produces `x` to also write silu(ChanRMSNorm(concat(x, skip)) * pa) — the next Block's block1 input through its
    prologue (ACT_PREP's contract) — and return that tensor; None if x has no such producer, it already serves another consumer, or the
    statistics of `skip` are not in memory yet (they would have to be computed by a launch BEHIND the producer)."""
    op = getattr(x, "res_op", None)
    if op is None:
        return None
    p, plan = op
    C2 = skip.C if skip is not None else 0
    if p.prep_out or x.ld != x.C or (skip is not None and (ssq_skip is None or skip.C % 8 or skip.C > 256 or skip.ld * skip.H * skip.W != skip.bs)):
        return None
    xa = new_act(x.B, x.H, x.W, x.C + C2, x.t.device)
    p.prep_out, p.ld_prep, p.prep_pa, p.prep_ssq_wb = xa.ptr, xa.ld, pa.data_ptr(), float(ssq_wb)
    plan.keep += [xa.t, pa]
    if skip is not None:
        p.prep_x2, p.prep_C2, p.ld_prep_x2, p.prep_ssq_b = skip.ptr, skip.C, skip.ld, ssq_skip.data_ptr()
        plan.keep += [skip.t, ssq_skip]
    return xa


def select_rows(plan: Plan, a: torch.Tensor, nul: torch.Tensor, mask: Optional[torch.Tensor], src: torch.Tensor, keep: torch.Tensor,
                dst: torch.Tensor, *, R, L, C, label: str = ""):
    """dst[r,l,:] = keep[r] & mask[src[r],l] ? a[src[r],l,:] : nul[l,:]   (mask/keep uint8, src int32)."""
    p = STRUCTS["ImagenSelectRowsParams"]()
    p.a, p.nul, p.mask, p.src, p.keep, p.dst = a.data_ptr(), nul.data_ptr(), ptr(mask), src.data_ptr(), keep.data_ptr(), dst.data_ptr()
    p.R, p.L, p.C = R, L, C
    plan.add(p, label or "select_rows", [a, nul, mask, src, keep, dst])
    return p


def mean_rows(plan: Plan, x: Act, out: Act, label: str = ""):
    """out[b,:] = mean over the H*W rows of x[b]."""
    p = STRUCTS["ImagenMeanRowsParams"]()
    p.x, p.out = x.ptr, out.ptr
    p.B, p.rows, p.C, p.bs_x, p.ld_x, p.ld_out = x.B, x.H * x.W, x.C, x.bs, x.ld, out.ld
    assert out.rows == x.B and out.C == x.C
    plan.add(p, label or "mean_rows", [x.t, out.t])
    return p


def time_embed(plan: Plan, *, times, coef, step_ptr, freqs, w, bias, hid: Act, label: str = ""):
    p = STRUCTS["ImagenTimeEmbedParams"]()
    p.times, p.coef, p.step_ptr = ptr(times), ptr(coef), ptr(step_ptr)
    p.freqs, p.w, p.bias, p.hid = freqs.data_ptr(), w.data_ptr(), bias.data_ptr(), hid.ptr
    p.B, p.half_dim, p.out_dim, p.ld_hid = hid.B * hid.H * hid.W, freqs.numel(), hid.C, hid.ld
    p.steps = coef.shape[0] if (coef is not None and step_ptr is not None) else 0   # rows of coef: the kernel clamps *step_ptr to them
    plan.add(p, label or "time_embed", [times, coef, step_ptr, freqs, w, bias, hid.t])
    return p


def scale_shift(plan: Plan, ss, gamma_s, idx_scale, idx_shift, pa, ps, label: str = ""):
    """ss: the batched time-MLP output — an fp16 Act (an IGEMM's rows) or a 2-D fp32 tensor [rows, width] (LINEAR_F32's rows)."""
    p = STRUCTS["ImagenScaleShiftParams"]()
    if isinstance(ss, Act):
        p.ss, p.B, p.ld_ss, p.ss_f32, keep = ss.ptr, ss.rows, ss.ld, 0, ss.t
    else:
        assert ss.dtype == torch.float32 and ss.ndim == 2 and ss.is_contiguous()
        p.ss, p.B, p.ld_ss, p.ss_f32, keep = ss.data_ptr(), ss.shape[0], ss.shape[1], 1, ss
    p.gamma_s, p.idx_scale, p.idx_shift, p.pa, p.ps = gamma_s.data_ptr(), idx_scale.data_ptr(), idx_shift.data_ptr(), pa.data_ptr(), ps.data_ptr()
    p.total_c = gamma_s.numel()
    plan.add(p, label or "scale_shift", [keep, gamma_s, idx_scale, idx_shift, pa, ps])
    return p


def linear_f32(plan: Plan, x, wt: torch.Tensor, bias: Optional[torch.Tensor], y: torch.Tensor, *, res: Optional[Act] = None,
               act_in: int = ACT_NONE, label: str = ""):
    """LINEAR_F32 (include/imagen_hip.h): y[r, :] = bias + f(x[r, :]) @ wt (+ res[r, :]), fp32 end to end.  x: an fp16 Act of rows or a 2-D fp32
    tensor; wt: fp32 [K, Cout] (the module's weight transposed); y: 2-D fp32 [rows, Cout]; res: fp16 Act of rows."""
    p = STRUCTS["ImagenLinearF32Params"]()
    assert wt.dtype == torch.float32 and wt.ndim == 2 and wt.is_contiguous() and y.dtype == torch.float32 and y.ndim == 2 and y.is_contiguous()
    K, Cout = wt.shape
    if isinstance(x, Act):
        assert x.C == K, f"{label}: {x.C} input channels, weight has {K}"
        p.x, p.rows, p.ld_x, p.x_f32, keep = x.ptr, x.rows, x.ld, 0, x.t
    else:
        assert x.dtype == torch.float32 and x.ndim == 2 and x.is_contiguous() and x.shape[1] == K
        p.x, p.rows, p.ld_x, p.x_f32, keep = x.data_ptr(), x.shape[0], x.shape[1], 1, x
    assert tuple(y.shape) == (p.rows, Cout), f"{label}: output {tuple(y.shape)} for {p.rows} rows of {Cout}"
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == Cout)
    p.wt, p.bias, p.y = wt.data_ptr(), (bias.data_ptr() if bias is not None else None), y.data_ptr()
    p.K, p.Cout, p.ld_y, p.act_in = K, Cout, Cout, act_in
    if res is not None:
        assert res.rows == p.rows and res.C == Cout
        p.res, p.ld_res = res.ptr, res.ld
    plan.add(p, label or "linear_f32", [keep, wt, bias, y, res.t if res is not None else None])
    return p


def step_slice(plan: Plan, segments, step_ptr: torch.Tensor, label: str = ""):
    """segments: up to four (table, dst) tensor pairs; table = [steps, <dst's bytes>] (same dtype / element count per step as dst, a multiple
    of 16 bytes): each launch copies row *step_ptr of every table into its dst (ImagenStepSliceParams)."""
    assert 1 <= len(segments) <= 4
    p = STRUCTS["ImagenStepSliceParams"]()
    keep = [step_ptr]
    for k, (tab, dst) in enumerate(segments):
        nbytes = dst.numel() * dst.element_size()
        assert nbytes % 16 == 0 and tab.dtype == dst.dtype and tab.is_contiguous() and dst.is_contiguous() and tab.numel() % dst.numel() == 0
        setattr(p, f"src{k}", tab.data_ptr())
        setattr(p, f"dst{k}", dst.data_ptr())
        setattr(p, f"words{k}", nbytes // 16)
        keep += [tab, dst]
    p.step_ptr = step_ptr.data_ptr()
    p.steps = segments[0][0].numel() // segments[0][1].numel()   # rows of the tables: the kernel clamps *step_ptr to them
    assert all(tab.numel() // dst.numel() == p.steps for tab, dst in segments), "step_slice: tables of one launch hold the same number of steps"
    plan.add(p, label or "step_slice", keep)
    return p


def pack_image(plan: Plan, a: torch.Tensor, b: Optional[torch.Tensor], out: Act, brep: int, label: str = ""):
    p = STRUCTS["ImagenPackImageParams"]()
    B, Ca, H, W = a.shape
    p.a, p.b, p.out = a.data_ptr(), ptr(b), out.ptr
    p.B, p.Brep, p.H, p.W, p.Ca, p.Cb, p.Cpad = B, brep, H, W, Ca, (b.shape[1] if b is not None else 0), out.C
    assert out.B == B * brep and out.ld == out.C
    plan.add(p, label or "pack_image", [a, b, out.t])
    return p


def rows_copy(plan: Plan, src: torch.Tensor, dst: torch.Tensor, *, B, rows, C, src_bs, src_rs, dst_bs, dst_rs, src_off=0, dst_off=0,
              label: str = ""):
    p = STRUCTS["ImagenRowsCopyParams"]()
    p.src, p.dst = src.data_ptr() + 2 * src_off, dst.data_ptr() + 2 * dst_off
    p.B, p.rows, p.C, p.src_bs, p.src_rs, p.dst_bs, p.dst_rs = B, rows, C, src_bs, src_rs, dst_bs, dst_rs
    plan.add(p, label or "rows_copy", [src, dst])
    return p


def memset32(plan: Plan, dst: torch.Tensor, value: int, count: Optional[int] = None, label: str = ""):
    p = STRUCTS["ImagenMemset32Params"]()
    p.dst, p.value, p.count = dst.data_ptr(), value, count if count is not None else dst.numel()
    plan.add(p, label or "memset32", [dst])
    return p


OBJECTIVES = {"noise": 0, "x_start": 1, "v": 2}


def cfg_x0(plan: Plan, x, pred, coef, step_ptr, x0, absx0, *, B, n_per_sample, cfg: bool, cond_scale: float, objective: str = "noise",
           label: str = ""):
    p = STRUCTS["ImagenCfgX0Params"]()
    p.x, p.pred, p.coef, p.step_ptr, p.x0, p.absx0 = x.data_ptr(), pred.data_ptr(), coef.data_ptr(), step_ptr.data_ptr(), x0.data_ptr(), absx0.data_ptr()
    p.B, p.n_per_sample, p.cfg, p.cond_scale, p.objective = B, n_per_sample, int(cfg), cond_scale, OBJECTIVES[objective]
    plan.add(p, label or "cfg_x0", [x, pred, coef, step_ptr, x0, absx0])
    return p


def quantile(plan: Plan, absx0, out, scratch, *, B, n, q: float, label: str = ""):
    p = STRUCTS["ImagenQuantileParams"]()
    p.absx0, p.out, p.scratch, p.B, p.n, p.q = absx0.data_ptr(), out.data_ptr(), scratch.data_ptr(), B, n, q
    W = ENUMS["IMAGEN_QUANTILE_SCRATCH_WORDS"]
    assert scratch.numel() >= B * W and scratch.dtype == torch.int32
    sv = scratch.view(-1)[: B * W].view(B, W)   # cleared once here; the op's last kernel re-clears it after every use
    sv.zero_()
    sv[:, 1025] = -1
    plan.add(p, label or "quantile", [absx0, out, scratch])
    return p


def ddpm_update(plan: Plan, x, x0, quant, coef, noise, final_out, step_ptr, *, B, n_per_sample, dynamic_threshold: bool,
                total_steps: int, seed: int, stream_id: int, sample_offset: int = 0, seed_ptr: Optional[torch.Tensor] = None,
                advance: bool = True, x0_thr: Optional[torch.Tensor] = None, row_keys: Optional[torch.Tensor] = None, label: str = ""):
    """row_keys: device int32 [B, 4] = (Philox key lo, key hi, global sample index, 0) per row — overrides seed / seed_ptr / sample_offset
    (requests merged into one batch: every row draws the noise of its own request)."""
    p = STRUCTS["ImagenDdpmUpdateParams"]()
    p.x0_thr = ptr(x0_thr)
    p.row_keys = ptr(row_keys)
    if row_keys is not None:
        assert row_keys.dtype == torch.int32 and tuple(row_keys.shape) == (B, 4) and row_keys.is_contiguous()
        plan.keep.append(row_keys)
    p.x, p.x0, p.quant, p.coef, p.noise, p.final_out, p.step_ptr = (x.data_ptr(), x0.data_ptr(), ptr(quant), coef.data_ptr(), ptr(noise),
                                                                   ptr(final_out), step_ptr.data_ptr())
    p.B, p.n_per_sample, p.dynamic_threshold, p.total_steps = B, n_per_sample, int(dynamic_threshold), total_steps
    p.sample_offset = sample_offset
    p.no_advance = 0 if advance else 1
    p.seed_ptr = ptr(seed_ptr)
    if seed_ptr is not None:
        plan.keep.append(seed_ptr)
    p.seed_lo, p.seed_hi, p.stream_id = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, stream_id
    plan.add(p, label or "ddpm_update", [x, x0, quant, coef, noise, final_out, step_ptr, x0_thr])
    return p


def randn(plan: Plan, out: torch.Tensor, *, seed: int, stream_id: int, tag: int, sample_offset: int = 0, label: str = ""):
    """out: fp32 [B, ...]; per-sample counter streams keyed by the GLOBAL sample index (shard-invariant noise)."""
    p = STRUCTS["ImagenRandnParams"]()
    p.out, p.B, p.n_per_sample, p.sample_offset = out.data_ptr(), out.shape[0], out[0].numel(), sample_offset
    p.seed_lo, p.seed_hi, p.stream_id, p.tag = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, stream_id, tag
    plan.add(p, label or "randn", [out])
    return p


def lowres_prep(plan: Plan, img: torch.Tensor, noise: torch.Tensor, out: torch.Tensor, *, alpha: float, sigma: float, label: str = ""):
    p = STRUCTS["ImagenLowresPrepParams"]()
    B, C, Hin, Win = img.shape
    p.img, p.noise, p.out = img.data_ptr(), noise.data_ptr(), out.data_ptr()
    p.B, p.C, p.Hin, p.Win, p.Hout, p.Wout, p.alpha, p.sigma = B, C, Hin, Win, out.shape[2], out.shape[3], alpha, sigma
    plan.add(p, label or "lowres_prep", [img, noise, out])
    return p


def lincomb(plan: Plan, t0, out, coef, step_ptr, *, B, n_per_sample, t1=None, t2=None, t3=None, q1=None, q3=None, out2=None,
            final_out=None, thr_mode: int = 0, final: bool = False, advance: bool = False, seed: int = 0, stream_id: int = 0,
            sample_offset: int = 0, seed_ptr: Optional[torch.Tensor] = None, mask=None, mask_else=None, label: str = ""):
    """Per-step state update (ImagenLincombParams): out = w0*t0 + w1*thr(t1) + w2*t2 + w3*thr(t3) + w4*z, out2 = w5*out,
    weights = coef[*step_ptr, 0:6]; with `mask` (fp32 0/1, same shape) out keeps `mask_else` where the mask is 0."""
    p = STRUCTS["ImagenLincombParams"]()
    p.t0, p.t1, p.t2, p.t3 = t0.data_ptr(), ptr(t1), ptr(t2), ptr(t3)
    p.q1, p.q3, p.out, p.out2, p.final_out = ptr(q1), ptr(q3), out.data_ptr(), ptr(out2), ptr(final_out)
    p.coef, p.step_ptr, p.seed_ptr = coef.data_ptr(), step_ptr.data_ptr(), ptr(seed_ptr)
    p.B, p.n_per_sample, p.thr_mode, p.final, p.advance, p.sample_offset = B, n_per_sample, thr_mode, int(final), int(advance), sample_offset
    p.seed_lo, p.seed_hi, p.stream_id = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, stream_id
    p.mask, p.mask_else = ptr(mask), ptr(mask_else)
    assert coef.dtype == torch.float32 and coef.shape[-1] == 8
    assert mask is None or (mask.dtype == torch.float32 and mask_else is not None and mask.numel() == B * n_per_sample)
    plan.add(p, label or "lincomb", [t0, t1, t2, t3, q1, q3, out, out2, final_out, coef, step_ptr, seed_ptr, mask, mask_else])
    return p


# ------------------------------------------------------------------------------------------------ Imagen-Video ops

def temporal_peg(plan: Plan, x: Act, w: torch.Tensor, bias: torch.Tensor, out: Act, *, B: int, F: int, causal: bool, label: str = ""):
    """x / out: the clip as B*F consecutive NHWC frames (Act.B == B*F, dense); w: fp32 [C, 3] depthwise taps; bias: fp32 [C]."""
    assert x.B == B * F and out.B == B * F and x.ld == x.C == out.ld == out.C and x.bs == x.H * x.W * x.C == out.bs
    assert tuple(w.shape) == (x.C, 3) and w.dtype == torch.float32 and bias.numel() == x.C
    p = STRUCTS["ImagenTemporalPegParams"]()
    p.x, p.w, p.bias, p.out = x.ptr, w.data_ptr(), bias.data_ptr(), out.ptr
    p.B, p.F, p.P, p.C, p.causal = B, F, x.H * x.W, x.C, int(causal)
    plan.add(p, label or "temporal_peg", [x.t, w, bias, out.t])
    return p


def temporal_attention(plan: Plan, qkv: Act, null_kv: torch.Tensor, q_scale: torch.Tensor, k_scale: torch.Tensor, bias: torch.Tensor,
                       o: Act, *, B: int, F: int, P: int, heads: int, causal: bool, scale: float, label: str = ""):
    """qkv: rows (b, f, p) of q (heads*64) | k (64) | v (64); bias: fp32 [heads, F, F+1] (column 0 = null key); o: rows of heads*64."""
    assert qkv.rows == B * F * P == o.rows and qkv.C == heads * 64 + 128 and o.C == heads * 64
    assert tuple(bias.shape) == (heads, F, F + 1) and bias.dtype == torch.float32 and F <= 32
    p = STRUCTS["ImagenTemporalAttentionParams"]()
    p.qkv, p.null_kv, p.q_scale, p.k_scale, p.bias, p.o = qkv.ptr, null_kv.data_ptr(), q_scale.data_ptr(), k_scale.data_ptr(), bias.data_ptr(), o.ptr
    p.B, p.F, p.P, p.heads, p.ld, p.ld_o, p.causal, p.scale = B, F, P, heads, qkv.ld, o.ld, int(causal), scale
    plan.add(p, label or "temporal_attention", [qkv.t, null_kv, q_scale, k_scale, bias, o.t])
    return p
