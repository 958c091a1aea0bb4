"""MI355X: EVERY igemm tile configuration (both kernel families) x every k-loop instantiation x every tile shape the launcher accepts,
with an explicit cfg=, against the fp32 torch restatement of the op contract (tests/igemm_case.py).

Round-1 gap this closes: pick_cfg only selects the big-tile configurations for layers with >= 1024 workgroups, so tests that let the
planner choose never reached the instantiations the benchmark spends its time in.  Here the configuration is forced, the shapes are
small and ragged (partial tiles in both directions, several output-channel tiles, a concat boundary inside the channel range).
Tolerance: 1e-3 normwise (BASELINE.json north_star), weights' fp16 rounding included.
"""
import pytest
import torch

from igemm_case import run_case

pytestmark = pytest.mark.gpu
TOL = 1e-3
EMULATED = __import__("os").environ.get("IMAGEN_EMUL_TESTS") == "1"   # on the CPU through tools/emul (conftest.py)


@pytest.fixture(scope="module")
def ops():
    from imagen_pytorch_amd import ops as o

    return o


@pytest.fixture(scope="module")
def dev():
    import os
    return torch.device("cpu" if os.environ.get("IMAGEN_EMUL_TESTS") == "1" else "cuda:0")   # cpu: through the emulated library (tools/emul)


def _num_cfgs():
    try:
        from imagen_pytorch_amd import ops as o

        return len(o.cfg_table())
    except Exception:   # library not built yet: collection must still work
        return 44


NUM_CFGS = _num_cfgs()


def _shapes(ops, cfg, K, stride, H, W):
    return [(th, tw) for _, _, th, tw in ops.launchable_shapes(cfg, (H - K) // stride + 1 if stride > 1 else H, (W - K) // stride + 1 if stride > 1 else W,
                                                               K, K, stride)]


@pytest.mark.parametrize("cfg", range(NUM_CFGS))
def test_conv3x3_block_every_cfg_and_tile_shape(ops, dev, cfg):
    """Block conv (prologue rs * pa + ps -> SiLU, concat input, bias), plain NHWC output with ssq_out where one tile covers Cout."""
    tp, bn, G, fam = ops.cfg_table()[cfg]
    if G not in (1, 4) or fam in (2, 3, 5, 6, 7):
        pytest.skip("3x3 convs with a prologue use 8- or 32-channel chunks of families 0 / 1")
    C1, C2 = (64, 32) if G == 4 else (16, 8)
    H, W = (40, 36) if tp >= 128 else (20, 24)
    shapes = _shapes(ops, cfg, 3, 1, H, W)
    if not shapes:
        pytest.skip("cfg has no 3x3 instantiation")
    for th, tw in shapes:
        for Cout in sorted({bn, 2 * bn + 32 if bn < 128 else bn + 64}):
            full = Cout <= bn
            r = run_case(ops, dev, B=2, H=H, W=W, C1=C1, C2=C2, Cout=Cout, K=3, G=G, cfg=(cfg, th, tw), prologue="rs", ssq_out=full)
            assert r["err"] < TOL, (cfg, th, tw, Cout, r)
            if full:
                assert r["err_ssq"] < 2e-3, (cfg, th, tw, Cout, r)


@pytest.mark.parametrize("cfg", range(NUM_CFGS))
def test_conv3x3_raw_post_and_ssq_prologue_every_cfg(ops, dev, cfg):
    """The two fused forms of a ResnetBlock: conv1 with ssq statistics (ssq_a + wb * ssq_b over the concat) and the output-side
    Block prologue (post_pa); conv2 staging an already activated input with no arithmetic (prologue none)."""
    tp, bn, G, fam = ops.cfg_table()[cfg]
    if G != 4 or fam in (2, 3, 5, 6, 7):
        pytest.skip("32-channel chunks of families 0 / 1 only")
    H, W = (32, 48) if tp >= 128 else (16, 24)
    shapes = _shapes(ops, cfg, 3, 1, H, W)
    if not shapes:
        pytest.skip("cfg has no 3x3 instantiation")
    th, tw = shapes[0]
    Cout = bn
    r = run_case(ops, dev, B=2, H=H, W=W, C1=32, C2=32, Cout=Cout, K=3, G=G, cfg=(cfg, th, tw), prologue="ssq", affine=False, epilogue="post")
    assert r["err"] < TOL, ("ssq+post", cfg, r)
    r = run_case(ops, dev, B=2, H=H, W=W, C1=64, C2=0, Cout=Cout, K=3, G=G, cfg=(cfg, th, tw), prologue="none", act_in="none", gca=(fam == 1))
    assert r["err"] < TOL and r.get("err_gca", 0.0) < 2e-3, ("raw", cfg, r)


@pytest.mark.parametrize("cfg", range(NUM_CFGS))
def test_conv_dma_every_cfg(ops, dev, cfg):
    """The all-DMA family (csrc/conv_dma.hip): prologue-free single-input 3x3 convs; ragged image sizes (partial tiles, zero padding
    from the zero page), 1-3 channel chunks (both chunk parities and the ring wrap), several output-channel tiles, every epilogue."""
    tp, bn, G, fam = ops.cfg_table()[cfg]
    if fam != 2:
        pytest.skip("family 2 only")
    H, W = (40, 36) if tp >= 128 else (20, 24)
    (th, tw), = _shapes(ops, cfg, 3, 1, H, W)
    raw = dict(prologue="none", act_in="none", K=3, G=4, cfg=(cfg, th, tw))
    for Cin in (32, 64, 96):
        r = run_case(ops, dev, B=2, H=H, W=W, C1=Cin, Cout=bn, ssq_out=True, gca=True, **raw)
        assert r["err"] < TOL and r["err_ssq"] < 2e-3 and r["err_gca"] < 2e-3, (cfg, Cin, r)
    r = run_case(ops, dev, B=2, H=H - 5, W=W - 3, C1=64, Cout=bn - 8, gca=True, **raw)   # ragged tiles, couts that do not fill the tile
    assert r["err"] < TOL and r["err_gca"] < 2e-3, (cfg, "gca ragged", r)
    r = run_case(ops, dev, B=3, H=H - 3, W=W + 5, C1=128, Cout=bn + 64 if bn >= 128 else 2 * bn + 32, **raw)
    assert r["err"] < TOL, (cfg, "cout tiles", r)
    for ep in ("post", "addend", "res"):
        r = run_case(ops, dev, B=2, H=H, W=W, C1=64, Cout=bn, epilogue=ep, **raw)
        assert r["err"] < TOL, (cfg, ep, r)
    if bn <= 128:
        r = run_case(ops, dev, B=2, H=H, W=W, C1=32, Cout=3, epilogue="nchw", **raw)
        assert r["err"] < TOL, (cfg, "nchw", r)


@pytest.mark.parametrize("cfg", range(NUM_CFGS))
def test_conv_big_every_cfg(ops, dev, cfg):
    """The big-tile all-DMA family (csrc/conv_big.hip): prologue-free single-input 3x3 convs to 128-cout tiles, 64 x 64 per wave, shared
    weight ring stages of one tap row, one barrier per tap row (and the K split over two wave groups on the 128-pixel tiles); ragged
    images (partial tiles, zero padding from the zero page), 1-6 channel chunks (every ring / halo-buffer phase, the stream's repeated last
    stage), one and two output-channel tile columns, every epilogue incl. the GlobalContext partials."""
    tp, bn, G, fam = ops.cfg_table()[cfg]
    if fam != 5:
        pytest.skip("family 5 only")
    H, W = 40, 36
    (th, tw), = _shapes(ops, cfg, 3, 1, H, W)
    raw = dict(prologue="none", act_in="none", K=3, G=4, cfg=(cfg, th, tw))
    for Cin in (32, 64, 96, 128, 192):
        r = run_case(ops, dev, B=2, H=H, W=W, C1=Cin, Cout=bn, ssq_out=True, gca=True, **raw)
        assert r["err"] < TOL and r["err_ssq"] < 2e-3 and r["err_gca"] < 2e-3, (cfg, Cin, r)
    r = run_case(ops, dev, B=2, H=H - 5, W=W - 3, C1=64, Cout=bn - 8, gca=True, **raw)   # ragged tiles, couts that do not fill the tile
    assert r["err"] < TOL and r["err_gca"] < 2e-3, (cfg, "gca ragged", r)
    r = run_case(ops, dev, B=3, H=H - 3, W=W + 5, C1=128, Cout=2 * bn, **raw)
    assert r["err"] < TOL, (cfg, "cout tiles", r)
    for ep in ("post", "addend", "res"):
        r = run_case(ops, dev, B=2, H=H, W=W, C1=64, Cout=bn, epilogue=ep, **raw)
        assert r["err"] < TOL, (cfg, ep, r)
    r = run_case(ops, dev, B=2, H=H, W=W, C1=32, Cout=3, epilogue="nchw", **raw)
    assert r["err"] < TOL, (cfg, "nchw", r)


def test_conv_stream_family(ops, dev):
    """The streaming family (csrc/conv_stream.hip): 3x3 convs to <= 32 channels from one or two 32-channel inputs — raw inputs and the
    ssq-statistics Block prologue (per-pixel sums of squares of both inputs, per-channel gain or per-(batch, channel) affine, SiLU),
    every epilogue, ragged images (partial tiles, zero padding), and a map with more tiles than resident workgroups so that every
    workgroup walks several tiles (the cross-tile prefetch / double buffer / statistics hand-over)."""
    sid = ops.stream_cfg()
    if sid is None and EMULATED:
        pytest.skip("the emulated library holds the wave-specialised family only")
    assert sid is not None
    cfg = (sid, 16, 16)
    base = dict(K=3, G=4, cfg=cfg, Cout=32)
    raw = dict(prologue="none", act_in="none")
    for C2 in (0, 32):
        for kw in (dict(raw, ssq_out=True), dict(prologue="ssq", affine=False, ssq_out=True), dict(prologue="ssq", affine=True),
                   dict(prologue="ssq", affine=True, act_in="none"), dict(raw, epilogue="post"), dict(prologue="ssq", affine=False, epilogue="post"),
                   dict(raw, epilogue="addend"), dict(prologue="ssq", affine=False, epilogue="res")):
            r = run_case(ops, dev, B=2, H=40, W=36, C1=32, C2=C2, **base, **kw)
            assert r["err"] < TOL and r.get("err_ssq", 0.0) < 2e-3, (C2, kw, r)
        r = run_case(ops, dev, B=3, H=27, W=45, C1=32, C2=C2, K=3, G=4, cfg=cfg, Cout=24, prologue="ssq", affine=False)   # couts that do not fill the tile
        assert r["err"] < TOL, (C2, "ragged", r)
        r = run_case(ops, dev, B=2, H=33, W=20, C1=32, C2=C2, K=3, G=4, cfg=cfg, Cout=3, epilogue="nchw", **raw)
        assert r["err"] < TOL, (C2, "nchw", r)
    # 2 x 17 x 17 = 578 tiles (one input: 512 resident workgroups) / 2 x 23 x 12 = 552 tiles (two inputs: 256)
    r = run_case(ops, dev, B=2, H=272, W=272, C1=32, C2=0, **base, prologue="ssq", affine=True, ssq_out=True)
    assert r["err"] < TOL and r["err_ssq"] < 2e-3, ("many tiles", r)
    r = run_case(ops, dev, B=2, H=360, W=190, C1=32, C2=32, **base, prologue="ssq", affine=False, epilogue="post")
    assert r["err"] < TOL, ("many tiles, two inputs", r)
    r = run_case(ops, dev, B=2, H=272, W=272, C1=32, C2=0, **base, **raw, ssq_out=True)
    assert r["err"] < TOL and r["err_ssq"] < 2e-3, ("many tiles, raw", r)


def test_conv_pro_family(ops, dev):
    """The streaming family with the Block prologue on register-staged rows (csrc/conv_pro.hip): 3x3 convs to exactly 32 channels from one
    or two 32-channel inputs — the ssq-statistics SiLU prologue (shared gain, per-(batch, channel) affine with shifts) and raw inputs, the
    plain / ssq_out / post epilogues, ragged images (partial 8 x 16 tiles, zero padding of the ACTIVATED tensor), images of a single tile
    (the per-image parameter sets alternate every tile), and maps with several tiles per persistent workgroup (contiguous tile ranges, the
    register-staged rows two tiles ahead, the double-buffered tile image)."""
    pid = ops.pro_cfg(32)
    assert pid is not None and ops.pro_cfg(64) is not None
    cfg = (pid, 8, 16)
    base = dict(K=3, G=4, cfg=cfg, Cout=32, C1=32)
    raw = dict(prologue="none", act_in="none")
    for C2 in (0, 32):
        for kw in (dict(prologue="ssq", affine=False, ssq_out=True), dict(prologue="ssq", affine=True, ssq_out=True), dict(prologue="ssq", affine=False, epilogue="post"),
                   dict(prologue="ssq", affine=True, epilogue="post"), dict(raw, ssq_out=True), dict(raw, epilogue="post"), dict(prologue="ssq", affine=False)):
            r = run_case(ops, dev, B=2, H=40, W=36, C2=C2, **base, **kw)
            assert r["err"] < TOL and r.get("err_ssq", 0.0) < 2e-3, (C2, kw, r)
        r = run_case(ops, dev, B=3, H=27, W=45, C2=C2, **base, prologue="ssq", affine=True, ssq_out=True)   # ragged right / bottom edges
        assert r["err"] < TOL and r["err_ssq"] < 2e-3, (C2, "ragged", r)
        r = run_case(ops, dev, B=9, H=8, W=16, C2=C2, **base, prologue="ssq", affine=True, epilogue="post")   # one tile per image
        assert r["err"] < TOL, (C2, "single-tile images", r)
    # 64 output channels: eight waves (two cout blocks per pixel block), the last chunks' weights from LDS, the output-side norm across the two
    # cout waves of a pixel; 64 | 64 + 32 | 32 + 32 input channels; ranges of odd length compute their last tile twice
    base64 = dict(K=3, G=4, cfg=(ops.pro_cfg(64), 8, 16), Cout=64)
    for C1, C2 in ((64, 32), (32, 32), (64, 0)):
        for kw in (dict(prologue="ssq", affine=False, epilogue="post"), dict(prologue="ssq", affine=True, epilogue="post"), dict(prologue="ssq", affine=True),
                   dict(raw), dict(raw, epilogue="post")):
            r = run_case(ops, dev, B=3, H=27, W=45, C1=C1, C2=C2, **base64, **kw)
            assert r["err"] < TOL, ("64 couts", C1, C2, kw, r)
    r = run_case(ops, dev, B=5, H=24, W=16, C1=32, C2=32, **base64, prologue="ssq", affine=True, epilogue="post")
    assert r["err"] < TOL, ("64 couts, odd ranges", r)
    if not EMULATED:   # 2 x 34 x 17 = 1156 tiles (one input: 768 resident workgroups) / 2 x 45 x 12 = 1080 tiles (two inputs: 512)
        r = run_case(ops, dev, B=16, H=128, W=128, C1=64, C2=32, **base64, prologue="ssq", affine=False, epilogue="post")
        assert r["err"] < TOL, ("96 -> 64 at the benchmark's size", r)
        r = run_case(ops, dev, B=2, H=272, W=272, C2=0, **base, prologue="ssq", affine=True, ssq_out=True)
        assert r["err"] < TOL and r["err_ssq"] < 2e-3, ("many tiles", r)
        r = run_case(ops, dev, B=2, H=360, W=190, C2=32, **base, prologue="ssq", affine=False, epilogue="post")
        assert r["err"] < TOL, ("many tiles, two inputs", r)
        r = run_case(ops, dev, B=16, H=64, W=64, C2=32, **base, **raw, ssq_out=True)
        assert r["err"] < TOL and r["err_ssq"] < 2e-3, ("many tiles, raw, two inputs", r)


def test_conv_pw_family(ops, dev):
    """The streaming pointwise family (csrc/conv_pw.hip): 1x1 convs over one or two raw 32-channel-chunk inputs with the res_conv epilogues
    (bias + gate * addend | residual | plain) and the per-pixel sum of squares — every instantiation, ragged pixel counts (partial last
    tile), several tiles per persistent workgroup, Cout below the tile width; and the planner's own pick for a benchmark-sized launch."""
    tab = ops.cfg_table()
    cfgs = [i for i, c in enumerate(tab) if c[3] == 4]
    if not cfgs:
        pytest.skip("the library holds no streaming pointwise family")
    for cfg in cfgs:
        tp, bn, kch, _ = tab[cfg]
        for (C1, C2) in {(32 * (kch - 1), 32), (32 * kch, 0)}:
            if C1 == 0:
                continue
            for ep, Cout, H, W in (("addend", bn, 40, 52), ("res", bn - 8, 16, 16), ("plain", bn, 9, 31)):
                G = 8 if (C1 + C2) % 64 == 0 else 4
                r = run_case(ops, dev, B=3, H=H, W=W, C1=C1, C2=C2, Cout=Cout, K=1, G=G, cfg=(cfg, 1, tp), prologue="none", act_in="none", epilogue=ep,
                             ssq_out=True)
                assert r["err"] < TOL and r["err_ssq"] < 2e-3, (cfg, C1, C2, ep, r)
    # the planner picks the family for the large res_conv launches by itself
    r = run_case(ops, dev, B=16, H=128, W=128, C1=64, C2=32, Cout=64, K=1, prologue="none", act_in="none", epilogue="addend", ssq_out=True)
    assert tab[r["cfg"][0]][3] == 4 and r["err"] < TOL and r["err_ssq"] < 2e-3, r
    r = run_case(ops, dev, B=16, H=64, W=64, C1=128, C2=64, Cout=128, K=1, prologue="none", act_in="none", epilogue="addend", ssq_out=True)   # two waves per pixel block
    assert tab[r["cfg"][0]][3] == 4 and tab[r["cfg"][0]][0] == 128 and r["err"] < TOL and r["err_ssq"] < 2e-3, r


def test_conv_gemm_family(ops, dev):
    """The tiled pointwise GEMM (csrc/conv_gemm.hip): 1x1 layers in 32-channel chunks — raw rows, the LayerNorm prologue (mu + rs statistics,
    per-(batch, channel) or shared affine) and a bare per-row scale; one or two inputs; every epilogue (plain + ssq_out, GELU, gate * addend,
    residual, pixel shuffle, fp32 NCHW, post_pa, GlobalContext partials); several cout slabs, couts below the slab, ragged rows (partial
    tiles in both tile directions), an odd and an even number of chunks; and the planner's own pick for the benchmark's token GEMMs."""
    tab = ops.cfg_table()
    gid = ops.gemm_cfg()
    if gid is None:
        pytest.skip("the library holds no tiled pointwise GEMM family")
    raw = dict(prologue="none", act_in="none")
    ln = dict(prologue="ln", act_in="none")
    cases = [
        dict(raw, B=2, H=1, W=300, C1=256, Cout=256, cfg=(gid, 1, 128), ssq_out=False),                         # tokens, two slabs, ragged rows
        dict(ln, B=2, H=1, W=256, C1=128, Cout=640, cfg=(gid, 1, 128), affine=True),                           # qkv: LN prologue, five slabs
        dict(ln, B=2, H=1, W=130, C1=256, Cout=512, cfg=(gid, 1, 128), affine=False, act_out="gelu"),          # FeedForward lin1: shared gain, GELU
        dict(raw, B=2, H=1, W=256, C1=512, Cout=256, cfg=(gid, 1, 128), epilogue="res"),                       # to_out / lin2 + residual
        dict(raw, B=2, H=20, W=32, C1=256, C2=128, Cout=256, cfg=(gid, 4, 32), epilogue="addend"),             # res_conv of the 32^2 level
        dict(raw, B=2, H=9, W=20, C1=96, C2=32, Cout=120, cfg=(gid, 8, 16), ssq_out=True),                      # ragged both ways, couts below the slab, ssq_out
        dict(raw, B=2, H=16, W=64, C1=128, Cout=256, cfg=(gid, 2, 64), epilogue="shuffle"),                    # pixel-shuffle upsample GEMM
        dict(raw, B=2, H=12, W=32, C1=160, Cout=3, cfg=(gid, 4, 32), epilogue="nchw"),                         # fp32 NCHW, odd chunk count
        dict(dict(prologue="rs", act_in="none"), B=2, H=1, W=200, C1=128, Cout=128, cfg=(gid, 1, 128), epilogue="post"),   # per-row scale only; post_pa
        dict(raw, B=2, H=16, W=16, C1=128, Cout=128, cfg=(gid, 8, 16), gca=True, ssq_out=True),                 # GlobalContext partials
    ]
    for kw in cases:
        r = run_case(ops, dev, K=1, **kw)
        assert r["err"] < TOL and r.get("err_ssq", 0.0) < 2e-3 and r.get("err_gca", 0.0) < 2e-3, (kw, r)
    if not EMULATED:
        # the planner picks the family for the benchmark's token GEMMs and the 32^2 res_conv by itself
        r = run_case(ops, dev, B=16, H=1, W=1024, C1=256, Cout=640, K=1, **ln)
        assert tab[r["cfg"][0]][3] == 7 and r["err"] < TOL, r
        r = run_case(ops, dev, B=16, H=1, W=1024, C1=512, Cout=256, K=1, **raw)
        assert tab[r["cfg"][0]][3] == 7 and r["err"] < TOL, r
        # ... and leaves the generic-epilogue launches where they were (measured slower here: DESIGN 9.7)
        r = run_case(ops, dev, B=16, H=32, W=32, C1=256, C2=128, Cout=256, K=1, epilogue="addend", **raw)
        assert tab[r["cfg"][0]][3] != 7 and r["err"] < TOL, r


def test_conv_small_family(ops, dev):
    """The small-map 3x3 family (csrc/conv_small.hip): 32-pixel tiles as 4 x 8 / 2 x 16 / 1 x 32, the K loop split over the 8 / 4 / 2 waves of a
    cout fragment (even and uneven splits: 2, 3, 8, 12 and 16 channel chunks), every prologue (none, per-pixel scale, ssq statistics over a
    concat, LayerNorm statistics; shared and per-batch affine), every epilogue incl. the all-cout ones on the 64- / 128-cout tiles (ssq_out,
    post_pa, GlobalContext partials), several cout tiles, couts below the tile, ragged maps; and the planner's own pick on the benchmark's
    8^2 / 16^2 layers."""
    tab = ops.cfg_table()
    fam8 = {tab[i][1]: i for i in range(len(tab)) if tab[i][3] == 8}
    if not fam8:
        pytest.skip("the library holds no small-map family")
    raw = dict(prologue="none", act_in="none")
    cases = [
        dict(B=2, H=8, W=8, C1=128, Cout=128, cfg=(fam8[32], 4, 8), prologue="ssq", affine=False),                          # 8^2: four cout tiles, 8 K slices
        dict(B=2, H=8, W=8, C1=256, C2=128, Cout=256, cfg=(fam8[32], 4, 8), prologue="ssq", affine=True),                   # the 384 -> 256 concat Block, per-batch scale / shift
        dict(raw, B=2, H=8, W=8, C1=128, Cout=128, cfg=(fam8[128], 4, 8), gca=True, ssq_out=True),                           # all couts in one tile: GlobalContext partials + ssq_out (2 K slices)
        dict(raw, B=2, H=8, W=8, C1=128, Cout=128, cfg=(fam8[128], 4, 8), epilogue="post"),                                  # the next Block's prologue applied by the producer
        dict(B=2, H=16, W=16, C1=128, C2=64, Cout=128, cfg=(fam8[128], 2, 16), prologue="ssq", affine=False, epilogue="post"),   # 16^2 concat Block, 2 x 16 tiles
        dict(raw, B=2, H=16, W=16, C1=64, Cout=64, cfg=(fam8[64], 2, 16), gca=True),                                          # 64-cout tile, 4 K slices
        dict(B=2, H=16, W=16, C1=64, Cout=64, cfg=(fam8[64], 2, 16), prologue="ssq", affine=False, ssq_out=True),
        dict(B=2, H=10, W=20, C1=96, Cout=72, cfg=(fam8[32], 2, 16), prologue="rs"),                                           # ragged both ways, 3 chunks (54 steps over 8 slices), couts below the last tile
        dict(B=2, H=6, W=40, C1=64, C2=32, Cout=96, cfg=(fam8[32], 1, 32), prologue="ln", affine=True),                       # 1 x 32 tiles, LayerNorm statistics
        dict(raw, B=2, H=8, W=8, C1=512, Cout=64, cfg=(fam8[64], 4, 8), epilogue="addend"),                                   # 16 chunks; gate * addend
        dict(raw, B=2, H=12, W=8, C1=64, Cout=128, cfg=(fam8[32], 4, 8), epilogue="res"),
        dict(raw, B=2, H=8, W=16, C1=64, Cout=3, cfg=(fam8[32], 2, 16), epilogue="nchw"),
        dict(raw, B=2, H=8, W=8, C1=64, Cout=64, cfg=(fam8[32], 4, 8), epilogue="shuffle", act_out="silu"),
        dict(raw, B=3, H=8, W=8, C1=32, Cout=32, cfg=(fam8[32], 4, 8), bias=False),                                            # one chunk: 18 steps over 8 slices
        dict(raw, B=4, H=16, W=16, C1=32, Cout=96, cfg=(fam8[32], 2, 16)),                                                     # the map outweighs the weights: cout slab fastest in the tile order
    ]
    for kw in cases:
        r = run_case(ops, dev, K=3, G=4, **kw)
        assert r["err"] < TOL and r.get("err_ssq", 0.0) < 2e-3 and r.get("err_gca", 0.0) < 2e-3, (kw, r)
    # the same kernel with one tap and no halo: the 1x1 res_conv (gate * addend epilogue over a concat) and upsample (SiLU + pixel shuffle) GEMMs of the
    # small maps, every packing (G = 4 | 8 | 16 by the channel count), the LayerNorm prologue, a plain output with statistics
    cases_1x1 = [
        dict(raw, B=2, H=8, W=8, C1=256, C2=128, Cout=256, cfg=(fam8[32], 4, 8), epilogue="addend"),
        dict(raw, B=2, H=16, W=16, C1=128, C2=64, Cout=128, cfg=(fam8[32], 2, 16), epilogue="addend"),
        dict(raw, B=2, H=8, W=8, C1=256, Cout=512, cfg=(fam8[32], 4, 8), epilogue="shuffle", act_out="silu"),
        dict(raw, B=2, H=6, W=40, C1=96, Cout=64, cfg=(fam8[64], 1, 32), epilogue="res", ssq_out=True),
        dict(B=2, H=8, W=8, C1=64, Cout=96, cfg=(fam8[32], 4, 8), prologue="ln", affine=True, act_in="none"),   # 2 chunks over 8 K slices: waves without a unit
        dict(raw, B=2, H=8, W=8, C1=32, Cout=32, cfg=(fam8[32], 4, 8)),                                          # one chunk: seven of the eight waves idle
    ]
    for kw in cases_1x1:
        r = run_case(ops, dev, K=1, **kw)
        assert r["err"] < TOL and r.get("err_ssq", 0.0) < 2e-3, (kw, r)
    if not EMULATED:
        r = run_case(ops, dev, B=16, H=8, W=8, C1=256, C2=128, Cout=256, K=1, epilogue="addend", **raw)    # README unet1's res_conv of the 8^2 level
        assert tab[r["cfg"][0]][3] == 8 and r["err"] < TOL, r
        # the planner takes the benchmark's small maps (16 rows of 8^2 / 16^2) by itself, with the tile the epilogue needs ...
        r = run_case(ops, dev, B=16, H=8, W=8, C1=256, C2=128, Cout=256, K=3, prologue="ssq", affine=True)
        assert tab[r["cfg"][0]][3] == 8 and tab[r["cfg"][0]][1] == 32 and r["err"] < TOL, r
        r = run_case(ops, dev, B=16, H=16, W=16, C1=128, Cout=128, K=3, gca=True, **raw)
        assert tab[r["cfg"][0]][3] == 8 and tab[r["cfg"][0]][1] == 128 and r["err"] < TOL and r["err_gca"] < 2e-3, r
        # ... and leaves the larger maps where they were
        r = run_case(ops, dev, B=16, H=32, W=32, C1=128, Cout=128, K=3, **raw)
        assert tab[r["cfg"][0]][3] != 8 and r["err"] < TOL, r


def test_act_prep(ops, dev):
    """ACT_PREP: the Block prologue as its own pass (ssq statistics over a two-tensor concat, per-channel gain, SiLU; and the
    LayerNorm form with a per-(batch, channel) affine) vs fp32 torch."""
    import torch.nn.functional as F

    torch.manual_seed(0)
    B, H, W, C1, C2 = 2, 12, 20, 64, 32
    x1, x2 = torch.randn(B, C1, H, W).half().float(), (torch.randn(B, C2, H, W) * 0.7).half().float()
    wb = 0.5
    g = 1 + 0.2 * torch.randn(C1 + C2)
    q = (x1 * x1).sum(1, keepdim=True) + wb * (x2 * x2).sum(1, keepdim=True)
    ref = F.silu(torch.cat((x1, x2), 1) / q.sqrt() * g.view(1, -1, 1, 1))
    a1, a2 = ops.act_from_nchw(x1.to(dev)), ops.act_from_nchw(x2.to(dev))
    y = ops.new_act(B, H, W, C1 + C2, dev)
    plan = ops.Plan()
    ops.act_prep(plan, a1, y, x2=a2, ssq_a=(x1 * x1).sum(1).reshape(-1).to(dev), ssq_b=(x2 * x2).sum(1).reshape(-1).to(dev), ssq_wb=wb,
                 pa=g.to(dev), pstride=0, act_in=ops.ACT_SILU)
    plan.run()
    torch.cuda.synchronize()
    assert nerr_(ops.act_to_nchw(y), ref) < 5e-4
    xin = x1
    mu, rs = xin.mean(1, keepdim=True), torch.rsqrt(xin.var(1, unbiased=False, keepdim=True) + 1e-5)
    pa, ps = 1 + 0.2 * torch.randn(B, C1), 0.2 * torch.randn(B, C1)
    ref = (xin - mu) * rs * pa.view(B, C1, 1, 1) + ps.view(B, C1, 1, 1)
    y = ops.new_act(B, H, W, C1, dev)
    plan = ops.Plan()
    ops.act_prep(plan, a1, y, mu=mu.reshape(-1).contiguous().to(dev), rs=rs.reshape(-1).contiguous().to(dev), pa=pa.to(dev), ps=ps.to(dev), pstride=C1)
    plan.run()
    torch.cuda.synchronize()
    assert nerr_(ops.act_to_nchw(y), ref) < 5e-4
    # self_stat: the launch reduces x1's sum of squares itself (16 / 32 / 64 lanes per pixel), the skip tensor's comes from its producer;
    # per-channel gain, or per-(batch, channel) scale + shift
    for (c1, c2, affine) in ((64, 32, False), (256, 0, True), (256, 128, False), (128, 0, False)):
        x1 = torch.randn(B, c1, H, W).half().float()
        x2 = (torch.randn(B, c2, H, W) * 0.7).half().float() if c2 else None
        C = c1 + c2
        q = (x1 * x1).sum(1, keepdim=True) + (wb * (x2 * x2).sum(1, keepdim=True) if c2 else 0.0)
        pa = 1 + 0.2 * torch.randn(B if affine else 1, C)
        ps = 0.2 * torch.randn(B, C) if affine else None
        xin = x1 if x2 is None else torch.cat((x1, x2), 1)
        ref = xin / q.sqrt() * pa.view(-1, C, 1, 1)
        if ps is not None:
            ref = ref + ps.view(B, C, 1, 1)
        ref = F.silu(ref)
        a1 = ops.act_from_nchw(x1.to(dev))
        a2 = ops.act_from_nchw(x2.to(dev)) if c2 else None
        y = ops.new_act(B, H, W, C, dev)
        plan = ops.Plan()
        ops.act_prep(plan, a1, y, x2=a2, ssq_b=(x2 * x2).sum(1).reshape(-1).to(dev) if c2 else None, ssq_wb=wb, pa=pa.contiguous().to(dev),
                     ps=ps.to(dev) if ps is not None else None, pstride=C if affine else 0, act_in=ops.ACT_SILU, self_stat=True)
        plan.run()
        torch.cuda.synchronize()
        assert nerr_(ops.act_to_nchw(y), ref) < 5e-4, (c1, c2, affine)


def nerr_(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


@pytest.mark.parametrize("cfg", range(NUM_CFGS))
def test_1x1_every_cfg_and_epilogue(ops, dev, cfg):
    """1x1 convs / linears: LayerNorm prologue + GELU, res_conv forms (gate * addend, residual), pixel-shuffle + SiLU, fp32 NCHW."""
    tp, bn, G, fam = ops.cfg_table()[cfg]
    if fam == 4:
        pytest.skip("the streaming pointwise family has its own test (test_conv_pw_family)")
    if fam == 7:
        pytest.skip("the tiled pointwise GEMM has its own test (test_conv_gemm_family)")
    Cin = {1: 24, 4: 96, 8: 192, 16: 256}[G]
    H, W = (24, 40) if tp >= 128 else (12, 20)
    shapes = _shapes(ops, cfg, 1, 1, H, W)
    if not shapes:
        pytest.skip("cfg has no 1x1 instantiation")
    th, tw = shapes[-1]
    Cout = 2 * bn if bn <= 64 else bn + 32
    cases = [dict(prologue="ln", affine=False, act_in="none", act_out="gelu"),
             dict(prologue="none", act_in="none", epilogue="addend", C2=Cin // 3 // 8 * 8, C1=Cin - Cin // 3 // 8 * 8, ssq_out=False),
             dict(prologue="none", act_in="none", epilogue="res"),
             dict(prologue="none", act_in="none", act_out="silu", epilogue="shuffle"),
             dict(prologue="rs", epilogue="nchw", Cout=3)]
    if bn > 128:
        cases.pop()   # (a 3-channel output is packed to 128 couts: narrower than this tile)
    for kw in cases:
        kw = dict(dict(C1=Cin, C2=0, Cout=Cout), **kw)
        r = run_case(ops, dev, B=2, H=H, W=W, K=1, G=G, cfg=(cfg, th, tw), **kw)
        assert r["err"] < TOL, (cfg, kw, r)
    # one tile covering all couts: ssq_out behind a gated addend
    r = run_case(ops, dev, B=2, H=H, W=W, C1=Cin, Cout=bn, K=1, G=G, cfg=(cfg, th, tw), prologue="none", act_in="none", epilogue="addend", ssq_out=True)
    assert r["err"] < TOL and r["err_ssq"] < 2e-3, (cfg, r)
    # token layout (H = 1)
    tsh = _shapes(ops, cfg, 1, 1, 1, 200)
    if tsh:
        r = run_case(ops, dev, B=3, H=1, W=200, C1=Cin, Cout=Cout, K=1, G=G, cfg=(cfg,) + tsh[0], prologue="ln", affine=False, act_in="none")
        assert r["err"] < TOL, (cfg, "tokens", r)


@pytest.mark.parametrize("cfg", range(NUM_CFGS))
def test_strided_and_wide_kernels_every_cfg(ops, dev, cfg):
    """2x2 stride-2 downsample (ip.py:633-640) and the 15x15 cross-embed window (ip.py:1051-1076): family-0 instantiations only."""
    tp, bn, G, fam = ops.cfg_table()[cfg]
    ran = False
    if G == 4:
        H, W = (80, 48) if tp >= 128 else (40, 24)
        sh = [(th, tw) for _, _, th, tw in ops.launchable_shapes(cfg, H // 2, W // 2, 2, 2, 2)]
        if sh:
            r = run_case(ops, dev, B=2, H=H, W=W, C1=32, Cout=bn, K=2, stride=2, pad=0, G=G, cfg=(cfg,) + sh[0], prologue="none", act_in="none", ssq_out=True)
            assert r["err"] < TOL and r["err_ssq"] < 2e-3, (cfg, "2x2s2", r)
            ran = True
    if G == 1:
        H, W = 40, 40
        sh = [(th, tw) for _, _, th, tw in ops.launchable_shapes(cfg, H, W, 15, 15, 1)]
        if sh:
            r = run_case(ops, dev, B=2, H=H, W=W, C1=8, Cout=32, K=15, G=G, cfg=(cfg,) + sh[0], prologue="none", act_in="none", ssq_out=bn >= 32)
            assert r["err"] < TOL, (cfg, "15x15", r)
            ran = True
    if not ran:
        pytest.skip("cfg has neither instantiation")
