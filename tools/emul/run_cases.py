#!/usr/bin/env python
"""Run IGEMM test cases (tests/igemm_case.py: one launch against the fp32 torch restatement of the op contract) on the CPU through an
EMULATED kernel library (tools/emul/build_emul_lib.sh).  Must be started with IMAGEN_LIB_PATH pointing at that library:

    IMAGEN_LIB_PATH=imagen-pytorch_amd/libimagen_emul.so python tools/emul/run_cases.py --out /tmp/emul_default.pt [--cases NAME ...]

Writes {case name: dict(err=..., err_ssq=..., cfg=..., y=<output tensor>)}; the caller compares libraries (tests/test_igemm_emulated.py).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

# one case per structural path of igemm_kernel: tile arrangements (WM x WN), k-loop instantiations (3x3 / 1x1 / 2x2 / generic), prologues,
# both epilogue instantiations (plain / generic) with every output mode, partial tiles, several cout tiles, a persistent walk (more
# tiles than the emulated chip holds workgroups)
CASES = {
    # cfg 3 = 64 px x 128 co (the dominant MFMA-bound instantiation): 3x3 with concat + statistics prologue, plain epilogue + ssq_out
    "cfg3_block_ssq": dict(B=2, H=20, W=24, C1=64, C2=32, Cout=128, K=3, G=4, cfg=(3, 8, 8), prologue="ssq", affine=False, ssq_out=True),
    "cfg3_block_rs_2tiles": dict(B=1, H=16, W=16, C1=32, C2=0, Cout=160, K=3, G=4, cfg=(3, 8, 8), prologue="rs"),
    "cfg3_post": dict(B=2, H=16, W=24, C1=32, C2=32, Cout=128, K=3, G=4, cfg=(3, 8, 8), prologue="ssq", affine=False, epilogue="post"),
    "cfg3_raw": dict(B=1, H=16, W=16, C1=64, Cout=128, K=3, G=4, cfg=(3, 8, 8), prologue="none", act_in="none"),
    "cfg3_addend": dict(B=2, H=12, W=20, C1=64, C2=32, Cout=128, K=1, G=4, cfg=(3, 2, 32), prologue="none", act_in="none", epilogue="addend", ssq_out=True),
    "cfg3_res_gelu": dict(B=1, H=1, W=96, C1=64, Cout=128, K=1, G=4, cfg=(3, 1, 64), prologue="ln", act_in="none", act_out="gelu", epilogue="res"),
    "cfg3_nchw": dict(B=1, H=16, W=16, C1=32, Cout=3, K=3, G=4, cfg=(3, 8, 8), prologue="none", act_in="none", epilogue="nchw"),
    "cfg3_shuffle": dict(B=1, H=8, W=8, C1=32, Cout=128, K=1, G=4, cfg=(3, 8, 8), prologue="none", act_in="none", act_out="silu", epilogue="shuffle"),
    "cfg3_down2x2": dict(B=1, H=16, W=16, C1=32, Cout=128, K=2, stride=2, G=4, cfg=(3, 8, 8), prologue="none", act_in="none"),
    # cfg 0 = 256 px x 32 co (WM = 4) and cfg 2 = 256 px x 64 co (2 x 2 waves): the HBM-bound tilings
    "cfg0_block": dict(B=1, H=20, W=36, C1=32, C2=32, Cout=32, K=3, G=4, cfg=(0, 16, 16), prologue="ssq", affine=False, ssq_out=True),
    "cfg0_post": dict(B=1, H=16, W=32, C1=32, Cout=32, K=3, G=4, cfg=(0, 8, 32), prologue="rs", epilogue="post"),
    "cfg2_block": dict(B=1, H=16, W=32, C1=64, C2=32, Cout=64, K=3, G=4, cfg=(2, 16, 16), prologue="rs", ssq_out=True),
    # 8-channel chunks (generic k loop) and the deep 1x1 chunks
    "cfg9_generic": dict(B=1, H=12, W=20, C1=16, C2=8, Cout=32, K=3, G=1, cfg=(9, 8, 16), prologue="rs"),
    "cfg7_ni2": dict(B=1, H=8, W=8, C1=24, Cout=128, K=1, G=1, cfg=(7, 8, 8), prologue="none", act_in="none", epilogue="res"),
    "cfg10_linear": dict(B=1, H=1, W=100, C1=256, Cout=128, K=1, G=16, cfg=(10, 1, 64), prologue="ln", act_in="none", act_out="gelu"),
    "cfg14_addend": dict(B=2, H=8, W=16, C1=128, C2=64, Cout=128, K=1, G=8, cfg=(14, 4, 16), prologue="none", act_in="none", epilogue="addend", ssq_out=True),
    "cfg13_resconv": dict(B=1, H=16, W=32, C1=32, C2=32, Cout=32, K=1, G=8, cfg=(13, 2, 64), prologue="none", act_in="none", epilogue="addend", ssq_out=True),
    # the all-DMA family (conv_dma.hip; direct-to-LDS copies land at the covering vmcnt wait, in issue order) and the streaming family
    # (conv_stream.hip: persistent tile walk, in-place LDS prologue); cfg = "dma:<tile px>x<tile couts>" | "stream" is resolved at run time
    "dma_64x128_gca": dict(B=2, H=16, W=16, C1=64, Cout=128, K=3, G=4, cfg="dma:64x128", prologue="none", act_in="none", gca=True),
    "dma_128x128_post": dict(B=1, H=16, W=32, C1=128, Cout=128, K=3, G=4, cfg="dma:128x128", prologue="none", act_in="none", epilogue="post"),
    "dma_256x32_ssq": dict(B=1, H=20, W=36, C1=32, Cout=32, K=3, G=4, cfg="dma:256x32", prologue="none", act_in="none", ssq_out=True),
    # the big-tile all-DMA family (conv_big.hip): cfg = "big:<n>" = the n-th configuration of family 5 (0: 256 px, 1: 128 px with the K split,
    # 2: ... three halo buffers, 3: 256 px with a 3-stage ring); several tiles per image and partial tiles, 4 / 6 / 3 chunks, 256 couts = two tile columns
    "big0_gca": dict(B=2, H=32, W=16, C1=128, Cout=128, K=3, G=4, cfg="big:0", prologue="none", act_in="none", gca=True),
    "big0_post_ragged": dict(B=1, H=20, W=36, C1=96, Cout=128, K=3, G=4, cfg="big:0", prologue="none", act_in="none", epilogue="post"),
    "big1_ssq": dict(B=2, H=16, W=16, C1=128, Cout=128, K=3, G=4, cfg="big:1", prologue="none", act_in="none", ssq_out=True),
    "big1_256co_addend": dict(B=1, H=12, W=20, C1=192, Cout=256, K=3, G=4, cfg="big:1", prologue="none", act_in="none", epilogue="addend"),
    "big2_gca": dict(B=1, H=16, W=32, C1=128, Cout=128, K=3, G=4, cfg="big:2", prologue="none", act_in="none", gca=True),
    "big3_res": dict(B=1, H=16, W=16, C1=160, Cout=128, K=3, G=4, cfg="big:3", prologue="none", act_in="none", epilogue="res"),
    "stream_raw": dict(B=2, H=40, W=36, C1=32, Cout=32, K=3, G=4, cfg="stream", prologue="none", act_in="none", ssq_out=True),
    "stream_pro_concat_post": dict(B=2, H=40, W=36, C1=32, C2=32, Cout=32, K=3, G=4, cfg="stream", prologue="ssq", affine=False, epilogue="post"),
    "stream_pro_affine_ragged": dict(B=3, H=27, W=45, C1=32, Cout=24, K=3, G=4, cfg="stream", prologue="ssq", affine=True),
    # the streaming family with the prologue on register-staged rows (conv_pro.hip): persistent contiguous tile ranges over several images,
    # ragged right / bottom edges, one and two inputs, shared and per-row affines (+ shifts), the three epilogues
    "pro_concat_post": dict(B=3, H=40, W=36, C1=32, C2=32, Cout=32, K=3, G=4, cfg="pro", prologue="ssq", affine=False, epilogue="post"),
    "pro_concat_ssq": dict(B=2, H=24, W=48, C1=32, C2=32, Cout=32, K=3, G=4, cfg="pro", prologue="ssq", affine=False, ssq_out=True),
    "pro_single_affine_ragged": dict(B=3, H=27, W=45, C1=32, Cout=32, K=3, G=4, cfg="pro", prologue="ssq", affine=True, ssq_out=True),
    "pro_single_post_tiny_images": dict(B=9, H=8, W=16, C1=32, Cout=32, K=3, G=4, cfg="pro", prologue="ssq", affine=True, epilogue="post"),
    "pro_raw": dict(B=2, H=40, W=36, C1=32, Cout=32, K=3, G=4, cfg="pro", prologue="none", act_in="none", ssq_out=True),
    "pro_raw_concat": dict(B=1, H=16, W=64, C1=32, C2=32, Cout=32, K=3, G=4, cfg="pro", prologue="none", act_in="none"),
    # ... 64 output channels (eight waves: two cout blocks per pixel block; the last chunks' weights from LDS; the output-side norm crosses the two waves)
    "pro64_concat3_post": dict(B=3, H=27, W=45, C1=64, C2=32, Cout=64, K=3, G=4, cfg="pro64", prologue="ssq", affine=False, epilogue="post"),
    "pro64_concat2_post_odd_range": dict(B=5, H=24, W=16, C1=32, C2=32, Cout=64, K=3, G=4, cfg="pro64", prologue="ssq", affine=True, epilogue="post"),
    "pro64_single_plain": dict(B=2, H=16, W=48, C1=64, Cout=64, K=3, G=4, cfg="pro64", prologue="ssq", affine=True),
    "pro64_raw": dict(B=2, H=24, W=32, C1=64, Cout=64, K=3, G=4, cfg="pro64", prologue="none", act_in="none"),
}


def resolve_cfg(ops, spec, kw):
    """'stream' | 'dma:<tp>x<bn>' -> (cfg id, th, tw) of this library (family cfg ids depend on which families a library holds)."""
    if not isinstance(spec, str):
        return spec
    if spec == "stream":
        return (ops.stream_cfg(), 16, 16)
    if spec == "pro":
        return (ops.pro_cfg(32), 8, 16)
    if spec == "pro64":
        return (ops.pro_cfg(64), 8, 16)
    if spec.startswith("big:"):
        i = [j for j, c in enumerate(ops.cfg_table()) if c[3] == 5][int(spec.split(":")[1])]
        sh = ops.launchable_shapes(i, kw["H"], kw["W"], 3, 3, 1)
        return (i, sh[0][2], sh[0][3])
    tp, bn = map(int, spec.split(":")[1].split("x"))
    for i, (t, b, g, fam) in enumerate(ops.cfg_table()):
        if fam == 2 and (t, b) == (tp, bn):
            sh = ops.launchable_shapes(i, kw["H"], kw["W"], 3, 3, 1)
            if sh:
                return (i, sh[0][2], sh[0][3])
    raise KeyError(spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--cases", nargs="*", default=None)
    args = ap.parse_args()
    assert "emul" in os.path.basename(os.environ.get("IMAGEN_LIB_PATH", "")), "start with IMAGEN_LIB_PATH=<an emulated library>"
    torch.cuda.synchronize = lambda *a, **k: None        # the case runner syncs the GPU it thinks it is on
    from imagen_pytorch_amd import ops
    ops.current_stream_handle = lambda: 0
    from igemm_case import run_case

    dev = torch.device("cpu")
    results = {}
    for name, kw in CASES.items():
        if args.cases and name not in args.cases:
            continue
        captured = {}
        real_to_nchw, real_igemm = ops.act_to_nchw, ops.igemm

        def grab(a, _c=captured, _f=real_to_nchw):   # NHWC outputs: keep the raw result for the library-vs-library comparison
            out = _f(a)
            _c["y"] = out.clone()
            return out

        def grab_y(plan, x1, pw, y, *a, _c=captured, **k):   # fp32 NCHW outputs never pass through act_to_nchw: keep the tensor itself
            if isinstance(y, torch.Tensor):
                _c["y_t"] = y
            return real_igemm(plan, x1, pw, y, *a, **k)

        ops.act_to_nchw, ops.igemm = grab, grab_y
        try:
            r = run_case(ops, dev, **dict(kw, cfg=resolve_cfg(ops, kw["cfg"], kw)))
        finally:
            ops.act_to_nchw, ops.igemm = real_to_nchw, real_igemm
        r["y"] = captured["y"] if "y" in captured else captured["y_t"].clone()
        results[name] = r
        print(f"{name:24s} err {r['err']:.2e}" + (f"  ssq {r['err_ssq']:.2e}" if 'err_ssq' in r else "") + f"  cfg {r['cfg']}", flush=True)
    torch.save(results, args.out)


if __name__ == "__main__":
    main()
