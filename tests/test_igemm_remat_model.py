"""CPU: the index arithmetic of the -DIGEMM_EPI_REMAT build of csrc/igemm.hip (epilogue constants re-derived from the thread id inside
the tile loop) equals the product build's, for every consumer thread, every wave arrangement of the 16 tile
configurations and every tile width the host can choose.  Transliteration of the two code blocks; the kernels themselves are compared
on the GPU (tests/test_igemm_cfgs_gpu.py under IMAGEN_LIB_PATH=.../libimagen_hip_remat.so)."""
import itertools


def product_build(tid, WM, WN, MI, TW):
    lane, wave = tid & 63, tid >> 6
    half, l31 = lane >> 5, lane & 31
    wm, wn = wave // WN, wave % WN
    pix = []
    for mi in range(MI):
        tp = (wm * MI + mi) * 32 + l31
        py = tp // TW
        pix.append((py, tp - py * TW))
    return half, l31, wm, wn, pix


def remat_build(tid, WM, WN, MI, TW):
    half, l31 = (tid >> 5) & 1, tid & 31
    wave_e = tid >> 6                      # readfirstlane of a wave-uniform value
    wm, wn = wave_e // WN, wave_e % WN
    tw_sh = (TW & -TW).bit_length() - 1    # __builtin_ctz
    pix = []
    for mi in range(MI):
        tp = (wm * MI + mi) * 32 + l31
        pix.append((tp >> tw_sh, tp & (TW - 1)))
    return half, l31, wm, wn, pix


def test_remat_epilogue_constants_equal_the_product_build():
    arrangements = {(4, 1), (1, 4), (2, 2)}             # (WM, WN) of kCfgs
    for (WM, WN), MI, TW in itertools.product(arrangements, (1, 2, 4), (1, 2, 4, 8, 16, 32, 64, 128, 256)):
        for tid in range(256):                          # consumer waves 0-3
            assert product_build(tid, WM, WN, MI, TW) == remat_build(tid, WM, WN, MI, TW), (WM, WN, MI, TW, tid)


def test_host_tile_widths_are_powers_of_two():
    from imagen_pytorch_amd import ops

    for tp in (64, 128, 256):
        for OH in (1, 8, 64):
            for th, tw in ops._tile_shapes(tp, OH, 64):
                assert tw > 0 and tw & (tw - 1) == 0 and th * tw == tp
