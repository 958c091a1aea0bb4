"""Constructor-flag combinations of the image `Unet` outside the two README unets: shared by the oracle-vs-live-reference sweep
(tests/test_oracle_vs_reference.py) and the planner-vs-oracle sweep (tests/test_plan_interp.py)."""

_T = dict(dim=8, cond_dim=32, text_embed_dim=32, dim_mults=(1, 2), attn_heads=2, max_text_len=16, attn_pool_num_latents=8)

SWEEP = {
    "memory_efficient": dict(_T, num_resnet_blocks=(1, 2), layer_attns=(False, True), layer_cross_attns=(False, True), memory_efficient=True),
    "memory_efficient_lowres": dict(_T, num_resnet_blocks=2, layer_attns=(False, True), layer_cross_attns=(False, True), memory_efficient=True,
                                    lowres_cond=True),
    "no_attn_pool": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(True, True), attn_pool_text=False),
    "unconditional": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=False, cond_on_text=False),
    "depth2_attn_everywhere": dict(_T, num_resnet_blocks=1, layer_attns=True, layer_attns_depth=2, layer_mid_attns_depth=2, layer_cross_attns=True),
    "no_final_resnet_no_skip_scale": dict(_T, num_resnet_blocks=2, layer_attns=(False, True), layer_cross_attns=(False, True),
                                          final_resnet_block=False, scale_skip_connection=False),
    "plain_init_conv_no_mid_attn": dict(_T, num_resnet_blocks=1, layer_attns=False, layer_cross_attns=(False, True), init_cross_embed=False,
                                        attend_at_middle=False, init_conv_kernel_size=7),
    "three_levels_no_gca": dict(_T, dim_mults=(1, 2, 4), num_resnet_blocks=(1, 1, 2), layer_attns=(False, False, True),
                                layer_cross_attns=(False, True, True), use_global_context_attn=False),
    "four_time_tokens_init_dim": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), num_time_tokens=4,
                                      init_dim=16, ff_mult=4.),
    "dim32_three_levels": dict(dim=32, cond_dim=64, text_embed_dim=32, dim_mults=(1, 2, 4), attn_heads=4, max_text_len=16, attn_pool_num_latents=8,
                               num_resnet_blocks=(1, 2, 2), layer_attns=(False, True, True), layer_cross_attns=(False, True, True)),
    # 256 channels at the coarsest level: the ResnetBlocks whose block2 cannot get its norm from block1's epilogue (two output-channel tiles)
    "dim64_three_levels": dict(dim=64, cond_dim=64, text_embed_dim=32, dim_mults=(1, 2, 4), attn_heads=2, max_text_len=16, attn_pool_num_latents=8,
                               num_resnet_blocks=(1, 1, 2), layer_attns=(False, False, True), layer_cross_attns=(False, True, True)),
    "dim24_lowres": dict(dim=24, cond_dim=40, text_embed_dim=32, dim_mults=(1, 2), attn_heads=2, max_text_len=16, attn_pool_num_latents=8,
                         num_resnet_blocks=2, layer_attns=(False, True), layer_cross_attns=(True, True), lowres_cond=True),
    "nearest_upsample_init_residual": dict(_T, dim_mults=(1, 2, 4), num_resnet_blocks=1, layer_attns=(False, False, True),
                                           layer_cross_attns=(False, True, True), pixel_shuffle_upsample=False, init_conv_to_final_conv_residual=True),
    "init_residual_memory_efficient": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), memory_efficient=True,
                                           init_conv_to_final_conv_residual=True),
    # the reference's UnetConfig default head geometry (configs.py:48-49: 32-dim heads); the mid-block cross attention keeps 8 x 64
    "head_dim_32": dict(_T, attn_dim_head=32, attn_heads=4, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True)),
    "head_dim_32_three_levels": dict(dim=32, cond_dim=64, text_embed_dim=32, dim_mults=(1, 2, 4), attn_dim_head=32, attn_heads=16, max_text_len=16,
                                     attn_pool_num_latents=8, num_resnet_blocks=(1, 2, 2), layer_attns=(False, True, True),
                                     layer_cross_attns=(False, True, True)),
    # image conditioning (ip.py:1191-1194, 1555-1560): extra init-conv input channels, given at another resolution (nearest-resized)
    "cond_images_3": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), cond_images_channels=3),
    "cond_images_10_lowres_plain_init": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True),
                                             cond_images_channels=10, lowres_cond=True, init_cross_embed=False),
    # self-conditioning (ip.py:1541-1543): the previous x0 estimate as three more init-conv input channels
    "self_cond": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), self_cond=True),
    "self_cond_lowres_cond_images_5": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), self_cond=True,
                                           lowres_cond=True, cond_images_channels=5),
    # UpsampleCombiner (ip.py:1078-1110): every up level's feature map resized to the output resolution, through its own Block, concatenated
    "combine_upsample_fmaps": dict(_T, dim_mults=(1, 2, 4), num_resnet_blocks=(1, 1, 2), layer_attns=(False, False, True),
                                   layer_cross_attns=(False, True, True), combine_upsample_fmaps=True),
    "combine_fmaps_init_residual_memory_efficient": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True),
                                                         memory_efficient=True, init_conv_to_final_conv_residual=True, combine_upsample_fmaps=True,
                                                         lowres_cond=True),
    "init_residual_no_final_resnet_lowres": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True),
                                                 init_conv_to_final_conv_residual=True, final_resnet_block=False, lowres_cond=True),
    "channels_out_6": dict(_T, num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), channels_out=6),
}


def cond_images_for(kw, B, seed=9):
    """The conditioning image of a sweep configuration ((B, cond_images_channels, 8, 8): half the 16x16 input's size), or None."""
    import torch
    cc = kw.get("cond_images_channels", 0)
    if not cc:
        return None
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, cc, 8, 8, generator=g)


def self_cond_for(kw, B, S=16, seed=10):
    """The self-conditioning image of a sweep configuration, or None."""
    import torch
    if not kw.get("self_cond", False):
        return None
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, kw.get("channels", 3), S, S, generator=g) * 2 - 1
