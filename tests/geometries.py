"""Network geometries other than config_256.yaml's that the step-parity tests run (GPU: tests/test_gpu_step.py; CPU pin of the
oracle to the REFERENCE modules on every one of them: tests/golden/make_golden_geometries.py ->
tests/golden/golden_geometries.json -> tests/test_oracle_golden.py).  One list, so a geometry cannot enter the GPU suite
without its reference-generated fixture: the CPU test fails on a name the fixture file does not hold.

Entry = (name, size, overrides): `size` an int or (crop_image_height, crop_image_width); `overrides` merged into
oracle.default_hp (nested dicts key by key).  Every entry runs at batch 2, gen_state 1 unless overridden."""

GEOMETRIES = [
    # three down-samplings on a 72x72 crop: a 9x9 trunk (odd extents: no Winograd tiles, the implicit-GEMM forms), zero padding
    # everywhere, narrow networks (32 channels at the first layer, 8-dim style, 64-wide MLP), two discriminator scales of
    # three layers with instance norm
    ("deep_zero_pad", 72, dict(gen=dict(dim=32, mlp_dim=64, style_dim=8, n_downsample=3, n_res=2, pad_type="zero"),
                               dis=dict(dim=32, n_layer=3, num_scales=2, pad_type="zero", norm="in"))),
    # one down-sampling on a 40x40 crop (20x20 trunk), six residual blocks, LeakyReLU generator, one discriminator scale
    ("shallow_lrelu", 40, dict(gen=dict(dim=64, n_downsample=1, n_res=6, activ="lrelu"),
                               dis=dict(dim=16, n_layer=2, num_scales=1))),
    # config_256.yaml's networks on crops that are not square (crop_image_height != crop_image_width, utils.py:229-249):
    # 64 x 96 -> a 16 x 24 trunk, 2 x 3 maps in the last discriminator layer; a kernel that mixes up H and W cannot pass
    ("non_square", (64, 96), dict()),
    # gen_state 0 (two AdaINGen, trainer.py:84-97), taller than wide
    ("non_square_two_generators", (80, 64), dict(gen_state=0)),
    # loss terms switched off by the config (trainer.py:501-537 skips what has weight 0): no cycle reconstruction, no style /
    # content reconstruction -- other tensors reach the backward pass, other branches of gen_update run
    ("no_cycle_no_latent_recon", 64, dict(recon_x_cyc_w=0, recon_s_w=0, recon_c_w=0)),
    # config_256.yaml's widths with ZERO padding in both networks: the Winograd kernels' zero-pad border paths (forward,
    # backward-data without the reflect fold, guarded backward-weight loads) and the zero-padded sub-pixel / stride-2 forms
    ("zero_pad_full_width", 64, dict(gen=dict(pad_type="zero"), dis=dict(pad_type="zero"))),
    # a ReLU discriminator of four scales and three layers (MsImageDis takes all three from the config, networks.py:22-30)
    ("relu_discriminator_4_scales", 64, dict(dis=dict(activ="relu", num_scales=4, n_layer=3))),
    # a tanh generator with a LeakyReLU-free instance-normed discriminator: the norm kernels' fused tanh / LeakyReLU codes
    # (networks.py:668-681 allows any activation behind any norm)
    ("tanh_generator_in_lrelu_dis", 48, dict(gen=dict(dim=16, mlp_dim=32, n_res=2, activ="tanh"),
                                              dis=dict(dim=16, n_layer=2, num_scales=2, norm="in", activ="lrelu"))),
]

# input_dim_a = input_dim_b = 1 (networks.py:121-186 builds the first and last layers from them)
ONE_CHANNEL = ("one_channel_domains", 64, dict(input_dim_a=1, input_dim_b=1))

ALL = GEOMETRIES + [ONE_CHANNEL]
BATCH = 2


def merged_hp(default_hp, size, over, batch=BATCH):
    """oracle.default_hp(size, batch, gen_state) with `over` applied the way tests/parity.run_step_parity applies it."""
    hp = default_hp(size, batch, over.get("gen_state", 1))
    for k, v in over.items():
        if isinstance(v, dict):
            hp[k] = dict(hp[k], **v)
        else:
            hp[k] = v
    return hp
