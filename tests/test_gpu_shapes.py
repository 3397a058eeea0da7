"""GPU: HIP kernels vs the fp64 CPU oracle AT THE LAYER SHAPES OF BASELINE.json's GPU configurations -- one test per
dominant kernel kind, forward + backward-data + backward-weight, through the C ABI (munit_amd.ops), same tolerances as the
small-shape op tests (2e-5 forward, 1e-4 gradients, normalised max error):

  config #2  256x256, batch 8, fp32  (configs/config_256.yaml geometry)
      3x3 256->256 @64x64           residual trunk, scripts/networks.py:603-624        Winograd F(2x2,3x3)
      up x2 + 5x5 256->128 @64->128  decoder, scripts/networks.py:532-546              sub-pixel Winograd + frame
      up x2 + 5x5 128->64 @128->256  decoder
      4x4 s2 64->128 @256            content / style encoder, scripts/networks.py:490-503, 451-470   F(3x3,2x2)
      4x4 s2 128->256 @128
      7x7 3->64 @256                 first layer, scripts/networks.py:488
      7x7 64->3 @256 + tanh          image head, scripts/networks.py:548-559
      4x4 s2 3->64 @256 lrelu        discriminator first layer, scripts/networks.py:46-47
      4x4 s2 256->512 @32 lrelu      discriminator last layer (split-K implicit GEMM)
      IN / AdaIN / LayerNorm at their real extents (scripts/networks.py:657, 823-845, 862-878)
  config #4  512x512, batch 4, fp32  (configs/config_HD.yaml:73-75)
      3x3 256->256 @128x128 (B=4), up x2 + 5x5 128->64 @256->512 (B=4)
  config #3  256x256, batch 32, bf16 storage (build extension)
      3x3 256->256 @64x64 on bf16 tensors, vs fp64 on the same bf16-valued operands
  and ONE whole dis_update + gen_update at 256x256 (batch 1) against the fp64 oracle (tests/parity.py); the same at 512x512
  is tests/test_gpu_hd_step.py (two minutes of oracle: collected last).

The fp64 reference convolutions cost seconds each on the host (torch CPU fp64: ~10-60 GMAC/s); the file as a whole a few minutes."""
import pytest
import torch

from oracle import munit_oracle as O
from tests.parity import KINK_FRAC, KINK_NOISE, nerr, run_step_parity

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-5
BWD_TOL = 1e-4


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


LAYERS = [
    # id, cin, cout, k, stride, pad, ups, act, B, H, W
    ("cfg2_trunk3x3", 256, 256, 3, 1, 1, 0, "none", 8, 64, 64),
    ("cfg2_up5x5_256_128", 256, 128, 5, 1, 2, 1, "none", 8, 64, 64),
    ("cfg2_up5x5_128_64", 128, 64, 5, 1, 2, 1, "none", 8, 128, 128),
    ("cfg2_down4x4_64_128", 64, 128, 4, 2, 1, 0, "none", 8, 256, 256),
    ("cfg2_down4x4_128_256", 128, 256, 4, 2, 1, 0, "relu", 8, 128, 128),
    ("cfg2_first7x7", 3, 64, 7, 1, 3, 0, "none", 8, 256, 256),
    ("cfg2_head7x7", 64, 3, 7, 1, 3, 0, "tanh", 8, 256, 256),
    ("cfg2_dis_first", 3, 64, 4, 2, 1, 0, "lrelu", 8, 256, 256),
    ("cfg2_dis_last", 256, 512, 4, 2, 1, 0, "lrelu", 8, 32, 32),
    ("cfg4_trunk3x3", 256, 256, 3, 1, 1, 0, "none", 4, 128, 128),
    ("cfg4_up5x5_128_64", 128, 64, 5, 1, 2, 1, "none", 4, 256, 256),
]


TWIN_BATCH = ("cfg2_up5x5_128_64", "cfg4_up5x5_128_64")


@pytest.mark.parametrize("case", LAYERS, ids=lambda c: c[0])
def test_conv_at_baseline_shape(case):
    from munit_amd import ops
    _, cin, cout, k, stride, pad, ups, act, B, H, W = case
    # The two most expensive fp64 references (the 128->64 up-sampling layers: 1.3 / 0.7 TFLOP of fp64 on the host) run on a batch
    # whose second half repeats its first: the kernel still works on the full batch -- every tile block, split and phase of the
    # real launch -- the reference on half of it; the twin samples must reproduce their originals bit for bit in y and dx, and
    # the weight gradient is twice the half-batch one.
    twin = case[0] in TWIN_BATCH
    x = rnd((B // 2 if twin else B, cin, H, W), 1)
    if twin:
        x = torch.cat([x, x])
    w = rnd((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5)
    b = rnd((cout,), 3, 0.1)
    xd = x.float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.float().to(dev()).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride, pad, "reflect", bool(ups), act)

    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    # A ReLU / LeakyReLU layer of 10^7 outputs has a few pre-activations within fp32 rounding of 0; there the fp64 reference
    # would take the other branch and one whole element of the incoming gradient would switch (the step tests pin these kinks
    # for the same reason, tests/parity.py).  The reference takes the device's branches, and the audit below requires that
    # they differ from its own only where |pre-activation| is within rounding noise.
    kinks = O.KinkMasks([(y[i:i + 1] > 0).cpu() for i in range(B)]) if act in ("relu", "lrelu") else None
    # sample by sample: torch's fp64 CPU convolution unfolds the input (Cin*k*k*Ho*Wo doubles per sample: 6.7 GB for the
    # config-#4 up-sampling layer); the parameter gradients accumulate over the loop exactly as over a batch
    ys, dy = [], None
    O.KINK_MASKS = kinks
    try:
        nref = B // 2 if twin else B
        for i in range(nref):
            xi = xr[i:i + 1]
            yi = O.conv_block(O.upsample2(xi) if ups else xi, wr, br, stride, pad, "reflect", None, act)
            if dy is None:
                dy = rnd((nref,) + tuple(yi.shape[1:]), 4)
                if twin:
                    dy = torch.cat([dy, dy])
            yi.backward(dy[i:i + 1])
            ys.append(yi.detach())
    finally:
        O.KINK_MASKS = None
    yr = torch.cat(ys + ys) if twin else torch.cat(ys)
    if twin:      # the reference saw each distinct sample once: its parameter gradients count half of what the device sums
        with torch.no_grad():
            wr.grad *= 2
            br.grad *= 2
            xr.grad[B // 2:] = xr.grad[:B // 2]
        assert torch.equal(y[B // 2:], y[:B // 2]), "twin samples differ in the forward"
    if kinks is not None:
        assert kinks.done() and kinks.worst_rel <= KINK_NOISE, (kinks.worst_rel, kinks.worst_at)
        assert kinks.n_disagree <= KINK_FRAC * kinks.n_total, (kinks.n_disagree, kinks.n_total)

    assert tuple(y.shape) == tuple(yr.shape)
    e = nerr(y, yr)
    assert e <= FWD_TOL, ("fwd", e)
    y.backward(dy.float().to(dev()).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    if twin:
        assert torch.equal(xd.grad[B // 2:], xd.grad[:B // 2]), "twin samples differ in the backward-data"
    errs = {"dx": nerr(xd.grad, xr.grad), "dw": nerr(wd.grad, wr.grad), "db": nerr(bd.grad, br.grad)}
    print(case[0], "fwd %.2e" % e, {k_: "%.2e" % v for k_, v in errs.items()},
          "kink disagreements %d (worst %.1e)" % (kinks.n_disagree, kinks.worst_rel) if kinks is not None else "")
    for name, v in errs.items():
        assert v <= BWD_TOL, (name, v)


NORMS = [
    # id, kind, B, C, H, W
    ("cfg2_in_64x256", "in_relu", 8, 64, 256, 256),
    ("cfg2_in_128x128", "in_relu", 8, 128, 128, 128),
    ("cfg2_in_256x64_res", "in_res", 8, 256, 64, 64),
    ("cfg2_adain_256x64_relu", "adain_relu", 8, 256, 64, 64),
    ("cfg2_adain_256x64_res", "adain_res", 8, 256, 64, 64),
    ("cfg2_ln_128x128", "ln_relu", 8, 128, 128, 128),
    ("cfg2_ln_64x256", "ln_relu", 8, 64, 256, 256),
    ("cfg4_ln_64x512", "ln_relu", 4, 64, 512, 512),
]


def pinned_relu(pre, y_dev):
    """ReLU of the fp64 reference taking the DEVICE's branches (mask = device output > 0), audited: the two sides may
    disagree only where the pre-activation is within rounding noise of 0 (see test_conv_at_baseline_shape)."""
    mask = (y_dev > 0).cpu()
    dis = (pre.detach() > 0) != mask
    n = int(dis.sum())
    if n:
        worst = float(pre.detach()[dis].abs().max()) / float(pre.detach().abs().max())
        assert worst <= KINK_NOISE and n <= KINK_FRAC * pre.numel(), (n, worst)
    return torch.where(mask, pre, torch.zeros_like(pre))


@pytest.mark.parametrize("case", NORMS, ids=lambda c: c[0])
def test_norm_at_baseline_shape(case):
    from munit_amd import ops
    _, kind, B, C, H, W = case
    shape = (B, C, H, W)
    x = rnd(shape, 1, 1.7) + 0.4
    dy = rnd(shape, 4)
    xr = x.clone().requires_grad_(True)
    xd = x.float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    extra = []
    relu, residual = kind.endswith("relu"), kind.endswith("res")
    if kind.startswith("ln"):
        g = torch.rand(C, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
        bt = rnd((C,), 3, 0.3)
        gr, btr = g.clone().requires_grad_(True), bt.clone().requires_grad_(True)
        gd, btd = (t.float().to(dev()).requires_grad_(True) for t in (g, bt))
        y = ops.layer_norm(xd, gd, btd, True)
        yr = pinned_relu(O.munit_layer_norm(xr, gr, btr), y)
        extra = [("dgamma", gd, gr), ("dbeta", btd, btr)]
    else:
        res = rnd(shape, 5)
        rr = res.clone().requires_grad_(True)
        rd = res.float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        if kind.startswith("adain"):
            params = rnd((B, 8 * C), 2) + 0.5            # the MLP output: 8 AdaIN layers x (bias C, weight C)
            w_off, b_off = 3 * C, 2 * C
            pr = params.clone().requires_grad_(True)
            pd = params.float().to(dev()).requires_grad_(True)
            yr = O.adain(xr, pr[:, w_off:w_off + C], pr[:, b_off:b_off + C])
            y = ops.adain(xd, pd, w_off, b_off, relu, rd if residual else None)
            extra = [("dparams", pd, pr)]
        else:
            yr = O.instance_norm(xr)
            y = ops.instance_norm(xd, relu, rd if residual else None)
        if relu:
            yr = pinned_relu(yr, y)
        if residual:
            yr = yr + rr
    yr.backward(dy)
    e = nerr(y, yr)
    assert e <= FWD_TOL, ("fwd", e)
    y.backward(dy.float().to(dev()).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    ex = nerr(xd.grad, xr.grad)
    assert ex <= BWD_TOL, ("dx", ex)
    for name, mine, ref in extra:
        assert nerr(mine.grad, ref.grad) <= BWD_TOL, (name, nerr(mine.grad, ref.grad))


def test_bf16s_trunk_layer_at_config3_batch():
    """BASELINE.json configs[2] (256x256, batch 32, bf16 storage): the residual-trunk layer on bf16 tensors at B = 32 against
    fp64 on the same bf16-valued operands -- bf16 outputs to bf16 rounding, the fp32 weight gradient to fp32-accumulation
    accuracy (tolerances of tests/test_gpu_bf16s.py)."""
    from munit_amd import ops
    BF = torch.bfloat16
    r16 = lambda t: t.float().bfloat16().double()
    B, C, H, W = 32, 256, 64, 64
    ops.set_compute("bf16s")
    try:
        x, w, b = r16(rnd((B, C, H, W), 1)), rnd((C, C, 3, 3), 2, (2.0 / (C * 9)) ** 0.5), rnd((C,), 3, 0.1)
        wq = r16(w)
        xr, wr, br = x.clone().requires_grad_(True), wq.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = O.conv_block(xr, wr, br, 1, 1, "reflect", None, "none")
        dy = r16(rnd(tuple(yr.shape), 4))
        yr.backward(dy)
        xd = x.to(BF).to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd = w.float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bd = b.float().to(dev()).requires_grad_(True)
        y = ops.conv2d(xd, wd, bd, 1, 1, "reflect", False, "none", out_dtype=BF)
        assert y.dtype == BF
        assert nerr(y, yr) <= 8e-3, ("fwd", nerr(y, yr))
        y.backward(dy.to(BF).to(dev()).contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
        assert xd.grad.dtype == BF
        assert nerr(xd.grad, xr.grad) <= 8e-3, ("dx", nerr(xd.grad, xr.grad))
        assert nerr(wd.grad, wr.grad) <= 3e-5, ("dw", nerr(wd.grad, wr.grad))
        assert nerr(bd.grad, br.grad) <= 3e-5, ("db", nerr(bd.grad, br.grad))
    finally:
        ops.set_compute("f32")


def test_step_matches_oracle_at_256():
    """One dis_update + gen_update at BASELINE.json configs[1]'s resolution (256x256; batch 1 keeps the fp64 oracle to about half a
    minute on the host) with the tolerances of the 64x64 step test: losses 1e-5, every gradient tensor <= 5e-5 with the kinks
    pinned AND the recorded branches audited against the oracle's own (tests/parity.py KINK_NOISE), Adam moments, weight step."""
    rep = run_step_parity(size=256, batch=1, gen_state=1, iters=1, device="cuda:0")
    print({k: v for k, v in rep.items() if not isinstance(v, list)})
