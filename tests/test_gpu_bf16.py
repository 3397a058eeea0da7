"""GPU: the bf16-compute mode of the convolutions (BASELINE.json config #3; a build extension with no reference
counterpart).  Operands are rounded to bf16 (nearest-even) when staged into LDS and accumulated in fp32, so a
layer must equal the fp64 convolution of the bf16-ROUNDED operands to fp32-accumulation accuracy (1e-5
normalised) -- a much sharper check than the loose 2e-2 the mode is allowed against the unrounded fp64 oracle,
which is asserted as well.  Layers with channel counts that are not multiples of 32 run fp32 in both modes."""
import pytest
import torch

from oracle import munit_oracle as O
from tests.parity import nerr

pytestmark = pytest.mark.gpu

BF16_VS_ORACLE = 2e-2     # stated tolerance of the bf16 configuration (forward)
SHARP = 2e-5              # vs the fp64 result on bf16-rounded operands


def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def r16(t):
    return t.float().bfloat16().double()


@pytest.fixture(autouse=True)
def bf16_mode():
    from munit_amd import ops
    ops.set_compute("bf16")
    yield
    ops.set_compute("f32")


CASES = [
    # cin, cout, k, stride, pad, pad_type, ups, act, B, H, W
    (64, 128, 4, 2, 1, "reflect", 0, "relu", 2, 16, 16),
    (256, 256, 3, 1, 1, "reflect", 0, "none", 2, 8, 8),
    (256, 128, 5, 1, 2, "reflect", 1, "none", 2, 6, 8),      # sub-pixel path
    (128, 64, 5, 1, 2, "reflect", 1, "none", 1, 9, 7),
    (256, 512, 4, 2, 1, "reflect", 0, "lrelu", 2, 4, 4),     # split-K path
    (32, 48, 3, 1, 1, "zero", 0, "relu", 2, 9, 11),
    (256, 256, 3, 1, 1, "reflect", 0, "relu", 3, 33, 17),    # several tiles, ragged
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%d-%d_k%ds%d_u%d" % (c[0], c[1], c[2], c[3], c[6]))
def test_conv_bf16_forward_and_dgrad(case):
    from munit_amd import ops
    cin, cout, k, stride, pad, pt, ups, act, B, H, W = case
    x = rnd((B, cin, H, W), 1)
    w = rnd((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5)
    b = rnd((cout,), 3, 0.1)

    def ref(xx, ww):
        xr, wr = xx.clone().requires_grad_(True), ww.clone().requires_grad_(True)
        if ups:
            y = O.conv_block(O.upsample2(xr), wr, b, stride, pad, pt, None, act)
        else:
            y = O.conv_block(xr, wr, b, stride, pad, pt, None, act)
        return xr, wr, y

    xr, wr, y_exact = ref(x, w)
    xq, wq, y_rounded = ref(r16(x), r16(w))

    xd = x.float().to(dev()).requires_grad_(True)
    wd = w.float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.float().to(dev()).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride, pad, pt, bool(ups), act)
    if not ups:   # the sub-pixel form merges weights in fp32 BEFORE rounding them: only the loose bound applies
        assert nerr(y, y_rounded) <= SHARP, ("fwd sharp", nerr(y, y_rounded))
    assert nerr(y, y_exact) <= BF16_VS_ORACLE, ("fwd", nerr(y, y_exact))

    # Gradients are compared with the oracle evaluated on the rounded operands: its ReLU mask is then the
    # kernel's own mask (a mask taken from the unrounded forward differs on the elements whose pre-activation
    # moved across zero by the 1e-2 rounding error -- each such flip is an O(1) change of one dx element).
    dy = rnd(tuple(y_exact.shape), 4)
    y_rounded.backward(dy)
    y.backward(dy.float().to(dev()))
    assert nerr(xd.grad, xq.grad) <= BF16_VS_ORACLE, ("dx", nerr(xd.grad, xq.grad))
    assert nerr(wd.grad, wq.grad) <= BF16_VS_ORACLE, ("dw", nerr(wd.grad, wq.grad))


def test_bf16_mode_is_off_by_default_and_validated():
    from munit_amd import ops
    ops.set_compute("f32")
    assert ops.get_compute() == "f32"
    with pytest.raises(ValueError):
        ops.set_compute("fp8")
    ops.set_compute("bf16")
    assert ops.get_compute() == "bf16"
