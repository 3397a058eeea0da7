"""GPU: the bf16-compute mode of the convolutions (BASELINE.json config #3; a build extension with no reference
counterpart).  Operands are rounded to bf16 (nearest-even) when staged into LDS and accumulated in fp32, so a
layer must equal the fp64 convolution of the bf16-ROUNDED operands to fp32-accumulation accuracy (1e-5
normalised) -- a much sharper check than the loose 2e-2 the mode is allowed against the unrounded fp64 oracle,
which is asserted as well.  Layers with channel counts that are not multiples of 32 run fp32 in both modes."""
import pytest
import torch

from oracle import munit_oracle as O
from tests.parity import nerr

# frozen_mode: an arithmetic mode no BASELINE.json config asks for (f32x3; bf16 MFMA operands on fp32 storage -- superseded by the
# bf16-storage mode `bf16s`).  Frozen since round 3: kept working, no further work; `-m "gpu and not frozen_mode"` leaves them out.
pytestmark = [pytest.mark.gpu, pytest.mark.frozen_mode]

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


WGRAD_CASES = [
    # cin, cout, k, stride, pad, pad_type, ups, B, H, W
    (256, 256, 3, 1, 1, "reflect", 0, 2, 8, 8),       # 128x128 tile
    (64, 128, 4, 2, 1, "reflect", 0, 2, 16, 16),
    (128, 64, 3, 1, 1, "zero", 0, 2, 9, 11),           # Cout <= 64: 64x256 tile, two X images
    (256, 128, 5, 1, 2, "reflect", 1, 2, 6, 8),        # sub-pixel phases (bf16) + frame (fp32)
    (32, 48, 3, 1, 1, "zero", 0, 3, 33, 17),           # partial tiles, several pixel splits
    (256, 512, 4, 2, 1, "reflect", 0, 2, 4, 4),
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "c%d-%d_k%ds%d_u%d" % (c[0], c[1], c[2], c[3], c[6]))
def test_conv_bf16_wgrad(case):
    """dw / db of the bf16 backward-weight kernel = fp64 gradient on bf16-rounded x and dy (fp32 accumulation)."""
    from munit_amd import ops
    cin, cout, k, stride, pad, pt, ups, B, H, W = case
    x = rnd((B, cin, H, W), 11)
    w = rnd((cout, cin, k, k), 12, 0.05)
    xq = r16(x)
    wr = w.clone().requires_grad_(True)
    br = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    src = O.upsample2(xq) if ups else xq
    y = O.conv_block(src, wr, br, stride, pad, pt, None, "none")
    dy = rnd(tuple(y.shape), 13)
    y.backward(r16(dy))
    xd = x.float().to(dev()).contiguous(memory_format=torch.channels_last)
    dyd = dy.float().to(dev()).contiguous(memory_format=torch.channels_last)
    dw, db = ops.conv2d_wgrad_raw(xd, dyd, tuple(w.shape), stride, pad, pt, bool(ups))
    tol = SHARP if not ups else 5e-3      # the frame pixels of the sub-pixel form are computed in fp32 from unrounded data
    assert nerr(dw, wr.grad) <= tol, ("dw", nerr(dw, wr.grad))
    assert nerr(db, br.grad) <= tol, ("db", nerr(db, br.grad))


def test_step_bf16_tracks_fp32_step():
    """One dis_update + gen_update in bf16-compute mode against the same step in fp32 mode (same weights, same
    batch): every loss within 2e-2 relative (the stated tolerance of config #3), weights moved the same way."""
    from munit_amd import ops
    from munit_amd.trainer import MUNIT_Trainer
    from tests.parity import l2err
    import bench
    size, batch = 64, 2
    x_a, x_b, m_a, m_b = (t.to(dev()) for t in bench.make_batch(batch, size))
    out = {}
    for prec in ("f32", "bf16"):
        hp = bench.bench_hp(size, batch)
        hp["precision"] = prec
        torch.manual_seed(1234)
        tr = MUNIT_Trainer(hp)
        tr.to(dev())
        torch.manual_seed(5)
        tr.dis_update(x_a, x_b, hp)
        tr.gen_update(x_a, x_b, hp, m_a, m_b)
        names = [n for n in vars(tr) if n.startswith("loss_")]
        out[prec] = ({n: float(getattr(tr, n)) for n in names},
                     {k: v.detach().clone() for k, v in tr.gen.state_dict().items() if v.dtype == torch.float32})
        assert ops.get_compute() == prec
    ops.set_compute("bf16")
    lf, lb = out["f32"][0], out["bf16"][0]
    assert set(lf) == set(lb) and len(lf) >= 10
    for n in lf:
        assert abs(lb[n] - lf[n]) <= 2e-2 * max(abs(lf[n]), 1e-3), (n, lf[n], lb[n])
    # Adam's first step is sign-like (|dw| = lr): the two runs must agree on the direction for almost all weights
    wf, wb = out["f32"][1], out["bf16"][1]
    agree, total = 0, 0
    torch.manual_seed(1234)
    init = MUNIT_Trainer(bench.bench_hp(size, batch)).gen.state_dict()
    for k in wf:
        d0 = (wf[k].cpu() - init[k]).flatten()
        d1 = (wb[k].cpu() - init[k]).flatten()
        nz = d0.abs() > 0
        agree += int((torch.sign(d0[nz]) == torch.sign(d1[nz])).sum())
        total += int(nz.sum())
    assert total > 1e6 and agree / total > 0.9, (agree, total)
