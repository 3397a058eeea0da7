"""GPU: every HIP kernel (through the C ABI via munit_amd.ops) against the fp64 CPU oracle
on the same seeded inputs.  Tolerances (normalised max error = max|d| / max|ref|, SURVEY.md
section 8c): forward 1e-4 (we assert the tighter 2e-5 the fp32 MFMA chain actually gives),
gradients 1e-2 (asserted 1e-4)."""
import pytest
import torch

from oracle import munit_oracle as O
from tests.parity import nerr

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-5
BWD_TOL = 1e-4


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


CONV_CASES = [
    # cin, cout, k, stride, pad, pad_type, ups, act, B, H, W
    (3, 64, 7, 1, 3, "reflect", 0, "relu", 2, 20, 24),      # first encoder layer (K=147, unaligned)
    (64, 128, 4, 2, 1, "reflect", 0, "relu", 2, 16, 16),    # down-sampling
    (128, 256, 4, 2, 1, "reflect", 0, "none", 1, 12, 20),
    (256, 256, 3, 1, 1, "reflect", 0, "none", 2, 8, 8),     # resblock conv
    (256, 128, 5, 1, 2, "reflect", 1, "none", 2, 6, 8),     # upsample x2 + 5x5
    (128, 64, 5, 1, 2, "reflect", 1, "none", 1, 9, 7),
    (256, 128, 5, 1, 2, "reflect", 1, "relu", 2, 12, 10),   # large enough for the box-sum backward-data (H, W >= 8)
    (128, 64, 5, 1, 2, "reflect", 1, "none", 1, 8, 19),
    (64, 3, 7, 1, 3, "reflect", 0, "tanh", 2, 16, 12),      # image head (N=3)
    (3, 64, 4, 2, 1, "reflect", 0, "lrelu", 2, 16, 16),     # D first layer (K=48)
    (256, 512, 4, 2, 1, "reflect", 0, "lrelu", 2, 4, 4),    # D last layer
    (512, 1, 1, 1, 0, "zero", 0, "none", 2, 4, 4),          # D head
    (256, 16, 1, 1, 0, "zero", 0, "none", 3, 1, 1),         # style head on 1x1
    (32, 48, 3, 1, 1, "zero", 0, "relu", 2, 9, 11),         # zero pad, odd sizes, odd channels
    (36, 20, 4, 2, 1, "reflect", 0, "none", 2, 9, 11),      # unaligned Cin, odd sizes with stride 2
    (64, 64, 4, 2, 1, "reflect", 0, "none", 2, 2, 2),       # tiniest reflect case (2x2 -> 1x1)
    # stride-1 reflect layers whose backward-data folds <= 2 padded positions per axis: direct-to-LDS tiles + LDS patch
    (256, 256, 3, 1, 1, "reflect", 0, "none", 2, 20, 24),   # several M-tiles, rows that straddle tile boundaries
    (64, 96, 3, 1, 1, "reflect", 0, "relu", 1, 4, 5),       # smallest extent with at most one mirror per row (H = 4)
    (64, 64, 5, 1, 2, "reflect", 0, "none", 2, 7, 9),       # pad 2: two mirrored rows per edge
    (32, 64, 7, 1, 3, "reflect", 0, "none", 1, 10, 13),     # pad 3
    (64, 64, 3, 1, 1, "reflect", 0, "none", 1, 3, 3),       # H = 3: row 1 mirrors both ways -> register-path fallback
    # 3-channel head on the 4x4x1 MFMA kernel: ragged tile edges, two 64-channel passes, zero padding
    (64, 3, 7, 1, 3, "reflect", 0, "tanh", 1, 9, 37),
    (128, 3, 7, 1, 3, "zero", 0, "none", 2, 8, 20),
    # three INPUT channels on the direct-to-LDS 4-channel-tap path: ragged sizes, stride 2, 5x5 (K-tile tail of zero taps)
    (3, 64, 7, 1, 3, "reflect", 0, "relu", 1, 13, 37),
    (3, 128, 4, 2, 1, "reflect", 0, "lrelu", 2, 10, 14),
    (3, 32, 5, 1, 2, "zero", 0, "none", 2, 9, 11),
    (32, 3, 5, 1, 2, "reflect", 0, "none", 2, 9, 11),        # 3 output channels, not 7x7: backward-data on that path too
    # Winograd F(2x2, 3x3) path (conv_wino.hip): partial 8x8-tile blocks, zero and reflect padding, fused activations,
    # K = 9 chunks of 8, two N-blocks, the smallest extent, and one block-aligned trunk-like shape
    (64, 128, 3, 1, 1, "zero", 0, "lrelu", 2, 20, 12),
    (72, 64, 3, 1, 1, "reflect", 0, "relu", 1, 18, 34),
    (8, 64, 3, 1, 1, "reflect", 0, "none", 3, 2, 2),
    (128, 256, 3, 1, 1, "reflect", 0, "none", 1, 32, 48),
    (64, 64, 3, 1, 1, "reflect", 0, "none", 1, 2, 2),       # one tile: the backward-weight chunk is 7/8 empty
    (64, 192, 3, 1, 1, "zero", 0, "none", 3, 6, 10),        # 45 tiles: ragged last chunk, zero padding, three N-blocks
    # 4x4 / stride 2 layers as F(3x3, 2x2) over the four input phases: zero padding, ragged 3x3 tiles, a 26x26 output
    (64, 64, 4, 2, 1, "zero", 0, "lrelu", 2, 10, 14),
    (8, 128, 4, 2, 1, "reflect", 0, "none", 1, 52, 52),
    # sub-pixel layers at the extents of the 64x64 step tests: several 8x8-tile blocks per axis in the backward-data
    (256, 128, 5, 1, 2, "reflect", 1, "none", 2, 16, 16),
    (128, 64, 5, 1, 2, "reflect", 1, "none", 1, 32, 32),
]


def ref_conv(x, w, b, stride, pad, pad_type, ups, act):
    if ups:
        x = O.upsample2(x)
    return O.conv_block(x, w, b, stride, pad, pad_type, None, act)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "c%d-%d_k%ds%d_%s_u%d_%s" % (c[0], c[1], c[2], c[3], c[5], c[6], c[7]))
def test_conv_fwd_bwd(case):
    from munit_amd import ops
    cin, cout, k, stride, pad, pt, ups, act, B, H, W = case
    x = rnd((B, cin, H, W), 1)
    w = rnd((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5)
    b = rnd((cout,), 3, 0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = ref_conv(xr, wr, br, stride, pad, pt, ups, act)
    dy = rnd(tuple(yr.shape), 4)
    yr.backward(dy)

    xd = x.float().to(dev()).requires_grad_(True)
    wd = w.float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.float().to(dev()).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride, pad, pt, bool(ups), act)
    assert tuple(y.shape) == tuple(yr.shape)
    assert nerr(y, yr) <= FWD_TOL, nerr(y, yr)
    y.backward(dy.float().to(dev()))
    assert nerr(xd.grad, xr.grad) <= BWD_TOL, ("dx", nerr(xd.grad, xr.grad))
    assert nerr(wd.grad, wr.grad) <= BWD_TOL, ("dw", nerr(wd.grad, wr.grad))
    assert nerr(bd.grad, br.grad) <= BWD_TOL, ("db", nerr(bd.grad, br.grad))


def test_conv_stride2_winograd_on_small_shapes():
    """The F(3x3, 2x2) kernel of the 4x4 / stride 2 layers only takes shapes that fill the chip (no split over K); the
    library reads MUNIT_WINO_S2_MIN_BLOCKS once per process, so a child process with the threshold at 1 pushes the small
    stride-2 cases of CONV_CASES (ragged 3x3 tiles, zero padding, one-block grids) through it."""
    import os, subprocess, sys
    env = dict(os.environ, MUNIT_WINO_S2_MIN_BLOCKS="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_ops.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_conv_fwd_bwd and k4s2", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


def test_conv_wgrad_accumulates_into_buffer():
    """The trainer path: backward-weight adds into a preallocated buffer (beta = 1)."""
    from munit_amd import ops
    x = rnd((2, 64, 8, 8), 1).float().to(dev())
    w = rnd((64, 64, 3, 3), 2, 0.05).float().to(dev()).contiguous(memory_format=torch.channels_last)
    b = rnd((64,), 3).float().to(dev())
    dy = rnd((2, 64, 8, 8), 4).float().to(dev())
    dw0, db0 = ops.conv2d_wgrad_raw(x, dy, w.shape, 1, 1, "reflect", False)
    dw, db = dw0.clone(), db0.clone()
    ops.conv2d_wgrad_raw(x, dy, w.shape, 1, 1, "reflect", False, dw=dw, db=db, beta=1.0)
    assert nerr(dw, 2 * dw0) <= 1e-6 and nerr(db, 2 * db0) <= 1e-6


def test_linear():
    from munit_amd import ops
    x = rnd((5, 16), 1)
    w = rnd((256, 16), 2, 0.2)
    b = rnd((256,), 3, 0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = torch.clamp_min(torch.nn.functional.linear(xr, wr, br), 0)
    dy = rnd(tuple(yr.shape), 4)
    yr.backward(dy)
    xd, wd, bd = (t.float().to(dev()).requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xd, wd, bd, "relu")
    assert nerr(y, yr) <= FWD_TOL
    y.backward(dy.float().to(dev()))
    assert nerr(xd.grad, xr.grad) <= BWD_TOL
    assert nerr(wd.grad, wr.grad) <= BWD_TOL
    assert nerr(bd.grad, br.grad) <= BWD_TOL


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (3, 256, 8, 8), (1, 128, 31, 17), (2, 48, 5, 7)])
@pytest.mark.parametrize("mode", ["in", "in_relu", "adain_relu", "adain_res", "in_lrelu", "adain_tanh", "adain_lrelu"])
def test_instance_norm(shape, mode):
    from munit_amd import ops
    B, C, H, W = shape
    x = rnd(shape, 1, 1.7) + 0.4
    res = rnd(shape, 5)
    params = rnd((B, 4 * C), 2) + 0.5
    w_off, b_off = 3 * C, C
    xr = x.clone().requires_grad_(True)
    pr = params.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True)
    if mode.startswith("adain"):
        yr = O.adain(xr, pr[:, w_off:w_off + C], pr[:, b_off:b_off + C])
    else:
        yr = O.instance_norm(xr)
    if mode.endswith("lrelu"):        # networks.py:672 nn.LeakyReLU(0.2) behind a norm (networks.py:695-701): fused into the norm kernels
        yr = torch.nn.functional.leaky_relu(yr, 0.2)
    elif mode.endswith("tanh"):
        yr = torch.tanh(yr)
    elif mode.endswith("relu"):
        yr = torch.clamp_min(yr, 0)
    if mode.endswith("res"):
        yr = yr + rr
    dy = rnd(shape, 4)
    yr.backward(dy)

    xd = x.float().to(dev()).requires_grad_(True)
    pd = params.float().to(dev()).requires_grad_(True)
    rd = res.float().to(dev()).requires_grad_(True)
    relu = "lrelu" if mode.endswith("lrelu") else "tanh" if mode.endswith("tanh") else mode.endswith("relu")
    residual = rd if mode.endswith("res") else None
    if mode.startswith("adain"):
        y = ops.adain(xd, pd, w_off, b_off, relu, residual)
    else:
        y = ops.instance_norm(xd, relu, residual)
    assert nerr(y, yr) <= FWD_TOL, nerr(y, yr)
    y.backward(dy.float().to(dev()))
    assert nerr(xd.grad, xr.grad) <= BWD_TOL, nerr(xd.grad, xr.grad)
    if mode.startswith("adain"):
        assert nerr(pd.grad, pr.grad) <= BWD_TOL, nerr(pd.grad, pr.grad)
    if mode.endswith("res"):
        assert nerr(rd.grad, rr.grad) <= 1e-6


@pytest.mark.parametrize("shape", [(2, 128, 16, 16), (1, 64, 33, 9), (3, 64, 8, 8)])
@pytest.mark.parametrize("relu", [False, True, "lrelu", "tanh"])
def test_layer_norm(shape, relu):
    from munit_amd import ops
    B, C, H, W = shape
    x = rnd(shape, 1, 2.0) - 0.7
    g = torch.rand(C, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    bt = rnd((C,), 3, 0.3)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, g, bt))
    yr = O.munit_layer_norm(xr, gr, br)
    if relu == "lrelu":
        yr = torch.nn.functional.leaky_relu(yr, 0.2)
    elif relu == "tanh":
        yr = torch.tanh(yr)
    elif relu:
        yr = torch.clamp_min(yr, 0)
    dy = rnd(shape, 4)
    yr.backward(dy)
    xd, gd, bd = (t.float().to(dev()).requires_grad_(True) for t in (x, g, bt))
    y = ops.layer_norm(xd, gd, bd, relu)
    assert nerr(y, yr) <= FWD_TOL, nerr(y, yr)
    y.backward(dy.float().to(dev()))
    assert nerr(xd.grad, xr.grad) <= BWD_TOL, nerr(xd.grad, xr.grad)
    assert nerr(gd.grad, gr.grad) <= BWD_TOL
    assert nerr(bd.grad, br.grad) <= BWD_TOL


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (1, 3, 9, 13), (2, 3, 2, 2), (1, 5, 1, 4)])
def test_avgpool(shape):
    from munit_amd import ops
    x = rnd(shape, 1)
    xr = x.clone().requires_grad_(True)
    yr = O.avgpool_3s2(xr)
    dy = rnd(tuple(yr.shape), 2)
    yr.backward(dy)
    xd = x.float().to(dev()).requires_grad_(True)
    y = ops.avgpool3s2(xd)
    assert tuple(y.shape) == tuple(yr.shape)
    assert nerr(y, yr) <= 1e-6
    y.backward(dy.float().to(dev()))
    assert nerr(xd.grad, xr.grad) <= 1e-6


def test_global_avgpool():
    from munit_amd import ops
    x = rnd((3, 256, 16, 16), 1)
    xr = x.clone().requires_grad_(True)
    yr = xr.mean(dim=(2, 3), keepdim=True)
    dy = rnd(tuple(yr.shape), 2)
    yr.backward(dy)
    xd = x.float().to(dev()).requires_grad_(True)
    y = ops.global_avgpool(xd)
    assert nerr(y, yr) <= 1e-6
    y.backward(dy.float().to(dev()))
    assert nerr(xd.grad, xr.grad) <= 1e-6


@pytest.mark.parametrize("masked", [False, True])
def test_l1_mean(masked):
    from munit_amd import ops
    a, b = rnd((2, 3, 17, 19), 1), rnd((2, 3, 17, 19), 2)
    m = (torch.rand(2, 1, 17, 19, generator=torch.Generator().manual_seed(3)) > 0.5).double() if masked else None
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    lr = O.l1_masked(ar, br, m) if masked else O.l1(ar, br)
    (lr * 12.0).backward()
    ad, bd = (t.float().to(dev()).requires_grad_(True) for t in (a, b))
    md = m.float().to(dev()) if masked else None
    l = ops.l1_mean(ad, bd, md)
    assert abs(float(l) - float(lr)) <= 1e-6 * abs(float(lr))
    torch.autograd.backward([l], [torch.tensor(12.0, device=dev())])
    assert nerr(ad.grad, ar.grad) <= 1e-6 and nerr(bd.grad, br.grad) <= 1e-6


def test_l1_mean_content_and_style_shapes():
    from munit_amd import ops
    for shape in [(2, 256, 8, 8), (3, 16, 1, 1)]:
        a, b = rnd(shape, 1), rnd(shape, 2)
        l = ops.l1_mean(a.float().to(dev()), b.float().to(dev()))
        assert abs(float(l) - float(O.l1(a, b))) <= 1e-6 * float(O.l1(a, b))


def test_mse_const_and_scalar_sum():
    from munit_amd import ops
    xs = [rnd((2, 1, 4, 4), 1), rnd((2, 1, 2, 2), 2), rnd((2, 1, 1, 1), 3)]
    xr = [t.clone().requires_grad_(True) for t in xs]
    lr = sum(torch.mean((t - 1) ** 2) for t in xr)
    (3.0 * lr).backward()
    xd = [t.float().to(dev()).requires_grad_(True) for t in xs]
    l = ops.scalar_sum([ops.mse_const(t, 1.0) for t in xd])
    assert abs(float(l) - float(lr)) <= 1e-6 * float(lr)
    torch.autograd.backward([l], [torch.tensor(3.0, device=dev())])
    for d, r in zip(xd, xr):
        assert nerr(d.grad, r.grad) <= 1e-6


def test_adam_matches_torch_semantics():
    from munit_amd import ops
    n = 1003
    p, g = rnd((n,), 1), rnd((n,), 2, 0.01)
    m, v = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    pd, gd = p.float().to(dev()).contiguous(), g.float().to(dev()).contiguous()
    md, vd = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    pr = p.clone()
    for step in (1, 2, 3):
        O.adam_update(pr, g, m, v, step, 1e-4, 0.5, 0.999, 1e-8, 1e-4)
        ops.adam_step(pd, gd, md, vd, 1e-4, 0.5, 0.999, 1e-8, 1e-4, step)
    # p is O(1) in fp32: each of the 3 updates rounds to ~6e-8 * |p|
    ep, em, ev = float((pd.double().cpu() - pr).abs().max()), nerr(md, m), nerr(vd, v)
    assert ep <= 2e-6 and em <= 1e-6 and ev <= 1e-6, (ep, em, ev)


def test_extraadam_kernel_matches_oracle():
    from munit_amd import ops
    n = 777
    p0 = rnd((n,), 1)
    gs = [rnd((n,), 10 + k, 0.3) for k in range(5)]
    pr = [p0.clone()]
    st = O.ExtraAdamState(pr, 1e-3, (0.5, 0.999), 1e-4)
    pd = p0.float().to(dev()).contiguous()
    md, vd, sd = torch.zeros(n, device=dev()), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    has_copy = False
    for k, mode in enumerate(["extrapolation", "step", "extrapolation", "extrapolation", "step"]):
        getattr(st, mode)([gs[k]])
        code = 2 if mode == "step" else (1 if has_copy else 0)
        has_copy = mode != "step"
        ops.extraadam_step(pd, gs[k].float().to(dev()).contiguous(), md, vd, sd, 1e-3, 0.5, 0.999, 1e-8, 1e-4, k + 1, code)
        assert float((pd.double().cpu() - pr[0]).abs().max()) <= 2e-6, (k, mode)
    assert nerr(md, st.m[0]) <= 1e-6 and nerr(vd, st.v[0]) <= 1e-6


def test_cpu_tensor_is_refused():
    """No CPU fallback: the product path must fail loudly off-device."""
    from munit_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), None, 1, 1, "reflect")


def test_linear_entry_points():
    """munit_linear_fwd / munit_linear_bwd (the named C entry points of nn.Linear) straight through ctypes."""
    import ctypes
    from munit_amd import _lib
    lib = _lib.load()
    B, K, N = 5, 256, 4096
    x, w, b, dy = rnd((B, K), 1), rnd((N, K), 2, 0.05), rnd((N,), 3, 0.1), rnd((B, N), 4)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = torch.relu(xr @ wr.t() + br)
    yr.backward(dy)
    d = dev()
    xd, wd, bd = (t.float().to(d).contiguous() for t in (x, w, b))
    y = torch.empty(B, N, device=d)
    nws = lib.munit_linear_workspace_bytes(B, K, N)
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=d)
    vp = ctypes.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.munit_linear_fwd(vp(xd.data_ptr()), vp(wd.data_ptr()), vp(bd.data_ptr()), vp(y.data_ptr()), B, K, N,
                                    _lib.ACT["relu"], ctypes.c_float(0.2), vp(ws.data_ptr()), ctypes.c_size_t(nws), st), "linear_fwd")
    assert nerr(y, yr) <= FWD_TOL
    g = (dy * (yr > 0)).float().to(d).contiguous()      # gradient at the pre-activation output
    dx, dw, db = torch.empty_like(xd), torch.empty_like(wd), torch.empty_like(bd)
    _lib.check(lib.munit_linear_bwd(vp(xd.data_ptr()), vp(wd.data_ptr()), vp(g.data_ptr()), vp(dx.data_ptr()),
                                    vp(dw.data_ptr()), vp(db.data_ptr()), B, K, N, ctypes.c_float(0.0), vp(ws.data_ptr()),
                                    ctypes.c_size_t(nws), st), "linear_bwd")
    assert nerr(dx, xr.grad) <= BWD_TOL and nerr(dw, wr.grad) <= BWD_TOL and nerr(db, br.grad) <= BWD_TOL
