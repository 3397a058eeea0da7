"""GPU: bf16 STORAGE mode (`precision: bf16s`, BASELINE.json config #3 "256x256 batch 32 bf16"; a build extension with no
reference counterpart -- the reference is fp32 throughout).  Activations of the content encoder /
decoder trunk live in HBM as bf16; weights, norm statistics, AdaIN parameters, accumulators, weight gradients, losses stay
fp32.  Checks, all against the fp64 oracle evaluated on the SAME bf16-valued inputs (so that only the kernel's own
arithmetic is measured):
  * a bf16 output must equal the fp64 result to bf16 rounding (2^-9 relative to the element, asserted as 8e-3 of the
    tensor maximum) and fp32 outputs (weight gradients, statistics, the image head) to fp32-accumulation accuracy;
  * the stated 2e-2 of the mode against the unrounded oracle, and the whole step against the fp32 step."""
import pytest
import torch

from oracle import munit_oracle as O
from tests.parity import l2err, nerr, trainer_named_params

pytestmark = pytest.mark.gpu

BF16_OUT = 8e-3        # bf16 rounding of an output tensor (normalised max)
SHARP = 3e-5           # fp32 accumulation of exact bf16 products vs fp64
MODE_TOL = 2e-2        # stated tolerance of the bf16 configuration vs the unrounded oracle
GRAD_TENSOR_TOL = 6e-2 # step level: relative L2 of EVERY gradient tensor of a bf16-storage step against the fp64 oracle taking the SAME
                       # ReLU / LeakyReLU / L1 branches (measured on MI355X: worst 4.0e-2 at 64x64 batch 2, 4.5e-2 at 256x256 batch 1;
                       # medians 1.6e-2 / 0.9e-2)


def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def r16(t):
    return t.float().bfloat16().double()


def to_dev(t, dtype):
    return t.to(dtype).to(dev()).contiguous(memory_format=torch.channels_last)


@pytest.fixture(autouse=True)
def bf16_mode():
    from munit_amd import ops
    ops.set_compute("bf16s")
    yield
    ops.set_compute("f32")


BF, F32 = torch.bfloat16, torch.float32
CASES = [
    # cin, cout, k, stride, pad, ups, act, B, H, W, in dtype, out dtype
    (64, 128, 4, 2, 1, 0, "none", 2, 16, 16, BF, BF),      # down-sampling (strided backward-data phases + fold)
    (128, 256, 4, 2, 1, 0, "none", 1, 12, 20, BF, BF),
    (256, 256, 3, 1, 1, 0, "none", 2, 8, 8, BF, BF),       # resblock conv (LDS-patch backward-data)
    (256, 256, 3, 1, 1, 0, "none", 3, 33, 17, BF, BF),     # several tiles, ragged
    (256, 128, 5, 1, 2, 1, "none", 2, 6, 8, BF, BF),       # up-sampling conv: sub-pixel forward, correlation + fold backward
    (128, 64, 5, 1, 2, 1, "none", 1, 9, 7, BF, BF),
    (3, 64, 7, 1, 3, 0, "none", 2, 20, 24, F32, BF),       # first layer: fp32 image in, bf16 out
    (64, 3, 7, 1, 3, 0, "tanh", 2, 16, 12, BF, F32),       # image head: bf16 in, fp32 image out
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%d-%d_k%ds%d_u%d" % (c[0], c[1], c[2], c[3], c[5]))
def test_conv_bf16_storage(case):
    from munit_amd import ops
    cin, cout, k, stride, pad, ups, act, B, H, W, din, dout = case
    x = rnd((B, cin, H, W), 1)
    w = rnd((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5)
    b = rnd((cout,), 3, 0.1)
    xq = r16(x) if din == BF else x.float().double()
    wq = r16(w) if cin % 64 == 0 and cout != 3 else w.float().double()    # layers off the MFMA path multiply in fp32
    xr, wr, br = xq.clone().requires_grad_(True), wq.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = O.conv_block(O.upsample2(xr) if ups else xr, wr, br, stride, pad, "reflect", None, act)
    dy = rnd(tuple(yr.shape), 4)
    dyq = r16(dy) if dout == BF else dy.float().double()
    yr.backward(dyq)

    xd = to_dev(xq, din).requires_grad_(True)
    wd = w.float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.float().to(dev()).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride, pad, "reflect", bool(ups), act, out_dtype=dout)
    assert y.dtype == dout and tuple(y.shape) == tuple(yr.shape)
    loose = bool(ups)      # the sub-pixel form merges the 5x5 weights in fp32 BEFORE rounding them to bf16
    assert nerr(y, yr) <= (MODE_TOL if loose else (BF16_OUT if dout == BF else 1e-4)), ("fwd", nerr(y, yr))
    y.backward(to_dev(dyq, dout))
    assert xd.grad.dtype == din
    assert nerr(xd.grad, xr.grad) <= (BF16_OUT if din == BF else 1e-4), ("dx", nerr(xd.grad, xr.grad))
    assert nerr(wd.grad, wr.grad) <= SHARP, ("dw", nerr(wd.grad, wr.grad))
    assert nerr(bd.grad, br.grad) <= SHARP, ("db", nerr(bd.grad, br.grad))


def test_prepared_bf16_images_follow_the_optimizer():
    """The bf16 weight images of a layer bound to an optimizer are built once and refreshed by the optimizer step: the
    forward after a step must use the new weights (not a stale image)."""
    from munit_amd import ops
    from munit_amd.trainer import FusedAdam
    conv_w = torch.nn.Parameter((rnd((128, 64, 3, 3), 5, 0.05)).float().contiguous(memory_format=torch.channels_last))
    opt = FusedAdam([conv_w], lr=1e-2, betas=(0.5, 0.999), weight_decay=0.0)
    opt.bind(dev())
    x = to_dev(r16(rnd((2, 64, 8, 8), 6)), BF)

    def fwd():
        return ops.conv2d(x, conv_w, None, 1, 1, "reflect", False, "none")

    y0 = fwd()
    ref0 = O.conv_block(x.double().cpu(), r16(conv_w.detach().cpu().double()), None, 1, 1, "reflect", None, "none")
    assert nerr(y0, ref0) <= BF16_OUT
    opt.zero_grad()
    y0.float().pow(2).sum().backward()
    torch.cuda.synchronize()
    assert float(opt.flat_g.abs().max()) > 0
    opt.step()
    y1 = fwd()
    ref1 = O.conv_block(x.double().cpu(), r16(conv_w.detach().cpu().double()), None, 1, 1, "reflect", None, "none")
    assert nerr(ref1, ref0) > 1e-2              # the step really moved the weights
    assert nerr(y1, ref1) <= BF16_OUT, nerr(y1, ref1)


@pytest.mark.parametrize("kind", ["in", "in_relu_res", "adain", "ln"])
def test_norms_bf16_storage(kind):
    from munit_amd import ops
    B, C, H, W = 2, 64, 12, 10
    xq = r16(rnd((B, C, H, W), 7) * 2 + 0.5)
    xr = xq.clone().requires_grad_(True)
    xd = to_dev(xq, BF).requires_grad_(True)
    extra_ref, extra_dev = [], []
    if kind == "in":
        yr = O.instance_norm(xr)
        y = ops.instance_norm(xd)
    elif kind == "in_relu_res":
        rq = r16(rnd((B, C, H, W), 8))
        yr = O.instance_norm(xr) + rq          # residual without relu: the second conv of a ResBlock
        y = ops.instance_norm(xd, relu=False, residual=to_dev(rq, BF))
    elif kind == "adain":
        params = rnd((B, 2 * C), 9)
        pr = params.clone().requires_grad_(True)
        pd = params.float().to(dev()).requires_grad_(True)
        yr = torch.clamp_min(O.adain(xr, pr[:, C:], pr[:, :C]), 0)
        y = ops.adain(xd, pd, C, 0, relu=True)
        extra_ref, extra_dev = [pr], [pd]
    else:
        gamma, beta = rnd((C,), 10).abs() + 0.1, rnd((C,), 11, 0.1)
        gr, btr = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        gd, btd = gamma.float().to(dev()).requires_grad_(True), beta.float().to(dev()).requires_grad_(True)
        yr = torch.clamp_min(O.munit_layer_norm(xr, gr, btr), 0)
        y = ops.layer_norm(xd, gd, btd, relu=True)
        extra_ref, extra_dev = [gr, btr], [gd, btd]
    assert y.dtype == BF
    assert nerr(y, yr) <= BF16_OUT, nerr(y, yr)
    dyq = r16(rnd(tuple(yr.shape), 12))
    # the ReLU mask of the reference must be the kernel's (taken from fp32 values before rounding): pin it via the output
    yr.backward(dyq)
    y.backward(to_dev(dyq, BF))
    assert xd.grad.dtype == BF
    assert nerr(xd.grad, xr.grad) <= 1.2e-2, ("dx", nerr(xd.grad, xr.grad))
    for a, r in zip(extra_dev, extra_ref):
        assert a.grad.dtype == F32 and nerr(a.grad, r.grad) <= 2e-3, nerr(a.grad, r.grad)


def test_l1_bf16_storage():
    from munit_amd import ops
    a, b = r16(rnd((2, 64, 9, 7), 13)), r16(rnd((2, 64, 9, 7), 14))
    ad, bd = to_dev(a, BF).requires_grad_(True), to_dev(b, BF).requires_grad_(True)
    out = ops.l1_mean(ad, bd)
    assert out.dtype == F32 and abs(float(out.detach()) - float((a - b).abs().mean())) <= 1e-5
    out.backward()
    g = torch.sign(a - b) / a.numel()
    assert ad.grad.dtype == BF and nerr(ad.grad, g) <= BF16_OUT and nerr(bd.grad, -g) <= BF16_OUT


@pytest.mark.parametrize("size,batch", [(64, 2), (256, 32)], ids=["64_b2", "config3_256_b32"])
def test_step_bf16_storage_tracks_fp32_step(size, batch):
    """One dis_update + gen_update with bf16 storage against the same step in fp32 (same weights, same batch): every
    loss within the mode's stated 2e-2; the activations of the trunk are bf16, the images fp32.  The second case is
    BASELINE.json config #3 at its own size (256x256, batch 32)."""
    from munit_amd import ops
    from munit_amd.trainer import MUNIT_Trainer
    import bench
    x_a, x_b, m_a, m_b = (t.to(dev()) for t in bench.make_batch(batch, size))
    out = {}
    for prec in ("f32", "bf16s"):
        hp = bench.bench_hp(size, batch)
        hp["precision"] = prec
        torch.manual_seed(1234)
        tr = MUNIT_Trainer(hp)
        tr.to(dev())
        with torch.no_grad():
            c, s = tr.gen.encode(x_a, 1)
            img = tr.gen.decode(c, s, 1)
        assert c.dtype == (BF if prec == "bf16s" else F32) and s.dtype == F32 and img.dtype == F32
        torch.manual_seed(5)
        tr.dis_update(x_a, x_b, hp)
        tr.gen_update(x_a, x_b, hp, m_a, m_b)
        names = [n for n in vars(tr) if n.startswith("loss_")]
        out[prec] = ({n: float(getattr(tr, n)) for n in names}, tr.gen_opt.flat_g.detach().clone(), img.detach().clone())
    ops.set_compute("bf16s")
    lf, lb = out["f32"][0], out["bf16s"][0]
    assert set(lf) == set(lb) and len(lf) >= 10
    for n in lf:
        assert abs(lb[n] - lf[n]) <= MODE_TOL * max(abs(lf[n]), 1e-3), (n, lf[n], lb[n])
    assert nerr(out["bf16s"][2], out["f32"][2]) <= MODE_TOL * 2.5     # decoded image: eleven bf16 layers deep
    # the flat generator gradient points the same way (cosine), and is finite everywhere
    # (Unpinned, bf16 rounding flips ~1 % of the L1 terms' signs and ReLU branches, each a +-1/N jump of a gradient: tensor by
    # tensor the two steps differ by 0.2 - 0.45 relative L2 at random initialisation -- measured -- although they point the same
    # way.  The per-tensor bound is therefore taken against the fp64 oracle with the kinks PINNED to the bf16 run:
    # test_step_bf16_storage_matches_the_pinned_oracle, and config #3's batch is tied to it by
    # test_config3_batch_gradient_is_the_mean_of_the_per_sample_gradients.)
    gf, gb = out["f32"][1].double(), out["bf16s"][1].double()
    assert torch.isfinite(gb).all()
    cos = float((gf * gb).sum() / (gf.norm() * gb.norm()))
    assert cos > 0.98, cos


def test_bf16_storage_two_generators_and_inference():
    """gen_state 0 (two AdaINGen) in bf16 storage: one update runs with finite losses, and the inference entry points
    (forward / sample) return fp32 images."""
    from munit_amd.trainer import MUNIT_Trainer
    import bench
    hp = bench.bench_hp(64, 2, gen_state=0)
    hp["precision"] = "bf16s"
    hp["display_size"] = 2
    torch.manual_seed(7)
    tr = MUNIT_Trainer(hp)
    tr.to(dev())
    x_a, x_b, m_a, m_b = (t.to(dev()) for t in bench.make_batch(2, 64))
    tr.update_learning_rate()
    tr.dis_update(x_a, x_b, hp)
    tr.gen_update(x_a, x_b, hp, m_a, m_b)
    for n in ("loss_gen_total", "loss_dis_total", "loss_gen_recon_c_a", "loss_gen_cycrecon_x_b"):
        v = float(getattr(tr, n).detach())
        assert v == v and 0 < v < 1e4, (n, v)
    x_ab, x_ba = tr.forward(x_a, x_b)
    outs = tr.sample(x_a, x_b)
    assert x_ab.dtype == F32 and all(o.dtype == F32 and tuple(o.shape) == (2, 3, 64, 64) for o in outs)
    assert torch.isfinite(x_ab).all() and torch.isfinite(outs[3]).all()


def test_bf16_storage_on_another_geometry():
    """bf16 storage on networks the benchmark does not use -- three down-samplings, zero padding, a LeakyReLU generator (its
    activation then follows the norms as an op of its own), an instance-normed discriminator: one update with finite losses
    that track the fp32 update on the same weights and batch within the mode's tolerance."""
    from munit_amd import ops
    from munit_amd.trainer import MUNIT_Trainer
    import bench
    size, batch = 72, 2
    x_a, x_b, m_a, m_b = (t.to(dev()) for t in bench.make_batch(batch, size))
    out = {}
    for prec in ("f32", "bf16s"):
        hp = bench.bench_hp(size, batch)
        hp["gen"] = dict(hp["gen"], dim=64, mlp_dim=64, style_dim=8, n_downsample=3, n_res=2, pad_type="zero", activ="lrelu")
        hp["dis"] = dict(hp["dis"], dim=32, n_layer=3, num_scales=2, pad_type="zero", norm="in")
        hp["precision"] = prec
        torch.manual_seed(1234)
        tr = MUNIT_Trainer(hp)
        tr.to(dev())
        torch.manual_seed(5)
        tr.dis_update(x_a, x_b, hp)
        tr.gen_update(x_a, x_b, hp, m_a, m_b)
        out[prec] = {n: float(getattr(tr, n)) for n in vars(tr) if n.startswith("loss_")}
    ops.set_compute("f32")
    hp = bench.bench_hp(size, batch)
    hp["gen"] = dict(hp["gen"], dim=32)
    hp["precision"] = "bf16s"
    with pytest.raises(ValueError, match="multiple of 64"):      # narrower generators are refused up front, not deep in a kernel
        MUNIT_Trainer(hp)
    for n, v in out["f32"].items():
        assert out["bf16s"][n] == out["bf16s"][n] and abs(out["bf16s"][n] - v) <= 3 * MODE_TOL * max(abs(v), 1e-3), (n, v, out["bf16s"][n])



@pytest.mark.parametrize("size,batch", [(64, 2), (256, 1)], ids=["64_b2", "256_b1"])
def test_step_bf16_storage_matches_the_pinned_oracle(size, batch):
    """One dis_update + gen_update in bf16 storage against the fp64 oracle that takes the bf16 run's own ReLU / LeakyReLU / L1
    branches (tests/parity.py: both sides then differentiate the same piecewise-linear function, so what remains is the bf16
    rounding of the trunk's activations and of their gradients): every loss within the mode's 2e-2, EVERY gradient tensor
    within GRAD_TENSOR_TOL relative L2 (median a quarter of it), Adam moments and weights after the step within the same.  The
    second case is config #3's resolution at batch 1; its batch of 32 is tied to per-sample steps by the next test."""
    from tests.parity import run_step_parity
    rep = run_step_parity(size=size, batch=batch, gen_state=1, iters=1, device="cuda:0", precision="bf16s", check=False)
    print({k: v for k, v in rep.items() if not isinstance(v, list)}, "loosest:", sorted(rep["grad_kinks"], key=lambda r: -r[2])[:4])
    assert rep["loss_rel"] <= MODE_TOL, rep["loss_rel"]
    assert rep["grad_l2"] <= GRAD_TENSOR_TOL and rep["grad_nerr"] <= 1.5 * GRAD_TENSOR_TOL, (rep["grad_l2"], rep["grad_nerr"])
    assert rep["grad_l2_median"] <= 2.5e-2, rep["grad_l2_median"]
    assert rep["moment_l2"] <= 2 * GRAD_TENSOR_TOL and rep["weight_abs"] <= 4.0 * 1e-4, (rep["moment_l2"], rep["weight_abs"])
    # the recorded branches differ from the oracle's own only near the kinks: within the bf16 rounding that eleven layers of
    # bf16 activations accumulate on a pre-activation (measured 5.1e-2 / 5.6e-2 of the tensor maximum on 0.6 % of the elements)
    assert rep["kink_worst_rel"] <= 0.1 and rep["kink_disagree_frac"] <= 2e-2, (rep["kink_worst_rel"], rep["kink_disagree_frac"])
    # biases ahead of an instance norm have a mathematically zero gradient: bf16 noise only
    assert rep["zero_grad_abs"] <= 2e-2, rep["zero_grad_abs"]


def test_config3_batch_statistics_are_the_means_of_the_per_sample_ones():
    """BASELINE.json config #3 at its own size (256x256, batch 32, bf16 storage), where the fp64 oracle is out of reach: every
    loss is a batch mean and every normalisation per sample, so the batch-32 dis_update / gen_update must
    reproduce the MEANS of the 32 batch-1 updates on the same weights -- a kernel that mixes samples or mis-tiles the large
    batch cannot.  Losses: every loss_* within 2e-3 relative.  Gradients, tensor by tensor: the comparison is UNPINNED between
    the two batch sizes, and in bf16 storage a rounding that lands differently when the batch is tiled differently is amplified
    layer by layer into flipped ReLU / L1 branches (measured: median 0.20 relative L2 -- the same signature the bf16-vs-fp32
    comparison shows; the PINNED per-tensor bound is test_step_bf16_storage_matches_the_pinned_oracle), so the bound here is
    the one that separates 'same gradient up to flipped kinks' from 'another gradient': relative L2 <= 0.6 and cosine >= 0.8
    on every tensor (uncorrelated tensors: 1.41 / 0), 1e-2 on the discriminators (fp32 throughout, their inputs bf16-made)."""
    import bench
    from munit_amd.trainer import MUNIT_Trainer
    size, batch = 256, 32
    data = bench.make_batch(batch, size)
    hp = bench.bench_hp(size, batch)
    hp["precision"] = "bf16s"
    torch.manual_seed(1234)
    tr = MUNIT_Trainer(hp)
    tr.to(dev())
    g0, d0 = tr.gen_opt.flat_p.detach().clone(), tr.dis_opt.flat_p.detach().clone()
    gnames, dnames = trainer_named_params(tr)
    loss_names = None

    def run(sel):
        nonlocal loss_names
        x_a, x_b, m_a, m_b = (t[sel].to(dev()) for t in data)
        with torch.no_grad():       # every run on the initial weights (the optimizers' moments do not enter a gradient)
            tr.gen_opt.flat_p.copy_(g0)
            tr.dis_opt.flat_p.copy_(d0)
        tr.gen_opt.invalidate_prepared()
        tr.dis_opt.invalidate_prepared()
        tr.dis_update(x_a, x_b, hp)
        gd = [p._munit_grad.detach().double().clone() for _, p in dnames]
        with torch.no_grad():
            tr.dis_opt.flat_p.copy_(d0)
        tr.dis_opt.invalidate_prepared()
        tr.gen_update(x_a, x_b, hp, m_a, m_b)
        gg = [p._munit_grad.detach().double().clone() for _, p in gnames]
        if loss_names is None:
            loss_names = sorted(n for n in vars(tr) if n.startswith("loss_") and torch.is_tensor(getattr(tr, n)))
        losses = torch.stack([getattr(tr, n).detach().double().reshape(()) for n in loss_names])
        return gd, gg, losses

    whole = run(list(range(batch)))
    mean = None
    for i in range(batch):
        part = run([i])
        mean = part if mean is None else ([a + b for a, b in zip(mean[0], part[0])], [a + b for a, b in zip(mean[1], part[1])],
                                          mean[2] + part[2])
    lrel = ((whole[2] - mean[2] / batch).abs() / (mean[2] / batch).abs().clamp_min(1e-3)).cpu()
    print("config #3 losses, batch 32 vs mean of per-sample:", {n: float(v) for n, v in zip(loss_names, lrel)})
    assert len(loss_names) >= 10 and float(lrel.max()) <= 2e-3, (loss_names, lrel)
    rows = [[], []]
    for k, names in ((0, dnames), (1, gnames)):
        for (n, _), w, m in zip(names, whole[k], mean[k]):
            assert torch.isfinite(w).all()
            # conv biases ahead of an instance norm / AdaIN: mathematically zero, bf16 noise on both sides
            if k == 1 and n.endswith("conv.bias") and ("_content." in n or ".model.0.model." in n):
                continue
            m = m / batch
            cos = float((w * m).sum() / (w.norm() * m.norm()).clamp_min(1e-30))
            rows[k].append((l2err(w, m), cos, n))
        rows[k].sort(reverse=True)
    print("config #3 batch 32 vs mean of per-sample gradients: worst dis", rows[0][:2], "gen", rows[1][:3],
          "median gen", rows[1][len(rows[1]) // 2][:2])
    assert len(rows[1]) >= 80
    assert rows[0][0][0] <= 1e-2, rows[0][:3]
    assert rows[1][0][0] <= 0.6 and min(r[1] for r in rows[1]) >= 0.8, rows[1][:4]
