"""CPU: the oracle (oracle/munit_oracle.py) against the golden fixtures produced from the
reference's own network modules (tests/golden/make_golden.py).  Tolerances: the fixtures'
f64 entries were produced in float64, so the f64 oracle must agree to ~1e-10 relative;
the f32 oracle to 1e-4 (PyTorch-fp32's own error is ~2e-6 forward)."""
import numpy as np
import pytest
import torch

from oracle import munit_oracle as O


def dg(t):
    f = t.detach().double().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, 8).long()
    return [float(f.sum()), float(f.abs().sum()), float(f.norm())] + [float(v) for v in f[idx]]


def close_digest(a, b, rtol, atol=0.0):
    """atol covers tensors that are mathematically zero (e.g. the gradient of a conv bias
    that feeds an instance norm), where both sides hold only rounding noise."""
    a, b = np.asarray(a), np.asarray(b)
    assert abs(a[0] - b[0]) <= rtol * abs(b[1]) + atol, (a[0], b[0])
    assert abs(a[1] - b[1]) <= rtol * abs(b[1]) + atol, (a[1], b[1])
    assert abs(a[2] - b[2]) <= rtol * abs(b[2]) + atol, (a[2], b[2])
    peak = max(np.abs(b[3:]).max(), b[2] / 100)
    assert np.abs(a[3:] - b[3:]).max() <= rtol * 10 * peak + atol, (a[3:], b[3:])


def states(hp, dtype):
    gs = hp["gen_state"]
    if gs == 1:
        gen = O.make_state(O.gen_param_shapes(hp["gen"], 3, True), "gen.", dtype)
    else:
        sh = O.gen_param_shapes(hp["gen"], 3, False)
        gen = {}
        for tag in ("a", "b"):
            gen.update({tag + "." + k: v for k, v in O.make_state(sh, "gen_%s." % tag, dtype).items()})
    dsh = O.dis_param_shapes(hp["dis"], 3)
    return gen, O.make_state(dsh, "dis_a.", dtype), O.make_state(dsh, "dis_b.", dtype)


@pytest.mark.parametrize("gs", [1, 0])
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_forward_matches_reference_modules(golden, gs, dt):
    meta, arrays = golden
    dtype = torch.float64 if dt == "f64" else torch.float32
    rtol = 1e-10 if dt == "f64" else 1e-4
    hp = O.default_hp(64, 2, gs)
    gen, dis_a, _ = states(hp, dtype)
    x_a, x_b, _, _ = O.synthetic_batch(2, 64, seed=7)
    if gs == 1:
        ga = gb = O.GenView(gen, hp["gen"], True)
        ka, kb = 1, 2
    else:
        ga, gb = O.GenView(gen, hp["gen"], False, "a."), O.GenView(gen, hp["gen"], False, "b.")
        ka = kb = None
    with torch.no_grad():
        c, s = ga.encode(x_a.to(dtype), ka)
        x_rec = ga.decode(c, s, ka)
        c2, s2 = gb.encode(x_b.to(dtype), kb)
        x_ab = gb.decode(c, s2, kb)
        d = O.dis_forward(dis_a, "", x_rec, hp["dis"])
    g = meta["fwd_gs%d_%s" % (gs, dt)]
    close_digest(dg(c), g["content"], rtol)
    close_digest(dg(s), g["style"], rtol)
    close_digest(dg(x_rec), g["x_rec"], rtol)
    close_digest(dg(x_ab), g["x_ab"], rtol)
    close_digest(dg(c2), g["content2"], rtol)
    for o, e in zip(d, g["dis"]):
        close_digest(dg(o), e, rtol)
    # full tensors against the float64 reference run
    key = "fwd_gs%d_f64" % gs
    tol = 1e-6 if dt == "f64" else 2e-4  # stored as f32 -> 1e-6 floor
    assert np.abs(x_rec.double().numpy() - arrays[key + "_x_rec"]).max() <= tol
    assert np.abs(x_ab.double().numpy() - arrays[key + "_x_ab"]).max() <= tol
    ref_s = arrays[key + "_style"]
    assert np.abs(s.double().numpy() - ref_s).max() <= tol * max(1.0, np.abs(ref_s).max())
    ref_c = arrays[key + "_content_slice"]
    assert np.abs(c[:, ::16, ::2, ::2].double().numpy() - ref_c).max() <= tol * max(1.0, np.abs(ref_c).max())
    for i, o in enumerate(d):
        r = arrays[key + "_dis%d" % i]
        assert np.abs(o.double().numpy() - r).max() <= tol * max(1.0, np.abs(r).max())


def test_norm_semantics(golden):
    """biased-variance IN/AdaIN, unbiased-std LayerNorm with eps on std."""
    _, arrays = golden
    x = torch.from_numpy(arrays["norm_x"])
    g = O.fill_det("ln.gamma", (8,), dtype=torch.float64)
    b = O.fill_det("ln.beta", (8,), dtype=torch.float64)
    assert np.abs(O.munit_layer_norm(x, g, b).numpy() - arrays["norm_ln_y"]).max() < 1e-12
    assert np.abs(O.munit_layer_norm(x[:1], g, b).numpy() - arrays["norm_ln1_y"]).max() < 1e-12
    w = O.fill_det("ad.w", (2, 8), 1.0, torch.float64)
    bb = O.fill_det("ad.b", (2, 8), 1.0, torch.float64)
    assert np.abs(O.adain(x, w, bb).numpy() - arrays["norm_adain_y"]).max() < 1e-12
    assert np.abs(O.instance_norm(x).numpy() - arrays["norm_in_y"]).max() < 1e-12


@pytest.mark.parametrize("gs,iters", [(1, 3), (0, 1)])
def test_step_matches_reference_sequence(golden, gs, iters):
    """losses, gradients (iteration 0) and weights after `iters` dis_update+gen_update pairs,
    float64, against reference modules + torch.optim.Adam + StepLR (make_golden.py)."""
    meta, _ = golden
    rec = meta["step_gs%d" % gs]
    hp = O.default_hp(64, 2, gs)
    hp["step_size"] = 2
    gen, dis_a, dis_b = states(hp, torch.float64)
    tr = O.OracleTrainer(hp, gen, dis_a, dis_b)
    x_a, x_b, m_a, m_b = (t.double() for t in O.synthetic_batch(2, 64, seed=7))
    for it in range(iters):
        tr.update_learning_rate()
        dgr = tr.dis_update(x_a, x_b)
        ggr = tr.gen_update(x_a, x_b, m_a, m_b)
        e = rec["iters"][it]
        assert abs(tr._lr() - e["lr"]) < 1e-18
        for k, v in e["losses"].items():
            assert abs(float(tr.losses[k]) - v) <= 1e-9 * max(1.0, abs(v)), (it, k)
        if it == 0:
            for g_, d_ in zip(dgr, e["dis_grad"]):
                close_digest(dg(g_), d_, 1e-8, 1e-11)
            for g_, d_ in zip(ggr, e["gen_grad"]):
                close_digest(dg(g_), d_, 1e-8, 1e-11)
    if iters == 3:
        for p, d_ in zip(tr.opt["gen"]["params"], rec["gen_after"]):
            close_digest(dg(p), d_, 1e-9)
        for p, d_ in zip(tr.opt["dis"]["params"], rec["dis_after"]):
            close_digest(dg(p), d_, 1e-9)


@pytest.mark.parametrize("name", ["gs1_guided0", "gs0_guided0", "gs1_nomask"])
def test_step_variants_match_reference_sequence(name):
    """The branches make_golden.py's fixtures do not drive, pinned to the reference MODULES the same way
    (tests/golden/make_golden_variants.py): `guided: 0` (sampled styles, trainer.py:377-379 / 405-407 / 438-440 /
    1155-1157) for both generator layouts and `recon_mask: 0` (trainer.py:476-487): losses, every gradient tensor, weights
    after one dis_update + gen_update, float64."""
    import json
    import os
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_variants.json")))[name]
    hp = O.default_hp(rec["size"], rec["batch"], rec["gen_state"])
    hp["guided"], hp["recon_mask"] = rec["guided"], rec["recon_mask"]
    gen, dis_a, dis_b = states(hp, torch.float64)
    tr = O.OracleTrainer(hp, gen, dis_a, dis_b)
    x_a, x_b, m_a, m_b = (t.double() for t in O.synthetic_batch(rec["batch"], rec["size"], seed=7))
    sd = hp["gen"]["style_dim"]

    def styles(seed):       # the reference draws s_a, s_b at the top of each update from the host RNG (trainer.py:366-367)
        torch.manual_seed(seed)
        return torch.randn(rec["batch"], sd, 1, 1).double(), torch.randn(rec["batch"], sd, 1, 1).double()

    tr.update_learning_rate()
    dgr = tr.dis_update(x_a, x_b, *styles(rec["style_seeds"][0]))
    ggr = tr.gen_update(x_a, x_b, m_a, m_b, *styles(rec["style_seeds"][1]))
    for k, v in rec["losses"].items():
        assert abs(float(tr.losses[k]) - v) <= 1e-9 * max(1.0, abs(v)), (k, float(tr.losses[k]), v)
    for g_, d_ in zip(dgr, rec["dis_grad"]):
        close_digest(dg(g_), d_, 1e-8, 1e-11)
    for g_, d_ in zip(ggr, rec["gen_grad"]):
        if d_ is not None:
            close_digest(dg(g_), d_, 1e-8, 1e-11)
    for p, d_ in zip(tr.opt["gen"]["params"], rec["gen_after"]):
        close_digest(dg(p), d_, 1e-9)
    for p, d_ in zip(tr.opt["dis"]["params"], rec["dis_after"]):
        close_digest(dg(p), d_, 1e-9)


def test_extraadam_matches_reference_file():
    """oracle ExtraAdamState vs the parameter trace produced by the reference's own
    scripts/extraadam.py (tests/golden/make_golden_extraadam.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_extraadam.npz"))
    p = [torch.from_numpy(g["p0"][:35].copy()).reshape(7, 5), torch.from_numpy(g["p0"][35:].copy())]
    st = O.ExtraAdamState(p, 1e-3, (0.5, 0.999), 1e-4)
    for k, mode in enumerate(g["modes"]):
        grads = [torch.from_numpy(g["g"][k][:35].copy()).reshape(7, 5), torch.from_numpy(g["g"][k][35:].copy())]
        getattr(st, str(mode))(grads)
        got = np.concatenate([t.numpy().reshape(-1) for t in p])
        assert np.abs(got - g["trace"][k]).max() <= 1e-14, (k, mode)
    with pytest.raises(RuntimeError):
        st.step([torch.zeros(7, 5, dtype=torch.float64), torch.zeros(11, dtype=torch.float64)])


def _geometry_names():
    from tests.geometries import ALL
    return [g[0] for g in ALL]


@pytest.mark.parametrize("name", _geometry_names())
def test_step_geometries_match_reference_sequence(name):
    """Every geometry of tests/geometries.py -- the list the GPU step-parity tests iterate -- pinned to the reference MODULES
    built from that geometry's config (tests/golden/make_golden_geometries.py: networks.py:170-209, 262-388, 442-563, 20-115
    driven per trainer.py:365-561 / 1145-1186): forward digests, every loss_*, every gradient tensor, weights after one
    dis_update + gen_update, float64.  A geometry without a fixture fails here (KeyError), so the GPU suite cannot compare the
    HIP path to an oracle the reference has not pinned."""
    import json
    import os
    from tests.geometries import ALL, merged_hp
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_geometries.json")))[name]
    _, size, over = [g for g in ALL if g[0] == name][0]
    assert rec["over"] == json.loads(json.dumps(over)) and rec["size"] == (size if isinstance(size, int) else list(size)), \
        "tests/geometries.py changed: regenerate tests/golden/golden_geometries.json"
    hp = merged_hp(O.default_hp, size, over, rec["batch"])
    nin = hp["input_dim_a"]
    dtype = torch.float64
    if hp["gen_state"] == 1:
        gen = O.make_state(O.gen_param_shapes(hp["gen"], nin, True), "gen.", dtype)
    else:
        sh = O.gen_param_shapes(hp["gen"], nin, False)
        gen = {}
        for tag in ("a", "b"):
            gen.update({tag + "." + k: v for k, v in O.make_state(sh, "gen_%s." % tag, dtype).items()})
    dsh = O.dis_param_shapes(hp["dis"], nin)
    dis_a, dis_b = O.make_state(dsh, "dis_a.", dtype), O.make_state(dsh, "dis_b.", dtype)
    tr = O.OracleTrainer(hp, gen, dis_a, dis_b)
    x_a, x_b, m_a, m_b = (t.double() for t in O.synthetic_batch(rec["batch"], size, seed=7))
    x_a, x_b = x_a[:, :nin].contiguous(), x_b[:, :nin].contiguous()
    # forward digests on the initial weights
    (ga, ka), (gb, kb) = tr._views()
    with torch.no_grad():
        c_a, s_a = ga.encode(x_a, ka)
        c_b, s_b = gb.encode(x_b, kb)
        x_ba, x_ab = ga.decode(c_b, s_a, ka), gb.decode(c_a, s_b, kb)
        d = O.dis_forward(tr.dis_a, "", x_ba, hp["dis"])
    f = rec["forward"]
    close_digest(dg(c_a), f["content"], 1e-10)
    close_digest(dg(s_b), f["style"], 1e-10)
    close_digest(dg(x_ba), f["x_ba"], 1e-10)
    close_digest(dg(x_ab), f["x_ab"], 1e-10)
    assert len(d) == len(f["dis"])
    for o, e in zip(d, f["dis"]):
        close_digest(dg(o), e, 1e-10)
    tr.update_learning_rate()
    dgr = tr.dis_update(x_a, x_b)
    ggr = tr.gen_update(x_a, x_b, m_a, m_b)
    for k, v in rec["losses"].items():
        assert abs(float(tr.losses[k]) - v) <= 1e-9 * max(1.0, abs(v)), (k, float(tr.losses[k]), v)
    assert len(dgr) == len(rec["dis_grad"]) and len(ggr) == len(rec["gen_grad"])
    for g_, d_ in zip(dgr, rec["dis_grad"]):
        close_digest(dg(g_), d_, 1e-8, 1e-11)
    for g_, d_ in zip(ggr, rec["gen_grad"]):
        if d_ is not None:
            close_digest(dg(g_), d_, 1e-8, 1e-11)
        else:
            assert g_ is None or float(g_.abs().max()) == 0.0
    for p, d_ in zip(tr.opt["gen"]["params"], rec["gen_after"]):
        close_digest(dg(p), d_, 1e-9)
    for p, d_ in zip(tr.opt["dis"]["params"], rec["dis_after"]):
        close_digest(dg(p), d_, 1e-9)
