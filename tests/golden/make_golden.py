"""Generate the golden fixtures under tests/golden/ from the REFERENCE network modules.

Runs only in the build container (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports /root/reference/scripts/networks.py (the only importable part of the
reference's hot path), loads deterministic weights
(oracle.munit_oracle.fill_det, recipe shared with the tests so weights are never stored),
and records inputs-by-recipe + expected outputs.  The trainer module itself cannot be
imported (extraadam.py has no imports, torchvision/comet_ml absent, hard-coded .cuda()),
so the step-level fixtures drive the reference *modules* through the update sequence of
trainer.py:365-561 / :1145-1186 with torch.optim.Adam + StepLR as trainer.py:109-122 does.

Fixtures are data only: seeds/recipes, expected tensors or per-tensor digests.
"""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/scripts")

import networks as ref  # noqa: E402  (reference, read-only)
from oracle import munit_oracle as O  # noqa: E402

torch.set_num_threads(8)


def digest(t: torch.Tensor):
    """Per-tensor digest: sum, abs-sum, l2 and 8 strided samples, all float64."""
    f = t.detach().double().reshape(-1)
    n = f.numel()
    idx = torch.linspace(0, n - 1, 8).long()
    return [float(f.sum()), float(f.abs().sum()), float(f.norm())] + [float(v) for v in f[idx]]


def load_ref(module, state, strict_params=True):
    sd = module.state_dict()
    for k in sd:
        if k in state:
            sd[k] = state[k].detach().clone()
    missing = [k for k, _ in module.named_parameters() if k not in state]
    assert not missing, missing
    module.load_state_dict(sd)
    return module


def build_ref(hp, dtype):
    gs = hp["gen_state"]
    out = {}
    if gs == 1:
        shapes = O.gen_param_shapes(hp["gen"], 3, True)
        m = ref.AdaINGen_double(3, hp["gen"])
        assert [k for k, _ in m.named_parameters()] == list(shapes.keys()), "param order/name mismatch"
        assert {k: tuple(v.shape) for k, v in m.named_parameters()} == shapes
        st = O.make_state(shapes, "gen.", dtype)
        out["gen"] = load_ref(m.to(dtype), st)
        out["gen_state_oracle"] = st
    else:
        shapes = O.gen_param_shapes(hp["gen"], 3, False)
        st = {}
        for tag in ("a", "b"):
            m = ref.AdaINGen(3, hp["gen"])
            assert [k for k, _ in m.named_parameters()] == list(shapes.keys())
            s = O.make_state(shapes, "gen_%s." % tag, dtype)
            out["gen_" + tag] = load_ref(m.to(dtype), s)
            st.update({tag + "." + k: v for k, v in s.items()})
        out["gen_state_oracle"] = st
    dshapes = O.dis_param_shapes(hp["dis"], 3)
    for tag in ("a", "b"):
        m = ref.MsImageDis(3, hp["dis"])
        assert [k for k, _ in m.named_parameters()] == list(dshapes.keys())
        s = O.make_state(dshapes, "dis_%s." % tag, dtype)
        # discriminators use gaussian(0, 0.02)-scale weights in the reference (trainer.py:126-127)
        out["dis_" + tag] = load_ref(m.to(dtype), s)
        out["dis_%s_state_oracle" % tag] = s
    return out


def ref_encode(R, hp, x, k):
    if hp["gen_state"] == 1:
        return R["gen"].encode(x, k)
    return R["gen_a" if k == 1 else "gen_b"].encode(x)


def ref_decode(R, hp, c, s, k):
    if hp["gen_state"] == 1:
        return R["gen"].decode(c, s, k)
    return R["gen_a" if k == 1 else "gen_b"].decode(c, s)


def ref_gen_losses(R, hp, x_a, x_b, m_a, m_b):
    """Reference modules driven per trainer.py:401-558 (guided == 1 branch)."""
    l1 = lambda a, b: torch.mean(torch.abs(a - b))
    l1m = lambda a, b, m: torch.mean(torch.abs(torch.mul((a - b), 1 - m)))
    c_a, s_a = ref_encode(R, hp, x_a, 1)
    c_b, s_b = ref_encode(R, hp, x_b, 2)
    x_a_recon = ref_decode(R, hp, c_a, s_a, 1)
    x_b_recon = ref_decode(R, hp, c_b, s_b, 2)
    x_ba = ref_decode(R, hp, c_b, s_a, 1)
    x_ab = ref_decode(R, hp, c_a, s_b, 2)
    c_b_recon, s_a_recon = ref_encode(R, hp, x_ba, 1)
    c_a_recon, s_b_recon = ref_encode(R, hp, x_ab, 2)
    x_aba = ref_decode(R, hp, c_a_recon, s_a, 1)
    x_bab = ref_decode(R, hp, c_b_recon, s_b, 2)
    L = {}
    L["loss_gen_recon_x_a"] = l1(x_a_recon, x_a)
    L["loss_gen_recon_x_b"] = l1(x_b_recon, x_b)
    L["loss_gen_recon_s_a"] = l1(s_a_recon, s_a)
    L["loss_gen_recon_s_b"] = l1(s_b_recon, s_b)
    L["loss_gen_recon_c_a"] = l1(c_a_recon, c_a)
    L["loss_gen_recon_c_b"] = l1(c_b_recon, c_b)
    if hp["recon_mask"] == 1:
        L["loss_gen_cycrecon_x_a"] = l1m(x_aba, x_a, m_a)
        L["loss_gen_cycrecon_x_b"] = l1m(x_bab, x_b, m_b)
    else:
        L["loss_gen_cycrecon_x_a"] = l1(x_aba, x_a)
        L["loss_gen_cycrecon_x_b"] = l1(x_bab, x_b)
    L["loss_gen_adv_a"] = R["dis_a"].calc_gen_loss(x_ba)
    L["loss_gen_adv_b"] = R["dis_b"].calc_gen_loss(x_ab)
    L["loss_gen_total"] = (
        hp["gan_w"] * L["loss_gen_adv_a"] + hp["gan_w"] * L["loss_gen_adv_b"]
        + hp["recon_x_w"] * L["loss_gen_recon_x_a"] + hp["recon_s_w"] * L["loss_gen_recon_s_a"]
        + hp["recon_c_w"] * L["loss_gen_recon_c_a"] + hp["recon_x_w"] * L["loss_gen_recon_x_b"]
        + hp["recon_s_w"] * L["loss_gen_recon_s_b"] + hp["recon_c_w"] * L["loss_gen_recon_c_b"]
        + hp["recon_x_cyc_w"] * L["loss_gen_cycrecon_x_a"] + hp["recon_x_cyc_w"] * L["loss_gen_cycrecon_x_b"])
    return L, dict(x_ba=x_ba, x_ab=x_ab, x_a_recon=x_a_recon, c_a=c_a, s_a=s_a)


def ref_dis_losses(R, hp, x_a, x_b):
    """trainer.py:1163-1184."""
    c_a, s_a = ref_encode(R, hp, x_a, 1)
    c_b, s_b = ref_encode(R, hp, x_b, 2)
    x_ba = ref_decode(R, hp, c_b, s_a, 1)
    x_ab = ref_decode(R, hp, c_a, s_b, 2)
    L = {}
    L["loss_dis_a"] = R["dis_a"].calc_dis_loss(x_ba.detach(), x_a)
    L["loss_dis_b"] = R["dis_b"].calc_dis_loss(x_ab.detach(), x_b)
    L["loss_dis_total"] = hp["gan_w"] * L["loss_dis_a"] + hp["gan_w"] * L["loss_dis_b"]
    return L


def gen_modules(R, hp):
    return [R["gen"]] if hp["gen_state"] == 1 else [R["gen_a"], R["gen_b"]]


def main():
    out = {}
    arrays = {}

    # ---------------- module-level forward fixtures (fp32 reference forward + fp64 truth) -------------
    S, B = 64, 2
    for gs in (1, 0):
        hp = O.default_hp(S, B, gs)
        x_a, x_b, m_a, m_b = O.synthetic_batch(B, S, seed=7)
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            R = build_ref(hp, dtype)
            with torch.no_grad():
                c, s = ref_encode(R, hp, x_a.to(dtype), 1)
                x_rec = ref_decode(R, hp, c, s, 1)
                c2, s2 = ref_encode(R, hp, x_b.to(dtype), 2)
                x_ab = ref_decode(R, hp, c, s2, 2)
                d_out = R["dis_a"](x_rec)
            key = "fwd_gs%d_%s" % (gs, tag)
            if tag == "f64":
                arrays[key + "_style"] = s.numpy().astype(np.float64)
                arrays[key + "_x_rec"] = x_rec.numpy().astype(np.float32)
                arrays[key + "_x_ab"] = x_ab.numpy().astype(np.float32)
                arrays[key + "_content_slice"] = c[:, ::16, ::2, ::2].numpy().astype(np.float32)
                for i, o in enumerate(d_out):
                    arrays[key + "_dis%d" % i] = o.numpy().astype(np.float64)
            out[key] = dict(content=digest(c), style=digest(s), x_rec=digest(x_rec), x_ab=digest(x_ab),
                            content2=digest(c2), dis=[digest(o) for o in d_out])
            print(key, out[key]["x_rec"][:3])

    # ---------------- norm-layer semantics (Appendix B of SURVEY.md) ----------------------------------
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 6, 5, generator=g, dtype=torch.float64) * 1.7 + 0.3
    ln = ref.LayerNorm(8).double()
    with torch.no_grad():
        ln.gamma.copy_(O.fill_det("ln.gamma", (8,), dtype=torch.float64))
        ln.beta.copy_(O.fill_det("ln.beta", (8,), dtype=torch.float64))
    ad = ref.AdaptiveInstanceNorm2d(8).double()
    w = O.fill_det("ad.w", (2, 8), 1.0, torch.float64)
    b_ = O.fill_det("ad.b", (2, 8), 1.0, torch.float64)
    ad.weight, ad.bias = w.reshape(-1), b_.reshape(-1)
    arrays["norm_x"] = x.numpy()
    arrays["norm_ln_y"] = ln(x).detach().numpy()
    arrays["norm_ln1_y"] = ln(x[:1]).detach().numpy()
    arrays["norm_adain_y"] = ad(x).detach().numpy()
    arrays["norm_in_y"] = torch.nn.InstanceNorm2d(8)(x).numpy()

    # ---------------- step-level fixtures: losses, grad digests, weights after 3 iterations -----------
    S, B = 64, 2
    for gs in (1, 0):
        hp = O.default_hp(S, B, gs)
        hp["step_size"] = 2  # exercise the StepLR decay inside 3 iterations
        x_a, x_b, m_a, m_b = O.synthetic_batch(B, S, seed=7)
        dtype = torch.float64
        R = build_ref(hp, dtype)
        xa, xb, ma, mb = (t.to(dtype) for t in (x_a, x_b, m_a, m_b))
        gen_params = [p for m in gen_modules(R, hp) for p in m.parameters()]
        dis_params = list(R["dis_a"].parameters()) + list(R["dis_b"].parameters())
        mk = lambda ps: torch.optim.Adam(ps, lr=hp["lr"], betas=(hp["beta1"], hp["beta2"]),
                                         weight_decay=hp["weight_decay"])
        gen_opt, dis_opt = mk(gen_params), mk(dis_params)
        sch = lambda o: torch.optim.lr_scheduler.StepLR(o, step_size=hp["step_size"], gamma=hp["gamma"])
        gen_s, dis_s = sch(gen_opt), sch(dis_opt)
        rec = dict(iters=[])
        import warnings
        warnings.filterwarnings("ignore")
        for it in range(3):
            dis_s.step(); gen_s.step()          # train.py:172 (scheduler stepped first)
            dis_opt.zero_grad()
            Ld = ref_dis_losses(R, hp, xa, xb)
            Ld["loss_dis_total"].backward()
            d_gr = [digest(p.grad) for p in dis_params] if it == 0 else None
            dis_opt.step()
            gen_opt.zero_grad()
            Lg, aux = ref_gen_losses(R, hp, xa, xb, ma, mb)
            Lg["loss_gen_total"].backward()
            g_gr = [digest(p.grad) for p in gen_params] if it == 0 else None
            gen_opt.step()
            entry = dict(losses={k: float(v) for k, v in list(Ld.items()) + list(Lg.items())},
                         lr=gen_opt.param_groups[0]["lr"])
            if it == 0:
                entry["dis_grad"] = d_gr
                entry["gen_grad"] = g_gr
                entry["x_ba"] = digest(aux["x_ba"])
            rec["iters"].append(entry)
            print("gs", gs, "iter", it, entry["losses"]["loss_gen_total"], entry["losses"]["loss_dis_total"], entry["lr"])
        rec["gen_after"] = [digest(p) for p in gen_params]
        rec["dis_after"] = [digest(p) for p in dis_params]
        out["step_gs%d" % gs] = rec

    # ---------------- init parity: RNG stream of construction + weights_init (utils.py:1093-1115) -----
    # The reference trainer constructs gen, dis_a, dis_b in this order (trainer.py:67-88), draws the
    # display noise (trainer.py:94-95), then applies kaiming to everything and gaussian to D.
    import torch.nn.init as init

    def weights_init(kind):
        def f(m):
            cn = m.__class__.__name__
            if (cn.find("Conv") == 0 or cn.find("Linear") == 0) and hasattr(m, "weight"):
                if kind == "gaussian":
                    init.normal_(m.weight.data, 0.0, 0.02)
                elif kind == "kaiming":
                    init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
                if hasattr(m, "bias") and m.bias is not None:
                    init.constant_(m.bias.data, 0.0)
        return f

    for gs in (1, 0):
        hp = O.default_hp(64, 1, gs)
        torch.manual_seed(1234)
        holder = torch.nn.Module()
        if gs == 1:
            holder.gen = ref.AdaINGen_double(3, hp["gen"])
        else:
            holder.gen_a = ref.AdaINGen(3, hp["gen"])
            holder.gen_b = ref.AdaINGen(3, hp["gen"])
        holder.dis_a = ref.MsImageDis(3, hp["dis"])
        holder.dis_b = ref.MsImageDis(3, hp["dis"])
        holder.instancenorm = torch.nn.InstanceNorm2d(512, affine=False)
        s_a = torch.randn(8, 16, 1, 1)
        s_b = torch.randn(8, 16, 1, 1)
        holder.apply(weights_init("kaiming"))
        holder.dis_a.apply(weights_init("gaussian"))
        holder.dis_b.apply(weights_init("gaussian"))
        out["init_gs%d" % gs] = dict(
            keys=[k for k, _ in holder.named_parameters()],
            state_keys=list(holder.state_dict().keys()),
            digests=[digest(p) for _, p in holder.named_parameters()],
            s_a=digest(s_a), s_b=digest(s_b))

    np.savez_compressed(os.path.join(HERE, "golden_arrays.npz"), **arrays)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.getsize(os.path.join(HERE, "golden_arrays.npz")), "bytes npz")


if __name__ == "__main__":
    main()
