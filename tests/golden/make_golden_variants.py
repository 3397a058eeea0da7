"""Second fixture set from the REFERENCE network modules (build container only; needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_variants.py

1. golden_variants.json -- the update sequence of scripts/trainer.py with the branches make_golden.py does not drive:
     * `guided: 0`  -- translation with the sampled styles s_a, s_b (trainer.py:377-379 / 405-407 in gen_update,
       :1155-1157 / :1167-1169 in dis_update) and the style-reconstruction target s_a / s_b (trainer.py:438-440);
     * `recon_mask: 0` -- unmasked cycle reconstruction (trainer.py:476-487).
   One dis_update + gen_update each in float64 over reference modules + torch.optim.Adam: losses, gradient digests, weight
   digests after the step.  The sampled styles are drawn from the host RNG in the reference's order (two draws at the top of
   each update, trainer.py:366-367, 1146-1147) after torch.manual_seed(seed); the seeds are stored.

2. ckpt_ref/ -- a checkpoint directory WRITTEN BY THE REFERENCE's objects in the reference's format (trainer.py:1387-1429):
   gen_00000003.pt = {"2": AdaINGen_double.state_dict()}, dis_00000003.pt = {"a": ..., "b": MsImageDis.state_dict()},
   optimizer.pt = {"gen": torch.optim.Adam.state_dict(), "dis": ...} after two real Adam steps (so the moments and step
   counters are populated), for a SMALL geometry (gen dim 8, 1 resblock; dis dim 8, 2 layers, 2 scales: 120 kB in total), and
   ckpt_ref/expect.json: digests of a forward pass of the reference modules holding those weights.  The files are plain tensor
   containers: tests load them with torch.load(weights_only=True) through MUNIT_Trainer.resume.
"""
import json
import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/scripts")

import networks as ref  # noqa: E402  (reference, read-only)
from oracle import munit_oracle as O  # noqa: E402
from make_golden import build_ref, digest, gen_modules, ref_decode, ref_encode  # noqa: E402

torch.set_num_threads(8)
warnings.filterwarnings("ignore")


def gen_losses(R, hp, x_a, x_b, m_a, m_b, s_a, s_b):
    """Reference modules driven per trainer.py:366-558, both `guided` branches and both `recon_mask` branches."""
    l1 = lambda a, b: torch.mean(torch.abs(a - b))
    l1m = lambda a, b, m: torch.mean(torch.abs(torch.mul((a - b), 1 - m)))
    c_a, s_a_prime = ref_encode(R, hp, x_a, 1)
    c_b, s_b_prime = ref_encode(R, hp, x_b, 2)
    x_a_recon = ref_decode(R, hp, c_a, s_a_prime, 1)
    x_b_recon = ref_decode(R, hp, c_b, s_b_prime, 2)
    if hp["guided"] == 0:
        x_ba = ref_decode(R, hp, c_b, s_a, 1)
        x_ab = ref_decode(R, hp, c_a, s_b, 2)
    else:
        x_ba = ref_decode(R, hp, c_b, s_a_prime, 1)
        x_ab = ref_decode(R, hp, c_a, s_b_prime, 2)
    c_b_recon, s_a_recon = ref_encode(R, hp, x_ba, 1)
    c_a_recon, s_b_recon = ref_encode(R, hp, x_ab, 2)
    x_aba = ref_decode(R, hp, c_a_recon, s_a_prime, 1)
    x_bab = ref_decode(R, hp, c_b_recon, s_b_prime, 2)
    L = {}
    L["loss_gen_recon_x_a"] = l1(x_a_recon, x_a)
    L["loss_gen_recon_x_b"] = l1(x_b_recon, x_b)
    if hp["guided"] == 0:
        L["loss_gen_recon_s_a"] = l1(s_a_recon, s_a)
        L["loss_gen_recon_s_b"] = l1(s_b_recon, s_b)
    else:
        L["loss_gen_recon_s_a"] = l1(s_a_recon, s_a_prime)
        L["loss_gen_recon_s_b"] = l1(s_b_recon, s_b_prime)
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
    return L, x_ba


def dis_losses(R, hp, x_a, x_b, s_a, s_b):
    """trainer.py:1146-1184."""
    c_a, s_a_prime = ref_encode(R, hp, x_a, 1)
    c_b, s_b_prime = ref_encode(R, hp, x_b, 2)
    if hp["guided"] == 0:
        x_ba = ref_decode(R, hp, c_b, s_a, 1)
        x_ab = ref_decode(R, hp, c_a, s_b, 2)
    else:
        x_ba = ref_decode(R, hp, c_b, s_a_prime, 1)
        x_ab = ref_decode(R, hp, c_a, s_b_prime, 2)
    L = {}
    L["loss_dis_a"] = R["dis_a"].calc_dis_loss(x_ba.detach(), x_a)
    L["loss_dis_b"] = R["dis_b"].calc_dis_loss(x_ab.detach(), x_b)
    L["loss_dis_total"] = hp["gan_w"] * L["loss_dis_a"] + hp["gan_w"] * L["loss_dis_b"]
    return L


def draw_styles(seed, batch, style_dim, dtype):
    torch.manual_seed(seed)
    return torch.randn(batch, style_dim, 1, 1).to(dtype), torch.randn(batch, style_dim, 1, 1).to(dtype)


def step_variant(gs, guided, recon_mask, S=64, B=2, seeds=(100, 200)):
    hp = O.default_hp(S, B, gs)
    hp["guided"], hp["recon_mask"] = guided, recon_mask
    dtype = torch.float64
    x_a, x_b, m_a, m_b = (t.to(dtype) for t in O.synthetic_batch(B, S, seed=7))
    R = build_ref(hp, dtype)
    gen_params = [p for m in gen_modules(R, hp) for p in m.parameters()]
    dis_params = list(R["dis_a"].parameters()) + list(R["dis_b"].parameters())
    mk = lambda ps: torch.optim.Adam(ps, lr=hp["lr"], betas=(hp["beta1"], hp["beta2"]), weight_decay=hp["weight_decay"])
    gen_opt, dis_opt = mk(gen_params), mk(dis_params)
    sd = hp["gen"]["style_dim"]
    dis_opt.zero_grad()
    s_a, s_b = draw_styles(seeds[0], B, sd, dtype)
    Ld = dis_losses(R, hp, x_a, x_b, s_a, s_b)
    Ld["loss_dis_total"].backward()
    d_gr = [digest(p.grad) for p in dis_params]
    dis_opt.step()
    gen_opt.zero_grad()
    s_a, s_b = draw_styles(seeds[1], B, sd, dtype)
    Lg, x_ba = gen_losses(R, hp, x_a, x_b, m_a, m_b, s_a, s_b)
    Lg["loss_gen_total"].backward()
    g_gr = [digest(p.grad) if p.grad is not None else None for p in gen_params]
    gen_opt.step()
    return dict(gen_state=gs, guided=guided, recon_mask=recon_mask, size=S, batch=B, style_seeds=list(seeds),
                losses={k: float(v) for k, v in list(Ld.items()) + list(Lg.items())},
                dis_grad=d_gr, gen_grad=g_gr, x_ba=digest(x_ba),
                gen_after=[digest(p) for p in gen_params], dis_after=[digest(p) for p in dis_params])


SMALL_GEN = dict(dim=8, mlp_dim=16, style_dim=8, activ="relu", n_downsample=2, n_res=1, pad_type="reflect")
SMALL_DIS = dict(dim=8, norm="none", activ="lrelu", n_layer=2, gan_type="lsgan", num_scales=2, pad_type="reflect")


def write_checkpoint(out_dir):
    """A checkpoint the way the reference's MUNIT_Trainer.save writes it (trainer.py:1387-1429), from reference modules."""
    os.makedirs(out_dir, exist_ok=True)
    torch.manual_seed(4321)
    gen = ref.AdaINGen_double(3, SMALL_GEN)
    dis_a, dis_b = ref.MsImageDis(3, SMALL_DIS), ref.MsImageDis(3, SMALL_DIS)
    gen_params = list(gen.parameters())
    dis_params = list(dis_a.parameters()) + list(dis_b.parameters())
    mk = lambda ps: torch.optim.Adam([p for p in ps if p.requires_grad], lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-4)
    gen_opt, dis_opt = mk(gen_params), mk(dis_params)
    g = torch.Generator().manual_seed(11)
    x_a = 2 * torch.rand(2, 3, 32, 32, generator=g) - 1
    x_b = 2 * torch.rand(2, 3, 32, 32, generator=g) - 1
    for _ in range(2):                       # two real updates: exp_avg / exp_avg_sq / step are populated
        dis_opt.zero_grad()
        c_a, s_a = gen.encode(x_a, 1)
        c_b, s_b = gen.encode(x_b, 2)
        x_ba, x_ab = gen.decode(c_b, s_a, 1), gen.decode(c_a, s_b, 2)
        (dis_a.calc_dis_loss(x_ba.detach(), x_a) + dis_b.calc_dis_loss(x_ab.detach(), x_b)).backward()
        dis_opt.step()
        gen_opt.zero_grad()
        c_a, s_a = gen.encode(x_a, 1)
        c_b, s_b = gen.encode(x_b, 2)
        x_ba, x_ab = gen.decode(c_b, s_a, 1), gen.decode(c_a, s_b, 2)
        x_rec = gen.decode(c_a, s_a, 1)
        (dis_a.calc_gen_loss(x_ba) + dis_b.calc_gen_loss(x_ab) + torch.mean(torch.abs(x_rec - x_a))).backward()
        gen_opt.step()
    iterations = 2                            # save(snapshot_dir, iterations) names the files iterations + 1
    torch.save({"2": gen.state_dict()}, os.path.join(out_dir, "gen_%08d.pt" % (iterations + 1)))
    torch.save({"a": dis_a.state_dict(), "b": dis_b.state_dict()}, os.path.join(out_dir, "dis_%08d.pt" % (iterations + 1)))
    torch.save({"gen": gen_opt.state_dict(), "dis": dis_opt.state_dict()}, os.path.join(out_dir, "optimizer.pt"))
    with torch.no_grad():
        c_a, s_a = gen.encode(x_a, 1)
        c_b, s_b = gen.encode(x_b, 2)
        x_ab = gen.decode(c_a, s_b, 2)
        d = dis_a(x_ab)
    st = gen_opt.state_dict()["state"]
    expect = dict(gen_cfg=SMALL_GEN, dis_cfg=SMALL_DIS, input_seed=11, input_shape=[2, 3, 32, 32], iterations=iterations + 1,
                  content=digest(c_a), style=digest(s_b), x_ab=digest(x_ab), dis=[digest(o) for o in d],
                  gen_keys=list(gen.state_dict().keys()), dis_keys=list(dis_a.state_dict().keys()),
                  gen_opt_step=float(st[0]["step"]), gen_exp_avg0=digest(st[0]["exp_avg"]),
                  gen_exp_avg_sq_last=digest(st[len(st) - 1]["exp_avg_sq"]),
                  gen_params=[digest(p) for p in gen.parameters()])
    with open(os.path.join(out_dir, "expect.json"), "w") as f:
        json.dump(expect, f)
    print("checkpoint:", {n: os.path.getsize(os.path.join(out_dir, n)) for n in sorted(os.listdir(out_dir))})


def main():
    out = {}
    for name, (gs, guided, recon_mask) in {"gs1_guided0": (1, 0, 1), "gs0_guided0": (0, 0, 1),
                                           "gs1_nomask": (1, 1, 0)}.items():
        out[name] = step_variant(gs, guided, recon_mask)
        print(name, out[name]["losses"]["loss_gen_total"], out[name]["losses"]["loss_dis_total"])
    with open(os.path.join(HERE, "golden_variants.json"), "w") as f:
        json.dump(out, f)
    write_checkpoint(os.path.join(HERE, "ckpt_ref"))


if __name__ == "__main__":
    main()
