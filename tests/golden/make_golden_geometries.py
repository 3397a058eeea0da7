"""Third fixture set from the REFERENCE network modules (build container only; needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_geometries.py

golden_geometries.json -- for EVERY network geometry the GPU step-parity tests run besides config_256.yaml's
(tests/geometries.py: other depths / widths / paddings / activations / norms, non-square crops, two generators, loss terms
at weight 0, one-channel domains) one dis_update + gen_update in float64 over the reference's own modules
(scripts/networks.py AdaINGen / AdaINGen_double :170-388, MsImageDis :20-115, built from the geometry's config exactly as
trainer.py:67-88 builds them) driven through the update sequence of trainer.py:365-561 / :1145-1186 with torch.optim.Adam:
forward digests (content, style, x_ba, x_ab, the discriminator maps), every loss_* scalar, per-tensor gradient digests, weights
after the step.  tests/test_oracle_golden.py::test_step_geometries_match_reference_sequence holds the fp64 oracle to 1e-9 on
them, so an oracle that hard-codes what the reference takes from the config (round 3: the MLP's activation) cannot pass.

Data only: names, config overrides, digests.
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
from make_golden import digest, load_ref  # noqa: E402
from tests.geometries import ALL, BATCH, merged_hp  # noqa: E402

torch.set_num_threads(8)
warnings.filterwarnings("ignore")


def build(hp, dtype):
    """Reference modules for `hp` (trainer.py:67-88), holding the oracle's deterministic weights (names / shapes / order are
    asserted equal to the oracle's own parameter table, i.e. to what the HIP trainer's state_dict must hold)."""
    nin = hp["input_dim_a"]
    R = {}
    if hp["gen_state"] == 1:
        shapes = O.gen_param_shapes(hp["gen"], nin, True)
        m = ref.AdaINGen_double(nin, hp["gen"])
        assert {k: tuple(v.shape) for k, v in m.named_parameters()} == shapes
        assert [k for k, _ in m.named_parameters()] == list(shapes.keys())
        R["gens"] = [load_ref(m.to(dtype), O.make_state(shapes, "gen.", dtype))]
        R["enc"] = lambda x, k: R["gens"][0].encode(x, k)
        R["dec"] = lambda c, s, k: R["gens"][0].decode(c, s, k)
    else:
        shapes = O.gen_param_shapes(hp["gen"], nin, False)
        R["gens"] = []
        for tag in ("a", "b"):
            m = ref.AdaINGen(nin, hp["gen"])
            assert {k: tuple(v.shape) for k, v in m.named_parameters()} == shapes
            R["gens"].append(load_ref(m.to(dtype), O.make_state(shapes, "gen_%s." % tag, dtype)))
        R["enc"] = lambda x, k: R["gens"][k - 1].encode(x)
        R["dec"] = lambda c, s, k: R["gens"][k - 1].decode(c, s)
    dshapes = O.dis_param_shapes(hp["dis"], nin)
    for tag in ("a", "b"):
        m = ref.MsImageDis(nin, hp["dis"])
        assert {k: tuple(v.shape) for k, v in m.named_parameters()} == dshapes
        R["dis_" + tag] = load_ref(m.to(dtype), O.make_state(dshapes, "dis_%s." % tag, dtype))
    return R


def l1(a, b):
    return torch.mean(torch.abs(a - b))                      # trainer.py:290


def l1m(a, b, m):
    return torch.mean(torch.abs(torch.mul((a - b), 1 - m)))  # trainer.py:305


def run(name, size, over):
    hp = merged_hp(O.default_hp, size, over)
    dtype = torch.float64
    nin = hp["input_dim_a"]
    x_a, x_b, m_a, m_b = (t.to(dtype) for t in O.synthetic_batch(BATCH, size, seed=7))
    x_a, x_b = x_a[:, :nin].contiguous(), x_b[:, :nin].contiguous()
    R = build(hp, dtype)
    gen_params = [p for m in R["gens"] for p in m.parameters()]
    dis_params = list(R["dis_a"].parameters()) + list(R["dis_b"].parameters())
    mk = lambda ps: torch.optim.Adam(ps, lr=hp["lr"], betas=(hp["beta1"], hp["beta2"]), weight_decay=hp["weight_decay"])
    gen_opt, dis_opt = mk(gen_params), mk(dis_params)
    enc, dec = R["enc"], R["dec"]

    # dis_update, trainer.py:1145-1186 (guided == 1)
    dis_opt.zero_grad()
    c_a, s_a = enc(x_a, 1)
    c_b, s_b = enc(x_b, 2)
    x_ba, x_ab = dec(c_b, s_a, 1), dec(c_a, s_b, 2)
    L = {}
    L["loss_dis_a"] = R["dis_a"].calc_dis_loss(x_ba.detach(), x_a)
    L["loss_dis_b"] = R["dis_b"].calc_dis_loss(x_ab.detach(), x_b)
    L["loss_dis_total"] = hp["gan_w"] * L["loss_dis_a"] + hp["gan_w"] * L["loss_dis_b"]
    L["loss_dis_total"].backward()
    d_gr = [digest(p.grad) for p in dis_params]
    with torch.no_grad():
        fwd = dict(content=digest(c_a), style=digest(s_b), x_ba=digest(x_ba), x_ab=digest(x_ab),
                   dis=[digest(o) for o in R["dis_a"](x_ba)])
    dis_opt.step()

    # gen_update, trainer.py:365-561 (guided == 1; the cycle decodes and their terms are skipped and read 0 when recon_x_cyc_w is 0, trainer.py:388-398, 466-487)
    gen_opt.zero_grad()
    c_a, s_a = enc(x_a, 1)
    c_b, s_b = enc(x_b, 2)
    x_a_recon, x_b_recon = dec(c_a, s_a, 1), dec(c_b, s_b, 2)
    x_ba, x_ab = dec(c_b, s_a, 1), dec(c_a, s_b, 2)
    c_b_recon, s_a_recon = enc(x_ba, 1)
    c_a_recon, s_b_recon = enc(x_ab, 2)
    G = {}
    G["loss_gen_recon_x_a"] = l1(x_a_recon, x_a)
    G["loss_gen_recon_x_b"] = l1(x_b_recon, x_b)
    G["loss_gen_recon_s_a"] = l1(s_a_recon, s_a)        # always evaluated (trainer.py:435-449); weight 0 only removes
    G["loss_gen_recon_s_b"] = l1(s_b_recon, s_b)        # them from loss_gen_total
    G["loss_gen_recon_c_a"] = l1(c_a_recon, c_a)
    G["loss_gen_recon_c_b"] = l1(c_b_recon, c_b)
    if hp["recon_x_cyc_w"] > 0:
        x_aba, x_bab = dec(c_a_recon, s_a, 1), dec(c_b_recon, s_b, 2)
        G["loss_gen_cycrecon_x_a"] = l1m(x_aba, x_a, m_a)
        G["loss_gen_cycrecon_x_b"] = l1m(x_bab, x_b, m_b)
    else:
        G["loss_gen_cycrecon_x_a"] = G["loss_gen_cycrecon_x_b"] = 0
    G["loss_gen_adv_a"] = R["dis_a"].calc_gen_loss(x_ba)
    G["loss_gen_adv_b"] = R["dis_b"].calc_gen_loss(x_ab)
    G["loss_gen_total"] = (
        hp["gan_w"] * G["loss_gen_adv_a"] + hp["gan_w"] * G["loss_gen_adv_b"]
        + hp["recon_x_w"] * G["loss_gen_recon_x_a"] + hp["recon_s_w"] * G["loss_gen_recon_s_a"]
        + hp["recon_c_w"] * G["loss_gen_recon_c_a"] + hp["recon_x_w"] * G["loss_gen_recon_x_b"]
        + hp["recon_s_w"] * G["loss_gen_recon_s_b"] + hp["recon_c_w"] * G["loss_gen_recon_c_b"]
        + hp["recon_x_cyc_w"] * G["loss_gen_cycrecon_x_a"] + hp["recon_x_cyc_w"] * G["loss_gen_cycrecon_x_b"])
    G["loss_gen_total"].backward()
    g_gr = [digest(p.grad) if p.grad is not None else None for p in gen_params]
    gen_opt.step()
    L.update(G)
    return dict(size=size if isinstance(size, int) else list(size), batch=BATCH, over=over,
                losses={k: float(v) for k, v in L.items()}, forward=fwd, dis_grad=d_gr, gen_grad=g_gr,
                gen_after=[digest(p) for p in gen_params], dis_after=[digest(p) for p in dis_params])


def main():
    out = {}
    for name, size, over in ALL:
        out[name] = run(name, size, over)
        print(name, out[name]["losses"]["loss_gen_total"], out[name]["losses"]["loss_dis_total"])
    with open(os.path.join(HERE, "golden_geometries.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.getsize(os.path.join(HERE, "golden_geometries.json")), "bytes")


if __name__ == "__main__":
    main()
