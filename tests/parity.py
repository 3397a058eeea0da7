"""Shared parity harness (tests/, __graft_entry__.smoke(), bench.py's checker leg): runs the
HIP trainer and the CPU oracle on the same deterministic weights and inputs and reports
normalised errors.  The oracle is the CHECKER here, never the thing measured or shipped."""
import math

import numpy as np
import torch

from oracle import munit_oracle as O


def nerr(a, ref):
    """max|a - ref| / max(|ref|)  (SURVEY.md section 8c 'normalised max error')."""
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    den = float(ref.abs().max())
    if den == 0.0:
        return float(a.abs().max())
    return float((a - ref).abs().max()) / den


def oracle_states(hp, dtype):
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


def load_into_trainer(trainer, gen, dis_a, dis_b):
    """Copy the oracle's deterministic weights into the HIP trainer through load_state_dict
    (buffers such as AdaIN running_mean/var keep their defaults)."""
    def load(module, state):
        sd = module.state_dict()
        for k, v in state.items():
            assert k in sd, k
            sd[k] = v.detach().float()
        module.load_state_dict(sd)

    if trainer.gen_state == 1:
        load(trainer.gen, gen)
    else:
        load(trainer.gen_a, {k[2:]: v for k, v in gen.items() if k.startswith("a.")})
        load(trainer.gen_b, {k[2:]: v for k, v in gen.items() if k.startswith("b.")})
    load(trainer.dis_a, dis_a)
    load(trainer.dis_b, dis_b)


def trainer_named_params(trainer):
    """(name, parameter) in the oracle's ordering for both optimizers."""
    if trainer.gen_state == 1:
        g = [(k, p) for k, p in trainer.gen.named_parameters()]
    else:
        g = [("a." + k, p) for k, p in trainer.gen_a.named_parameters()] + \
            [("b." + k, p) for k, p in trainer.gen_b.named_parameters()]
    d = [("a." + k, p) for k, p in trainer.dis_a.named_parameters()] + \
        [("b." + k, p) for k, p in trainer.dis_b.named_parameters()]
    return g, d


def run_step_parity(size=64, batch=2, gen_state=1, iters=1, device="cuda:0", oracle_dtype=torch.float64,
                    step_size=2, check=True):
    """dis_update + gen_update pairs on the HIP trainer vs the oracle.  Returns a report dict;
    with check=True asserts the tolerances of SURVEY.md section 8c (vs the fp64 oracle:
    losses 1e-5 relative, gradients 1e-2 normalised; weights after Adam: see the comment at the end)."""
    from munit_amd.trainer import MUNIT_Trainer

    hp = O.default_hp(size, batch, gen_state)
    hp["step_size"] = step_size
    gen, dis_a, dis_b = oracle_states(hp, oracle_dtype)
    orc = O.OracleTrainer(hp, gen, dis_a, dis_b)
    tr = MUNIT_Trainer(dict(hp))
    load_into_trainer(tr, gen, dis_a, dis_b)
    tr.to(device)
    x_a, x_b, m_a, m_b = O.synthetic_batch(batch, size, seed=7)
    dx_a, dx_b, dm_a, dm_b = (t.to(device) for t in (x_a, x_b, m_a, m_b))
    ox = [t.to(oracle_dtype) for t in (x_a, x_b, m_a, m_b)]
    rep = {"loss_rel": 0.0, "grad_nerr": 0.0, "weight_nerr": 0.0}
    gnames, dnames = trainer_named_params(tr)
    null = set()  # parameters whose true gradient is identically zero (bias ahead of IN/AdaIN):
    # Adam turns their rounding noise into +-lr steps on both sides, so their values are not comparable
    for it in range(iters):
        tr.update_learning_rate()
        orc.update_learning_rate()
        tr.dis_update(dx_a, dx_b, hp)
        d_ref = orc.dis_update(ox[0], ox[1])
        if it == 0:
            for (n, p), g in zip(dnames, d_ref):
                e = nerr(p._munit_grad, g)
                rep["grad_nerr"] = max(rep["grad_nerr"], e)
                if check:
                    assert e <= 1e-2 or float(g.abs().max()) < 1e-9, ("dis grad", n, e)
        tr.gen_update(dx_a, dx_b, hp, dm_a, dm_b)
        g_ref = orc.gen_update(ox[0], ox[1], ox[2], ox[3])
        if it == 0:
            for (n, p), g in zip(gnames, g_ref):
                if g is None:
                    continue
                gmax = float(g.abs().max())
                e = nerr(p._munit_grad, g)
                # gradients that are mathematically zero (conv bias ahead of an instance norm)
                # hold only rounding noise on both sides
                if gmax < 1e-7:
                    assert float(p._munit_grad.abs().max()) < 1e-3, ("zero grad", n)
                    null.add(n)
                    continue
                rep["grad_nerr"] = max(rep["grad_nerr"], e)
                if check:
                    assert e <= 1e-2, ("gen grad", n, e, gmax)
        for k, v in orc.losses.items():
            mine = float(getattr(tr, k))
            rel = abs(mine - float(v)) / max(1.0, abs(float(v)))
            rep["loss_rel"] = max(rep["loss_rel"], rel)
            rep[k] = mine
            if check:
                assert rel <= 1e-5 * (1 if it == 0 else 50), (it, k, mine, float(v))
    # Weights after Adam.  Adam's first steps are sign-like (|step| ~ lr whatever |g| is), so an
    # element whose gradient is below fp32 noise may step the other way: bound = a few lr per
    # iteration in absolute terms, and a relative L2 error over each tensor for the bulk.
    rep["weight_abs"] = 0.0
    rep["weight_l2"] = 0.0
    for (n, p), q in list(zip(gnames, orc.opt["gen"]["params"])) + list(zip(dnames, orc.opt["dis"]["params"])):
        if n in null:
            continue
        a, r = p.detach().double().cpu(), q.detach().double()
        rep["weight_abs"] = max(rep["weight_abs"], float((a - r).abs().max()))
        rep["weight_l2"] = max(rep["weight_l2"], float((a - r).norm() / r.norm().clamp_min(1e-30)))
    rep["weight_nerr"] = rep["weight_abs"]
    if check:
        assert rep["weight_abs"] <= 4.0 * hp["lr"] * iters, rep["weight_abs"]
        assert rep["weight_l2"] <= 2e-3, rep["weight_l2"]
    return rep
