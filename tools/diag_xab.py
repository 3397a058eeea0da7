"""Diagnostic (GPU box): gradient w.r.t. the translated images x_ba / x_ab, HIP vs fp64 oracle (gen_state 0/1)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import munit_oracle as O
from tests.parity import oracle_states, load_into_trainer, nerr, l2err
from munit_amd.trainer import MUNIT_Trainer

gs = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hp = O.default_hp(64, 2, gs)
gen, da, db = oracle_states(hp, torch.float64)
orc = O.OracleTrainer(hp, gen, da, db)
tr = MUNIT_Trainer(dict(hp)); load_into_trainer(tr, gen, da, db); tr.to("cuda:0")
x = O.synthetic_batch(2, 64, seed=7)
dx = [t.cuda() for t in x]; ox = [t.double() for t in x]
tr.update_learning_rate(); orc.update_learning_rate()
tr.dis_update(dx[0], dx[1], hp); orc.dis_update(ox[0], ox[1])
with torch.no_grad():
    for (p, q) in zip(list(tr.dis_a.parameters()) + list(tr.dis_b.parameters()), orc.opt["dis"]["params"]):
        q.copy_(p.detach().double().cpu())
caps = []
orig_dec = tr._dec
def dec(c, s, k):
    y = orig_dec(c, s, k)
    rec = {"y": y}
    if y.requires_grad:
        y.register_hook(lambda g, rec=rec: rec.__setitem__("g", g.detach().clone()))
    caps.append(rec)
    return y
tr._dec = dec
tr.gen_update(dx[0], dx[1], hp, dx[2], dx[3])
L = orc.gen_losses(*ox)
names = ["x_a_recon", "x_b_recon", "x_ba", "x_ab"]
ref = torch.autograd.grad(L["loss_gen_total"], [orc._last[n] for n in names], retain_graph=True)
for i, n in enumerate(names):
    g = caps[i]["g"].double().cpu(); r = ref[i]
    e = (g - r)
    print(n, "fwd nerr %.2e" % nerr(caps[i]["y"], orc._last[n]), "grad nerr %.3e l2 %.3e" % (nerr(g, r), l2err(g, r)),
          "| err by row-band:", ["%.1e" % float(e[:, :, a:a + 8].abs().max()) for a in range(0, 64, 8)],
          "| by col-band:", ["%.1e" % float(e[:, :, :, a:a + 8].abs().max()) for a in range(0, 64, 8)], "| gmax %.2e" % float(r.abs().max()))
(ga, ka), (gb, kb) = orc._views()
with torch.no_grad():
    _, s_b_rec = gb.encode(orc._last["x_ab"], kb)
    _, s_a_rec = ga.encode(orc._last["x_ba"], ka)
    d_b = (s_b_rec - orc._last["s_b_prime"]).reshape(-1)
    d_a = (s_a_rec - orc._last["s_a_prime"]).reshape(-1)
    # HIP side
    _, hs_b_rec = tr._enc(caps[3]["y"], 2)
    _, hs_b = tr._enc(dx[1], 2)
    _, hs_a_rec = tr._enc(caps[2]["y"], 1)
    _, hs_a = tr._enc(dx[0], 1)
    hd_b = (hs_b_rec - hs_b).reshape(-1).double().cpu()
    hd_a = (hs_a_rec - hs_a).reshape(-1).double().cpu()
print("recon_s_b: min |s_recon - s'| oracle %.3e  HIP %.3e ; sign mismatches %d" % (float(d_b.abs().min()), float(hd_b.abs().min()), int((torch.sign(d_b) != torch.sign(hd_b)).sum())))
print("recon_s_a: min |s_recon - s'| oracle %.3e  HIP %.3e ; sign mismatches %d" % (float(d_a.abs().min()), float(hd_a.abs().min()), int((torch.sign(d_a) != torch.sign(hd_a)).sum())))
print("oracle d_b sorted |.| smallest:", ["%.2e" % v for v in d_b.abs().sort().values[:4].tolist()])
