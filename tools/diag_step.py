"""Diagnostic (GPU box): per-parameter gradient error and per-loss error of one HIP step vs the fp64 oracle."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import munit_oracle as O
from tests.parity import oracle_states, nerr, load_into_trainer, trainer_named_params
from munit_amd.trainer import MUNIT_Trainer

gs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hp = O.default_hp(size, 2, gs)
gen, da, db = oracle_states(hp, torch.float64)
orc = O.OracleTrainer(hp, gen, da, db)
tr = MUNIT_Trainer(dict(hp)); load_into_trainer(tr, gen, da, db); tr.to("cuda:0")
x = O.synthetic_batch(2, size, seed=7)
dx = [t.cuda() for t in x]; ox = [t.double() for t in x]
gn, dn = trainer_named_params(tr)
tr.update_learning_rate(); orc.update_learning_rate()
tr.dis_update(dx[0], dx[1], hp); dref = orc.dis_update(ox[0], ox[1], apply=True)
errs = sorted(((nerr(p._munit_grad, g), n) for (n, p), g in zip(dn, dref)), reverse=True)
print("DIS grads worst:"); [print("  %.3e %s" % e) for e in errs[:8]]
tr.gen_update(dx[0], dx[1], hp, dx[2], dx[3]); gref = orc.gen_update(*ox, apply=False)
from tests.parity import l2err
errs = sorted(((l2err(p._munit_grad, g), n, float(g.abs().max())) for (n, p), g in zip(gn, gref) if g is not None and float(g.abs().max()) > 1e-7), reverse=True)
print("GEN grads worst:"); [print("  %.3e %s gmax=%.3e" % e) for e in errs[:40]]
print("GEN grads best:"); [print("  %.3e %s gmax=%.3e" % e) for e in errs[-10:]]
for k, v in orc.losses.items():
    print(k, float(v), float(getattr(tr, k)), abs(float(v) - float(getattr(tr, k))) / max(1, abs(float(v))))
print("---- anatomy of the worst generator gradients ----")
gd = {n: (p, g) for (n, p), g in zip(gn, gref) if g is not None}
for e, n, _ in errs[:3]:
    p, g = gd[n]
    a = p._munit_grad.detach().double().cpu(); r = g.double()
    d = (a - r).abs()
    print(n, "max", float(d.max()), "l2rel", float((a - r).norm() / r.norm()), "median", float(d.median()),
          "n>1e-3*gmax", int((d > 1e-3 * r.abs().max()).sum()), "of", d.numel())
    if d.dim() == 4:
        print("   per-cin max:", [round(float(v), 5) for v in d.amax(dim=(0, 2, 3)).topk(5).values], d.amax(dim=(0, 2, 3)).topk(5).indices.tolist())
        print("   per-cout max:", [round(float(v), 5) for v in d.amax(dim=(1, 2, 3)).topk(5).values], d.amax(dim=(1, 2, 3)).topk(5).indices.tolist())
