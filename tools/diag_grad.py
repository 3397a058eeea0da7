"""Diagnostic (GPU box): gradient w.r.t. the output of one conv (ahead of its instance norm) on the HIP path vs
the fp64 oracle, per channel."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import munit_oracle as O
from tests.parity import oracle_states, load_into_trainer
from munit_amd.trainer import MUNIT_Trainer

path = sys.argv[1] if len(sys.argv) > 1 else "enc1_content.model.3.model.3.model.0"
hp = O.default_hp(64, 2, 1)
gen, da, db = oracle_states(hp, torch.float64)
orc = O.OracleTrainer(hp, gen, da, db)
tr = MUNIT_Trainer(dict(hp)); load_into_trainer(tr, gen, da, db); tr.to("cuda:0")
x = O.synthetic_batch(2, 64, seed=7)
dx = [t.cuda() for t in x]; ox = [t.double() for t in x]

target_w = gen[path + ".conv.weight"]
ocap = []
orig = O.conv_block
def patched(x_, w, b, stride, pad, pad_type, norm_fn=None, activ="none"):
    if w is target_w and norm_fn is not None:
        def nf(t):
            if t.requires_grad:
                t.retain_grad(); ocap.append(t)
            return norm_fn(t)
        return orig(x_, w, b, stride, pad, pad_type, nf, activ)
    return orig(x_, w, b, stride, pad, pad_type, norm_fn, activ)
O.conv_block = patched

mod = tr.gen
for part in path.split("."):
    mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
gcap = []
def hook(m, i, o):
    rec = {"y": o.detach().clone()}
    if o.requires_grad:
        o.register_hook(lambda g, rec=rec: rec.__setitem__("g", g.detach().clone()))
    gcap.append(rec)
mod.conv.register_forward_hook(hook)

tr.update_learning_rate(); orc.update_learning_rate()
tr.dis_update(dx[0], dx[1], hp); orc.dis_update(ox[0], ox[1])
gcap.clear(); ocap.clear()
tr.gen_update(dx[0], dx[1], hp, dx[2], dx[3]); orc.gen_update(*ox, apply=False)
print("captures", len(gcap), len(ocap))
for k, (rec, t) in enumerate(zip(gcap, ocap)):
    y, g = rec["y"].double().cpu(), rec["g"].double().cpu()
    gr = t.grad
    ey = (y - t.detach()).abs().amax(dim=(0, 2, 3))
    eg = (g - gr)
    gscale = float(gr.abs().max())
    per_max = eg.abs().amax(dim=(0, 2, 3)) / gscale
    per_mean = eg.mean(dim=(2, 3)).abs().amax(dim=0) / gscale
    top = per_max.topk(4)
    print("call %d: fwd max err %.2e | grad scale %.3e | worst channels by max err:" % (k, float(ey.max()), gscale))
    for v, i in zip(top.values.tolist(), top.indices.tolist()):
        print("   ch %3d  max|dg|/gmax %.3e  |mean dg|/gmax %.3e   sum_p g_ref %.3e" % (i, v, float(per_mean[i]), float(gr[:, i].sum())))
    print("   median over channels of max|dg|/gmax %.3e" % float(per_max.median()))
# anatomy of the worst channel of call 0
rec, t = gcap[0], ocap[0]
g, gr, yr = rec["g"].double().cpu(), t.grad, t.detach()
eg = g - gr
ch = int((eg.abs().amax(dim=(0, 2, 3))).argmax())
for b in range(g.shape[0]):
    e = eg[b, ch].reshape(-1); r = gr[b, ch].reshape(-1); yy = yr[b, ch].reshape(-1)
    xh = (yy - yy.mean()) / yy.var(unbiased=False).add(1e-5).sqrt()
    slope = float((e * xh).sum() / (xh * xh).sum())
    resid = e - slope * xh
    print("b=%d ch=%d: |e|max %.3e  slope-on-xhat %.3e  resid max %.3e  |g_ref|max %.3e  n(|e|>0.1*max) %d  relu-active frac %.2f"
          % (b, ch, float(e.abs().max()), slope, float(resid.abs().max()), float(r.abs().max()),
             int((e.abs() > 0.1 * e.abs().max()).sum()), float((xh > 0).float().mean())))
    top = e.abs().topk(5)
    print("    top pixels", top.indices.tolist(), ["%.2e" % v for v in top.values.tolist()], " xhat there", ["%.3f" % float(xh[i]) for i in top.indices])
