"""Diagnostic (GPU box): per-channel error of one resblock conv of the HIP path vs fp64, relative to the channel's
own spatial std (what the following instance norm divides by)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import munit_oracle as O
from tests.parity import oracle_states, load_into_trainer
from munit_amd.trainer import MUNIT_Trainer
import torch.nn.functional as F

hp = O.default_hp(64, 2, 1)
gen, da, db = oracle_states(hp, torch.float64)
tr = MUNIT_Trainer(dict(hp)); load_into_trainer(tr, gen, da, db); tr.to("cuda:0")
x = O.synthetic_batch(2, 64, seed=7)
mod = tr.gen.enc1_content.model[3].model[3].model[0].conv
cap = {}
mod.register_forward_hook(lambda m, i, o: cap.update(i=i[0].detach().clone(), o=o.detach().clone()))
with torch.no_grad():
    tr.gen.encode(x[0].cuda(), 1)
xi = cap["i"].double().cpu().contiguous(); yo = cap["o"].double().cpu()
w = mod.weight.detach().double().cpu().contiguous(); b = mod.bias.detach().double().cpu()
yr = F.conv2d(F.pad(xi, (1, 1, 1, 1), mode="reflect"), w, b)
err = (yo - yr).abs().amax(dim=(0, 2, 3))
std = yr.std(dim=(2, 3)).amin(dim=0)
mean = yr.mean(dim=(2, 3)).abs().amax(dim=0)
ratio = err / std
top = ratio.topk(6)
print("global max err", float(err.max()), "typ |y|", float(yr.abs().mean()))
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    print("ch %3d err/std %.3e  err %.3e std %.3e |mean| %.3e" % (i, v, float(err[i]), float(std[i]), float(mean[i])))
print("median err/std %.3e" % float(ratio.median()))
# same conv in fp32 on the CPU (what the oracle's fp32 run would give)
y32 = F.conv2d(F.pad(xi.float(), (1, 1, 1, 1), mode="reflect"), w.float(), b.float()).double()
err32 = (y32 - yr).abs().amax(dim=(0, 2, 3))
print("torch-CPU-fp32 conv: global max err", float(err32.max()), "median err/std %.3e" % float((err32 / std).median()))
