"""Where does a replayed step start to differ from the eager one?  python tools/graph_diff.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from munit_amd.graph import GraphedStep
from munit_amd.trainer import MUNIT_Trainer
dev = torch.device("cuda:0")
size, batch = 64, 2
warm = tuple(t.to(dev) for t in bench.make_batch(batch, size, rank=9))
b0 = tuple(t.to(dev) for t in bench.make_batch(batch, size, rank=0))
def fresh():
    hp = bench.bench_hp(size, batch)
    torch.manual_seed(1234)
    tr = MUNIT_Trainer(hp); tr.to(dev)
    return tr, hp
def eager(tr, hp, b):
    tr.update_learning_rate(); tr.dis_update(b[0], b[1], hp); tr.gen_update(b[0], b[1], hp, b[2], b[3])
def snap(tr):
    torch.cuda.synchronize()
    return dict(gp=tr.gen_opt.flat_p.clone(), dp=tr.dis_opt.flat_p.clone(), gg=tr.gen_opt.flat_g.clone(), dg=tr.dis_opt.flat_g.clone(),
                gm=tr.gen_opt.flat_m.clone(), ld=tr.loss_dis_total.detach().clone(), lg=tr.loss_gen_total.detach().clone())
def diff(a, b, tag):
    print(tag, {k: (bool(torch.equal(a[k], b[k])), float((a[k].double() - b[k].double()).abs().max())) for k in a})
tr, hp = fresh()
for _ in range(2): eager(tr, hp, warm)
r2 = snap(tr)
tr2, hp2 = fresh()
g = GraphedStep(tr2, hp2, *warm, warmup=2)
diff(r2, snap(tr2), "after the 2 eager warm-up steps + capture:")
eager(tr, hp, b0); g(*b0)
diff(snap(tr), snap(tr2), "after step 3 (eager vs replay):")
eager(tr, hp, b0); g(*b0)
diff(snap(tr), snap(tr2), "after step 4:")
