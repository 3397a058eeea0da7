"""Bisect tool for hipGraph capture of the step: python tools/graph_probe.py <what>  (what: fwd | dis | gen | both)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from munit_amd.trainer import MUNIT_Trainer
what = sys.argv[1]
dev = torch.device("cuda:0")
hp = bench.bench_hp(64, 2)
torch.manual_seed(1234)
tr = MUNIT_Trainer(hp); tr.to(dev)
x_a, x_b, m_a, m_b = (t.to(dev) for t in bench.make_batch(2, 64))
if os.environ.get("MUNIT_PROBE_NO_RECORD_STREAM"):
    torch.Tensor.record_stream = lambda self, s: None
from munit_amd.trainer import _Branches
def run():
    if what.startswith("v"):
        with torch.no_grad():
            br = _Branches(dev)
            y = z = None
            if what == "v0":                      # a torch stream context without the fork helper
                st = torch.cuda.Stream(device=dev) if not hasattr(run, "st") else run.st
                run.st = st
                from munit_amd import ops
                ops.stream_wait(st, torch.cuda.current_stream(dev))
                with torch.cuda.stream(st):
                    y = tr.gen.encode(x_a, 1)[0]
                ops.stream_wait(torch.cuda.current_stream(dev), st)
                return y
            if what in ("v2", "v3", "v4"):
                if what == "v4": br.share(x_a, x_b)
                y = br.run(0, lambda: tr.gen.encode(x_a, 1))
            if what in ("v3", "v4"):
                z = br.run(1, lambda: tr.gen.encode(x_b, 2))
            br.join(y, z)
            return y, z
    if what == "brfwd":      # the two encodes on the branch streams, no autograd
        with torch.no_grad():
            br = _Branches(dev)
            br.adopt(x_a, x_b)
            c_a, s_a = br.run(0, lambda: tr.gen.encode(x_a, 1))
            c_b, s_b = br.run(1, lambda: tr.gen.encode(x_b, 2))
            br.share(c_a, c_b, s_a, s_b)
            y = br.run(0, lambda: tr.gen.decode(c_b, s_a, 1))
            z = br.run(1, lambda: tr.gen.decode(c_a, s_b, 2))
            br.join(y, z)
            return y, z
    if what == "fwd":
        with torch.no_grad():
            c, s = tr.gen.encode(x_a, 1); return tr.gen.decode(c, s, 2)
    if what in ("dis", "both"): tr.dis_update(x_a, x_b, hp)
    if what in ("gen", "both"): tr.gen_update(x_a, x_b, hp, m_a, m_b)
for _ in range(2): run()
torch.cuda.synchronize()
for opt in (tr.dis_opt, tr.gen_opt):
    opt.dyn = torch.tensor([2e-4, 0.0316], device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()
print("captured", what, flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", what, float(tr.loss_dis_total) if what != "fwd" and what != "gen" else "", flush=True)
