"""Race soak (GPU box): N steps with the three-stream schedule vs the single-stream schedule from the same seed;
every weight must agree bitwise.  python tools/soak_streams.py [steps] [size] [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from munit_amd import ops
from munit_amd import trainer as T
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
x_a, x_b, m_a, m_b = (t.to(dev) for t in bench.make_batch(batch, size))
def run(streams):
    ops.SIDE_STREAM_WGRAD = T.BRANCH_STREAMS = streams
    hp = bench.bench_hp(size, batch)
    torch.manual_seed(1234)
    tr = T.MUNIT_Trainer(hp); tr.to(dev)
    torch.manual_seed(3)
    for it in range(steps):
        tr.iterations = it
        tr.update_learning_rate(); tr.dis_update(x_a, x_b, hp); tr.gen_update(x_a, x_b, hp, m_a, m_b)
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in tr.state_dict().items() if torch.is_tensor(v)}
    return sd, float(tr.loss_gen_total.detach()), float(tr.loss_dis_total.detach())
ref, lg, ld = run(False)
got, lg2, ld2 = run(True)
bad = [k for k in ref if not torch.equal(ref[k], got[k])]
print("steps %d size %d batch %d: loss_gen %.6f / %.6f  loss_dis %.6f / %.6f  mismatching tensors: %d of %d"
      % (steps, size, batch, lg, lg2, ld, ld2, len(bad), len(ref)))
sys.exit(1 if bad or lg != lg2 or ld != ld2 else 0)
