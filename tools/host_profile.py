"""Where does the HOST spend the enqueue time of a step?  cProfile over a few steps (no device sync inside), top functions by
own time and by cumulative time.  python tools/host_profile.py [size batch]"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from munit_amd.trainer import MUNIT_Trainer
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
hp = bench.bench_hp(size, batch)
torch.manual_seed(1234)
tr = MUNIT_Trainer(hp); tr.to(dev)
x_a, x_b, m_a, m_b = (t.to(dev) for t in bench.make_batch(batch, size))
def step():
    tr.update_learning_rate(); tr.dis_update(x_a, x_b, hp); tr.gen_update(x_a, x_b, hp, m_a, m_b)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3): step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    print("==== by", key, "(3 steps)")
    pstats.Stats(pr).sort_stats(key).print_stats(45)
