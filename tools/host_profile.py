"""Where does the HOST time of one step go?  cProfile over 3 steps on the GPU box (python tools/host_profile.py)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from munit_amd.trainer import MUNIT_Trainer
dev = torch.device("cuda:0")
hp = bench.bench_hp(256, 8)
torch.manual_seed(1234)
tr = MUNIT_Trainer(hp); tr.to(dev)
x_a, x_b, m_a, m_b = (t.to(dev) for t in bench.make_batch(8, 256))
def step():
    tr.update_learning_rate(); tr.dis_update(x_a, x_b, hp); tr.gen_update(x_a, x_b, hp, m_a, m_b)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(30)
