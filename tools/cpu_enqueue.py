"""How long does the host need to ENQUEUE one step (no device sync)?  If this approaches the device time per
step the Python launch path, not the GPU, is the limit.  python tools/cpu_enqueue.py"""
import os, sys, time, torch
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
enq, t_all = [], 0.0
for _ in range(6):
    # every measured step starts on an idle device: once the host is several hundred launches ahead the runtime makes it wait for
    # queue space, and a back-to-back loop then reads the DEVICE time per step instead of the host's
    torch.cuda.synchronize()
    a = time.perf_counter(); step(); enq.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    t_all += time.perf_counter() - a
print("host enqueue per step (each from an idle device): %s ms ; wall per step incl. device: %.1f ms" % (" ".join("%.1f" % (1e3 * e) for e in enq), 1e3 * t_all / 6))
