"""Race soak of the data-parallel schedule (GPU box, one rank over RCCL with MUNIT_FORCE_ALLREDUCE): N steps with the staged
generator exchange + the discriminator exchange / optimizer step on the communication stream, against N steps with one in-line
all-reduce per update, from the same seed; every weight and Adam moment must agree bitwise.
    python tools/soak_exchange.py [steps] [size] [batch]"""
import os, sys, tempfile
os.environ["MUNIT_FORCE_ALLREDUCE"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from munit_amd import trainer as T
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="file://" + os.path.join(tempfile.mkdtemp(), "rdzv"), rank=0, world_size=1, device_id=dev)
x_a, x_b, m_a, m_b = (t.to(dev) for t in bench.make_batch(batch, size))
def run(overlap, guided):
    T.OVERLAP_EXCHANGE = overlap
    hp = bench.bench_hp(size, batch)
    hp["guided"] = guided
    torch.manual_seed(1234)
    tr = T.MUNIT_Trainer(hp); tr.to(dev)
    torch.manual_seed(3)
    for it in range(steps):
        tr.iterations = it
        tr.update_learning_rate(); tr.dis_update(x_a, x_b, hp); tr.gen_update(x_a, x_b, hp, m_a, m_b)
    torch.cuda.synchronize()
    out = [tr.gen_opt.flat_p.clone(), tr.gen_opt.flat_m.clone(), tr.gen_opt.flat_v.clone(), tr.dis_opt.flat_p.clone(),
           tr.dis_opt.flat_m.clone(), tr.dis_opt.flat_v.clone()]
    return out, float(tr.loss_gen_total.detach()), float(tr.loss_dis_total.detach()), (tr.last_exchange is not None, tr._dis_pending is not None)
rc = 0
for guided in (1, 0):
    ref, lg, ld, f0 = run(False, guided)
    got, lg2, ld2, f1 = run(True, guided)
    bad = sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, got))
    print("guided %d, steps %d size %d batch %d: loss_gen %.6f / %.6f  loss_dis %.6f / %.6f  mismatching buffers: %d of %d  (staged %s, serial %s)"
          % (guided, steps, size, batch, lg, lg2, ld, ld2, bad, len(ref), f1, f0))
    assert f1 == (True, True) and f0 == (False, False)
    rc |= 1 if bad or lg != lg2 or ld != ld2 else 0
dist.destroy_process_group()
sys.exit(rc)
