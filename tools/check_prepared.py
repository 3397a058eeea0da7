import sys, os, torch, ctypes
sys.path.insert(0, ".")
import bench
from munit_amd import ops, _lib
from munit_amd.trainer import MUNIT_Trainer
dev = torch.device("cuda:0")
hp = bench.bench_hp(64, 2)
torch.manual_seed(1234)
tr = MUNIT_Trainer(hp); tr.to(dev)
x_a, x_b, m_a, m_b = (t.to(dev) for t in bench.make_batch(2, 64))
lib = _lib.load()
def check(tag):
    torch.cuda.synchronize()
    bad = 0; tot = 0
    for name, p in list(tr.gen.named_parameters()) + list(tr.dis_a.named_parameters()):
        reg = getattr(p, "_munit_prep", None)
        if not reg: continue
        for key, ent in reg.items():
            buf, ver, item = ent
            fresh = torch.empty_like(buf)
            it2 = _lib.PrepItem(item.w, fresh.data_ptr(), item.Cout, item.KH, item.KW, item.Cin, item.kind, item.ps, item.bf16)
            _lib.check(lib.munit_conv2d_prepare_weights(ctypes.byref(it2), ops._stream()), "prep")
            torch.cuda.synchronize()
            tot += 1
            if not torch.equal(buf, fresh):
                bad += 1
                print(tag, "STALE", name, key, float((buf.view(torch.float32) - fresh.view(torch.float32)).abs().max()))
    print(tag, "images", tot, "stale", bad)
for it in range(3):
    tr.update_learning_rate(); tr.dis_update(x_a, x_b, hp); tr.gen_update(x_a, x_b, hp, m_a, m_b)
    check("after step %d" % it)
