"""Forward time of the trunk layer only (GPU box): python tools/time_fwd.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from munit_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 256, 64, 64, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).contiguous(memory_format=torch.channels_last)
w._munit_prep = {}
fn = lambda: ops.conv2d_fwd_raw(x, w, None, 1, 1, "reflect", False, "none", owner=w)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): fn()
e1.record(); torch.cuda.synchronize()
print("%s fwd %.1f us" % (os.environ.get("MUNIT_HIP_LIB", "base").split("_")[-1], e0.elapsed_time(e1) / 30 * 1e3))
