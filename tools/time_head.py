"""Time the three passes of the 64 -> 3 image head (7x7, 256x256, B=8) and of the 3 -> 64 first layer (GPU box)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from munit_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cin, cout in ((64, 3), (3, 64)):
    x = torch.randn(8, cin, 256, 256, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 7, 7, generator=g) * 0.03).to(dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(8, cout, 256, 256, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    fl = 2 * 8 * 256 * 256 * cin * cout * 49
    for name, fn in (("fwd", lambda: ops.conv2d_fwd_raw(x, w, None, 1, 3, "reflect", False, "none")),
                     ("dgrad", lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, 1, 3, "reflect", False)),
                     ("wgrad", lambda: ops.conv2d_wgrad_raw(x, dy, w.shape, 1, 3, "reflect", False, want_bias=True))):
        us = timeit(fn)
        print("%d->%d %s %8.1f us  %6.1f TFLOP/s" % (cin, cout, name, us, fl / us / 1e6))
