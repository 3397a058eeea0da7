"""Micro-benchmark (GPU box): the resblock 3x3 256->256 @64^2 B=8 conv, forward / dgrad / wgrad, HIP events."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from munit_amd import ops
if len(sys.argv) > 1: ops.set_compute(sys.argv[1])   # "f32" (default) or "bf16"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 256, 64, 64, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).contiguous(memory_format=torch.channels_last)
dy = torch.randn(8, 256, 64, 64, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w._munit_prep = {}     # keep the prepared weight images with the tensor, as the trainer's parameters do
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2 * 8 * 64 * 64 * 256 * 2304
for name, fn in (("fwd", lambda: ops.conv2d_fwd_raw(x, w, None, 1, 1, "reflect", False, "none", owner=w)),
                 ("dgrad", lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, 1, 1, "reflect", False, owner=w)),
                 ("wgrad", lambda: ops.conv2d_wgrad_raw(x, dy, w.shape, 1, 1, "reflect", False, want_bias=False))):
    us = timeit(fn)
    print("%s %8.1f us  %6.1f TFLOP/s" % (name, us, fl / us / 1e6))
