"""Micro-benchmark (GPU box): forward 3x3 conv Cin -> 256 @64^2 B=8 for several Cin: fixed cost vs per-K-tile slope."""
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
prev = None
for cin in (32, 64, 128, 256, 512, 1024):
    x = torch.randn(8, cin, 64, 64, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, cin, 3, 3, generator=g) * 0.03).to(dev).contiguous(memory_format=torch.channels_last)
    w._munit_prep = {}   # prepared weight image kept with the tensor (Winograd U / none for the direct kernel)
    us = timeit(lambda: ops.conv2d_fwd_raw(x, w, None, 1, 1, "reflect", False, "none", owner=w))
    fl = 2 * 8 * 64 * 64 * 256 * 9 * cin
    kt = 9 * cin // 32
    extra = "" if prev is None else "  slope %.2f us/K-tile -> %.1f TFLOP/s in-loop" % ((us - prev[0]) / (kt - prev[1]), 2 * 8 * 64 * 64 * 256 * 32 / ((us - prev[0]) / (kt - prev[1])) / 1e6)
    print("Cin %4d  K-tiles %4d  %8.1f us  %6.1f TFLOP/s%s" % (cin, kt, us, fl / us / 1e6, extra))
    prev = (us, kt)
