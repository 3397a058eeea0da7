"""Micro-benchmark (GPU box) of chosen layers, one pass each, HIP events over back-to-back launches (weights' prepared images kept
with the tensor as the trainer's parameters keep them):

    python tools/time_layers.py [set]         set = s2 (4x4 / stride 2 layers of config_256 at B=8, default) | up | img | trunk | all
    MUNIT_HIP_LIB=.../libmunit_hip_old.so python tools/time_layers.py s2      # the same on another build (A/B in one session)

Prints us per launch and TFLOP/s on executed FLOPs (munit_conv2d_executed_flops) per (layer, pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from munit_amd import ops  # noqa: E402

# (B, H, W, Cin, Cout, k, stride, pad, upsample)
SETS = {
    "s2": [(8, 256, 256, 64, 128, 4, 2, 1, False), (8, 128, 128, 128, 256, 4, 2, 1, False), (8, 64, 64, 256, 256, 4, 2, 1, False),
           (16, 128, 128, 64, 128, 4, 2, 1, False), (16, 64, 64, 128, 256, 4, 2, 1, False), (16, 32, 32, 256, 512, 4, 2, 1, False)],
    # the smaller 4x4 / stride 2 layers: style encoder tail, discriminator scales 1 and 2 (fake + real batched: B = 16)
    "s2small": [(8, 32, 32, 256, 256, 4, 2, 1, False), (8, 16, 16, 256, 256, 4, 2, 1, False),
                (16, 64, 64, 64, 128, 4, 2, 1, False), (16, 32, 32, 128, 256, 4, 2, 1, False), (16, 16, 16, 256, 512, 4, 2, 1, False),
                (16, 32, 32, 64, 128, 4, 2, 1, False), (16, 16, 16, 128, 256, 4, 2, 1, False), (16, 8, 8, 256, 512, 4, 2, 1, False),
                (8, 128, 128, 64, 128, 4, 2, 1, False), (8, 64, 64, 128, 256, 4, 2, 1, False), (8, 32, 32, 256, 512, 4, 2, 1, False)],
    "up": [(8, 64, 64, 256, 128, 5, 1, 2, True), (8, 128, 128, 128, 64, 5, 1, 2, True)],
    "img": [(8, 256, 256, 3, 64, 7, 1, 3, False), (8, 256, 256, 64, 3, 7, 1, 3, False)],
    "trunk": [(8, 64, 64, 256, 256, 3, 1, 1, False)],
}
SETS["all"] = SETS["trunk"] + SETS["s2"] + SETS["up"] + SETS["img"]
which = sys.argv[1] if len(sys.argv) > 1 else "s2"
passes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fwd", "dgrad", "wgrad"]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
cl = torch.channels_last


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("lib:", os.environ.get("MUNIT_HIP_LIB", "munit_amd/libmunit_hip.so"))
for (B, H, W, ci, co, k, s, p, up) in SETS[which]:
    x = torch.randn(B, ci, H, W, generator=g).to(dev).contiguous(memory_format=cl)
    w = (torch.randn(co, ci, k, k, generator=g) * 0.03).to(dev).contiguous(memory_format=cl)
    w._munit_prep = {}
    y = ops.conv2d_fwd_raw(x, w, None, s, p, "reflect", up, "none", owner=w)
    dy = torch.randn(y.shape, generator=g).to(dev).contiguous(memory_format=cl)
    pl = ops._plan(B, H, W, ci, co, k, k, s, p, "reflect", up, "none", 0.2, 0, 0)
    dw, db = torch.empty_like(w), torch.empty(co, device=dev)
    fns = {"fwd": (0, lambda: ops.conv2d_fwd_raw(x, w, None, s, p, "reflect", up, "none", owner=w)),
           "dgrad": (1, lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, s, p, "reflect", up, owner=w)),
           "wgrad": (2, lambda: ops.conv2d_wgrad_raw(x, dy, w.shape, s, p, "reflect", up, dw=dw, db=db))}
    for name in passes:
        idx, fn = fns[name]
        us = timeit(fn)
        print("B%d %dx%d %d->%d k%d s%d%s  %-5s %8.1f us  %6.1f TFLOP/s executed  (%s)" % (
            B, H, W, ci, co, k, s, " up2" if up else "", name, us, pl.flop_exec[idx] / us / 1e6, pl.kname[idx]))
