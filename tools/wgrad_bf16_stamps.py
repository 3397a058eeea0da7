"""Diagnostic (GPU box): per-wave cycle accounting of the bf16-storage backward-weight kernel on the trunk layer at B = 32.
Needs the stamped build:  make -C munit_amd/csrc alt ALTNAME=wgbstamp ALTFLAGS=-DWGB_STAMP ; run with
MUNIT_HIP_LIB=$PWD/munit_amd/libmunit_hip_wgbstamp.so python tools/wgrad_bf16_stamps.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from munit_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(B, 256, 64, 64, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, 256, 64, 64, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
ops.set_compute("bf16s")
try:
    for _ in range(5):
        ops.conv2d_wgrad_raw(x, dy, (256, 256, 3, 3), 1, 1, "reflect", False, want_bias=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d_wgrad_raw(x, dy, (256, 256, 3, 3), 1, 1, "reflect", False, want_bias=False)
    e1.record()
    torch.cuda.synchronize()
    print("launch + reduce: %.1f us" % (e0.elapsed_time(e1) * 100))
finally:
    ops.set_compute("f32")
buf = (ctypes.c_longlong * 256)()
fn = lib.munit_debug_wgrad_bf16_stamps
fn.restype = ctypes.c_int
assert fn(buf) == 0
names = ["DMA issue", "fragments + MFMA issue", "vmcnt(0) wait", "barrier"]
for blk in (0, 5):
    print("logical block", blk, ": shader cycles per 64-pixel step")
    for wv in range(8):
        v = [buf[(blk * 8 + wv) * 4 + i] for i in range(4)]
        steps = max(1, round(B * 4096 / 28 / 64))
        print("  wave %d: " % wv + "  ".join("%s %6.0f" % (names[i], v[i] / steps) for i in range(4)) + "   total/step %6.0f" % (sum(v) / steps))
