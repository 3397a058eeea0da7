"""Diagnostic (GPU box): per-wave cycle accounting of the Winograd forward kernel's main loop on the trunk layer.
Needs the stamped build:  make -C munit_amd/csrc alt ALTNAME=stamp ALTFLAGS=-DWINO_STAMP ; run with
MUNIT_HIP_LIB=$PWD/munit_amd/libmunit_hip_stamp.so python tools/wino_stamps.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from munit_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 256, 64, 64, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).contiguous(memory_format=torch.channels_last)
w._munit_prep = {}
for _ in range(5):
    ops.conv2d_fwd_raw(x, w, None, 1, 1, "reflect", False, "none", owner=w)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
fn = lib.munit_debug_wino_stamps
fn.restype = ctypes.c_int
assert fn(buf) == 0
names = ["pre-MFMA (loads, transform, V stores)", "compute() issue", "barrier wait", "prologue"]
nc = 32
for blk in (0, 3):
    print("block", blk, "cycles per chunk (s_memtime ticks), 32 chunks")
    for wv in range(8):
        v = [buf[(blk * 8 + wv) * 4 + i] for i in range(4)]
        tot = sum(v[:3])
        print("  wave %d (%s): " % (wv, "loader" if wv < 4 else "multiplier") +
              "  ".join("%s %6.0f" % (names[i].split(" (")[0], v[i] / nc) for i in range(3)) + "   total/chunk %6.0f   prologue %d" % (tot / nc, v[3]))
