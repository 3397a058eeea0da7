"""Per-layer efficiency table of the step's convolutions (GPU box): python tools/conv_table.py [size batch]

One step is run with a hook on ops._count that collects every (layer geometry, pass) and how often the step calls it;
each distinct one is then timed alone (HIP events, 20 launches, everything warm) and the table lists, sorted by the time
the step LOSES to it: launches per step, us per launch, executed TFLOP/s, fraction of the fp32 MFMA peak, and
lost ms = n * (t - flops / (REF_FRAC * peak)) with REF_FRAC = what the best kernel (the resblock forward) reaches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from munit_amd import ops, _lib
from munit_amd.trainer import MUNIT_Trainer

PEAK, REF_FRAC = 157.3e12, 0.80
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
hp = bench.bench_hp(size, batch)
torch.manual_seed(1234)
tr = MUNIT_Trainer(hp); tr.to(dev)
x_a, x_b, m_a, m_b = (t.to(dev) for t in bench.make_batch(batch, size))
def step():
    tr.update_learning_rate(); tr.dis_update(x_a, x_b, hp); tr.gen_update(x_a, x_b, hp, m_a, m_b)
for _ in range(2): step()
torch.cuda.synchronize()
calls = {}
orig = ops._count
def hook(pl, which):
    k = (id(pl), which)
    calls[k] = (pl, which, calls.get(k, (0, 0, 0))[2] + 1)
ops._count = hook
step()
torch.cuda.synchronize()
ops._count = orig
del tr
torch.cuda.empty_cache()

PADN = {v: k for k, v in _lib.PAD.items()}
DT = (torch.float32, torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

rows = []
g = torch.Generator().manual_seed(1)
for pl, which, n in calls.values():
    d = pl.d
    cl = torch.channels_last
    x = torch.randn(d.B, d.Cin, d.H, d.W, generator=g).to(dev).contiguous(memory_format=cl).to(DT[d.in_dtype])
    w = (torch.randn(d.Cout, d.Cin, d.KH, d.KW, generator=g) * 0.03).to(dev).contiguous(memory_format=cl)
    dy = torch.randn(d.B, d.Cout, pl.ho, pl.wo, generator=g).to(dev).contiguous(memory_format=cl).to(DT[d.out_dtype])
    pt, up = PADN[d.pad_mode], bool(d.upsample)
    if which == 0:
        fn = lambda: ops.conv2d_fwd_raw(x, w, None, d.stride, d.pad, pt, up, "none", out_dtype=DT[d.out_dtype])
    elif which == 1:
        fn = lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, d.stride, d.pad, pt, up, x_dtype=DT[d.in_dtype])
    else:
        dw = torch.empty_like(w); db = torch.empty(d.Cout, device=dev)
        fn = lambda: ops.conv2d_wgrad_raw(x, dy, w.shape, d.stride, d.pad, pt, up, dw=dw, db=db)
    t = timeit(fn)
    fe = pl.flop_exec[which]
    rows.append((n * (t - fe / (REF_FRAC * PEAK)) * 1e3, n, t * 1e6, fe / t / 1e12, fe / t / PEAK, pl.flop / max(fe, 1.0),
                 ("fwd", "dgrad", "wgrad")[which],
                 "B%d %dx%d %d->%d k%dx%d s%d p%d %s%s" % (d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, pt,
                                                          " up2" if up else "")))
rows.sort(reverse=True)
tot = sum(r[1] * r[2] for r in rows) * 1e-3
print("convolution passes of one step: %.1f ms summed (alone, one stream); lost = n*(t - flops/(%.2f*peak))" % (tot, REF_FRAC))
print("%8s %5s %9s %8s %6s %6s  %-6s %s" % ("lost ms", "n", "us", "TFLOP/s", "frac", "alg/ex", "pass", "layer"))
for r in rows:
    print("%8.2f %5d %9.1f %8.1f %6.3f %6.2f  %-6s %s" % r)
print("total lost %.1f ms" % sum(r[0] for r in rows))
