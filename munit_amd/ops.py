"""torch.autograd.Function wrappers over the C ABI (include/munit_hip.h).

PyTorch supplies device memory, the stream and the autograd tape; every arithmetic op of the
hot path is a HIP kernel in libmunit_hip.so.  Tensors keep the reference's logical NCHW / OIHW
shapes but live in channels_last memory (= NHWC / [Cout][KH][KW][Cin]), which is what the
kernels read.  There is no CPU path: a non-HIP tensor raises.

Parameter gradients: when a parameter carries a `_munit_grad` buffer (the trainer's flat
gradient storage) backward-weight accumulates straight into it (beta = 1) and autograd gets
None; otherwise the gradient is returned the ordinary way (used by the op-level tests).
"""
import ctypes
import os
from ctypes import byref, c_float, c_int, c_void_p

import torch
from torch.autograd import Function

from . import _lib
from ._lib import ACT, COMPUTE, PAD, ConvDesc

_ws_cache = {}
# bench.py sets this to a list to time every convolution pass (forward, backward-data, backward-weight) with HIP events on
# the launch stream: entries are (pass 0/1/2, the layer's _Plan, start event, end event).
PROFILE = None
# tests/parity.py sets this to a list to record, in call order, the sign pattern (output > 0) behind every ReLU /
# LeakyReLU of a forward pass, so that the fp64 oracle can take the same branch at every kink.
MASK_SINK = None
# likewise for the other kink of the objective: the sign of (input - target) behind every L1 term, in call order
L1_SINK = None
# bench.py sets this to {"alg": 0.0, "exec": 0.0} to add up, over one step, the algorithmic FLOPs of every convolution /
# linear pass (SURVEY.md section 8d's definition) and the FLOPs the kernels actually issue (sub-pixel and box-sum
# forms execute fewer).
FLOPS = None


def _count(pl, which):
    if FLOPS is not None:
        FLOPS["alg"] += pl.flop
        FLOPS["exec"] += pl.flop_exec[which]


# Arithmetic of the conv / linear contractions: "f32" (exact fp32 MFMA, the reference's arithmetic) or "bf16"
# (operands rounded to bf16 in LDS, fp32 accumulate; BASELINE.json config #3).  Process-wide: set by the trainer.
_COMPUTE = 0


def set_compute(mode):
    """Arithmetic of the contractions.  "bf16s" (bf16 STORAGE, BASELINE.json config #3) multiplies like "bf16"; what it
    adds -- bf16 activation tensors in HBM -- is a property of the tensors handed to the ops, not of this switch."""
    global _COMPUTE
    if mode == "bf16s":
        mode = "bf16"
    if mode not in COMPUTE:
        raise ValueError("munit_amd: compute mode must be one of %s, got %r" % (sorted(COMPUTE) + ["bf16s"], mode))
    _COMPUTE = COMPUTE[mode]


def get_compute():
    return [k for k, v in COMPUTE.items() if v == _COMPUTE][0]


def _require(t, name="tensor", bf16_ok=False):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("munit_amd: %s must live on a HIP device (no CPU fallback exists); got %s"
                           % (name, getattr(t, "device", type(t))))
    if t.dtype != torch.float32 and not (bf16_ok and t.dtype == torch.bfloat16):
        raise RuntimeError("munit_amd: %s must be float32%s, got %s" % (name, " or bfloat16" if bf16_ok else "", t.dtype))


def _dt(t):
    """MUNIT_DTYPE_* of an activation tensor (or of a torch dtype)."""
    return 1 if (t.dtype if isinstance(t, torch.Tensor) else t) == torch.bfloat16 else 0


_TORCH_DT = (torch.float32, torch.bfloat16)


_raw_stream = torch._C._cuda_getCurrentRawStream     # (device index) -> hipStream_t as int; no Stream object per call
_cur_device = torch._C._cuda_getDevice


def _stream():
    """Current stream of the current device (every launch sits inside _on(tensor), so that is the tensor's device)."""
    return c_void_p(_raw_stream(_cur_device()))


def _p(t):
    return None if t is None else c_void_p(t.data_ptr())


_CL = torch.channels_last


def _is_nhwc(t):
    if t.dim() != 4:
        return False
    # one C++ call instead of a Python loop over sizes and strides (this check runs ~1400 times per training step); torch's
    # answer agrees with the loop below on every size / stride pattern tried, size-1 axes included -- the loop stays as the
    # second opinion when torch says no
    if t.is_contiguous(memory_format=_CL):
        return True
    b, c, h, w = t.shape
    want = (h * w * c, 1, w * c, c)
    for size, st, ws in zip(t.shape, t.stride(), want):
        if size > 1 and st != ws:
            return False
    return True


def nhwc(t):
    """Return t (logical NCHW) with NHWC memory; copies only when needed (layout plumbing)."""
    if _is_nhwc(t):
        return t
    return t.contiguous(memory_format=torch.channels_last)


def empty_nhwc(b, c, h, w, like, dtype=torch.float32):
    return torch.empty((b, c, h, w), device=like.device, dtype=dtype, memory_format=torch.channels_last)


def workspace(nbytes, device, stream=None):
    """Grow-only scratch buffer per (device, stream): reuse is ordered by the stream it is used on (`stream`: a
    torch.cuda.Stream other than the current one, e.g. the backward-weight side stream)."""
    key = (device.index, _raw_stream(device.index) if stream is None else stream.cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        size = max(int(nbytes * 1.25), 1 << 20)
        if stream is None:
            buf = torch.empty(size, dtype=torch.uint8, device=device)
        else:
            with torch.cuda.stream(stream):      # the block must belong to the stream that uses it (allocator reuse rule)
                buf = torch.empty(size, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


class _Plan(object):
    """Everything about one layer geometry that does not change between calls: the descriptor, output extent,
    workspace sizes and prepared-weight sizes of the three passes.  Built once per (shape, config) -- the C-ABI
    queries behind it cost more host time than the launch itself when they are repeated ~500 times per step."""
    __slots__ = ("d", "ref", "ho", "wo", "ws_fwd", "ws_dgrad", "ws_wgrad", "prep_bytes", "prep_sig", "kname", "layer", "flop",
                 "flop_exec", "bytes_alg")


_plans = {}


def _plan(b, h, w, cin, cout, kh, kw, stride, pad, pad_type, upsample, act="none", slope=0.2, in_dt=0, out_dt=0):
    key = (b, h, w, cin, cout, kh, kw, stride, pad, pad_type, bool(upsample), act, slope, _COMPUTE, in_dt, out_dt)
    pl = _plans.get(key)
    if pl is not None:
        return pl
    lib = _lib.load()
    pl = _Plan()
    pl.d = ConvDesc(b, h, w, cin, cout, kh, kw, stride, pad, PAD[pad_type], int(bool(upsample)), ACT[act],
                    float(slope), _COMPUTE, in_dt, out_dt)
    pl.ref = byref(pl.d)
    ho, wo = c_int(), c_int()
    _lib.check(lib.munit_conv2d_out_hw(pl.ref, byref(ho), byref(wo)), "conv2d_out_hw")
    pl.ho, pl.wo = ho.value, wo.value
    pl.ws_fwd = lib.munit_conv2d_fwd_workspace_bytes(pl.ref)
    pl.ws_dgrad = lib.munit_conv2d_dgrad_workspace_bytes(pl.ref)
    pl.ws_wgrad = lib.munit_conv2d_wgrad_workspace_bytes(pl.ref)
    pl.prep_bytes = (lib.munit_conv2d_prepared_weight_bytes(pl.ref, 0), lib.munit_conv2d_prepared_weight_bytes(pl.ref, 1))
    # what kind of image each pass wants (kind, stride phases, bf16): one weight may meet several (an fp32 and a bf16
    # use of the same layer), and each gets an image of its own
    sig = []
    for which in (0, 1):
        item = _lib.PrepItem()
        rc = lib.munit_conv2d_prep_item(pl.ref, which, None, None, byref(item))
        sig.append((which, item.kind, item.ps, item.bf16) if rc == 0 else (which, 0, 0, 0))
    pl.prep_sig = tuple(sig)
    # measurement only (bench.py): profiler name of the kernel behind each pass, a readable layer label, and the
    # algorithmic HBM bytes of a pass = its operand tensors once (fwd: x + w + y; dgrad: dy + w + dx; wgrad: x + dy + dw)
    pl.kname = tuple(lib.munit_conv2d_kernel_name(pl.ref, k).decode() for k in range(3))
    pl.layer = "%s%dx%d s%d %d->%d @%dx%d B=%d" % ("up x2 + " if upsample else "", kh, kw, stride, cin, cout, h, w, b)
    nx, ny, nw = b * h * w * cin * (2 if in_dt else 4), b * pl.ho * pl.wo * cout * (2 if out_dt else 4), cout * kh * kw * cin * 4
    pl.bytes_alg = (nx + nw + ny,) * 3
    pl.flop = 2.0 * b * pl.ho * pl.wo * cout * kh * kw * cin        # algorithmic, the same for all three passes
    pl.flop_exec = tuple(lib.munit_conv2d_executed_flops(pl.ref, k) for k in range(3))
    _plans[key] = pl
    return pl


class _Here(object):
    """no-op context: the tensor already lives on the current device (the common case, kept off the slow path)"""
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_HERE = _Here()


def _on(t):
    """Context for launching on t's device: kernels and streams are per device, the process-wide current device may
    be another one (trainer on cuda:N without torch.cuda.set_device(N))."""
    idx = t.device.index
    if idx is None or idx == _cur_device():
        return _HERE
    return torch.cuda.device(idx)


def _same_device(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("munit_amd: operands live on different devices (%s vs %s)" % (dev, t.device))


def _guarded(fn):
    """Run a Function.forward / backward on the device of its first tensor argument (see _on)."""
    def wrapper(ctx, t, *args):
        idx = t.device.index
        if idx is None or idx == _cur_device():
            return fn(ctx, t, *args)
        with torch.cuda.device(idx):
            return fn(ctx, t, *args)
    return wrapper


# ------------------------------------------------------------------------------------------
# prepared weight images (include/munit_hip.h: munit_conv2d_prepare_weights*)
# ------------------------------------------------------------------------------------------
def _prepared(owner, w, pl, which):
    """Image of `w` for pass `which` (0 forward, 1 backward-data) kept with the parameter `owner`, or None when the
    pass needs none / the parameter is not bound to a flat optimizer buffer (then the library rebuilds the image in
    the workspace on every call).  Images are refreshed in ONE launch by the optimizer after every step
    (FusedAdam.refresh_prepared); here only first use and in-place edits of the parameter (load_state_dict) rebuild."""
    if owner is None or not pl.prep_bytes[which]:
        return None
    reg = getattr(owner, "_munit_prep", None)
    if reg is None or w.data_ptr() != owner.data_ptr():
        return None
    key = pl.prep_sig[which]
    ent = reg.get(key)
    # in-place edits of the parameter bump its version; writes through the optimizer's flat buffer (flat_p.copy_,
    # dist.broadcast(flat_p)) bump the buffer's.  Writes that bump neither (p.data.copy_, raw kernels) must be followed by
    # FusedAdam.invalidate_prepared().
    opt = getattr(owner, "_munit_opt", None)
    ver = (owner._version, opt.flat_p._version if opt is not None and opt.flat_p is not None else 0)
    if ent is not None and ent[1] == ver:
        return ent[0]
    lib = _lib.load()
    fresh = ent is None
    if fresh:
        buf = torch.empty(pl.prep_bytes[which], dtype=torch.uint8, device=w.device)   # fp32 or bf16 image
        item = _lib.PrepItem()
        _lib.check(lib.munit_conv2d_prep_item(pl.ref, which, _p(w), _p(buf), byref(item)), "conv2d_prep_item")
        ent = [buf, ver, item]
        reg[key] = ent
    _lib.check(lib.munit_conv2d_prepare_weights(byref(ent[2]), _stream()), "conv2d_prepare_weights")
    ent[1] = ver
    # other streams may use the image right away (the a / b branches share the style encoder): rare path, so simply
    # finish it before returning instead of carrying an event per parameter
    torch.cuda.current_stream().synchronize()
    if fresh and opt is not None:
        opt.register_prepared(ent[2])
    return ent[0]


def prepare_weights_batch(table, n):
    """table: device uint8 tensor holding n munit_prep_item structs."""
    lib = _lib.load()
    with _on(table):
        _lib.check(lib.munit_conv2d_prepare_weights_batch(_p(table), n, _stream()), "conv2d_prepare_weights_batch")


# ------------------------------------------------------------------------------------------
# raw (non-autograd) entry points, also used by the tests
# ------------------------------------------------------------------------------------------
def conv2d_fwd_raw(x, weight, bias, stride, pad, pad_type, upsample, act, slope=0.2, owner=None, out_dtype=None):
    """out_dtype: torch.float32 / torch.bfloat16 of y; default = x's dtype when Cout is a multiple of 64, else fp32
    (3-channel images, small heads).  A bf16 x runs the bf16-storage kernels (weights stay fp32 parameters)."""
    lib = _lib.load()
    x, weight = nhwc(x), nhwc(weight)
    _same_device(x, weight, bias)
    b, cin, h, w = x.shape
    cout, cin_w, kh, kw = weight.shape
    if cin_w != cin:
        raise RuntimeError("munit_amd.conv2d: weight expects %d input channels, input has %d" % (cin_w, cin))
    if out_dtype is None:
        out_dtype = x.dtype if cout % 64 == 0 else torch.float32
    pl = _plan(b, h, w, cin, cout, kh, kw, stride, pad, pad_type, upsample, act, slope, _dt(x), _dt(out_dtype))
    with _on(x):
        y = empty_nhwc(b, cout, pl.ho, pl.wo, x, out_dtype)
        ws = workspace(pl.ws_fwd, x.device) if pl.ws_fwd else None
        wp = _prepared(owner, weight, pl, 0)
        if PROFILE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.munit_conv2d_fwd_prepared(pl.ref, _p(x), _p(weight), _p(wp), _p(bias), _p(y), _p(ws),
                                                 ws.numel() if ws is not None else 0, _stream()), "conv2d_fwd")
        if PROFILE is not None:
            e1.record()
            PROFILE.append((0, pl, e0, e1))
        _count(pl, 0)
    if MASK_SINK is not None and act in ("relu", "lrelu"):
        MASK_SINK.append(y > 0)
    return y


def conv2d_dgrad_raw(dy, weight, x_shape, stride, pad, pad_type, upsample, add=None, owner=None,
                     x_dtype=torch.float32):
    """x_dtype: element type of the layer input, i.e. of the dx returned (dy carries the output's type)."""
    lib = _lib.load()
    dy, weight = nhwc(dy), nhwc(weight)
    _same_device(dy, weight, add)
    b, cin, h, w = x_shape
    cout, _, kh, kw = weight.shape
    pl = _plan(b, h, w, cin, cout, kh, kw, stride, pad, pad_type, upsample, in_dt=_dt(x_dtype), out_dt=_dt(dy))
    with _on(dy):
        ws = workspace(pl.ws_dgrad, dy.device)
        dx = empty_nhwc(b, cin, h, w, dy, x_dtype)
        if add is not None:
            add = nhwc(add)
        wp = _prepared(owner, weight, pl, 1)
        if PROFILE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.munit_conv2d_dgrad_prepared(pl.ref, _p(dy), _p(weight), _p(wp), _p(add), _p(dx), _p(ws),
                                                   ws.numel(), _stream()), "conv2d_dgrad")
        if PROFILE is not None:
            e1.record()
            PROFILE.append((1, pl, e0, e1))
        _count(pl, 1)
    return dx


def conv2d_wgrad_raw(x, dy, weight_shape, stride, pad, pad_type, upsample, dw=None, db=None, beta=0.0,
                     want_bias=True, stream=None):
    """dw/db given -> accumulate (beta) in place; else fresh tensors are returned.  `stream`: launch on this
    torch.cuda.Stream instead of the current one (dw / db must then be given: nothing is allocated for another stream)."""
    lib = _lib.load()
    x, dy = nhwc(x), nhwc(dy)
    _same_device(x, dy, dw, db)
    b, cin, h, w = x.shape
    cout, _, kh, kw = weight_shape
    pl = _plan(b, h, w, cin, cout, kh, kw, stride, pad, pad_type, upsample, in_dt=_dt(x), out_dt=_dt(dy))
    with _on(x):
        ws = workspace(pl.ws_wgrad, x.device, stream)
        if (dw is None or (db is None and want_bias)) and stream is not None:
            raise RuntimeError("munit_amd.conv2d_wgrad_raw: give dw / db when launching on another stream")
        if dw is None:
            dw = torch.empty(tuple(weight_shape), device=x.device, dtype=torch.float32,
                             memory_format=torch.channels_last)
            beta = 0.0
        if db is None and want_bias:
            db = torch.empty(cout, device=x.device, dtype=torch.float32)
        st = _stream() if stream is None else c_void_p(stream.cuda_stream)
        prof = PROFILE is not None and stream is None       # bench.py's instrumented steps run on one stream
        if prof:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.munit_conv2d_wgrad(pl.ref, _p(x), _p(dy), _p(dw), _p(db), c_float(beta), _p(ws), ws.numel(), st),
                   "conv2d_wgrad")
        if prof:
            e1.record()
            PROFILE.append((2, pl, e0, e1))
        _count(pl, 2)
    return dw, db


def act_bwd_raw(act, slope, y, dy):
    lib = _lib.load()
    dx = torch.empty_like(y)
    with _on(y):
        _lib.check(lib.munit_act_bwd(ACT[act], c_float(slope), _p(y), _p(dy), _p(dx), y.numel(), _stream()), "act_bwd")
    return dx


# ------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------
# Backward-weight runs on a side stream: it only feeds the optimizer, so it need not sit in the dy -> dx chain.
# Launched before the layer's backward-data, it fills the CUs that the chain leaves idle at kernel heads and tails
# (e.g. while the few blocks of a folded backward-data launch that hold 4-fold corner pixels finish).  All
# backward-weight launches share the one side stream, so accumulations into the same flat gradient stay ordered.
_SIDE = {}
SIDE_STREAM_WGRAD = not os.environ.get("MUNIT_NO_SIDE_STREAM")


def _side_stream(device):
    key = (device.type, device.index)
    st = _SIDE.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE[key] = st
    return st


def stream_wait(waiter, signaler):
    """`waiter` (torch.cuda.Stream) waits for everything enqueued so far on `signaler`, through the library's cached
    event (munit_stream_wait_stream) instead of a torch Event object per edge."""
    with torch.cuda.device(waiter.device):
        _lib.check(_lib.load().munit_stream_wait_stream(c_void_p(waiter.cuda_stream), c_void_p(signaler.cuda_stream)),
                   "stream_wait_stream")


def join_side_streams():
    """Make the current stream wait for every backward-weight launched so far (call before the optimizer step)."""
    for (_, index), st in _SIDE.items():
        stream_wait(torch.cuda.current_stream(torch.device("cuda", index)), st)


FUSE_SKIP_GRAD = not os.environ.get("MUNIT_NO_FUSE_SKIP_GRAD")   # A/B switch of ResidualLink


class ResidualLink(object):
    """Carries the gradient of a ResBlock's skip connection (networks.py:620-623, `out += residual`) from the block's last
    norm -- whose backward receives it -- to the block's FIRST convolution, whose backward-data adds it to dx in its epilogue
    (munit_conv2d_dgrad's `add` operand).  Autograd would otherwise sum the two gradients of the block input with a separate
    element-wise kernel over a 33 MB tensor per block (40 per gen_update).  One link per block call; the norm's backward runs
    first (it is the last node of the block), the convolution's last, both on the stream the block was recorded on."""
    __slots__ = ("grad",)

    def __init__(self):
        self.grad = None

    def park(self, dy):
        if self.grad is not None:
            raise RuntimeError("munit_amd: ResidualLink used twice in one backward pass")
        self.grad = dy

    def take(self):
        g, self.grad = self.grad, None
        return g


FUSE_ADAIN_GRAD = not os.environ.get("MUNIT_NO_FUSE_ADAIN_GRAD")   # A/B switch of AdainGradSink


class AdainGradSink(object):
    """One gradient buffer for the (B, n) AdaIN parameter tensor of a decode call (networks.py:230-239 slices it into the
    eight layers' weight / bias columns).  Every layer's backward writes its own columns in place; the FIRST layer of the
    module order -- whose backward runs last, all later layers being downstream of it -- returns the buffer to autograd, the
    others return nothing.  Replaces eight zero-filled (B, n) gradients summed by seven element-wise kernels."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, pad_type, upsample, act, slope, wbuf, bbuf, owner, out_dtype, link=None):
        _require(x, "conv input", bf16_ok=True)
        _require(weight, "conv weight")
        x, w = nhwc(x), nhwc(weight)
        y = conv2d_fwd_raw(x, w, bias, stride, pad, pad_type, upsample, act, slope, owner=owner, out_dtype=out_dtype)
        if act != "none" and y.dtype != torch.float32:
            raise RuntimeError("munit_amd.conv2d: a fused activation needs an fp32 output (bf16 layers are followed by a norm)")
        ctx.cfg = (stride, pad, pad_type, upsample, act, slope)
        ctx.has_bias = bias is not None
        ctx.wbuf = wbuf
        ctx.bbuf = bbuf if bias is not None else None
        ctx.owner = owner
        ctx.link = link
        ctx.save_for_backward(x, w, y if act != "none" else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, pad_type, upsample, act, slope = ctx.cfg
        dy = nhwc(dy)
        if act != "none":
            dy = act_bwd_raw(act, slope, y, dy)
        dx = dw = db = None
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if want_w and ctx.wbuf is not None:
            if SIDE_STREAM_WGRAD:
                side = _side_stream(dy.device)
                with _on(dy):                   # dy, x and the zeroed gradient buffer are ready on the current stream
                    _lib.check(_lib.load().munit_stream_wait_stream(c_void_p(side.cuda_stream), _stream()),
                               "stream_wait_stream")
                conv2d_wgrad_raw(x, dy, w.shape, stride, pad, pad_type, upsample, dw=ctx.wbuf, db=ctx.bbuf, beta=1.0,
                                 want_bias=False, stream=side)
                x.record_stream(side)
                dy.record_stream(side)
            else:
                conv2d_wgrad_raw(x, dy, w.shape, stride, pad, pad_type, upsample, dw=ctx.wbuf, db=ctx.bbuf, beta=1.0,
                                 want_bias=False)
        if ctx.needs_input_grad[0]:
            skip = ctx.link.take() if ctx.link is not None else None      # gradient of the ResBlock skip connection
            dx = conv2d_dgrad_raw(dy, w, x.shape, stride, pad, pad_type, upsample, add=skip, owner=ctx.owner, x_dtype=x.dtype)
        if want_w and ctx.wbuf is None:
            dw, db = conv2d_wgrad_raw(x, dy, w.shape, stride, pad, pad_type, upsample, want_bias=ctx.has_bias)
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None


def _gbuf(p):
    return getattr(p, "_munit_grad", None) if p is not None else None


def conv2d(x, weight, bias=None, stride=1, pad=0, pad_type="zero", upsample=False, act="none", slope=0.2,
           out_dtype=None, link=None):
    """pad -> conv -> bias -> activation (networks.py:695-701), optional fused nearest x2
    upsample of the input (networks.py:534).  out_dtype: see conv2d_fwd_raw.  link: ResidualLink of the ResBlock this
    convolution opens (its backward-data then adds the skip gradient parked there)."""
    return _Conv2d.apply(x, weight, bias, stride, pad, pad_type, upsample, act, slope, _gbuf(weight), _gbuf(bias),
                         weight, out_dtype, link)


def linear(x, weight, bias=None, act="none"):
    """nn.Linear (+ReLU) of LinearBlock (networks.py:743-749) as a 1x1 convolution."""
    b, k = x.shape
    n = weight.shape[0]
    wbuf = _gbuf(weight)
    if wbuf is not None:
        wbuf = wbuf.view(n, k, 1, 1)
    y = _Conv2d.apply(x.reshape(b, k, 1, 1), weight.view(n, k, 1, 1), bias, 1, 0, "zero", False, act, 0.2, wbuf,
                      _gbuf(bias), weight, torch.float32)
    return y.reshape(b, n)


_NORM_ACT = {False: 0, True: 1, None: 0, "none": 0, "relu": 1, "lrelu": 2, "tanh": 3}


def _norm_act(relu):
    """Activation fused behind a normalisation (networks.py:668-681, 695-701 pair any norm with any activation): the norm
    kernels take MUNIT_ACT_* in their `relu` argument -- 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 tanh.  Accepts the historical
    bool or the Conv2dBlock's activation name."""
    try:
        return _NORM_ACT[relu]
    except KeyError:
        raise NotImplementedError("munit_amd: activation %r after a normalisation layer" % (relu,))


class _InstNorm(Function):
    @staticmethod
    @_guarded
    def forward(ctx, x, adain, residual, w_off, b_off, relu, eps, link=None, sink=None, sink_first=False):
        _require(x, "instance-norm input", bf16_ok=True)
        lib = _lib.load()
        x = nhwc(x)
        b, c, h, w = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((b, c, 2), device=x.device, dtype=torch.float32)
        ws = workspace(lib.munit_instnorm_workspace_bytes(b, h * w, c), x.device)
        ld = 0
        if adain is not None:
            _require(adain, "adain params")
            if adain.dim() != 2 or not adain.is_contiguous():
                raise RuntimeError("munit_amd.adain: params must be a contiguous (B, n) tensor")
            ld = adain.shape[1]
        if residual is not None:
            residual = nhwc(residual)
            if residual.dtype != x.dtype:
                raise RuntimeError("munit_amd.instance_norm: residual must have the input's dtype")
        relu = _norm_act(relu)
        fn = lib.munit_instnorm_fwd_bf16 if x.dtype == torch.bfloat16 else lib.munit_instnorm_fwd
        _lib.check(fn(_p(x), _p(y), _p(stats), b, h * w, c, _p(adain), ld, w_off, b_off, _p(residual), relu,
                      c_float(eps), _p(ws), ws.numel(), _stream()), "instnorm_fwd")
        ctx.cfg = (w_off, b_off, relu, ld)
        ctx.has_res = residual is not None
        ctx.link = link if residual is not None else None
        ctx.sink, ctx.sink_first = (sink, sink_first) if adain is not None else (None, False)
        ctx.save_for_backward(x, stats, adain)
        if MASK_SINK is not None and relu in (1, 2):      # ReLU / LeakyReLU: the sign of the output is the branch taken
            if residual is not None:
                raise RuntimeError("munit_amd: kink recording needs relu and residual on different layers")
            MASK_SINK.append(y > 0)
        return y

    @staticmethod
    @_guarded
    def backward(ctx, dy):
        lib = _lib.load()
        x, stats, adain = ctx.saved_tensors
        w_off, b_off, relu, ld = ctx.cfg
        dy = nhwc(dy)
        b, c, h, w = x.shape
        dx = torch.empty_like(x)
        d_adain = None
        sink = ctx.sink if (adain is not None and ctx.needs_input_grad[1]) else None
        if sink is not None:
            if sink.buf is None:                     # the last AdaIN layer's backward comes first
                sink.buf = torch.empty_like(adain)   # every column is written by exactly one layer (checked at assignment)
            d_adain = sink.buf
        elif adain is not None and ctx.needs_input_grad[1]:
            d_adain = torch.zeros_like(adain)
        ws = workspace(lib.munit_instnorm_workspace_bytes(b, h * w, c), x.device)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        fn = lib.munit_instnorm_bwd_bf16 if x.dtype == torch.bfloat16 else lib.munit_instnorm_bwd
        _lib.check(fn(_p(x), _p(dy), _p(stats), _p(dx), b, h * w, c, _p(adain), _p(d_adain), ld, w_off, b_off,
                      relu, _p(ws), ws.numel(), _stream()), "instnorm_bwd")
        if sink is not None:
            if ctx.sink_first:
                sink.buf = None                      # handed to autograd: the buffer is complete
            else:
                d_adain = None
        if ctx.link is not None:        # the block's first convolution adds the skip gradient inside its backward-data
            ctx.link.park(dy)
            return dx, d_adain, None, None, None, None, None, None, None, None
        return dx, d_adain, (dy if ctx.has_res else None), None, None, None, None, None, None, None


def instance_norm(x, relu=False, residual=None, eps=1e-5, link=None):
    """nn.InstanceNorm2d(affine=False) (networks.py:657) [+activation: bool ReLU or "relu" / "lrelu" / "tanh" / "none"]
    [+residual].  link: see ResidualLink."""
    return _InstNorm.apply(x, None, residual, 0, 0, relu, eps, link)


def adain(x, params, w_off, b_off, relu=False, residual=None, eps=1e-5, link=None, sink=None, sink_first=False):
    """AdaptiveInstanceNorm2d (networks.py:823-845); weight/bias are columns
    [w_off, w_off+C) / [b_off, b_off+C) of the (B, n) MLP output.  sink / sink_first: see AdainGradSink."""
    return _InstNorm.apply(x, params, residual, w_off, b_off, relu, eps, link, sink, sink_first)


class _LayerNorm(Function):
    @staticmethod
    @_guarded
    def forward(ctx, x, gamma, beta, relu, eps):
        _require(x, "layer-norm input", bf16_ok=True)
        lib = _lib.load()
        x = nhwc(x)
        b, c, h, w = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((b, 2), device=x.device, dtype=torch.float32)
        ws = workspace(lib.munit_layernorm_workspace_bytes(b, h * w, c), x.device)
        relu = _norm_act(relu)
        fn = lib.munit_layernorm_fwd_bf16 if x.dtype == torch.bfloat16 else lib.munit_layernorm_fwd
        _lib.check(fn(_p(x), _p(y), _p(stats), b, h * w, c, _p(gamma), _p(beta), relu, c_float(eps), _p(ws),
                      ws.numel(), _stream()), "layernorm_fwd")
        ctx.cfg = (relu, eps)
        ctx.gbuf = getattr(gamma, "_munit_grad", None)
        ctx.bbuf = getattr(beta, "_munit_grad", None)
        ctx.save_for_backward(x, stats, gamma, beta)
        if MASK_SINK is not None and relu in (1, 2):
            MASK_SINK.append(y > 0)
        return y

    @staticmethod
    @_guarded
    def backward(ctx, dy):
        lib = _lib.load()
        x, stats, gamma, beta = ctx.saved_tensors
        relu, eps = ctx.cfg
        dy = nhwc(dy)
        b, c, h, w = x.shape
        dx = torch.empty_like(x)
        # accumulate straight into the flat gradient only when autograd actually wants these gradients (a frozen
        # gamma / beta gets throw-away buffers: the kernel always produces both)
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        into = want and ctx.gbuf is not None and ctx.bbuf is not None
        dgamma = ctx.gbuf if into else torch.empty_like(gamma)
        dbeta = ctx.bbuf if into else torch.empty_like(beta)
        ws = workspace(lib.munit_layernorm_workspace_bytes(b, h * w, c), x.device)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        fn = lib.munit_layernorm_bwd_bf16 if x.dtype == torch.bfloat16 else lib.munit_layernorm_bwd
        _lib.check(fn(_p(x), _p(dy), _p(stats), _p(dx), b, h * w, c, _p(gamma), _p(beta), _p(dgamma), _p(dbeta),
                      c_float(1.0 if into else 0.0), relu, c_float(eps), _p(ws), ws.numel(), _stream()),
                   "layernorm_bwd")
        if into or not want:
            return dx, None, None, None, None
        return (dx, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None, None,
                None)


def layer_norm(x, gamma, beta, relu=False, eps=1e-5):
    """MUNIT's LayerNorm (networks.py:862-878) [+activation, see instance_norm]."""
    return _LayerNorm.apply(x, gamma, beta, relu, eps)


class _AvgPool3s2(Function):
    @staticmethod
    @_guarded
    def forward(ctx, x):
        _require(x, "avgpool input")
        lib = _lib.load()
        x = nhwc(x)
        b, c, h, w = x.shape
        y = empty_nhwc(b, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, x)
        _lib.check(lib.munit_avgpool3s2_fwd(_p(x), _p(y), b, h, w, c, _stream()), "avgpool_fwd")
        ctx.shape = (b, c, h, w)
        return y

    @staticmethod
    @_guarded
    def backward(ctx, dy):
        lib = _lib.load()
        b, c, h, w = ctx.shape
        dy = nhwc(dy)
        dx = empty_nhwc(b, c, h, w, dy)
        _lib.check(lib.munit_avgpool3s2_bwd(_p(dy), _p(dx), b, h, w, c, _stream()), "avgpool_bwd")
        return dx


def avgpool3s2(x):
    """nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False) (networks.py:32-34)."""
    return _AvgPool3s2.apply(x)


class _GlobalAvgPool(Function):
    @staticmethod
    @_guarded
    def forward(ctx, x):
        _require(x, "global-avgpool input")
        lib = _lib.load()
        x = nhwc(x)
        b, c, h, w = x.shape
        y = empty_nhwc(b, c, 1, 1, x)
        _lib.check(lib.munit_gap_fwd(_p(x), _p(y), b, h * w, c, _stream()), "gap_fwd")
        ctx.shape = (b, c, h, w)
        return y

    @staticmethod
    @_guarded
    def backward(ctx, dy):
        lib = _lib.load()
        b, c, h, w = ctx.shape
        dy = dy.contiguous()
        dx = empty_nhwc(b, c, h, w, dy)
        _lib.check(lib.munit_gap_bwd(_p(dy), _p(dx), b, h * w, c, _stream()), "gap_bwd")
        return dx


def global_avgpool(x):
    """nn.AdaptiveAvgPool2d(1) (networks.py:471)."""
    return _GlobalAvgPool.apply(x)


class _L1Mean(Function):
    @staticmethod
    @_guarded
    def forward(ctx, a, b, mask):
        _require(a, "l1 input", bf16_ok=True)
        _require(b, "l1 target", bf16_ok=True)
        lib = _lib.load()
        if a.shape != b.shape:
            raise RuntimeError("munit_amd.l1_mean: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        if a.dtype != b.dtype:
            raise RuntimeError("munit_amd.l1_mean: dtype mismatch %s vs %s" % (a.dtype, b.dtype))
        if L1_SINK is not None:
            L1_SINK.append(a.float() > b.float())
        if a.dim() == 4:
            a, b = nhwc(a), nhwc(b)
            c = a.shape[1]
        else:
            a, b = a.contiguous(), b.contiguous()
            c = 1
        npix = a.numel() // c
        if mask is not None:
            _require(mask, "l1 mask")
            mask = mask.contiguous()
            if mask.numel() != npix:
                raise RuntimeError("munit_amd.l1_mean: mask must have one value per pixel (B,1,H,W)")
        out = torch.empty((), device=a.device, dtype=torch.float32)
        ws = workspace(lib.munit_loss_workspace_bytes(a.numel()), a.device)
        fn = lib.munit_l1_mean_fwd_bf16 if a.dtype == torch.bfloat16 else lib.munit_l1_mean_fwd
        _lib.check(fn(_p(a), _p(b), _p(mask), npix, c, _p(out), _p(ws), ws.numel(), _stream()), "l1_mean_fwd")
        ctx.c = c
        ctx.save_for_backward(a, b, mask)
        return out

    @staticmethod
    @_guarded
    def backward(ctx, gout):
        lib = _lib.load()
        a, b, mask = ctx.saved_tensors
        gout = gout.contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        fn = lib.munit_l1_mean_bwd_bf16 if a.dtype == torch.bfloat16 else lib.munit_l1_mean_bwd
        _lib.check(fn(_p(a), _p(b), _p(mask), a.numel() // ctx.c, ctx.c, _p(gout), _p(da), _p(db), _stream()),
                   "l1_mean_bwd")
        return da, db, None


def l1_mean(a, b, mask=None):
    """recon_criterion / recon_criterion_mask (trainer.py:279-305)."""
    return _L1Mean.apply(a, b, mask)


class _MseConst(Function):
    @staticmethod
    @_guarded
    def forward(ctx, x, target):
        _require(x, "mse input")
        lib = _lib.load()
        x = nhwc(x) if x.dim() == 4 else x.contiguous()
        out = torch.empty((), device=x.device, dtype=torch.float32)
        ws = workspace(lib.munit_loss_workspace_bytes(x.numel()), x.device)
        _lib.check(lib.munit_mse_const_fwd(_p(x), c_float(target), x.numel(), _p(out), _p(ws), ws.numel(), _stream()),
                   "mse_const_fwd")
        ctx.target = target
        ctx.save_for_backward(x)
        return out

    @staticmethod
    @_guarded
    def backward(ctx, gout):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        _lib.check(lib.munit_mse_const_bwd(_p(x), c_float(ctx.target), x.numel(), _p(gout.contiguous()), _p(dx),
                                           _stream()), "mse_const_bwd")
        return dx, None


def mse_const(x, target):
    """LSGAN term torch.mean((x - target) ** 2) (networks.py:91,109)."""
    return _MseConst.apply(x, float(target))


class _ScalarSum(Function):
    """sum of device scalars with unit weights; backward hands the upstream gradient to every
    term unchanged (no kernel)."""

    @staticmethod
    def forward(ctx, *terms):
        ctx.n = len(terms)
        return weighted_sum(terms, [1.0] * len(terms))

    @staticmethod
    def backward(ctx, gout):
        return (gout,) * ctx.n


def scalar_sum(terms):
    return _ScalarSum.apply(*terms)


def weighted_sum(terms, weights):
    """Device-side sum_i w_i * term_i of scalar tensors (no autograd; logging value of the
    total loss, trainer.py:539-558 / 1181-1184)."""
    lib = _lib.load()
    n = len(terms)
    out = torch.empty((), device=terms[0].device, dtype=torch.float32)
    _same_device(*terms)
    ptrs = (c_void_p * n)(*[t.data_ptr() for t in terms])
    ws_ = (c_float * n)(*[float(w) for w in weights])
    with _on(out):
        _lib.check(lib.munit_weighted_sum(ptrs, ws_, n, _p(out), _stream()), "weighted_sum")
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    lib = _lib.load()
    for t, nm in ((p, "param"), (g, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
        _require(t, "adam " + nm)
    _same_device(p, g, m, v)
    with _on(p):
        _lib.check(lib.munit_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
                                       float(eps), float(weight_decay), int(step), _stream()), "adam_step")


def extraadam_step(p, g, m, v, p_saved, lr, beta1, beta2, eps, weight_decay, step, mode):
    lib = _lib.load()
    for t, nm in ((p, "param"), (g, "grad"), (m, "exp_avg"), (v, "exp_avg_sq"), (p_saved, "saved params")):
        _require(t, "extraadam " + nm)
    _same_device(p, g, m, v, p_saved)
    with _on(p):
        _lib.check(lib.munit_extraadam_step(_p(p), _p(g), _p(m), _p(v), _p(p_saved), p.numel(), float(lr),
                                            float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                            int(mode), _stream()), "extraadam_step")


def scale_(x, alpha):
    lib = _lib.load()
    _require(x, "scale input")
    with _on(x):
        _lib.check(lib.munit_scale(_p(x), _p(x), x.numel(), c_float(alpha), 0, _stream()), "scale")
    return x
