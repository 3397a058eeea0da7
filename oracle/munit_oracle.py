"""CPU oracle for the MUNIT AdaINGen / AdaINGen_double + MsImageDis training step.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product path (munit_amd/) never
does; it fails loudly when the HIP library is missing.

It is a functional restatement (plain torch CPU ops over a flat {state_dict key: tensor}
mapping, no nn.Module classes) of the algorithm in the reference:

  networks      /root/reference/scripts/networks.py
  trainer step  /root/reference/scripts/trainer.py:336-561 (gen_update), :1133-1186 (dis_update)
  init/sched    /root/reference/scripts/utils.py:1066-1115

Each function cites the reference lines it follows.  Parity pin: the reference ships no
tests or golden vectors for this path, so the oracle is pinned by
fixtures generated in the build container from the reference's own importable
scripts/networks.py (tests/golden/make_golden.py, outputs committed under tests/golden/).
The trainer module itself is not importable anywhere (extraadam.py has no imports,
torchvision absent, hard-coded .cuda()), so the step-level fixtures come from the
reference *network modules* driven by the documented update sequence with
torch.optim.Adam; that is the strongest pin available.

dtype follows the tensors handed in (float32 for the "reference forward", float64 for
ground truth).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

IN_EPS = 1e-5  # nn.InstanceNorm2d / AdaptiveInstanceNorm2d default (networks.py:657, :811)
LN_EPS = 1e-5  # LayerNorm default (networks.py:852)


# --------------------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------------------
class KinkMasks:
    """Optional test hook (tests/parity.py): sign patterns of every ReLU / LeakyReLU, in call order, recorded from
    another run of the same network.  While one is installed in KINK_MASKS, `activation` takes the recorded branch
    instead of its own comparison, so two runs that differ only by rounding differentiate the SAME piecewise-linear
    function (a pre-activation within rounding noise of 0 otherwise switches an upstream gradient element on or
    off).  Values change by at most that noise; the default (None) is the plain activation."""

    def __init__(self, masks, l1_signs=None):
        self.masks, self.pos = list(masks), 0
        # the same for the L1 terms of the objective: recorded (input > target) patterns, in call order; None = plain abs
        self.l1_signs, self.l1_pos = (None if l1_signs is None else list(l1_signs)), 0
        # where a recorded branch differs from the branch this (fp64) run would have taken by itself: the largest
        # |pre-activation| among those elements relative to the tensor's max, and how many there are.  A recorded mask
        # may only differ within rounding noise of the kink -- tests/parity.py asserts it (a wrong mask on a LARGE
        # pre-activation would otherwise be followed, not flagged).
        self.worst_rel, self.n_disagree, self.n_total, self.worst_at = 0.0, 0, 0, None

    def _audit(self, own: Tensor, rec: Tensor, x: Tensor, where: str) -> None:
        # an element that is exactly 0 (the masked-out pixels of the masked L1 terms) has the same value and the same zero
        # gradient on both branches: not a disagreement
        dis = (own != rec) & (x.detach() != 0)
        self.n_total += x.numel()
        n = int(dis.sum())
        if n:
            self.n_disagree += n
            rel = float(x.detach()[dis].abs().max()) / max(float(x.detach().abs().max()), 1e-300)
            if rel > self.worst_rel:
                self.worst_rel, self.worst_at = rel, where

    def take_l1(self, d: Tensor) -> Tensor:
        assert self.l1_pos < len(self.l1_signs), "more L1 terms than recorded sign patterns"
        m = self.l1_signs[self.l1_pos]
        self.l1_pos += 1
        assert tuple(m.shape) == tuple(d.shape), (self.l1_pos, tuple(m.shape), tuple(d.shape))
        self._audit(d > 0, m, d, "l1 term %d" % (self.l1_pos - 1))
        return m

    def take(self, x: Tensor) -> Tensor:
        assert self.pos < len(self.masks), "more activations than recorded masks"
        m = self.masks[self.pos]
        self.pos += 1
        assert m.numel() == x.numel() and m.shape[0] == x.shape[0], (self.pos, tuple(m.shape), tuple(x.shape))
        m = m.reshape(x.shape)
        self._audit(x > 0, m, x, "activation %d %s" % (self.pos - 1, tuple(x.shape)))
        return m

    def done(self) -> bool:
        return self.pos == len(self.masks) and (self.l1_signs is None or self.l1_pos == len(self.l1_signs))


KINK_MASKS: Optional[KinkMasks] = None


def activation(x: Tensor, kind: str) -> Tensor:
    """networks.py:668-681 (relu / lrelu 0.2 / tanh / none)."""
    if KINK_MASKS is not None and kind in ("relu", "lrelu"):
        m = KINK_MASKS.take(x)
        return torch.where(m, x, x * (0.0 if kind == "relu" else 0.2))
    if kind == "relu":
        return torch.clamp_min(x, 0)
    if kind == "lrelu":
        return torch.where(x > 0, x, x * 0.2)
    if kind == "tanh":
        return torch.tanh(x)
    if kind == "none":
        return x
    raise ValueError("unsupported activation %r" % kind)


def pad2d(x: Tensor, p: int, pad_type: str) -> Tensor:
    """networks.py:642-649: ReflectionPad2d / ZeroPad2d ahead of an un-padded Conv2d."""
    if p == 0:
        return x
    if pad_type == "reflect":
        return F.pad(x, (p, p, p, p), mode="reflect")
    if pad_type == "zero":
        return F.pad(x, (p, p, p, p))
    raise ValueError("unsupported pad type %r" % pad_type)


def conv_block(x, w, b, stride, pad, pad_type, norm_fn=None, activ="none"):
    """Conv2dBlock.forward, networks.py:695-701: activation(norm(conv(pad(x))))."""
    y = F.conv2d(pad2d(x, pad, pad_type), w, b, stride=stride)
    if norm_fn is not None:
        y = norm_fn(y)
    return activation(y, activ)


def instance_norm(x: Tensor) -> Tensor:
    """nn.InstanceNorm2d(affine=False), networks.py:657: per-(b,c) biased variance."""
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - mu) / torch.sqrt(var + IN_EPS)


def adain(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """AdaptiveInstanceNorm2d.forward, networks.py:823-845: F.batch_norm(training=True)
    on the (1, B*C, H, W) view == instance norm with a per-(b,c) affine.
    weight / bias are (B, C)."""
    b, c = x.shape[:2]
    return instance_norm(x) * weight.reshape(b, c, 1, 1) + bias.reshape(b, c, 1, 1)


def munit_layer_norm(x: Tensor, gamma: Tensor, beta: Tensor) -> Tensor:
    """LayerNorm.forward, networks.py:862-878: per-sample mean and UNBIASED std over
    C*H*W, eps added to std, then per-channel gamma/beta."""
    n = x.shape[0]
    flat = x.reshape(n, -1)
    mu = flat.mean(dim=1).reshape(n, 1, 1, 1)
    sd = flat.std(dim=1, unbiased=True).reshape(n, 1, 1, 1)
    y = (x - mu) / (sd + LN_EPS)
    return y * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)


def upsample2(x: Tensor) -> Tensor:
    """nn.Upsample(scale_factor=2) (nearest), networks.py:534."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def avgpool_3s2(x: Tensor) -> Tensor:
    """nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False), networks.py:32-34."""
    return F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)


# --------------------------------------------------------------------------------------
# network structure (key layout == the reference's state_dict)
# --------------------------------------------------------------------------------------
def style_encoder(sd: State, pre: str, x: Tensor, hp_gen: dict) -> Tensor:
    """StyleEncoder, networks.py:442-477 (built with n_downsample=4, norm='none',
    networks.py:183-185)."""
    act, pt = hp_gen["activ"], hp_gen["pad_type"]
    h = conv_block(x, sd[pre + "model.0.conv.weight"], sd[pre + "model.0.conv.bias"], 1, 3, pt, None, act)
    for i in range(1, 5):
        h = conv_block(h, sd[pre + "model.%d.conv.weight" % i], sd[pre + "model.%d.conv.bias" % i], 2, 1, pt, None, act)
    h = h.mean(dim=(2, 3), keepdim=True)  # AdaptiveAvgPool2d(1), networks.py:471
    return F.conv2d(h, sd[pre + "model.6.weight"], sd[pre + "model.6.bias"])


def res_blocks(sd: State, pre: str, x: Tensor, n_res: int, pt: str, act: str, norm_fns) -> Tensor:
    """ResBlocks/ResBlock, networks.py:569-580, 603-624.  norm_fns(i, j) returns the norm
    callable of block i, conv j."""
    for i in range(n_res):
        p = pre + "model.%d.model." % i
        h = conv_block(x, sd[p + "0.conv.weight"], sd[p + "0.conv.bias"], 1, 1, pt, norm_fns(i, 0), act)
        h = conv_block(h, sd[p + "1.conv.weight"], sd[p + "1.conv.bias"], 1, 1, pt, norm_fns(i, 1), "none")
        x = h + x
    return x


def content_encoder(sd: State, pre: str, x: Tensor, hp_gen: dict) -> Tensor:
    """ContentEncoder, networks.py:480-512 (norm='in')."""
    act, pt = hp_gen["activ"], hp_gen["pad_type"]
    nd, nr = hp_gen["n_downsample"], hp_gen["n_res"]
    h = conv_block(x, sd[pre + "model.0.conv.weight"], sd[pre + "model.0.conv.bias"], 1, 3, pt, instance_norm, act)
    for i in range(1, nd + 1):
        h = conv_block(h, sd[pre + "model.%d.conv.weight" % i], sd[pre + "model.%d.conv.bias" % i], 2, 1, pt, instance_norm, act)
    return res_blocks(sd, pre + "model.%d." % (nd + 1), h, nr, pt, act, lambda i, j: instance_norm)


def mlp(sd: State, pre: str, style: Tensor, n_blk: int = 3, act: str = "relu") -> Tensor:
    """MLP / LinearBlock, networks.py:583-597, 704-749: the generator's activation (networks.py:202-209 passes
    activ=activ; 'relu' in every shipped config) on all but the last layer."""
    h = style.reshape(style.shape[0], -1)
    for i in range(n_blk):
        h = F.linear(h, sd[pre + "model.%d.fc.weight" % i], sd[pre + "model.%d.fc.bias" % i])
        if i < n_blk - 1:
            h = activation(h, act)
    return h


def decoder(sd: State, pre: str, content: Tensor, adain_params: Tensor, hp_gen: dict) -> Tensor:
    """Decoder, networks.py:515-563, with assign_adain_params, networks.py:230-239:
    AdaIN layer l (module order: block0.conv0, block0.conv1, block1.conv0, ...) takes
    bias = params[:, 2lC : 2lC+C], weight = params[:, 2lC+C : 2lC+2C]."""
    act, pt = hp_gen["activ"], hp_gen["pad_type"]
    nu, nr = hp_gen["n_downsample"], hp_gen["n_res"]
    c = content.shape[1]

    def norm_fns(i, j):
        l = 2 * i + j
        bias = adain_params[:, 2 * l * c: 2 * l * c + c]
        weight = adain_params[:, 2 * l * c + c: 2 * l * c + 2 * c]
        return lambda t: adain(t, weight, bias)

    h = res_blocks(sd, pre + "model.0.", content, nr, pt, act, norm_fns)
    idx = 1
    for _ in range(nu):
        h = upsample2(h)
        p = pre + "model.%d." % (idx + 1)
        g, bt = sd[p + "norm.gamma"], sd[p + "norm.beta"]
        h = conv_block(h, sd[p + "conv.weight"], sd[p + "conv.bias"], 1, 2, pt,
                       lambda t, g=g, bt=bt: munit_layer_norm(t, g, bt), act)
        idx += 2
    p = pre + "model.%d." % idx
    return conv_block(h, sd[p + "conv.weight"], sd[p + "conv.bias"], 1, 3, pt, None, "tanh")


def num_adain_params(hp_gen: dict) -> int:
    """get_num_adain_params, networks.py:241-247."""
    dim = hp_gen["dim"] * (2 ** hp_gen["n_downsample"])
    return 2 * dim * 2 * hp_gen["n_res"]


class GenView:
    """encode/decode dispatch for AdaINGen (gen_state 0: one state per domain,
    networks.py:217-228) and AdaINGen_double (gen_state 1: shared style encoder, content
    encoder / decoder / MLP picked by encoder_name in {1,2}, networks.py:331-356)."""

    def __init__(self, sd: State, hp_gen: dict, double: bool, prefix: str = ""):
        self.sd, self.hp, self.double, self.pre = sd, hp_gen, double, prefix

    def encode(self, x: Tensor, k: Optional[int] = None) -> Tuple[Tensor, Tensor]:
        style = style_encoder(self.sd, self.pre + "enc_style.", x, self.hp)
        name = ("enc%d_content." % k) if self.double else "enc_content."
        return content_encoder(self.sd, self.pre + name, x, self.hp), style

    def decode(self, content: Tensor, style: Tensor, k: Optional[int] = None) -> Tensor:
        m = ("mlp%d." % k) if self.double else "mlp."
        d = ("dec%d." % k) if self.double else "dec."
        params = mlp(self.sd, self.pre + m, style, act=self.hp["activ"])
        return decoder(self.sd, self.pre + d, content, params, self.hp)


def dis_forward(sd: State, pre: str, x: Tensor, hp_dis: dict) -> List[Tensor]:
    """MsImageDis.forward, networks.py:72-77 with _make_net, networks.py:39-70
    (norm 'none' -- the configs on the hot path -- or 'in'; the first layer never has a norm, networks.py:41-53)."""
    assert hp_dis["norm"] in ("none", "in")
    norm_fn = instance_norm if hp_dis["norm"] == "in" else None
    outs = []
    for s in range(hp_dis["num_scales"]):
        h = x
        p = pre + "cnns.%d." % s
        for l in range(hp_dis["n_layer"]):
            h = conv_block(h, sd[p + "%d.conv.weight" % l], sd[p + "%d.conv.bias" % l], 2, 1,
                           hp_dis["pad_type"], norm_fn if l > 0 else None, hp_dis["activ"])
        l = hp_dis["n_layer"]
        outs.append(F.conv2d(h, sd[p + "%d.weight" % l], sd[p + "%d.bias" % l]))
        x = avgpool_3s2(x)
    return outs


def dis_loss_d(sd, pre, fake, real, hp_dis) -> Tensor:
    """calc_dis_loss LSGAN branch, networks.py:79-91."""
    assert hp_dis["gan_type"] == "lsgan"
    loss = 0
    for o0, o1 in zip(dis_forward(sd, pre, fake, hp_dis), dis_forward(sd, pre, real, hp_dis)):
        loss = loss + torch.mean(o0 ** 2) + torch.mean((o1 - 1) ** 2)
    return loss


def dis_loss_g(sd, pre, fake, hp_dis) -> Tensor:
    """calc_gen_loss LSGAN branch, networks.py:103-109."""
    assert hp_dis["gan_type"] == "lsgan"
    loss = 0
    for o0 in dis_forward(sd, pre, fake, hp_dis):
        loss = loss + torch.mean((o0 - 1) ** 2)
    return loss


def _abs(d: Tensor) -> Tensor:
    """|d|, or -- under the KinkMasks test hook -- d times the recorded sign (the same value up to twice the rounding
    noise of an element within that noise of 0, and the recorded branch of the derivative there)."""
    if KINK_MASKS is not None and KINK_MASKS.l1_signs is not None:
        return torch.where(KINK_MASKS.take_l1(d), d, -d)
    return torch.abs(d)


def l1(a: Tensor, b: Tensor) -> Tensor:
    """recon_criterion, trainer.py:279-290."""
    return torch.mean(_abs(a - b))


def l1_masked(a: Tensor, b: Tensor, mask: Tensor) -> Tensor:
    """recon_criterion_mask, trainer.py:292-305: mean over ALL elements of |(a-b)(1-mask)|."""
    return torch.mean(_abs((a - b) * (1 - mask)))


# --------------------------------------------------------------------------------------
# state construction
# --------------------------------------------------------------------------------------
def gen_param_shapes(hp_gen: dict, input_dim: int, double: bool) -> Dict[str, Tuple[int, ...]]:
    """Parameter (not buffer) names and shapes of AdaINGen / AdaINGen_double in
    registration order (networks.py:172-209 / 265-323)."""
    dim, sdim, mdim = hp_gen["dim"], hp_gen["style_dim"], hp_gen["mlp_dim"]
    nd, nr = hp_gen["n_downsample"], hp_gen["n_res"]
    out: Dict[str, Tuple[int, ...]] = {}

    def conv(name, co, ci, k):
        out[name + ".weight"] = (co, ci, k, k)
        out[name + ".bias"] = (co,)

    # style encoder
    p = "enc_style.model."
    conv(p + "0.conv", dim, input_dim, 7)
    d = dim
    for i in (1, 2):
        conv(p + "%d.conv" % i, 2 * d, d, 4)
        d *= 2
    for i in (3, 4):
        conv(p + "%d.conv" % i, d, d, 4)
    conv(p + "6", sdim, d, 1)

    def content(pre):
        conv(pre + "model.0.conv", dim, input_dim, 7)
        d = dim
        for i in range(1, nd + 1):
            conv(pre + "model.%d.conv" % i, 2 * d, d, 4)
            d *= 2
        for i in range(nr):
            for j in (0, 1):
                conv(pre + "model.%d.model.%d.model.%d.conv" % (nd + 1, i, j), d, d, 3)
        return d

    def dec(pre, d):
        for i in range(nr):
            for j in (0, 1):
                conv(pre + "model.0.model.%d.model.%d.conv" % (i, j), d, d, 3)
        idx = 2
        for _ in range(nd):
            # registration order inside Conv2dBlock: norm (gamma, beta) before conv
            out[pre + "model.%d.norm.gamma" % idx] = (d // 2,)
            out[pre + "model.%d.norm.beta" % idx] = (d // 2,)
            conv(pre + "model.%d.conv" % idx, d // 2, d, 5)
            d //= 2
            idx += 2
        conv(pre + "model.%d.conv" % (idx - 1), input_dim, d, 7)

    def mlp_(pre, n_out):
        out[pre + "model.0.fc.weight"] = (mdim, sdim)
        out[pre + "model.0.fc.bias"] = (mdim,)
        out[pre + "model.1.fc.weight"] = (mdim, mdim)
        out[pre + "model.1.fc.bias"] = (mdim,)
        out[pre + "model.2.fc.weight"] = (n_out, mdim)
        out[pre + "model.2.fc.bias"] = (n_out,)

    n_ad = num_adain_params(hp_gen)
    if double:
        dd = content("enc1_content.")
        content("enc2_content.")
        dec("dec1.", dd)
        dec("dec2.", dd)
        mlp_("mlp1.", n_ad)
        mlp_("mlp2.", n_ad)
    else:
        dd = content("enc_content.")
        dec("dec.", dd)
        mlp_("mlp.", n_ad)
    return out


def dis_param_shapes(hp_dis: dict, input_dim: int) -> Dict[str, Tuple[int, ...]]:
    """Parameter names/shapes of MsImageDis (networks.py:39-70)."""
    out: Dict[str, Tuple[int, ...]] = {}
    for s in range(hp_dis["num_scales"]):
        d = hp_dis["dim"]
        ci = input_dim
        for l in range(hp_dis["n_layer"]):
            co = d if l == 0 else 2 * d
            out["cnns.%d.%d.conv.weight" % (s, l)] = (co, ci, 4, 4)
            out["cnns.%d.%d.conv.bias" % (s, l)] = (co,)
            ci = co
            if l > 0:
                d *= 2
        out["cnns.%d.%d.weight" % (s, hp_dis["n_layer"])] = (1, ci, 1, 1)
        out["cnns.%d.%d.bias" % (s, hp_dis["n_layer"])] = (1,)
    return out


def fill_det(name: str, shape, scale: Optional[float] = None, dtype=torch.float32) -> Tensor:
    """Deterministic, RNG-library-independent fill used by fixtures AND tests:
    values come from a 64-bit LCG seeded by a hash of the tensor name, mapped to
    (-1, 1), scaled like kaiming fan-in for weights.  Not a reference function."""
    import numpy as np
    n = int(np.prod(shape)) if len(shape) else 1
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        v = (idx * np.uint64(6364136223846793005) + np.uint64(h | 1))
        v ^= v >> np.uint64(33)
        v *= np.uint64(0xFF51AFD7ED558CCD)
        v ^= v >> np.uint64(33)
    u = (v >> np.uint64(11)).astype(np.float64) / float(1 << 53)  # [0,1)
    x = 2.0 * u - 1.0
    if scale is None:
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            scale = math.sqrt(3.0) * math.sqrt(2.0 / fan_in)  # uniform with kaiming variance
        elif name.endswith("gamma"):
            x = 0.5 * x + 0.5  # (0,1) like LayerNorm's uniform_() init (networks.py:859)
            scale = 1.0
        else:
            scale = 0.1
    return torch.from_numpy((x * scale).reshape(shape)).to(dtype)


def make_state(shapes: Dict[str, Tuple[int, ...]], tag: str, dtype=torch.float32) -> State:
    return {k: fill_det(tag + k, s, dtype=dtype) for k, s in shapes.items()}


# --------------------------------------------------------------------------------------
# trainer step
# --------------------------------------------------------------------------------------
def adam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                beta1: float, beta2: float, eps: float, wd: float) -> None:
    """torch.optim.Adam as the reference configures it (trainer.py:109-120: L2-coupled
    weight_decay, amsgrad off, eps default 1e-8); in-place on p, m, v."""
    g = g + wd * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def extraadam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float,
                     beta2: float, eps: float, wd: float) -> Tensor:
    """ExtraAdam.update, extraadam.py:121-168 (amsgrad off): advances m, v in place and returns the
    displacement u = -lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)."""
    g = g + wd * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    step_size = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    return -step_size * m / (v.sqrt() + eps)


class ExtraAdamState:
    """Extragradient.extrapolation / .step, extraadam.py:30-84, over a list of tensors."""

    def __init__(self, params, lr, betas, weight_decay, eps=1e-8):
        self.params, self.lr, self.betas, self.wd, self.eps = params, lr, betas, weight_decay, eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.step_count = 0
        self.copy = []

    def _updates(self, grads, lr):
        self.step_count += 1
        return [None if g is None else
                extraadam_update(p, g, m, v, self.step_count, lr, self.betas[0], self.betas[1], self.eps, self.wd)
                for p, g, m, v in zip(self.params, grads, self.m, self.v)]

    def extrapolation(self, grads, lr=None):
        us = self._updates(grads, self.lr if lr is None else lr)
        if not self.copy:
            self.copy = [p.detach().clone() for p in self.params]
        with torch.no_grad():
            for p, u in zip(self.params, us):
                if u is not None:
                    p.add_(u)

    def step(self, grads, lr=None):
        if not self.copy:
            raise RuntimeError("Need to call extrapolation before calling step.")
        us = self._updates(grads, self.lr if lr is None else lr)
        with torch.no_grad():
            for p, c, u in zip(self.params, self.copy, us):
                if u is not None:
                    p.copy_(c + u)
        self.copy = []


def step_lr(base_lr: float, n_sched_steps: int, hp: dict) -> float:
    """get_scheduler, utils.py:1066-1090: StepLR(step_size, gamma); after n scheduler
    steps lr = base * gamma ** (n // step_size).  'constant' or absent -> base."""
    if hp.get("lr_policy", "constant") == "constant":
        return base_lr
    return base_lr * hp["gamma"] ** (n_sched_steps // hp["step_size"])


class OracleTrainer:
    """Restatement of MUNIT_Trainer's hot path (trainer.py:29-127, 336-561, 1133-1186,
    1326-1335) with aux losses at weight 0.  Holds leaf tensors in self.gen / self.dis_a
    / self.dis_b (for gen_state 0: self.gen holds 'a.' and 'b.' prefixed states)."""

    def __init__(self, hp: dict, gen: State, dis_a: State, dis_b: State):
        self.hp = hp
        self.gen_state = hp["gen_state"]
        self.guided = hp["guided"]
        self.recon_mask = hp["recon_mask"] == 1
        self.gen, self.dis_a, self.dis_b = gen, dis_a, dis_b
        for t in list(gen.values()) + list(dis_a.values()) + list(dis_b.values()):
            t.requires_grad_(True)
        self.opt = {}
        for grp, params in (("gen", list(gen.values())),
                            ("dis", list(dis_a.values()) + list(dis_b.values()))):
            self.opt[grp] = dict(params=params, step=0,
                                 m=[torch.zeros_like(p) for p in params],
                                 v=[torch.zeros_like(p) for p in params])
        self.sched_steps = 0
        self.iterations = 0
        self.extra = "extra" in hp.get("optimizer", "adam")
        if self.extra:
            for grp in ("gen", "dis"):
                self.opt[grp]["extra"] = ExtraAdamState(self.opt[grp]["params"], hp["lr"], (hp["beta1"], hp["beta2"]),
                                                        hp["weight_decay"])
        self.losses: Dict[str, Tensor] = {}

    # trainer.py:1326-1335
    def update_learning_rate(self):
        self.sched_steps += 1

    def _lr(self):
        return step_lr(self.hp["lr"], self.sched_steps, self.hp)

    def _views(self):
        if self.gen_state == 1:
            g = GenView(self.gen, self.hp["gen"], True)
            return (g, 1), (g, 2)
        ga = GenView(self.gen, self.hp["gen"], False, "a.")
        gb = GenView(self.gen, self.hp["gen"], False, "b.")
        return (ga, None), (gb, None)

    def _opt_step(self, grp: str, grads):
        o = self.opt[grp]
        if self.extra:  # trainer.py:252-268: extrapolate on even iterations, step on odd ones
            with torch.no_grad():
                if self.iterations % 2 == 0:
                    o["extra"].extrapolation(grads, self._lr())
                else:
                    o["extra"].step(grads, self._lr())
            o["step"] = o["extra"].step_count
            o["m"], o["v"] = o["extra"].m, o["extra"].v
            return
        o["step"] += 1
        with torch.no_grad():
            for p, g, m, v in zip(o["params"], grads, o["m"], o["v"]):
                if g is None:
                    continue
                adam_update(p, g, m, v, o["step"], self._lr(), self.hp["beta1"], self.hp["beta2"],
                            1e-8, self.hp["weight_decay"])

    def gen_losses(self, x_a, x_b, mask_a=None, mask_b=None, s_a=None, s_b=None) -> Dict[str, Tensor]:
        """trainer.py:366-558 (loss graph only)."""
        hp = self.hp
        (ga, ka), (gb, kb) = self._views()
        c_a, s_a_p = ga.encode(x_a, ka)
        c_b, s_b_p = gb.encode(x_b, kb)
        x_a_recon = ga.decode(c_a, s_a_p, ka)
        x_b_recon = gb.decode(c_b, s_b_p, kb)
        if self.guided == 1:
            sa_use, sb_use = s_a_p, s_b_p
        else:
            sa_use, sb_use = s_a, s_b
        x_ba = ga.decode(c_b, sa_use, ka)
        x_ab = gb.decode(c_a, sb_use, kb)
        # the adversarial terms (trainer.py:515-516) are evaluated here, as soon as the translations exist: the value is
        # the same wherever they stand, and the HIP trainer issues them at this point (tests replay its activation order)
        adv_a = dis_loss_g(self.dis_a, "", x_ba, hp["dis"])
        adv_b = dis_loss_g(self.dis_b, "", x_ab, hp["dis"])
        c_b_recon, s_a_recon = ga.encode(x_ba, ka)
        c_a_recon, s_b_recon = gb.encode(x_ab, kb)
        L: Dict[str, Tensor] = {}
        L["loss_gen_recon_x_a"] = l1(x_a_recon, x_a)
        L["loss_gen_recon_x_b"] = l1(x_b_recon, x_b)
        L["loss_gen_recon_s_a"] = l1(s_a_recon, sa_use)
        L["loss_gen_recon_s_b"] = l1(s_b_recon, sb_use)
        L["loss_gen_recon_c_a"] = l1(c_a_recon, c_a)
        L["loss_gen_recon_c_b"] = l1(c_b_recon, c_b)
        zero = torch.zeros((), dtype=x_a.dtype)
        if hp["recon_x_cyc_w"] > 0:
            x_aba = ga.decode(c_a_recon, s_a_p, ka)
            x_bab = gb.decode(c_b_recon, s_b_p, kb)
            if self.recon_mask:
                L["loss_gen_cycrecon_x_a"] = l1_masked(x_aba, x_a, mask_a)
                L["loss_gen_cycrecon_x_b"] = l1_masked(x_bab, x_b, mask_b)
            else:
                L["loss_gen_cycrecon_x_a"] = l1(x_aba, x_a)
                L["loss_gen_cycrecon_x_b"] = l1(x_bab, x_b)
        else:
            L["loss_gen_cycrecon_x_a"] = zero
            L["loss_gen_cycrecon_x_b"] = zero
        L["loss_gen_adv_a"] = adv_a
        L["loss_gen_adv_b"] = adv_b
        L["loss_gen_total"] = (
            hp["gan_w"] * L["loss_gen_adv_a"] + hp["gan_w"] * L["loss_gen_adv_b"]
            + hp["recon_x_w"] * L["loss_gen_recon_x_a"] + hp["recon_s_w"] * L["loss_gen_recon_s_a"]
            + hp["recon_c_w"] * L["loss_gen_recon_c_a"] + hp["recon_x_w"] * L["loss_gen_recon_x_b"]
            + hp["recon_s_w"] * L["loss_gen_recon_s_b"] + hp["recon_c_w"] * L["loss_gen_recon_c_b"]
            + hp["recon_x_cyc_w"] * L["loss_gen_cycrecon_x_a"]
            + hp["recon_x_cyc_w"] * L["loss_gen_cycrecon_x_b"])
        self._last = dict(x_ba=x_ba, x_ab=x_ab, x_a_recon=x_a_recon, x_b_recon=x_b_recon,
                          c_a=c_a, c_b=c_b, s_a_prime=s_a_p, s_b_prime=s_b_p)
        return L

    def gen_update(self, x_a, x_b, mask_a=None, mask_b=None, s_a=None, s_b=None, apply=True):
        """trainer.py:365-561.  Returns the generator gradients (list aligned with
        self.opt['gen']['params'])."""
        L = self.gen_losses(x_a, x_b, mask_a, mask_b, s_a, s_b)
        grads = torch.autograd.grad(L["loss_gen_total"], self.opt["gen"]["params"], allow_unused=True)
        self.losses.update({k: v.detach() for k, v in L.items()})
        if apply:
            self._opt_step("gen", grads)
        return grads

    def dis_losses(self, x_a, x_b, s_a=None, s_b=None) -> Dict[str, Tensor]:
        """trainer.py:1146-1184."""
        hp = self.hp
        (ga, ka), (gb, kb) = self._views()
        with torch.no_grad():  # x_ba / x_ab are detached at trainer.py:1178-1179
            c_a, s_a_p = ga.encode(x_a, ka)
            c_b, s_b_p = gb.encode(x_b, kb)
            if self.guided == 1:
                x_ba = ga.decode(c_b, s_a_p, ka)
                x_ab = gb.decode(c_a, s_b_p, kb)
            else:
                x_ba = ga.decode(c_b, s_a, ka)
                x_ab = gb.decode(c_a, s_b, kb)
        L = {}
        L["loss_dis_a"] = dis_loss_d(self.dis_a, "", x_ba, x_a, hp["dis"])
        L["loss_dis_b"] = dis_loss_d(self.dis_b, "", x_ab, x_b, hp["dis"])
        L["loss_dis_total"] = hp["gan_w"] * L["loss_dis_a"] + hp["gan_w"] * L["loss_dis_b"]
        return L

    def dis_update(self, x_a, x_b, s_a=None, s_b=None, apply=True):
        """trainer.py:1145-1186."""
        L = self.dis_losses(x_a, x_b, s_a, s_b)
        grads = torch.autograd.grad(L["loss_dis_total"], self.opt["dis"]["params"], allow_unused=True)
        self.losses.update({k: v.detach() for k, v in L.items()})
        if apply:
            self._opt_step("dis", grads)
        return grads


def synthetic_batch(batch: int, size, seed: int = 7, dtype=torch.float32):
    """SURVEY.md section 8d synthetic inputs: x = 2U-1, mask = (U > 0.5).  `size`: an int (square crops) or (height, width)."""
    h, w = (size, size) if isinstance(size, int) else size
    g = torch.Generator().manual_seed(seed)
    x_a = (2 * torch.rand(batch, 3, h, w, generator=g) - 1).to(dtype)
    x_b = (2 * torch.rand(batch, 3, h, w, generator=g) - 1).to(dtype)
    m_a = (torch.rand(batch, 1, h, w, generator=g) > 0.5).to(dtype)
    m_b = (torch.rand(batch, 1, h, w, generator=g) > 0.5).to(dtype)
    return x_a, x_b, m_a, m_b


def default_hp(size=256, batch: int = 1, gen_state: int = 1) -> dict:
    """configs/config_256.yaml with the benchmark overrides of SURVEY.md section 8d
    (semantic_w 0, adaptation adv/dfeat 0).  `size`: an int or (crop_image_height, crop_image_width)."""
    size_h, size_w = (size, size) if isinstance(size, int) else size
    size = min(size_h, size_w)
    return dict(
        batch_size=batch, weight_decay=1e-4, beta1=0.5, beta2=0.999, init="kaiming", lr=1e-4,
        lr_policy="step", step_size=100000, gamma=0.5, gan_w=3, recon_x_w=12, recon_s_w=1,
        recon_c_w=2, recon_x_cyc_w=12, vgg_w=0,
        adaptation=dict(full_adaptation=0, output_classifier_lambda=0, output_adv_lambda=0,
                        output_classif_freq=1, adv_lambda=0, dfeat_lambda=0, classif_frequency=15,
                        sem_seg_lambda=0),
        semantic_w=0, recon_mask=1, domain_adv_w=0, recon_synth_w=0, gen_state=gen_state, guided=1,
        gen=dict(dim=64, mlp_dim=256, style_dim=16, activ="relu", n_downsample=2, n_res=4, pad_type="reflect"),
        dis=dict(dim=64, norm="none", activ="lrelu", n_layer=4, gan_type="lsgan", num_scales=3, pad_type="reflect"),
        ratio_disc_gen=5, input_dim_a=3, input_dim_b=3, display_size=8, optimizer="adam",
        crop_image_height=size_h, crop_image_width=size_w, new_size=size, num_workers=0)
