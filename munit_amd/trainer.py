"""MUNIT_Trainer: drop-in for the reference's scripts/trainer.py hot path on MI355X.

Same constructor / gen_update / dis_update / update_learning_rate / forward / sample / save /
resume surface and `self.loss_*` attribute names, so scripts/train.py
can swap `from trainer import MUNIT_Trainer` for `from munit_amd.trainer import MUNIT_Trainer`.

What is different underneath (none of it changes results beyond fp32 rounding):
  * every layer runs as a HIP kernel (munit_amd.ops); nothing falls back to torch ops;
  * generator and discriminator parameters, their gradients and the Adam moments live in flat
    fp32 buffers: backward-weight accumulates straight into the flat gradient, Adam is one
    fused kernel per optimizer, and data-parallel training is ONE all-reduce per update over
    RCCL/xGMI (torch.distributed backend "nccl") of that flat buffer -- 109 MB (G) / 66 MB (D);
  * work the reference does and then throws away is skipped: D weight-gradients inside
    gen_update (zeroed by dis_update, trainer.py:1145), the autograd graph of the generator
    inside dis_update (x_ba / x_ab are detached at trainer.py:1178-1179);
  * loss scalars stay on the device; nothing synchronises the host inside an update.

Aux losses outside the AdaINGen + MsImageDis path (VGG, semantic segmentation, domain
classifiers, synthetic pairs) raise NotImplementedError when their weight is non-zero.
"""
import os
import warnings

import torch
import torch.nn as nn
from torch.optim import Optimizer

from . import ops
from .networks import AdaINGen, AdaINGen_double, ContentEncoder, InstanceNorm2d, MsImageDis, _ApplyRefreshesImages
from .utils import get_model_list, get_scheduler, normalize_config, weights_init


class FusedAdam(Optimizer):
    """torch.optim.Adam semantics (L2-coupled weight decay, amsgrad off; trainer.py:109-120)
    executed as one HIP kernel over a flat parameter buffer.  state_dict()/load_state_dict()
    speak torch.optim.Adam's format so `optimizer.pt` files interchange with the reference."""

    def __init__(self, params, lr, betas, weight_decay, eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._plist = [p for g in self.param_groups for p in g["params"]]
        self._step = 0
        self.flat_p = self.flat_g = self.flat_m = self.flat_v = None
        self._views = None
        # prepared weight images (ops._prepared): items registered on first use, refreshed in one launch per step
        self._prep_items = []
        self._prep_table = None

    # ---- flat storage -----------------------------------------------------------------
    @staticmethod
    def _view(flat, off, p):
        n = p.numel()
        v = flat[off:off + n]
        if p.dim() == 4:
            o, i, kh, kw = p.shape
            return v.view(o, kh, kw, i).permute(0, 3, 1, 2)  # logical OIHW, memory [O][KH][KW][I]
        return v.view(p.shape)

    def bind(self, device):
        """(Re)build the flat buffers on `device`, re-pointing every parameter at its slice."""
        offs, total = [], 0
        for p in self._plist:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned
        flat_p = torch.zeros(total, dtype=torch.float32, device=device)
        flat_g = torch.zeros(total, dtype=torch.float32, device=device)
        flat_m = torch.zeros(total, dtype=torch.float32, device=device)
        flat_v = torch.zeros(total, dtype=torch.float32, device=device)
        if self.flat_m is not None and self.flat_m.numel() == total:
            flat_m.copy_(self.flat_m)
            flat_v.copy_(self.flat_v)
        views = []
        with torch.no_grad():
            for p, off in zip(self._plist, offs):
                if p.dtype != torch.float32:
                    raise RuntimeError("munit_amd: parameters must stay float32")
                pv = self._view(flat_p, off, p)
                pv.copy_(p.data)
                p.data = pv
                gv = self._view(flat_g, off, p)
                p._munit_grad = gv
                p.grad = gv
                p._munit_prep = {}       # the storage moved: every prepared image is void
                p._munit_opt = self
                views.append((self._view(flat_m, off, p), self._view(flat_v, off, p)))
        self.flat_p, self.flat_g, self.flat_m, self.flat_v = flat_p, flat_g, flat_m, flat_v
        self._views = views
        self._offsets = offs                  # start of every parameter in the flat buffers (16-byte aligned slots)
        self._total = total
        self._prep_items, self._prep_table = [], None

    def ranges_of(self, select):
        """Maximal contiguous [start, end) element ranges of the flat buffers covered by the parameters p with
        select(p) true, in buffer order (alignment gaps between two selected parameters belong to the range)."""
        out = []
        ends = self._offsets[1:] + [self._total]
        for p, a, b in zip(self._plist, self._offsets, ends):
            if not select(p):
                continue
            if out and out[-1][1] == a:
                out[-1][1] = b
            else:
                out.append([a, b])
        return [tuple(r) for r in out]

    # ---- prepared weight images ----------------------------------------------------------
    def register_prepared(self, item):
        """Called by ops._prepared the first time a layer needs a re-laid-out image of one of this optimizer's
        weights (backward-data transpose, sub-pixel phase merge)."""
        self._prep_items.append(item)
        self._prep_table = None

    def refresh_prepared(self):
        """Rebuild every registered image from the current weights: one kernel launch on the current stream.
        The reference re-derives nothing here (cuDNN transposes inside its kernels); this build re-laid each
        weight 4-6 times per step inside the conv calls before (452 launches), now once."""
        n = len(self._prep_items)
        if n == 0 or not self.flat_p.is_cuda:
            return
        if self._prep_table is None:
            import ctypes
            from ._lib import PrepItem
            arr = (PrepItem * n)(*self._prep_items)
            host = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))),
                                    dtype=torch.uint8)
            self._prep_table = host.to(self.flat_p.device)
        ops.prepare_weights_batch(self._prep_table, n)

    def invalidate_prepared(self):
        """Call after ANY write to the weights that is not an optimizer step, a load_state_dict or an in-place op on the
        parameter / the flat buffer itself (those are seen through the tensors' version counters): `p.data.copy_(...)`,
        EMA through `.data`, a raw kernel.  Rebuilds every registered image from the current weights."""
        self.refresh_prepared()

    # ---- optimizer API ----------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self._step += 1
        ops.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, g["lr"], g["betas"][0], g["betas"][1],
                      g["eps"], g["weight_decay"], self._step)
        self.refresh_prepared()

    def state_dict(self):
        state = {}
        for i, (m, v) in enumerate(self._views):
            state[i] = {"step": torch.tensor(float(self._step)),
                        "exp_avg": m.detach().clone(memory_format=torch.contiguous_format),
                        "exp_avg_sq": v.detach().clone(memory_format=torch.contiguous_format)}
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d["params"] = list(range(len(g["params"])))
            groups.append(d)
        return {"state": state if self._step > 0 else {}, "param_groups": groups}

    def load_state_dict(self, sd):
        st = sd["state"]
        with torch.no_grad():
            for i, (m, v) in enumerate(self._views):
                if i in st:
                    m.copy_(st[i]["exp_avg"])
                    v.copy_(st[i]["exp_avg_sq"])
                    self._step = int(float(st[i]["step"]))
        for g, lg in zip(self.param_groups, sd["param_groups"]):
            for k, val in lg.items():
                if k != "params":
                    g[k] = val


class FusedExtraAdam(FusedAdam):
    """ExtraAdam (scripts/extraadam.py:14-168) on the flat buffers: `extrapolation()` saves the
    parameters (first call since the last step), moves them by the Adam-style update and `step()`
    applies the update computed at the extrapolated point to the saved parameters."""

    def __init__(self, params, lr, betas, weight_decay, eps=1e-8):
        super().__init__(params, lr, betas, weight_decay, eps)
        self.flat_saved = None
        self._has_copy = False

    def bind(self, device):
        old = self.flat_saved
        super().bind(device)
        self.flat_saved = torch.zeros_like(self.flat_p)
        if old is not None and old.numel() == self.flat_saved.numel():
            self.flat_saved.copy_(old)

    def _update(self, mode):
        g = self.param_groups[0]
        self._step += 1
        ops.extraadam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.flat_saved, g["lr"],
                           g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._step, mode)
        self.refresh_prepared()

    @torch.no_grad()
    def extrapolation(self):
        self._update(1 if self._has_copy else 0)
        self._has_copy = True

    @torch.no_grad()
    def step(self, closure=None):
        if not self._has_copy:
            raise RuntimeError("Need to call extrapolation before calling step.")
        self._update(2)
        self._has_copy = False


FORCE_ALLREDUCE = bool(os.environ.get("MUNIT_FORCE_ALLREDUCE"))
# Data-parallel exchange of the generator gradient in two parts (SURVEY.md section 8e "launch on a side stream to overlap
# with the remaining backward"): see GradExchange.  MUNIT_NO_OVERLAP_EXCHANGE=1 = one all-reduce after backward (A/B, tests).
OVERLAP_EXCHANGE = not os.environ.get("MUNIT_NO_OVERLAP_EXCHANGE")


def dp_world():
    """World size of the data-parallel exchange, 0 when no exchange is to be issued (no process group, or a single rank
    without MUNIT_FORCE_ALLREDUCE)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size()
    return world if (world > 1 or FORCE_ALLREDUCE) else 0


def comm_stream(dev):
    """The communication stream of a device (one per process and device): early gradient exchanges, the deferred
    discriminator exchange + optimizer step."""
    key = (dev.type, dev.index)
    if key not in GradExchange._comm:
        GradExchange._comm[key] = torch.cuda.Stream(device=dev)
    return GradExchange._comm[key]


class GradExchange:
    """All-reduce (mean) of a flat gradient buffer in stages that start INSIDE the backward pass.

    Every generator weight is used by several sub-graph calls (4 encodes, 6 decodes per gen_update), so a gradient is final only
    when the LAST backward pass through its module has run.  Backward replays the forward in reverse:
      stage 1 -- the decoders' and the MLPs' last use is the pair of decodes right after the first encodes, so their gradients
        (about half of the flat buffer) are final when the gradients of the first encodes' outputs (c_a, c_b, s_a', s_b'; with
        guided == 0 also of the sampled styles, whose MLP passes have no other tensor downstream) have been formed, while the
        backward of those two encodes (about 8 % of the backward pass) is still to come;
      stage 2 -- inside that last stretch the content encoders' residual trunks (7/8 of an encoder's weights) finish first: their
        gradients are final when the gradient of the tensor ENTERING the trunk has been formed in both first encodes, while
        the down-sampling layers, the 7x7 first layers and the style encoder are still running.
    `arm(ranges, tensors)` declares a stage: it hooks the tensors, and when the last hook fires a communication stream is
    ordered behind everything enqueued so far on the producing streams (every kernel that writes the ranges has been enqueued
    by then: a node's backward-weight launch precedes the gradient it returns) and the ranges are all-reduced from it,
    asynchronously.  `finish()` (after backward, on the caller's stream) reduces what no stage covered, waits for the stages
    and scales by 1 / world.

    Every rank issues the same collectives in the same order (the autograd engine's order is a function of the graph; ranges
    of a stage in buffer order, then the rest in buffer order).  Each element is summed over the ranks exactly once either
    way; with two ranks the result is bitwise the single all-reduce's (tests), with more ranks it can differ in the last bit
    where the ring's chunk boundaries move -- as between any two bucket layouts."""

    _comm = {}

    def __init__(self, flat, world, streams=()):
        self.flat, self.world = flat, world
        self.streams = [s for s in streams if s is not None]
        self.stages, self.works = [], []

    @property
    def fired(self):
        return any(st["fired"] for st in self.stages)

    def arm(self, ranges, tensors):
        """A stage without a tensor that requires a gradient (or without elements) is dropped: finish() sends its ranges."""
        ranges = [tuple(r) for r in ranges if r[1] > r[0]]
        ts = [t for t in tensors if torch.is_tensor(t) and t.requires_grad]
        if not ranges or not ts:
            return self
        for a, b in ranges:
            for st in self.stages:
                assert all(b <= c or d <= a for c, d in st["ranges"]), "GradExchange: stages must not overlap"
        st = {"ranges": ranges, "pending": len(ts), "fired": False}
        self.stages.append(st)
        for t in ts:
            t.register_hook(lambda grad, st=st: self._hook(st))
        return self

    def _hook(self, st):
        st["pending"] -= 1
        if st["pending"] == 0:
            self._launch(st)
        return None

    def _launch(self, st):
        import torch.distributed as dist
        st["fired"] = True
        if self.flat.is_cuda:
            comm = comm_stream(self.flat.device)
            for s in self.streams:            # everything that writes the stage's ranges has been enqueued on these by now
                ops.stream_wait(comm, s)
            with torch.cuda.stream(comm):
                for a, b in st["ranges"]:
                    self.works.append(dist.all_reduce(self.flat[a:b], async_op=True))
        else:
            for a, b in st["ranges"]:
                self.works.append(dist.all_reduce(self.flat[a:b], async_op=True))

    @property
    def rest(self):
        """What no fired stage covers, in buffer order."""
        done = sorted(r for st in self.stages if st["fired"] for r in st["ranges"])
        rest, pos = [], 0
        for a, b in done:
            if a > pos:
                rest.append((pos, a))
            pos = b
        if pos < self.flat.numel():
            rest.append((pos, self.flat.numel()))
        return rest

    def finish(self):
        """Call on the stream that holds the complete gradient (after backward and the joins of the producing streams)."""
        import torch.distributed as dist
        for a, b in self.rest:                # a stage whose hooks never fired (no tensor required grad) goes now
            dist.all_reduce(self.flat[a:b])
        for w in self.works:
            w.wait()                          # on a device: the caller's stream waits for the communication stream
        if self.world > 1:
            if self.flat.is_cuda:
                ops.scale_(self.flat, 1.0 / self.world)
            else:
                self.flat.mul_(1.0 / self.world)

BRANCH_STREAMS = not os.environ.get("MUNIT_NO_BRANCH_STREAMS")   # bench.py clears it while it times single kernels


class _Branches:
    """The a / b halves of an update on two streams.  Every stage of gen_update / dis_update consists of two
    independent halves (encode x_a | encode x_b, decode ... ), so running them on two streams lets the head and
    tail of one kernel overlap the body of another and fills the small grids of the discriminators.  Autograd
    replays each node on the stream it was recorded on, so the backward pass forks the same way.
    `share` is the cross-over point: both streams wait for each other and the tensors named become usable (and
    allocator-safe) on both.  MUNIT_NO_BRANCH_STREAMS=1 keeps everything on the caller's stream."""
    _streams = {}

    def __init__(self, dev):
        self.enabled = torch.cuda.is_available() and BRANCH_STREAMS
        if not self.enabled:
            return
        self.main = torch.cuda.current_stream(dev)
        key = (dev.type, dev.index)
        if key not in _Branches._streams:
            _Branches._streams[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        self.s = _Branches._streams[key]
        for st in self.s:
            ops.stream_wait(st, self.main)

    def run(self, k, fn):
        if not self.enabled:
            return fn()
        with torch.cuda.stream(self.s[k]):
            return fn()

    def adopt(self, *tensors):
        """Tensors that already exist on the caller's stream become usable on both branches: the fork in __init__ ordered
        the branches behind the caller's stream, so only the allocator has to be told."""
        if not self.enabled:
            return
        for t in tensors:
            if torch.is_tensor(t):
                t.record_stream(self.s[0])
                t.record_stream(self.s[1])

    def share(self, *tensors):
        if not self.enabled:
            return
        # cross-over = join into the caller's stream + fork again (the caller's stream is idle meanwhile)
        for st in self.s:
            ops.stream_wait(self.main, st)
        for st in self.s:
            ops.stream_wait(st, self.main)
        for t in tensors:
            if torch.is_tensor(t):
                t.record_stream(self.s[0])
                t.record_stream(self.s[1])

    def join(self, *tensors):
        if not self.enabled:
            return
        for st in self.s:
            ops.stream_wait(self.main, st)
        for t in tensors:
            if torch.is_tensor(t):
                t.record_stream(self.main)


class MUNIT_Trainer(_ApplyRefreshesImages, nn.Module):
    def __init__(self, hyperparameters):
        super(MUNIT_Trainer, self).__init__()
        hyperparameters = normalize_config(hyperparameters)
        lr = hyperparameters["lr"]
        self.gen_state = hyperparameters["gen_state"]
        self.guided = hyperparameters["guided"]
        self.newsize = hyperparameters["crop_image_height"]
        self.semantic_w = hyperparameters["semantic_w"] > 0
        self.recon_mask = hyperparameters["recon_mask"] == 1
        self.dann_scheduler = None
        self.full_adaptation = hyperparameters["adaptation"]["full_adaptation"] == 1
        self.hyperparameters = hyperparameters
        self.iterations = 0
        # build extension (no reference counterpart): `precision: bf16` = BASELINE.json config #3 -- the conv /
        # linear contractions multiply bf16-rounded operands with fp32 accumulation; tensors, norm statistics,
        # losses and the optimizer stay fp32.  Default: the reference's fp32 arithmetic.
        # `precision: bf16s` = bf16 STORAGE on top of that: the content encoders write bf16 activations, and the content
        # code carries the type through the decoders up to the fp32 image head; norm statistics, AdaIN parameters,
        # weights, gradients of weights, optimizer state, losses, the style encoder and the discriminators stay fp32.
        self.precision = hyperparameters.get("precision", "f32")
        ops.set_compute(self.precision)
        # build extension, opt-in (`reuse_dis_forward: 1`): dis_update and the gen_update that follows it in the same
        # iteration (scripts/train.py:182-187) run the SAME generator forward -- encode x_a, encode x_b, decode x_ba,
        # decode x_ab -- on the same images with the same generator weights (only the discriminators step in between);
        # the reference computes it twice (trainer.py:1146-1179 under no autograd, then trainer.py:366-390).  With this
        # switch dis_update keeps that forward and its autograd tape and gen_update continues from it when it is handed
        # the very same tensors: 11 % fewer multiply-accumulates per step; the forward values (and therefore every
        # loss) are the same numbers, the gradients agree to fp32 summation order (the kept nodes are older on the autograd
        # tape, so the uses of a shared weight accumulate in another order).  Off by default: the benchmark's step is
        # defined on the reference's sequence of computations.
        self.reuse_dis_forward = bool(hyperparameters.get("reuse_dis_forward", 0))
        self._fwd_cache = None
        self.fwd_reused = False     # whether the last gen_update continued from dis_update's forward
        self.last_exchange = None   # the GradExchange of the last data-parallel gen_update (introspection)

        optimizer = FusedExtraAdam if "extra" in hyperparameters["optimizer"] else FusedAdam  # trainer.py:41-45
        self.domain_classif_ab = hyperparameters.get("domain_adv_w", 0) > 0
        self.use_classifier_sr = hyperparameters["adaptation"]["dfeat_lambda"] > 0
        self.train_seg = hyperparameters["adaptation"]["sem_seg_lambda"] > 0
        self.use_output_classifier_sr = hyperparameters["adaptation"]["output_classifier_lambda"] > 0
        self._check_aux(hyperparameters)

        if self.gen_state == 0:
            self.gen_a = AdaINGen(hyperparameters["input_dim_a"], hyperparameters["gen"])
            self.gen_b = AdaINGen(hyperparameters["input_dim_b"], hyperparameters["gen"])
        elif self.gen_state == 1:
            self.gen = AdaINGen_double(hyperparameters["input_dim_a"], hyperparameters["gen"])
        else:
            raise ValueError("self.gen_state unknown value: %r" % (self.gen_state,))
        if self.precision == "bf16s":
            if hyperparameters["gen"]["dim"] % 64 != 0:
                raise ValueError("munit_amd: precision 'bf16s' keeps the generator's activations as bf16 tensors, whose kernels "
                                 "work on 64-channel rows: gen.dim must be a multiple of 64 (got %d)" % hyperparameters["gen"]["dim"])
            for m in self.modules():
                if isinstance(m, ContentEncoder):
                    m.store_dtype = torch.bfloat16
        self.dis_a = MsImageDis(hyperparameters["input_dim_a"], hyperparameters["dis"])
        self.dis_b = MsImageDis(hyperparameters["input_dim_b"], hyperparameters["dis"])
        self.instancenorm = InstanceNorm2d(512)
        self.style_dim = hyperparameters["gen"]["style_dim"]

        # fixed display noise (trainer.py:93-95); kept on the host until a device is known
        display_size = int(hyperparameters["display_size"])
        self.s_a = torch.randn(display_size, self.style_dim, 1, 1)
        self.s_b = torch.randn(display_size, self.style_dim, 1, 1)

        beta1, beta2 = hyperparameters["beta1"], hyperparameters["beta2"]
        dis_params = list(self.dis_a.parameters()) + list(self.dis_b.parameters())
        if self.gen_state == 0:
            gen_params = list(self.gen_a.parameters()) + list(self.gen_b.parameters())
        else:
            gen_params = list(self.gen.parameters())
        self.dis_opt = optimizer([p for p in dis_params if p.requires_grad], lr=lr, betas=(beta1, beta2),
                                 weight_decay=hyperparameters["weight_decay"])
        self.gen_opt = optimizer([p for p in gen_params if p.requires_grad], lr=lr, betas=(beta1, beta2),
                                 weight_decay=hyperparameters["weight_decay"])
        self.dis_scheduler = get_scheduler(self.dis_opt, hyperparameters)
        self.gen_scheduler = get_scheduler(self.gen_opt, hyperparameters)

        # Network weight initialization (trainer.py:124-127)
        self.apply(weights_init(hyperparameters["init"]))
        self.dis_a.apply(weights_init("gaussian"))
        self.dis_b.apply(weights_init("gaussian"))

        self._consts = {}
        # deferred discriminator exchange + step (data parallel, _defer_dis_step)
        self._dis_pending, self._dis_event, self._dis_waited = None, None, set()
        for d in (self.dis_a, self.dis_b):
            d.__dict__["before_forward"] = self._wait_dis     # every path into MsImageDis.forward, direct calls included
            d.register_state_dict_pre_hook(self._wait_dis)
        self._bind(torch.device("cpu"))

    # ------------------------------------------------------------------------------------
    @staticmethod
    def _check_aux(hp):
        bad = []
        if hp.get("vgg_w", 0) > 0:
            bad.append("vgg_w")
        if hp.get("semantic_w", 0) > 0:
            bad.append("semantic_w")
        if hp.get("domain_adv_w", 0) > 0:
            bad.append("domain_adv_w")
        for k in ("adv_lambda", "dfeat_lambda", "sem_seg_lambda", "output_classifier_lambda", "output_adv_lambda"):
            if hp["adaptation"].get(k, 0) > 0:
                bad.append("adaptation." + k)
        if bad:
            raise NotImplementedError(
                "munit_amd covers the AdaINGen + MsImageDis training step only; set these weights to 0 "
                "(they need external checkpoints / models outside the hot path): " + ", ".join(bad))

    def _bind(self, device):
        self.dis_opt.bind(device)
        self.gen_opt.bind(device)
        self._consts = {}
        # flat-gradient ranges of the decoders and MLPs (final before the first encodes' backward: GradExchange)
        gens = [self.gen] if self.gen_state == 1 else [self.gen_a, self.gen_b]
        early = {id(p) for g in gens for n, p in g.named_parameters() if n.startswith("dec") or n.startswith("mlp")}
        self._early_ranges = self.gen_opt.ranges_of(lambda p: id(p) in early)
        # ... and of the content encoders' residual trunks (final before the first encodes' down-sampling layers run backward)
        trunk = {id(p) for k in (1, 2) for p in self._content_enc(k).model[-1].parameters()}
        self._trunk_ranges = self.gen_opt.ranges_of(lambda p: id(p) in trunk)

    def _apply(self, fn, *args, **kwargs):
        self._settle_dis()
        out = super()._apply(fn, *args, **kwargs)
        p = next(self.dis_a.parameters())
        self._bind(p.device)
        self.s_a = self.s_a.to(p.device)
        self.s_b = self.s_b.to(p.device)
        return out

    def _const(self, value, device):
        key = (float(value), device)
        t = self._consts.get(key)
        if t is None:
            t = torch.full((), float(value), dtype=torch.float32, device=device)
            self._consts[key] = t
        return t

    # ---- optimizer steps (trainer.py:252-268) -----------------------------------------
    def dis_opt_step(self):
        """ExtraAdam extrapolates on even iterations and steps on odd ones (trainer.py:252-259)."""
        if "extra" in self.hyperparameters["optimizer"] and (self.iterations % 2 == 0):
            self.dis_opt.extrapolation()
        else:
            self.dis_opt.step()

    def gen_opt_step(self):
        if "extra" in self.hyperparameters["optimizer"] and (self.iterations % 2 == 0):
            self.gen_opt.extrapolation()
        else:
            self.gen_opt.step()

    # ---- criteria (trainer.py:279-305) ------------------------------------------------
    def recon_criterion(self, input, target):
        return ops.l1_mean(input, target)

    def recon_criterion_mask(self, input, target, mask):
        return ops.l1_mean(input, target, mask)

    # ---- generator dispatch -----------------------------------------------------------
    def _enc(self, x, k):
        if self.gen_state == 1:
            return self.gen.encode(x, k)
        return (self.gen_a if k == 1 else self.gen_b).encode(x)

    def _content_enc(self, k):
        if self.gen_state == 1:
            return self.gen.enc1_content if k == 1 else self.gen.enc2_content
        return (self.gen_a if k == 1 else self.gen_b).enc_content

    def _enc_keep(self, x, k):
        """encode, also returning the tensor that enters the content encoder's residual trunk (GradExchange stage 2)."""
        enc = self._content_enc(k)
        enc.keep_trunk_in = True
        try:
            c, s = self._enc(x, k)
            t = enc.trunk_in
        finally:
            enc.keep_trunk_in, enc.trunk_in = False, None
        return c, s, t

    def _dec(self, c, s, k):
        if self.gen_state == 1:
            return self.gen.decode(c, s, k)
        return (self.gen_a if k == 1 else self.gen_b).decode(c, s)

    def forward(self, x_a, x_b):
        """trainer.py:307-334."""
        ops.set_compute(self.precision)
        self.eval()
        with torch.no_grad():
            s_a, s_b = self.s_a.to(x_a.device), self.s_b.to(x_a.device)
            c_a, _ = self._enc(x_a, 1)
            c_b, _ = self._enc(x_b, 2)
            x_ba = self._dec(c_b, s_a, 1)
            x_ab = self._dec(c_a, s_b, 2)
        self.train()
        return x_ab, x_ba

    @staticmethod
    def _all_reduce_mean(flat):
        """Data-parallel exchange: one all-reduce (RCCL over xGMI) of the flat gradient.  MUNIT_FORCE_ALLREDUCE=1
        issues it at world size 1 too (a 1-GPU box then exercises the RCCL path; the result is unchanged)."""
        import torch.distributed as dist
        world = dp_world()
        if world:
            dist.all_reduce(flat)
            if world > 1:
                if flat.is_cuda:
                    ops.scale_(flat, 1.0 / world)
                else:
                    flat.mul_(1.0 / world)

    # ---- gen_update (trainer.py:336-561) -----------------------------------------------
    def gen_update(self, x_a, x_b, hyperparameters, mask_a=None, mask_b=None, comet_exp=None, synth=False,
                   semantic_gt_a=None, semantic_gt_b=None):
        ops.set_compute(self.precision)
        hp = hyperparameters
        if synth and hp.get("recon_synth_w", 0) > 0:
            raise NotImplementedError("munit_amd: synthetic-pair reconstruction loss is outside the hot path")
        self._check_aux(normalize_config(hp))
        self.gen_opt.zero_grad()
        # the reference draws these even when guided == 1 leaves them unused (trainer.py:366-367)
        s_a = torch.randn(x_a.size(0), self.style_dim, 1, 1)
        s_b = torch.randn(x_b.size(0), self.style_dim, 1, 1)
        dev = x_a.device
        fwd_key = self._fwd_key(x_a, x_b)      # of the caller's tensors (the layout conversion below may copy)
        x_a, x_b = ops.nhwc(x_a), ops.nhwc(x_b)

        world = dp_world()
        overlap = bool(world and OVERLAP_EXCHANGE)      # stages of the gradient exchange start inside backward (GradExchange)
        d_params = list(self.dis_a.parameters()) + list(self.dis_b.parameters())
        for p in d_params:  # D weight gradients made here would be discarded (trainer.py:1145)
            p.requires_grad_(False)
        try:
            br = _Branches(dev)
            br.adopt(x_a, x_b, mask_a, mask_b)
            cached, self._fwd_cache = self._fwd_cache, None
            reuse = (cached is not None and self.guided == 1 and cached[0] == fwd_key)
            self.fwd_reused = reuse
            trunk_a = trunk_b = None
            if reuse:      # the forward dis_update just ran on these tensors with these generator weights
                c_a, s_a_prime, c_b, s_b_prime, x_ba_kept, x_ab_kept = cached[1]
                br.adopt(c_a, s_a_prime, c_b, s_b_prime, x_ba_kept, x_ab_kept)
            elif overlap:  # also keep the tensors entering the residual trunks: their gradients open stage 2 of the exchange
                c_a, s_a_prime, trunk_a = br.run(0, lambda: self._enc_keep(x_a, 1))
                c_b, s_b_prime, trunk_b = br.run(1, lambda: self._enc_keep(x_b, 2))
            else:
                c_a, s_a_prime = br.run(0, lambda: self._enc(x_a, 1))
                c_b, s_b_prime = br.run(1, lambda: self._enc(x_b, 2))
            del cached
            x_a_recon = br.run(0, lambda: self._dec(c_a, s_a_prime, 1))
            x_b_recon = br.run(1, lambda: self._dec(c_b, s_b_prime, 2))
            if self.guided == 0:
                s_a_use, s_b_use = s_a.to(dev), s_b.to(dev)
                if overlap:
                    # The MLP passes over the sampled styles have no hooked tensor downstream: without a gradient of their own
                    # the exchange's stage 1 would rely on the autograd engine running them before the first decodes' nodes (it
                    # does -- higher sequence numbers first -- but that is an implementation detail).  With requires_grad the
                    # styles' gradients are formed at the END of those MLP passes' backward and stage 1 waits for them.  Costs a
                    # (B, style_dim) backward-data; no parameter gradient changes.
                    s_a_use.requires_grad_(True)
                    s_b_use.requires_grad_(True)
            elif self.guided == 1:
                s_a_use, s_b_use = s_a_prime, s_b_prime
            else:
                raise ValueError("self.guided unknown value: %r" % (self.guided,))
            br.share(c_a, c_b, s_a_use, s_b_use)
            if reuse:
                x_ba, x_ab = x_ba_kept, x_ab_kept
            else:
                x_ba = br.run(0, lambda: self._dec(c_b, s_a_use, 1))
                x_ab = br.run(1, lambda: self._dec(c_a, s_b_use, 2))
            # adversarial terms (trainer.py:515-516), issued as soon as the translations exist.  (Giving the two
            # discriminators streams of their own, so that their small grids run beside the encoder / decoder kernels
            # that follow, measured 161.0-161.4 ms per step against 159.9-160.2 on the branch streams: not kept.)
            self.loss_gen_adv_a = br.run(0, lambda: self.dis_a.calc_gen_loss(x_ba))
            self.loss_gen_adv_b = br.run(1, lambda: self.dis_b.calc_gen_loss(x_ab))
            c_b_recon, s_a_recon = br.run(0, lambda: self._enc(x_ba, 1))
            c_a_recon, s_b_recon = br.run(1, lambda: self._enc(x_ab, 2))
            br.share(c_a_recon, c_b_recon)
            cyc = hp["recon_x_cyc_w"] > 0
            x_aba = br.run(0, lambda: self._dec(c_a_recon, s_a_prime, 1)) if cyc else None
            x_bab = br.run(1, lambda: self._dec(c_b_recon, s_b_prime, 2)) if cyc else None

            self.loss_gen_recon_x_a = br.run(0, lambda: self.recon_criterion(x_a_recon, x_a))
            self.loss_gen_recon_x_b = br.run(1, lambda: self.recon_criterion(x_b_recon, x_b))
            self.loss_gen_recon_s_a = br.run(0, lambda: self.recon_criterion(s_a_recon, s_a_use))
            self.loss_gen_recon_s_b = br.run(1, lambda: self.recon_criterion(s_b_recon, s_b_use))
            self.loss_gen_recon_c_a = br.run(1, lambda: self.recon_criterion(c_a_recon, c_a))
            self.loss_gen_recon_c_b = br.run(0, lambda: self.recon_criterion(c_b_recon, c_b))
            self.loss_gen_recon_synth = 0
            if cyc:
                if self.recon_mask:
                    if mask_a is None or mask_b is None:
                        raise ValueError("recon_mask == 1 needs mask_a and mask_b of shape (B,1,H,W)")
                    self.loss_gen_cycrecon_x_a = br.run(0, lambda: self.recon_criterion_mask(x_aba, x_a, mask_a))
                    self.loss_gen_cycrecon_x_b = br.run(1, lambda: self.recon_criterion_mask(x_bab, x_b, mask_b))
                else:
                    self.loss_gen_cycrecon_x_a = br.run(0, lambda: self.recon_criterion(x_aba, x_a))
                    self.loss_gen_cycrecon_x_b = br.run(1, lambda: self.recon_criterion(x_bab, x_b))
            else:
                self.loss_gen_cycrecon_x_a = 0
                self.loss_gen_cycrecon_x_b = 0
            br.join(self.loss_gen_recon_x_a, self.loss_gen_recon_x_b, self.loss_gen_recon_s_a, self.loss_gen_recon_s_b,
                    self.loss_gen_recon_c_a, self.loss_gen_recon_c_b, self.loss_gen_cycrecon_x_a,
                    self.loss_gen_cycrecon_x_b, self.loss_gen_adv_a, self.loss_gen_adv_b)
        finally:
            for p in d_params:
                p.requires_grad_(True)
        self.loss_gen_vgg_a = self.loss_gen_vgg_b = 0
        self.loss_sem_seg = self.domain_adv_loss = self.loss_classifier_sr = self.loss_output_classifier_sr = 0

        pairs = [(hp["gan_w"], self.loss_gen_adv_a), (hp["gan_w"], self.loss_gen_adv_b),
                 (hp["recon_x_w"], self.loss_gen_recon_x_a), (hp["recon_s_w"], self.loss_gen_recon_s_a),
                 (hp["recon_c_w"], self.loss_gen_recon_c_a), (hp["recon_x_w"], self.loss_gen_recon_x_b),
                 (hp["recon_s_w"], self.loss_gen_recon_s_b), (hp["recon_c_w"], self.loss_gen_recon_c_b)]
        if cyc:
            pairs += [(hp["recon_x_cyc_w"], self.loss_gen_cycrecon_x_a),
                      (hp["recon_x_cyc_w"], self.loss_gen_cycrecon_x_b)]
        self.loss_gen_total = ops.weighted_sum([t.detach() for _, t in pairs], [w for w, _ in pairs])
        live = [(w, t) for w, t in pairs if w != 0 and t.requires_grad]
        xch = None
        if overlap:
            # the decoder / MLP half of the flat gradient is exchanged while the first encodes' backward still runs, the content
            # encoders' residual trunks while their down-sampling and first layers (and the style encoder) still run
            streams = [torch.cuda.current_stream(dev)] if dev.type == "cuda" else []
            if br.enabled:
                streams += list(br.s)
            if dev.type == "cuda" and ops.SIDE_STREAM_WGRAD:
                streams.append(ops._side_stream(dev))
            xch = GradExchange(self.gen_opt.flat_g, world, streams)
            xch.arm(self._early_ranges, [c_a, c_b, s_a_prime, s_b_prime] + ([s_a_use, s_b_use] if self.guided == 0 else []))
            if trunk_a is not None and trunk_b is not None:
                xch.arm(self._trunk_ranges, [trunk_a, trunk_b])
            self.last_exchange = xch     # introspection (tests): which stages fired
        torch.autograd.backward([t for _, t in live], [self._const(w, dev) for w, _ in live])
        br.join()                        # the backward pass ran on the branch streams its nodes were recorded on
        ops.join_side_streams()          # backward-weight kernels run on a side stream
        if xch is not None:
            xch.finish()
        else:
            self._all_reduce_mean(self.gen_opt.flat_g)
        self.gen_opt_step()
        self._log(comet_exp, ("loss_gen_adv_a", "loss_gen_adv_b", "loss_gen_recon_x_a", "loss_gen_recon_s_a",
                              "loss_gen_recon_c_a", "loss_gen_recon_x_b", "loss_gen_recon_s_b",
                              "loss_gen_recon_c_b", "loss_gen_cycrecon_x_a", "loss_gen_cycrecon_x_b",
                              "loss_gen_total"))

    # ---- dis_update (trainer.py:1133-1190) ---------------------------------------------
    def dis_update(self, x_a, x_b, hyperparameters, comet_exp=None):
        ops.set_compute(self.precision)
        hp = hyperparameters
        self._settle_dis()
        self.dis_opt.zero_grad()
        s_a = torch.randn(x_a.size(0), self.style_dim, 1, 1)
        s_b = torch.randn(x_b.size(0), self.style_dim, 1, 1)
        dev = x_a.device
        fwd_key = self._fwd_key(x_a, x_b)
        x_a, x_b = ops.nhwc(x_a), ops.nhwc(x_b)
        br = _Branches(dev)
        br.adopt(x_a, x_b)
        keep = self.reuse_dis_forward and self.guided == 1     # guided 0: the two updates draw different styles
        self._fwd_cache = None
        with torch.set_grad_enabled(keep):  # without `keep` the generator graph would never be back-propagated here
            c_a, s_a_prime = br.run(0, lambda: self._enc(x_a, 1))
            c_b, s_b_prime = br.run(1, lambda: self._enc(x_b, 2))
            if self.guided == 0:
                s_a_use, s_b_use = s_a.to(dev), s_b.to(dev)
            elif self.guided == 1:
                s_a_use, s_b_use = s_a_prime, s_b_prime
            else:
                raise ValueError("self.guided unknown value: %r" % (self.guided,))
            br.share(c_a, c_b, s_a_use, s_b_use)
            x_ba = br.run(0, lambda: self._dec(c_b, s_a_use, 1))
            x_ab = br.run(1, lambda: self._dec(c_a, s_b_use, 2))
        if keep:
            self._fwd_cache = (fwd_key, (c_a, s_a_prime, c_b, s_b_prime, x_ba, x_ab))
        self.loss_dis_a = br.run(0, lambda: self.dis_a.calc_dis_loss(x_ba.detach(), x_a))
        self.loss_dis_b = br.run(1, lambda: self.dis_b.calc_dis_loss(x_ab.detach(), x_b))
        br.join(self.loss_dis_a, self.loss_dis_b)
        self.loss_dis_total = ops.weighted_sum([self.loss_dis_a.detach(), self.loss_dis_b.detach()],
                                               [hp["gan_w"], hp["gan_w"]])
        w = self._const(hp["gan_w"], dev)
        torch.autograd.backward([self.loss_dis_a, self.loss_dis_b], [w, w])
        br.join()
        ops.join_side_streams()
        if dp_world() and OVERLAP_EXCHANGE and dev.type == "cuda":
            self._defer_dis_step(dev)
        else:
            self._all_reduce_mean(self.dis_opt.flat_g)
            self.dis_opt_step()
        self._log(comet_exp, ("loss_dis_b", "loss_dis_a"))

    # ---- data parallel: the discriminator exchange beside the next generator forward ---------------------------------
    def _defer_dis_step(self, dev):
        """The all-reduce of the discriminator gradient (66 MB), its 1 / world scale and the discriminator's optimizer step
        (Adam + the refresh of the prepared weight images) are enqueued on the communication stream, behind the backward pass
        that just ended on the caller's stream.  The caller's stream goes on: in the reference's iteration the next thing
        is gen_update (scripts/train.py:182-187), whose first ~15 ms -- two encodes, four decodes -- touch no discriminator
        weight, so the exchange runs beside them.  Whatever reads or writes the discriminators next waits for the recorded event
        (`_wait_dis`: the top of MsImageDis.forward, dis_update, save / resume / state_dict)."""
        import torch.distributed as dist
        comm = comm_stream(dev)
        ops.stream_wait(comm, torch.cuda.current_stream(dev))
        world = dp_world()
        with torch.cuda.stream(comm):
            dist.all_reduce(self.dis_opt.flat_g)          # RCCL is ordered behind, and hands back to, the CURRENT stream = comm
            if world > 1:
                ops.scale_(self.dis_opt.flat_g, 1.0 / world)
            self.dis_opt_step()
            ev = self._dis_event
            if ev is None:
                ev = self._dis_event = torch.cuda.Event()
            ev.record(comm)
        self._dis_pending = ev

    def _wait_dis(self, *_):
        """Order the current stream behind a deferred discriminator step (no-op when none is pending on this stream: every
        stream that reaches the discriminators waits once)."""
        ev = self._dis_pending
        if ev is None:
            return
        st = torch.cuda.current_stream(self.dis_opt.flat_p.device)
        key = st.cuda_stream
        if key in self._dis_waited:
            return
        st.wait_event(ev)
        self._dis_waited.add(key)

    def _settle_dis(self):
        """A new discriminator update (or a host-side reader) is about to start: the caller's stream waits, nothing pends."""
        if self._dis_pending is not None:
            self._wait_dis()
            self._dis_pending = None
        self._dis_waited = set()

    def _fwd_key(self, x_a, x_b):
        """Identity of a generator forward: the input tensors (storage, layout, in-place version) and the state of the
        generator weights (optimizer step count and the parameters' in-place versions)."""
        gp = self.gen_opt._plist
        return (x_a.data_ptr(), x_a._version, tuple(x_a.shape), tuple(x_a.stride()), x_b.data_ptr(), x_b._version,
                tuple(x_b.shape), tuple(x_b.stride()), self.gen_opt._step, self.gen_opt.flat_p.data_ptr(),
                sum(p._version for p in gp), self.training, ops.get_compute())

    def _log(self, comet_exp, names):
        if comet_exp is not None and self.iterations % 100 == 0:
            for n in names:
                v = getattr(self, n)
                comet_exp.log_metric(n, v.cpu().detach() if torch.is_tensor(v) else v)

    # ---- schedule (trainer.py:1326-1335) ------------------------------------------------
    def update_learning_rate(self):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # scheduler-before-optimizer order is the reference's (train.py:172)
            if self.dis_scheduler is not None:
                self.dis_scheduler.step()
            if self.gen_scheduler is not None:
                self.gen_scheduler.step()

    # ---- sampling (trainer.py:773-928, core outputs only) -------------------------------
    def sample_fid(self, x_a, x_b):
        """trainer.py:1087-1131: per-sample cross-domain translation a -> b with the style encoded from x_b
        (guided == 1); returns x_ab1 (trainer.py:1126-1131)."""
        ops.set_compute(self.precision)
        self.eval()
        x_ab1 = []
        with torch.no_grad():
            for i in range(x_a.size(0)):
                c_a, _ = self._enc(x_a[i:i + 1], 1)
                _, s_b_fake = self._enc(x_b[i:i + 1], 2)
                if self.guided == 1:
                    x_ab1.append(self._dec(c_a, s_b_fake, 2))
                else:
                    print("self.guided unknown value:", self.guided)
        x_ab1 = torch.cat(x_ab1)
        self.train()
        return x_ab1

    def sample(self, x_a, x_b):
        ops.set_compute(self.precision)
        self.eval()
        outs = [[] for _ in range(6)]
        with torch.no_grad():
            s_a1, s_b1 = self.s_a.to(x_a.device), self.s_b.to(x_a.device)
            s_a2 = torch.randn(x_a.size(0), self.style_dim, 1, 1).to(x_a.device)
            s_b2 = torch.randn(x_b.size(0), self.style_dim, 1, 1).to(x_a.device)
            for i in range(x_a.size(0)):
                xa, xb = x_a[i:i + 1], x_b[i:i + 1]
                c_a, s_a_fake = self._enc(xa, 1)
                c_b, s_b_fake = self._enc(xb, 2)
                outs[0].append(self._dec(c_a, s_a_fake, 1))
                outs[1].append(self._dec(c_b, s_b_fake, 2))
                if self.guided == 0:
                    outs[2].append(self._dec(c_b, s_a1[i:i + 1], 1))
                    outs[3].append(self._dec(c_b, s_a2[i:i + 1], 1))
                    outs[4].append(self._dec(c_a, s_b1[i:i + 1], 2))
                    outs[5].append(self._dec(c_a, s_b2[i:i + 1], 2))
                else:
                    outs[2].append(self._dec(c_b, s_a_fake, 1))
                    outs[3].append(self._dec(c_b, s_a_fake, 1))
                    outs[4].append(self._dec(c_a, s_b_fake, 2))
                    outs[5].append(self._dec(c_a, s_b_fake, 2))
        x_a_recon, x_b_recon, x_ba1, x_ba2, x_ab1, x_ab2 = (torch.cat(o) for o in outs)
        self.train()
        return x_a, x_a_recon, x_ab1, x_ab2, x_b, x_b_recon, x_ba1, x_ba2

    def sample_syn(self, x_a, x_b):
        """trainer.py:930-1085: line for line the body of `sample` under a second name (for synthetic-domain pairs)."""
        return self.sample(x_a, x_b)

    # ---- checkpoints (trainer.py:1337-1429) ---------------------------------------------
    @staticmethod
    def _plain(sd):
        return {k: v.detach().clone(memory_format=torch.contiguous_format).cpu() for k, v in sd.items()}

    def save(self, snapshot_dir, iterations):
        self._settle_dis()
        gen_name = os.path.join(snapshot_dir, "gen_%08d.pt" % (iterations + 1))
        dis_name = os.path.join(snapshot_dir, "dis_%08d.pt" % (iterations + 1))
        opt_name = os.path.join(snapshot_dir, "optimizer.pt")
        if self.gen_state == 0:
            torch.save({"a": self._plain(self.gen_a.state_dict()), "b": self._plain(self.gen_b.state_dict())},
                       gen_name)
        else:
            torch.save({"2": self._plain(self.gen.state_dict())}, gen_name)
        torch.save({"a": self._plain(self.dis_a.state_dict()), "b": self._plain(self.dis_b.state_dict())}, dis_name)
        torch.save({"gen": self.gen_opt.state_dict(), "dis": self.dis_opt.state_dict()}, opt_name)

    def resume(self, checkpoint_dir, hyperparameters):
        self._settle_dis()
        last_model_name = get_model_list(checkpoint_dir, "gen")
        state_dict = torch.load(last_model_name, map_location="cpu", weights_only=True)
        if self.gen_state == 0:
            self.gen_a.load_state_dict(state_dict["a"])
            self.gen_b.load_state_dict(state_dict["b"])
        else:
            self.gen.load_state_dict(state_dict["2"])
        iterations = int(last_model_name[-11:-3])
        last_model_name = get_model_list(checkpoint_dir, "dis")
        state_dict = torch.load(last_model_name, map_location="cpu", weights_only=True)
        self.dis_a.load_state_dict(state_dict["a"])
        self.dis_b.load_state_dict(state_dict["b"])
        state_dict = torch.load(os.path.join(checkpoint_dir, "optimizer.pt"), map_location="cpu", weights_only=True)
        self.dis_opt.load_state_dict(state_dict["dis"])
        self.gen_opt.load_state_dict(state_dict["gen"])
        self.dis_scheduler = get_scheduler(self.dis_opt, hyperparameters, iterations)
        self.gen_scheduler = get_scheduler(self.gen_opt, hyperparameters, iterations)
        self.gen_opt.invalidate_prepared()      # whatever wrote the weights, the prepared images follow them
        self.dis_opt.invalidate_prepared()
        print("Resume from iteration %d" % iterations)
        return iterations
