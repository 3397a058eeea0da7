"""CPU (no GPU): the C-ABI library loads and exports every declared symbol, the host-side
mirror of the reference interface (module tree, state_dict keys, init RNG stream, config
defaults, LR schedule, optimizer state format) and the data-parallel exchange over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from oracle import munit_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return True


def test_library_exports_every_declared_symbol(built):
    from munit_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "munit_hip.h")).read()
    declared = set(re.findall(r"\b(munit_[a-z0-9_]+)\s*\(", header))
    declared -= {"munit_stream_t"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.munit_version() >= 1


def test_host_arg_checks_without_gpu(built):
    """Argument validation happens on the host before any launch."""
    from ctypes import byref, c_int
    from munit_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc(1, 4, 4, 8, 8, 3, 3, 1, 4, 1, 0, 0, 0.0)  # reflect pad 4 >= H
    ho, wo = c_int(), c_int()
    assert lib.munit_conv2d_out_hw(byref(d), byref(ho), byref(wo)) != 0
    assert b"reflect" in lib.munit_last_error()
    d = _lib.ConvDesc(2, 16, 16, 64, 128, 4, 4, 2, 1, 1, 0, 0, 0.0)
    assert lib.munit_conv2d_out_hw(byref(d), byref(ho), byref(wo)) == 0 and (ho.value, wo.value) == (8, 8)
    d = _lib.ConvDesc(2, 16, 16, 256, 128, 5, 5, 1, 2, 1, 1, 0, 0.0)
    assert lib.munit_conv2d_out_hw(byref(d), byref(ho), byref(wo)) == 0 and (ho.value, wo.value) == (32, 32)
    assert lib.munit_conv2d_dgrad_workspace_bytes(byref(d)) > 0
    assert lib.munit_conv2d_wgrad_workspace_bytes(byref(d)) > 0


def test_ops_refuse_cpu_tensors():
    from munit_amd import ops
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.instance_norm(torch.zeros(1, 4, 4, 4))


@pytest.mark.parametrize("gs", [1, 0])
def test_state_dict_layout_and_init_stream_match_reference(golden, gs):
    """Same seed -> same module tree, key order and initial weights as the reference's
    construction sequence (trainer.py:67-127 driven over the reference modules in
    make_golden.py)."""
    from munit_amd.trainer import MUNIT_Trainer
    meta, _ = golden
    g = meta["init_gs%d" % gs]
    hp = O.default_hp(64, 1, gs)
    torch.manual_seed(1234)
    tr = MUNIT_Trainer(hp)
    assert [k for k, _ in tr.named_parameters()] == g["keys"]
    assert list(tr.state_dict().keys()) == g["state_keys"]

    def dg(t):
        f = t.detach().double().reshape(-1)
        idx = torch.linspace(0, f.numel() - 1, 8).long()
        return [float(f.sum()), float(f.abs().sum()), float(f.norm())] + [float(v) for v in f[idx]]

    for (k, p), d in zip(tr.named_parameters(), g["digests"]):
        got = dg(p)
        assert max(abs(a - b) for a, b in zip(got, d)) <= 1e-6 * max(1.0, abs(d[1])), k
    assert max(abs(a - b) for a, b in zip(dg(tr.s_a), g["s_a"])) < 1e-6
    assert max(abs(a - b) for a, b in zip(dg(tr.s_b), g["s_b"])) < 1e-6


def test_param_counts_and_flat_buffers():
    from munit_amd.trainer import MUNIT_Trainer
    tr = MUNIT_Trainer(O.default_hp(64, 1, 1))
    assert sum(p.numel() for p in tr.gen.parameters()) == 27293590
    assert sum(p.numel() for p in tr.dis_a.parameters()) == 8271171
    # every parameter is a view into the optimizer's flat buffer, conv weights stored [O][KH][KW][I]
    w = tr.gen.enc_style.model[1].conv.weight
    assert w.shape == (128, 64, 4, 4) and w.stride() == (1024, 1, 256, 64)
    lo, hi = tr.gen_opt.flat_p.data_ptr(), tr.gen_opt.flat_p.data_ptr() + 4 * tr.gen_opt.flat_p.numel()
    for p in tr.gen.parameters():
        assert lo <= p.data_ptr() < hi and p.data_ptr() % 16 == 0
        assert p._munit_grad.shape == p.shape and p.grad is p._munit_grad
    tr2 = MUNIT_Trainer(O.default_hp(64, 1, 0))
    assert sum(p.numel() for p in tr2.gen_a.parameters()) == 15030291


def test_config_defaults_and_unsupported_losses(tmp_path):
    from munit_amd import get_config
    from munit_amd.trainer import MUNIT_Trainer
    import yaml
    hp = O.default_hp(64, 1, 1)
    del hp["adaptation"], hp["optimizer"]          # a stale config like config_HD.yaml
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(hp))
    conf = get_config(str(p))
    assert conf["optimizer"] == "adam" and conf["adaptation"]["adv_lambda"] == 0
    MUNIT_Trainer(conf)
    bad = O.default_hp(64, 1, 1)
    bad["semantic_w"] = 3
    with pytest.raises(NotImplementedError, match="semantic_w"):
        MUNIT_Trainer(bad)
    xa = O.default_hp(64, 1, 1)
    xa["optimizer"] = "extraadam"
    tr = MUNIT_Trainer(xa)
    assert type(tr.gen_opt).__name__ == "FusedExtraAdam"
    tr.iterations = 1
    with pytest.raises(RuntimeError, match="extrapolation"):
        tr.gen_opt_step()                           # step before any extrapolation, as in the reference


def test_lr_schedule_matches_reference_order():
    """scheduler stepped at the start of each iteration (train.py:172); StepLR(step_size, gamma)."""
    from munit_amd.trainer import MUNIT_Trainer
    hp = O.default_hp(64, 1, 1)
    hp["step_size"] = 2
    tr = MUNIT_Trainer(hp)
    lrs = []
    for _ in range(5):
        tr.update_learning_rate()
        lrs.append(tr.gen_opt.param_groups[0]["lr"])
    assert lrs == [O.step_lr(hp["lr"], n, hp) for n in range(1, 6)]
    assert tr.dis_opt.param_groups[0]["lr"] == lrs[-1]


def test_optimizer_state_dict_is_torch_adam_format():
    from munit_amd.trainer import MUNIT_Trainer
    tr = MUNIT_Trainer(O.default_hp(64, 1, 1))
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone(memory_format=torch.contiguous_format))
                            for p in tr.dis_opt._plist], lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-4)
    for p in ref.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    ref.step()
    mine = tr.dis_opt
    mine.load_state_dict(ref.state_dict())
    assert mine._step == 1
    sd = mine.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == len(mine._plist)
    for i, st in ref.state_dict()["state"].items():
        assert torch.equal(sd["state"][i]["exp_avg"], st["exp_avg"])
        assert sd["state"][i]["exp_avg"].is_contiguous()
    ref.load_state_dict(sd)  # and back


def test_checkpoint_names_and_key_layout(tmp_path):
    from munit_amd.trainer import MUNIT_Trainer
    hp = O.default_hp(64, 1, 1)
    tr = MUNIT_Trainer(hp)
    tr.save(str(tmp_path), 41)
    assert sorted(os.listdir(tmp_path)) == ["dis_00000042.pt", "gen_00000042.pt", "optimizer.pt"]
    sd = torch.load(tmp_path / "gen_00000042.pt", weights_only=True)
    assert list(sd) == ["2"] and sd["2"]["enc_style.model.0.conv.weight"].is_contiguous()
    tr2 = MUNIT_Trainer(hp)
    assert tr2.resume(str(tmp_path), hp) == 42
    for a, b in zip(tr.parameters(), tr2.parameters()):
        assert torch.equal(a, b)


def ckpt_ref_hp():
    """Hyper-parameters of the small geometry tests/golden/ckpt_ref was written for (make_golden_variants.py)."""
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_ref")
    exp = json.load(open(os.path.join(here, "expect.json")))
    hp = O.default_hp(32, 2, 1)
    hp["gen"], hp["dis"] = dict(exp["gen_cfg"]), dict(exp["dis_cfg"])
    return here, exp, hp


def test_resume_loads_a_checkpoint_written_by_the_reference_modules():
    """tests/golden/ckpt_ref holds gen_/dis_/optimizer.pt files produced by the REFERENCE's
    AdaINGen_double / MsImageDis `state_dict()` and torch.optim.Adam `state_dict()` in the reference's save format
    (scripts/trainer.py:1387-1429).  MUNIT_Trainer.resume must take them as they are (weights-only loader): key layout,
    iteration count from the file name, parameter values, Adam moments and step counter."""
    from munit_amd.trainer import MUNIT_Trainer
    from tests.test_oracle_golden import close_digest, dg
    here, exp, hp = ckpt_ref_hp()
    tr = MUNIT_Trainer(hp)
    assert list(tr.gen.state_dict().keys()) == exp["gen_keys"]
    assert list(tr.dis_a.state_dict().keys()) == exp["dis_keys"]
    assert tr.resume(here, hp) == exp["iterations"]
    for p, d in zip(tr.gen.parameters(), exp["gen_params"]):
        close_digest(dg(p), d, 1e-7)
    assert tr.gen_opt._step == int(exp["gen_opt_step"]) == 2 and tr.dis_opt._step == 2
    m0, _ = tr.gen_opt._views[0]
    _, v_last = tr.gen_opt._views[-1]
    close_digest(dg(m0), exp["gen_exp_avg0"], 1e-7)
    close_digest(dg(v_last), exp["gen_exp_avg_sq_last"], 1e-7)
    # and the files this build writes back carry the same keys and values (the reverse direction of the wire format)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        tr.save(d, 2)
        mine = torch.load(os.path.join(d, "gen_00000003.pt"), weights_only=True)["2"]
        theirs = torch.load(os.path.join(here, "gen_00000003.pt"), weights_only=True)["2"]
        assert list(mine) == list(theirs)
        for k in mine:
            assert mine[k].shape == theirs[k].shape and torch.equal(mine[k], theirs[k]), k
        o_mine = torch.load(os.path.join(d, "optimizer.pt"), weights_only=True)
        o_theirs = torch.load(os.path.join(here, "optimizer.pt"), weights_only=True)
        for which in ("gen", "dis"):
            assert set(o_mine[which]["state"]) == set(o_theirs[which]["state"])
            for i, st in o_theirs[which]["state"].items():
                assert torch.equal(o_mine[which]["state"][i]["exp_avg_sq"], st["exp_avg_sq"]), (which, i)
                assert float(o_mine[which]["state"][i]["step"]) == float(st["step"])


_DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from munit_amd.trainer import MUNIT_Trainer
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
r = dist.get_rank()
flat = torch.arange(10, dtype=torch.float32) * (r + 1)
MUNIT_Trainer._all_reduce_mean(flat)
assert torch.allclose(flat, torch.arange(10, dtype=torch.float32) * 1.5), flat
# bench.py's shard rule: rank r takes samples [r*B, (r+1)*B) of the global batch
import bench
xa, xb, ma, mb = bench.make_batch(2, 16, rank=r)
ga, _, _, _ = bench.make_batch(2, 16, rank=0), None, None, None
assert xa.shape == (2, 3, 16, 16)
g = [torch.zeros_like(xa) for _ in range(2)]
dist.all_gather(g, xa)
assert not torch.equal(g[0], g[1])          # ranks see different data
dist.barrier()
dist.destroy_process_group()
print("ok", r)
"""


def test_data_parallel_exchange_gloo_world2(tmp_path):
    """N > 1 path on CPU: mean all-reduce of the flat gradient + per-rank data shards."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(_DP_WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              cwd=ROOT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_XCH_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from munit_amd import trainer as T
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
r = dist.get_rank()
n = 1000
g = torch.Generator().manual_seed(50 + r)
grad = torch.randn(n, generator=g)                       # this rank's "flat gradient"
early = [(300, 520), (700, 1000)]                        # two ranges, as gen_state 0 has (decoder + MLP of gen_a and of gen_b)
# serial form: one all-reduce of the whole buffer
serial = grad.clone()
T.MUNIT_Trainer._all_reduce_mean(serial)
# overlapped form: stage 1 (the early ranges) goes out from a tensor hook in the middle of a backward pass, stage 2 (the "residual
# trunk" of the encoder) from a hook further down, the rest afterwards
flat = grad.clone()
enc_w = torch.ones(4, requires_grad=True)
trunk_w = torch.ones(4, requires_grad=True)
dec_w = torch.ones(4, requires_grad=True)
h = enc_w * 2.0                                           # "tensor entering the trunk": its gradient is formed after the trunk's
c = h * trunk_w                                           # "content code": its gradient is formed after the decoder's
fired_at = []
trunk = [(100, 260)]
xch = T.GradExchange(flat, T.dp_world()).arm(early, [c]).arm(trunk, [h])
trunk_w.register_hook(lambda g_: fired_at.append(("trunk", [st["fired"] for st in xch.stages])))
enc_w.register_hook(lambda g_: fired_at.append(("enc", [st["fired"] for st in xch.stages])))
loss = (dec_w * c).sum()
loss.backward()
# stage 1 was launched BEFORE the trunk's gradient existed, stage 2 BEFORE the encoder's first layer's
assert fired_at == [("trunk", [True, False]), ("enc", [True, True])], fired_at
assert xch.rest == [(0, 100), (260, 300), (520, 700)], xch.rest
xch.finish()
assert torch.equal(flat, serial), (flat - serial).abs().max()
# a stage whose tensors need no gradient is dropped: its ranges are exchanged in finish()
flat2 = grad.clone()
x2 = T.GradExchange(flat2, T.dp_world()).arm(early, [torch.ones(3)])
assert not x2.stages and not x2.fired and x2.rest == [(0, n)]
x2.finish()
assert torch.equal(flat2, serial)
# overlapping stages are a programming error
try:
    T.GradExchange(grad.clone(), T.dp_world()).arm(early, [c]).arm([(500, 600)], [c])
    raise SystemExit("overlap accepted")
except AssertionError:
    pass
other = [torch.zeros(n) for _ in range(2)]
dist.all_gather(other, grad)
assert torch.allclose(serial, (other[0] + other[1]) / 2)
dist.barrier()
dist.destroy_process_group()
print("ok", r)
"""


def test_overlapped_gradient_exchange_equals_the_serial_all_reduce_gloo_world2(tmp_path):
    """GradExchange (the decoder / MLP part of the flat generator gradient all-reduced from a hook inside backward, the
    rest after it) against the single all-reduce, two gloo ranks: bitwise equal, issued before the encoder gradient is formed."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "xch.py"
    script.write_text(_XCH_WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_early_exchange_ranges_cover_decoders_and_mlps():
    """The flat-buffer ranges handed to GradExchange are exactly the decoder and MLP parameters (gen_state 1: one contiguous
    tail of the buffer; gen_state 0: one range per generator), 16-byte aligned; stage 2's are exactly the residual trunks of the
    two content encoders (one range each), disjoint from stage 1's."""
    from munit_amd.trainer import MUNIT_Trainer
    for gs, n_ranges in ((1, 1), (0, 2)):
        tr = MUNIT_Trainer(O.default_hp(64, 1, gs))
        r = tr._early_ranges
        assert len(r) == n_ranges and all(a % 4 == 0 and b % 4 == 0 for a, b in r), r
        gens = [("", tr.gen)] if gs == 1 else [("a.", tr.gen_a), ("b.", tr.gen_b)]
        names = [pre + n for pre, g in gens for n, _ in g.named_parameters()]
        offs = tr.gen_opt._offsets
        for n, p, off in zip(names, tr.gen_opt._plist, offs):
            late = ".dec" in "." + n or ".mlp" in "." + n
            inside = any(a <= off and off + p.numel() <= b for a, b in r)
            assert inside == late, (n, off, r)
        if gs == 1:
            assert r[0][1] == tr.gen_opt._total
        t = tr._trunk_ranges
        assert len(t) == 2 and all(a % 4 == 0 and b % 4 == 0 for a, b in t), t
        for n, p, off in zip(names, tr.gen_opt._plist, offs):
            in_trunk = ("content.model.3." in n)          # config_256.yaml: two down-samplings -> the ResBlocks are model.3
            inside = any(a <= off and off + p.numel() <= b for a, b in t)
            assert inside == in_trunk, (n, off, t)
        assert all(b <= c or d <= a for a, b in t for c, d in r)
        # 4 blocks x 2 convs x (256 * 256 * 9 + 256) parameters per encoder
        assert all(b - a == 8 * (256 * 256 * 9 + 256) for a, b in t), t


def test_bench_self_launches_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` without a launcher must start the two ranks itself (children of a GPU-free parent),
    relay rank 0's single JSON line and return 0; --dry-run keeps it on the CPU (gloo rendezvous + all-reduce)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, cwd=ROOT, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2"
    # a launcher / --gpus mismatch is refused instead of silently running another world size
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, cwd=ROOT, env=env2, timeout=600)
    assert p.returncode != 0 and b"does not match" in p.stderr


def test_compute_mode_plumbing_without_gpu():
    """munit_conv_desc.compute / ops.set_compute: names, values and validation (no kernel is launched)."""
    import ctypes
    from munit_amd import _lib, ops
    assert _lib.COMPUTE == {"f32": 0, "bf16": 1, "f32x3": 2}
    fields = [f[0] for f in _lib.ConvDesc._fields_]
    assert fields[-4:] == ["slope", "compute", "in_dtype", "out_dtype"] and ctypes.sizeof(_lib.ConvDesc) == 16 * 4
    assert ops.get_compute() == "f32"
    ops.set_compute("f32x3")
    assert ops.get_compute() == "f32x3"
    ops.set_compute("f32")
    with pytest.raises(ValueError):
        ops.set_compute("fp16")
    lib = _lib.load()
    d = _lib.ConvDesc(1, 8, 8, 32, 32, 3, 3, 1, 1, 1, 0, 0, 0.0, 7, 0, 0)  # unknown compute mode
    ho, wo = ctypes.c_int(), ctypes.c_int()
    assert lib.munit_conv2d_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) != 0
    assert b"compute" in lib.munit_last_error()
    # bf16 storage plumbing: "bf16s" multiplies like "bf16"; tensor dtypes travel in the descriptor and are validated
    ops.set_compute("bf16s")
    assert ops.get_compute() == "bf16"
    ops.set_compute("f32")
    d = _lib.ConvDesc(1, 8, 8, 32, 64, 3, 3, 1, 1, 1, 0, 0, 0.0, 1, 1, 1)  # bf16 input with 32 channels: refused
    assert lib.munit_conv2d_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) != 0
    assert b"Cin % 64" in lib.munit_last_error()
    d = _lib.ConvDesc(2, 8, 8, 64, 128, 3, 3, 1, 1, 1, 0, 0, 0.0, 1, 1, 1)
    assert lib.munit_conv2d_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == 0
    # the forward of a bf16-input layer multiplies by a bf16 copy of the weights; backward-data by the bf16 transpose
    assert lib.munit_conv2d_prepared_weight_bytes(ctypes.byref(d), 0) == 128 * 9 * 64 * 2
    assert lib.munit_conv2d_prepared_weight_bytes(ctypes.byref(d), 1) == 128 * 9 * 64 * 2
    # fp32 3x3 / stride 1 / pad 1 with wide channels: both passes multiply by a Winograd image (16 frequencies per channel pair)
    d32 = _lib.ConvDesc(2, 8, 8, 64, 128, 3, 3, 1, 1, 1, 0, 0, 0.0, 0, 0, 0)
    assert lib.munit_conv2d_prepared_weight_bytes(ctypes.byref(d32), 0) == 16 * 64 * 128 * 4
    assert lib.munit_conv2d_prepared_weight_bytes(ctypes.byref(d32), 1) == 16 * 64 * 128 * 4
    for which, kind in ((0, 4), (1, 5)):      # MUNIT_PREP_WINOGRAD, MUNIT_PREP_WINOGRAD_DGRAD
        item = _lib.PrepItem()
        assert lib.munit_conv2d_prep_item(ctypes.byref(d32), which, None, None, ctypes.byref(item)) == 0 and item.kind == kind
    # odd extent: the direct kernels (forward: the weights as they are; backward-data: their transpose)
    dodd = _lib.ConvDesc(2, 9, 8, 64, 128, 3, 3, 1, 1, 1, 0, 0, 0.0, 0, 0, 0)
    assert lib.munit_conv2d_prepared_weight_bytes(ctypes.byref(dodd), 0) == 0
    assert lib.munit_conv2d_prepared_weight_bytes(ctypes.byref(dodd), 1) == 128 * 9 * 64 * 4
    # nearest-x2 + 5x5 reflect-pad-2: four merged 3x3 phase filters, as Winograd images when the source extent is even
    dup = _lib.ConvDesc(2, 8, 8, 128, 64, 5, 5, 1, 2, 1, 1, 0, 0.0, 0, 0, 0)
    item = _lib.PrepItem()
    assert lib.munit_conv2d_prep_item(ctypes.byref(dup), 0, None, None, ctypes.byref(item)) == 0 and item.kind == 6
    assert lib.munit_conv2d_prepared_weight_bytes(ctypes.byref(dup), 0) == 4 * 16 * 128 * 64 * 4


def test_winograd_algebra_of_the_kernels():
    """The identities conv_wino.hip is built on, in fp64 numpy: F(2x2,3x3) and F(3x3,2x2) forward, their backward-weight form
    dg = G^T [(A dY A^T) . (B^T d B)] G, the reflect fold of the 3x3 backward-data done inside the input patch, and the
    parity split of the 4x4 / stride 2 layer (forward over four input phases; backward-data as four 2-tap phase planes whose
    padded rows fold onto their neighbours)."""
    import numpy as np
    rng = np.random.default_rng(0)
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
    G3 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
    AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
    G2 = np.array([[1, 0], [.5, .5], [.5, -.5], [0, 1]], float)
    AT3 = np.array([[1, 1, 1, 0], [0, 1, -1, 0], [0, 1, 1, -1]], float)

    def corr(d, g, m):   # VALID correlation, m x m outputs
        r = g.shape[0]
        return np.array([[(d[i:i + r, j:j + r] * g).sum() for j in range(m)] for i in range(m)])

    d = rng.standard_normal((4, 4))
    for G, AT, r, m in ((G3, AT2, 3, 2), (G2, AT3, 2, 3)):
        g = rng.standard_normal((r, r))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        assert np.abs(y - corr(d, g, m)).max() < 1e-13
        dy = rng.standard_normal((m, m))
        dg = G.T @ ((AT.T @ dy @ AT) * (BT @ d @ BT.T)) @ G
        ref = np.array([[(dy * d[a:a + m, b:b + m]).sum() for b in range(r)] for a in range(r)])
        assert np.abs(dg - ref).max() < 1e-13

    # 1-D reflect-pad-1 3-tap layer, H = 8: dx = P^T C^T dy.  Kernel form: zero-padded correlation of dy with the flipped
    # filter, 2-output tiles over patches dy[2t-1 .. 2t+2]; top tile: patch row 3 += patch row 1, bottom tile: row 0 += row 2.
    H, w = 8, rng.standard_normal(3)
    x = rng.standard_normal(H)
    dyv = rng.standard_normal(H)
    xp = np.concatenate([[x[1]], x, [x[H - 2]]])
    dxp = np.zeros(H + 2)
    for o in range(H):
        dxp[o:o + 3] += dyv[o] * w
    dx_ref = dxp[1:H + 1].copy(); dx_ref[1] += dxp[0]; dx_ref[H - 2] += dxp[H + 1]
    dyz = np.concatenate([[0.0], dyv, [0.0, 0.0]])
    dx = np.zeros(H)
    for t in range(H // 2):
        p = dyz[2 * t:2 * t + 4].copy()                      # dy rows 2t-1 .. 2t+2
        if t == 0: p[3] += p[1]
        if t == H // 2 - 1: p[0] += p[2]
        for i in range(2):
            dx[2 * t + i] = sum(p[i + s] * w[2 - s] for s in range(3))
    assert np.abs(dx - dx_ref).max() < 1e-13
    assert abs(float((xp[0:3] * w).sum()) - float(sum(xp[k] * w[k] for k in range(3)))) < 1e-13   # (orientation of w)

    # 1-D 4-tap / stride 2 / reflect-pad-1 layer, H = 8 -> 4 outputs: forward = sum over the input parities r of 2-tap
    # correlations over phase plane X_r[q] = xp[2q + r]; backward-data = parity planes dxp[2u + par] = sum_a dy[u-a] w[2a+par],
    # image row = padded row - 1, padded rows 0 and H+1 folding onto rows 1 and H-2
    w4 = rng.standard_normal(4)
    y = np.array([(xp[2 * o:2 * o + 4] * w4).sum() for o in range(H // 2)])
    yp = np.zeros(H // 2)
    for r in range(2):
        Xr = xp[r::2]                                        # H/2 + 1 rows
        yp += np.array([Xr[o] * w4[r] + Xr[o + 1] * w4[2 + r] for o in range(H // 2)])
    assert np.abs(y - yp).max() < 1e-13
    dyo = rng.standard_normal(H // 2)
    dxp = np.zeros(H + 2)
    for o in range(H // 2):
        dxp[2 * o:2 * o + 4] += dyo[o] * w4
    dx_ref = dxp[1:H + 1].copy(); dx_ref[1] += dxp[0]; dx_ref[H - 2] += dxp[H + 1]
    dx = np.zeros(H)
    Hd = H // 2
    for par in range(2):
        plane = np.array([sum((dyo[u - a] if 0 <= u - a < Hd else 0.0) * w4[2 * a + par] for a in range(2)) for u in range(Hd + 1)])
        if par == 0:
            plane[1] += plane[0]                             # padded row 0 -> image row 1 (same plane, u = 1)
        else:
            plane[Hd - 1] += plane[Hd]                       # padded row H+1 -> image row H-2 (same plane, u = Hd-1)
        for u in range(Hd + 1):
            i = 2 * u + par - 1
            if 0 <= i < H and not (par == 0 and u == 0) and not (par == 1 and u == Hd):
                dx[i] = plane[u]
    assert np.abs(dx - dx_ref).max() < 1e-13


def test_trainer_method_surface_is_the_reference_s():
    """The methods of MUNIT_Trainer a training / test script of the reference calls (scripts/trainer.py: __init__ 29, the
    optimizer steps 252-268, recon_criterion(_mask) 279-305, forward 307, gen_update 336-346, sample 773, sample_syn 930,
    sample_fid 1087, dis_update 1133, update_learning_rate 1326, resume 1337, save 1387), by name and by parameter list."""
    import inspect
    from munit_amd.trainer import MUNIT_Trainer
    want = {
        "__init__": ["self", "hyperparameters"],
        "dis_opt_step": ["self"], "gen_opt_step": ["self"],
        "recon_criterion": ["self", "input", "target"],
        "recon_criterion_mask": ["self", "input", "target", "mask"],
        "forward": ["self", "x_a", "x_b"],
        "gen_update": ["self", "x_a", "x_b", "hyperparameters", "mask_a", "mask_b", "comet_exp", "synth", "semantic_gt_a",
                       "semantic_gt_b"],
        "dis_update": ["self", "x_a", "x_b", "hyperparameters", "comet_exp"],
        "sample": ["self", "x_a", "x_b"], "sample_syn": ["self", "x_a", "x_b"], "sample_fid": ["self", "x_a", "x_b"],
        "update_learning_rate": ["self"],
        "resume": ["self", "checkpoint_dir", "hyperparameters"],
        "save": ["self", "snapshot_dir", "iterations"],
    }
    for name, params in want.items():
        fn = getattr(MUNIT_Trainer, name)
        assert list(inspect.signature(fn).parameters) == params, (name, list(inspect.signature(fn).parameters))
    sig = inspect.signature(MUNIT_Trainer.gen_update).parameters
    assert sig["comet_exp"].default is None and sig["synth"].default is False and sig["semantic_gt_a"].default is None


def test_data_parallel_host_hooks_without_a_device():
    """Host wiring of the data-parallel schedule on a CPU-built trainer (no kernel runs): the discriminators call the
    trainer's wait at the top of forward() -- calc_gen_loss / calc_dis_loss reach forward() directly, as the reference's do
    (networks.py:84-85, 104), so a module pre-hook would miss them --, the wait and the settle are no-ops while nothing is
    pending, apply() on the trainer or on a sub-network works before any device buffer exists, the content encoders hand out the
    tensor that enters their residual trunk only when asked, and the optimizer-step deferral is not taken without a process group."""
    from munit_amd import trainer as T
    from munit_amd.utils import weights_init
    tr = T.MUNIT_Trainer(O.default_hp(64, 1, 1))
    for d in (tr.dis_a, tr.dis_b):
        assert d.before_forward == tr._wait_dis
    assert tr._dis_pending is None and tr._wait_dis() is None
    tr._settle_dis()
    assert tr._dis_waited == set() and tr.last_exchange is None
    w0 = tr.dis_a.cnns[0][0].conv.weight.detach().clone()
    torch.manual_seed(5)
    tr.dis_a.apply(weights_init("gaussian"))            # sub-network apply with the image refresh tail: no optimizer buffer on a device yet
    tr.apply(weights_init("kaiming"))
    assert not torch.equal(w0, tr.dis_a.cnns[0][0].conv.weight)
    enc = tr._content_enc(1)
    assert enc is tr.gen.enc1_content and tr._content_enc(2) is tr.gen.enc2_content
    assert enc.keep_trunk_in is False and enc.trunk_in is None
    assert T.dp_world() == 0                             # no process group: gen_update / dis_update take the in-line path
    assert T.OVERLAP_EXCHANGE
    # the state_dict of a discriminator goes through the wait as well (registered pre-hook) and is unchanged by it
    assert list(tr.dis_a.state_dict().keys())[:2] == ["cnns.0.0.conv.weight", "cnns.0.0.conv.bias"]



def test_bench_sizes_host_threads_from_the_granted_cpus(monkeypatch):
    """bench.py's thread pools (the ranks it launches, the CPU-baseline leg) are sized from the CPUs the process may use --
    affinity mask capped by the cgroup quota, the rule of tests/conftest.usable_cpus -- never from os.cpu_count(), which on
    the GPU boxes reports 256 for a grant of 16."""
    import bench
    from tests.conftest import usable_cpus
    assert bench.usable_cpus() == usable_cpus()
    monkeypatch.setattr(os, "cpu_count", lambda: 4096)         # what the box SHOWS must not matter
    n = bench.usable_cpus()
    assert n == usable_cpus() and n <= len(os.sched_getaffinity(0))
    assert bench.host_cores() <= n
    assert bench.rank_threads(1) == n and bench.rank_threads(8) == max(1, n // 8) and bench.rank_threads(10 * n) == 1
    monkeypatch.setattr(bench, "cpu_quota", lambda: 3)          # a quota below the affinity mask wins
    assert bench.usable_cpus() == min(3, len(os.sched_getaffinity(0)))
    assert bench.rank_threads(2) == max(1, bench.usable_cpus() // 2)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("os.cpu_count()") == 2                     # the docstring and the no-sched_getaffinity fallback of usable_cpus


def test_kernel_names_reported_by_the_library_exist_in_the_committed_trace():
    """munit_conv2d_kernel_name feeds bench.py's roofline table and its lookup in the PMC summary: a name of a kernel that does
    not run (round 3: the image head's backward-weight was reported as conv_lanes_wgrad_kernel while conv_lanes_wgrad_pk_kernel
    ran) silently drops the traffic figure.  For the dominant layers of BASELINE configs[1] every reported kernel must appear in
    the newest committed rocprofv3 by-grid trace of that workload -- Winograd instantiations with their exact template
    arguments, the others by kernel name."""
    import glob
    import re
    from munit_amd import _lib, ops
    traces = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_kernel_trace_by_grid_single_stream_bench_256_b8.txt")))
    assert traces, "no committed kernel trace"
    text = open(traces[-1]).read()
    lib = _lib.load()
    lib.munit_conv2d_kernel_name.restype = ctypes.c_char_p
    layers = [(8, 64, 64, 256, 256, 3, 1, 1, False), (8, 256, 256, 64, 128, 4, 2, 1, False), (8, 128, 128, 128, 256, 4, 2, 1, False),
              (8, 64, 64, 256, 128, 5, 1, 2, True), (8, 128, 128, 128, 64, 5, 1, 2, True), (8, 256, 256, 64, 3, 7, 1, 3, False),
              (8, 256, 256, 3, 64, 7, 1, 3, False), (16, 128, 128, 64, 128, 4, 2, 1, False)]
    seen = set()
    for (b, h, w, ci, co, k, s, p, up) in layers:
        pl = ops._plan(b, h, w, ci, co, k, k, s, p, "reflect", up, "none", 0.2, 0, 0)
        for which in (0, 1, 2):
            name = lib.munit_conv2d_kernel_name(pl.ref, which).decode()
            for part in re.split(r" \+ | x4 ", name):
                part = part.split(" (")[0].strip()
                m = re.match(r"([a-z_0-9]+_kernel)(<[^>]*>)?", part)
                if not m:
                    continue
                base, targs = m.group(1), m.group(2)
                exact = base + targs if (targs and "." not in targs and base.startswith("conv_wino")) else base
                seen.add(exact)
                assert exact[:44] in text, (name, exact, os.path.basename(traces[-1]))
    assert "conv_lanes_wgrad_pk_kernel" in seen and "conv_wino_kernel<0, 0>" in seen and len(seen) >= 10, seen
