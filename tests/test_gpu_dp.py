"""GPU: the data-parallel step end to end -- two ranks (two processes sharing cuda:0, gloo backend so that they can)
run dis_update + gen_update on different half-batches; the all-reduced flat gradients must equal the gradient a
single process computes on the concatenated batch (every loss is a batch mean and every norm is per-sample, SURVEY.md
section 8e), and both ranks must end the step with bitwise identical weights."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

SIZE, B = 64, 2


def _step(trainer, hp, batch, record=False):
    """update_learning_rate + dis_update + gen_update; returns the (all-reduced) flat gradients, the discriminator weights
    the generator update saw, and -- record=True -- the ReLU / LeakyReLU / L1 branches each update took (tests/parity.py)."""
    from munit_amd import ops
    x_a, x_b, m_a, m_b = batch
    torch.manual_seed(11)
    trainer.update_learning_rate()
    kinks = []

    def run(fn):
        ops.MASK_SINK, ops.L1_SINK = ([], []) if record else (None, None)
        try:
            fn()
            kinks.append(([m.cpu() for m in ops.MASK_SINK], [m.cpu() for m in ops.L1_SINK]) if record else None)
        finally:
            ops.MASK_SINK = ops.L1_SINK = None

    run(lambda: trainer.dis_update(x_a, x_b, hp))
    # no read of the discriminator buffers here: in data-parallel mode their exchange and optimizer step are still in flight on
    # the communication stream (trainer._defer_dis_step) while gen_update's generator forward runs -- gen_update freezes the
    # discriminators, so their gradient and weights are read after it
    run(lambda: trainer.gen_update(x_a, x_b, hp, m_a, m_b))
    torch.cuda.synchronize()
    g_dis = trainer.dis_opt.flat_g.detach().clone()
    dis_p = trainer.dis_opt.flat_p.detach().clone()
    g_gen = trainer.gen_opt.flat_g.detach().clone()
    torch.cuda.synchronize()
    return g_dis.cpu(), g_gen.cpu(), dis_p.cpu(), kinks


def _worker(rank, world, tmpdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    from munit_amd.trainer import MUNIT_Trainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(tmpdir, "rdzv"), rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    hp = bench.bench_hp(SIZE, B)
    batch = tuple(t.to(dev) for t in bench.make_batch(B, SIZE, rank))
    from munit_amd import trainer as T
    assert T.OVERLAP_EXCHANGE                       # default: the generator gradient goes out in stages inside backward, the
    # discriminator's exchange + optimizer step run on the communication stream beside the next generator forward
    # ONE trainer serves the four runs (guided 1 / 0 x staged / serial exchange): construction and the first-use build of the
    # prepared weight images dominate this test's time.  Between runs the weights, the Adam moments and the step counters go
    # back to the initial state, so every run is the FIRST update of the same trainer.
    torch.manual_seed(1234)
    tr = MUNIT_Trainer(dict(hp))
    tr.to(dev)
    init = [(o, o.flat_p.detach().clone()) for o in (tr.gen_opt, tr.dis_opt)]

    def fresh(guided):
        tr._settle_dis()
        torch.cuda.synchronize()
        with torch.no_grad():
            for o, p0 in init:
                o.flat_p.copy_(p0)
                o.flat_m.zero_()
                o.flat_v.zero_()
                o._step = 0
                o.invalidate_prepared()
        tr.guided = guided
        tr.last_exchange = None
        return dict(hp, guided=guided)

    for guided in (1, 0):                           # guided 0: the sampled styles' MLP passes are tied into stage 1 by a gradient
        hp_g = fresh(guided)
        out = _step(tr, hp_g, batch, record=(guided == 1))
        stages = tr.last_exchange.stages
        assert len(stages) == 2 and all(st["fired"] for st in stages), stages
        assert stages[0]["ranges"] == [tuple(r) for r in tr._early_ranges] and stages[1]["ranges"] == [tuple(r) for r in tr._trunk_ranges]
        assert tr._dis_pending is not None           # the discriminator step was deferred ...
        # ... and gen_update's discriminator forwards (calc_gen_loss -> forward, on the two branch streams) waited for it
        assert len(tr._dis_waited) >= (2 if T.BRANCH_STREAMS else 1), tr._dis_waited
        sd_g = {k: v.detach().cpu().clone() for k, v in tr.gen.state_dict().items()}
        sd_d = {k: v.detach().cpu().clone() for k, v in tr.dis_a.state_dict().items()}
        # the same step with ONE all-reduce after backward and the discriminator step in line: bitwise the same averaged
        # gradients and weights (two ranks)
        hp_g = fresh(guided)
        T.OVERLAP_EXCHANGE = False
        out2 = _step(tr, hp_g, batch)
        T.OVERLAP_EXCHANGE = True
        assert tr._dis_pending is None and tr.last_exchange is None
        assert torch.equal(out[1], out2[1]) and torch.equal(out[0], out2[0]) and torch.equal(out[2], out2[2]), guided
        for (k, a), b in zip(sd_g.items(), tr.gen.state_dict().values()):
            assert torch.equal(a, b.detach().cpu()), (guided, k)
        for (k, a), b in zip(sd_d.items(), tr.dis_a.state_dict().values()):
            assert torch.equal(a, b.detach().cpu()), (guided, k)
        if guided == 1:
            g_dis, g_gen, dis_p, kinks = out
            sd = sd_g
    torch.save({"g_dis": g_dis, "g_gen": g_gen, "gen": sd, "dis_p": dis_p, "kinks": kinks},
               os.path.join(tmpdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_matches_the_oracle_on_the_joint_batch(tmp_path):
    """Two ranks, half-batches, averaged flat gradients -- against the fp64 ORACLE's gradient on the joint batch (every loss is
    a batch mean and every norm per sample), with the kinks pinned: each rank records the ReLU /
    LeakyReLU / L1 branches of its own samples and the oracle takes their concatenation along the batch axis, so both sides
    differentiate the same piecewise-linear function and the comparison carries the step tests' bounds (every tensor <= 5e-5
    max and L2; tests/parity.GradCheck) instead of the 2e-3 an unpinned HIP-vs-HIP comparison of two batch sizes needs (a
    pre-activation within rounding noise of 0 takes the other branch when the batch is tiled differently).  The recorded
    branches are audited against the oracle's own as in every step test."""
    import torch.multiprocessing as mp
    import bench
    from munit_amd.trainer import MUNIT_Trainer
    from oracle import munit_oracle as O
    from tests.parity import GradCheck, KINK_FRAC, KINK_NOISE, oracle_states, trainer_named_params
    mp.spawn(_worker, args=(2, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    for k in r0["gen"]:                      # same averaged gradient, same update
        assert torch.equal(r0["gen"][k], r1["gen"][k]), k
    assert torch.equal(r0["g_gen"], r1["g_gen"]) and torch.equal(r0["g_dis"], r1["g_dis"])
    assert torch.equal(r0["dis_p"], r1["dis_p"])

    dev = torch.device("cuda:0")
    hp = bench.bench_hp(SIZE, 2 * B)
    torch.manual_seed(1234)
    tr = MUNIT_Trainer(hp)                   # the ranks' initial weights (same seed); hosts the per-tensor views of the flat buffers
    tr.to(dev)
    gnames, dnames = trainer_named_params(tr)
    orc = O.OracleTrainer(dict(hp), *oracle_states(hp, torch.float64))
    o_gen, o_dis = orc.opt["gen"]["params"], orc.opt["dis"]["params"]
    with torch.no_grad():
        for (n, p), q in list(zip(gnames, o_gen)) + list(zip(dnames, o_dis)):
            q.copy_(p.detach().double().cpu())
    parts = [bench.make_batch(B, SIZE, r) for r in range(2)]
    joint = [torch.cat([parts[0][i], parts[1][i]], 0).double() for i in range(4)]

    def pinned(fn, phase):
        (m0, s0), (m1, s1) = r0["kinks"][phase], r1["kinks"][phase]
        assert len(m0) == len(m1) and len(s0) == len(s1)
        km = O.KinkMasks([torch.cat([a, b], 0) for a, b in zip(m0, m1)], [torch.cat([a, b], 0) for a, b in zip(s0, s1)])
        O.KINK_MASKS = km
        try:
            out = fn()
            assert km.done()
        finally:
            O.KINK_MASKS = None
        assert km.worst_rel <= KINK_NOISE and km.n_disagree <= KINK_FRAC * km.n_total, (km.worst_rel, km.n_disagree, km.n_total)
        return out

    orc.update_learning_rate()
    gc = GradCheck(pinned=True)
    d_ref = pinned(lambda: orc.dis_update(joint[0], joint[1]), 0)
    tr.dis_opt.flat_g.copy_(r0["g_dis"].to(dev))
    for (n, p), g in zip(dnames, d_ref):
        gc.add("dis." + n, p._munit_grad, g)
    with torch.no_grad():                    # gen_update on the discriminators the ranks stepped to
        tr.dis_opt.flat_p.copy_(r0["dis_p"].to(dev))
        for (n, p), q in zip(dnames, o_dis):
            q.copy_(p.detach().double().cpu())
    g_ref = pinned(lambda: orc.gen_update(joint[0], joint[1], joint[2], joint[3]), 1)
    tr.gen_opt.flat_g.copy_(r0["g_gen"].to(dev))
    n_checked = 0
    for (n, p), g in zip(gnames, g_ref):
        if g is None or float(g.abs().max()) < 1e-7:     # conv bias ahead of an instance norm: mathematically zero
            continue
        gc.add("gen." + n, p._munit_grad, g)
        n_checked += 1
    gc.finish()
    assert n_checked >= 80
    print("two ranks vs fp64 oracle on the joint batch: worst max %.2e, worst L2 %.2e, median L2 %.2e" %
          (gc.worst_max, gc.worst_l2, gc.median))


_NCCL_WORKER = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
os.environ["MUNIT_FORCE_ALLREDUCE"] = "1"
import bench
from munit_amd import trainer as T
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
hp = bench.bench_hp(64, 2)
batch = tuple(t.to(dev) for t in bench.make_batch(2, 64, 0))

def run():
    torch.manual_seed(1234)
    tr = T.MUNIT_Trainer(hp); tr.to(dev)
    torch.manual_seed(11)
    for it in range(3):      # three iterations: the deferred discriminator step of one is settled by the next dis_update
        tr.iterations = it
        tr.update_learning_rate(); tr.dis_update(batch[0], batch[1], hp); tr.gen_update(batch[0], batch[1], hp, batch[2], batch[3])
    torch.cuda.synchronize()
    return tr

ref = run()                                      # no process group yet: no exchange
dist.init_process_group("nccl", init_method="file://" + %(rdzv)r, rank=0, world_size=1, device_id=dev)
assert T.FORCE_ALLREDUCE
got = run()                                      # same steps with the RCCL all-reduce of both flat gradients issued: staged
# generator exchange, discriminator exchange + Adam on the communication stream beside the next generator forward
assert got.last_exchange is not None and len(got.last_exchange.stages) == 2 and got._dis_pending is not None
for a, b in zip(ref.parameters(), got.parameters()):
    assert torch.equal(a, b)
assert torch.equal(ref.dis_opt.flat_m, got.dis_opt.flat_m) and torch.equal(ref.gen_opt.flat_v, got.gen_opt.flat_v)
# and the exchange on its own: 109 MB flat generator gradient, device events on the stream RCCL is enqueued from
g = got.gen_opt.flat_g
before = g.clone()
for _ in range(3):
    dist.all_reduce(g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    dist.all_reduce(g)
e1.record()
torch.cuda.synchronize()
assert torch.equal(g, before)                    # world size 1: SUM over one rank
print(json.dumps({"nccl_world1_allreduce_ms": e0.elapsed_time(e1) / 10, "mbytes": g.numel() * 4 / 1e6}))
dist.barrier()
dist.destroy_process_group()
"""


def test_nccl_backend_world1_step(tmp_path):
    """The RCCL branch of the data-parallel path, executed once on the one GPU of the box: init_process_group("nccl"),
    all_reduce of both flat gradient buffers inside dis_update / gen_update (MUNIT_FORCE_ALLREDUCE=1), destroy.  With
    one rank the exchange must leave every weight bit-identical to the step without a process group."""
    import subprocess
    script = tmp_path / "nccl_w.py"
    script.write_text(_NCCL_WORKER % dict(root=ROOT, rdzv=str(tmp_path / "rdzv")))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT, env=env,
                       timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    line = out.strip().splitlines()[-1]
    print(line)
    keep = os.path.join(ROOT, "gpurun_out")          # scratch directory of the GPU box runs: keep the measured figure
    if os.path.isdir(keep):
        with open(os.path.join(keep, "nccl_world1_allreduce.json"), "w") as f:
            f.write(line + "\n")


def test_bench_under_torchrun_one_rank():
    """The driver's launch form (python -m torch.distributed.run ... bench.py --gpus N) with N = 1: the nccl group,
    the barriers around the timed region and the MAX all-reduce of the elapsed time all execute."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MUNIT_FORCE_ALLREDUCE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--size", "64", "--batch", "2", "--no-cpu-baseline", "--no-roofline", "--no-modes"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, env=env, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["parallelism"] == "dp1"


_COMM_WORKER = r"""
import ctypes, sys, torch
sys.path.insert(0, %(root)r)
from munit_amd import _lib
lib = _lib.load()
torch.cuda.set_device(0)
uid = (ctypes.c_char * 128)()
assert lib.munit_comm_unique_id(uid, 128) == 0, lib.munit_last_error()
comm = ctypes.c_void_p()
assert lib.munit_comm_init(ctypes.byref(comm), 0, 1, uid) == 0, lib.munit_last_error()
g = torch.randn(1 << 20, device="cuda:0")
ref = g.clone()
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert lib.munit_comm_allreduce(comm, ctypes.c_void_p(g.data_ptr()), g.numel(), ctypes.c_void_p(st)) == 0, lib.munit_last_error()
torch.cuda.synchronize()
assert torch.equal(g, ref)                        # SUM over one rank
assert lib.munit_comm_allreduce(comm, None, 0, ctypes.c_void_p(st)) == 0
assert lib.munit_comm_init(ctypes.byref(comm), 2, 1, uid) != 0 and b"bad arguments" in lib.munit_last_error()
assert lib.munit_shutdown() != 0 and b"still alive" in lib.munit_last_error()     # refused while the communicator lives
assert lib.munit_comm_allreduce(comm, ctypes.c_void_p(g.data_ptr()), g.numel(), ctypes.c_void_p(st)) == 0   # ... and nothing was unloaded
torch.cuda.synchronize()
assert lib.munit_comm_destroy(comm) == 0
assert lib.munit_shutdown() == 0
print("comm ok")
"""


def test_c_abi_communicator_world1(tmp_path):
    """munit_comm_unique_id / init / allreduce / destroy + munit_shutdown (SURVEY.md section 8b's RCCL entry points for hosts
    without torch.distributed), at world size 1 on the box's one GPU, in a child process of its own."""
    import subprocess
    script = tmp_path / "comm_w.py"
    script.write_text(_COMM_WORKER % dict(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT, env=env, timeout=600)
    assert p.returncode == 0 and "comm ok" in p.stdout.decode(), p.stdout.decode()[-3000:]

