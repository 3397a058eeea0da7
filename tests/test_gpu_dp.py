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


def _step(trainer, hp, batch):
    x_a, x_b, m_a, m_b = batch
    torch.manual_seed(11)
    trainer.update_learning_rate()
    trainer.dis_update(x_a, x_b, hp)
    g_dis = trainer.dis_opt.flat_g.detach().clone()
    trainer.gen_update(x_a, x_b, hp, m_a, m_b)
    g_gen = trainer.gen_opt.flat_g.detach().clone()
    torch.cuda.synchronize()
    return g_dis.cpu(), g_gen.cpu()


def _worker(rank, world, tmpdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    from munit_amd.trainer import MUNIT_Trainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(tmpdir, "rdzv"), rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    hp = bench.bench_hp(SIZE, B)
    torch.manual_seed(1234)
    tr = MUNIT_Trainer(hp)
    tr.to(dev)
    batch = tuple(t.to(dev) for t in bench.make_batch(B, SIZE, rank))
    g_dis, g_gen = _step(tr, hp, batch)
    sd = {k: v.detach().cpu() for k, v in tr.gen.state_dict().items()}
    torch.save({"g_dis": g_dis, "g_gen": g_gen, "gen": sd}, os.path.join(tmpdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_matches_single_process_on_the_joint_batch(tmp_path):
    import torch.multiprocessing as mp
    import bench
    from munit_amd.trainer import MUNIT_Trainer
    from tests.parity import l2err
    mp.spawn(_worker, args=(2, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    for k in r0["gen"]:                      # same averaged gradient, same update
        assert torch.equal(r0["gen"][k], r1["gen"][k]), k
    assert torch.equal(r0["g_gen"], r1["g_gen"]) and torch.equal(r0["g_dis"], r1["g_dis"])

    dev = torch.device("cuda:0")
    hp = bench.bench_hp(SIZE, 2 * B)
    torch.manual_seed(1234)
    tr = MUNIT_Trainer(hp)
    tr.to(dev)
    parts = [bench.make_batch(B, SIZE, r) for r in range(2)]
    joint = tuple(torch.cat([parts[0][i], parts[1][i]], 0).to(dev) for i in range(4))
    g_dis, g_gen = _step(tr, hp, joint)
    # mean over ranks of per-rank batch means == mean over the joint batch (fp32 summation order differs)
    assert l2err(r0["g_dis"], g_dis.double()) <= 1e-4, l2err(r0["g_dis"], g_dis.double())
    assert l2err(r0["g_gen"], g_gen.double()) <= 2e-3, l2err(r0["g_gen"], g_gen.double())
