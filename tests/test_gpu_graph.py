"""GPU: hipGraph replay of the step (munit_amd/graph.py) must be bit-identical to eager stepping."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graphed_step_is_bitwise_the_eager_step():
    import bench
    from munit_amd.graph import GraphedStep
    from munit_amd.trainer import MUNIT_Trainer
    dev = torch.device("cuda:0")
    size, batch, n_eager_warm, n_steps = 64, 2, 2, 3
    batches = [tuple(t.to(dev) for t in bench.make_batch(batch, size, rank=r)) for r in range(n_steps)]
    warm = tuple(t.to(dev) for t in bench.make_batch(batch, size, rank=9))

    def fresh():
        hp = bench.bench_hp(size, batch)
        hp["step_size"] = 3                      # the LR schedule changes inside the run: lr must not be baked into the graph
        torch.manual_seed(1234)
        tr = MUNIT_Trainer(hp)
        tr.to(dev)
        return tr, hp

    # eager reference: the same sequence of batches (GraphedStep's warm-up steps are real steps on the warm-up batch)
    tr, hp = fresh()
    for _ in range(n_eager_warm):
        tr.update_learning_rate(); tr.dis_update(warm[0], warm[1], hp); tr.gen_update(warm[0], warm[1], hp, warm[2], warm[3])
    for b in batches:
        tr.update_learning_rate(); tr.dis_update(b[0], b[1], hp); tr.gen_update(b[0], b[1], hp, b[2], b[3])
    torch.cuda.synchronize()
    ref = [tr.gen_opt.flat_p.clone(), tr.dis_opt.flat_p.clone(), tr.gen_opt.flat_m.clone(), tr.gen_opt.flat_v.clone()]
    ref_loss = (float(tr.loss_gen_total.detach()), float(tr.loss_dis_total.detach()))
    ref_lr = tr.gen_opt.param_groups[0]["lr"]

    tr2, hp2 = fresh()
    g = GraphedStep(tr2, hp2, *warm, warmup=n_eager_warm)
    for b in batches:
        g(*b)
    torch.cuda.synchronize()
    got = [tr2.gen_opt.flat_p, tr2.dis_opt.flat_p, tr2.gen_opt.flat_m, tr2.gen_opt.flat_v]
    assert tr2.gen_opt._step == tr.gen_opt._step == n_eager_warm + n_steps
    assert tr2.gen_opt.param_groups[0]["lr"] == ref_lr and ref_lr < hp["lr"]
    assert (float(tr2.loss_gen_total.detach()), float(tr2.loss_dis_total.detach())) == ref_loss
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # and back to eager
    g.release()
    b = batches[0]
    tr2.update_learning_rate(); tr2.dis_update(b[0], b[1], hp2); tr2.gen_update(b[0], b[1], hp2, b[2], b[3])
    tr.update_learning_rate(); tr.dis_update(b[0], b[1], hp); tr.gen_update(b[0], b[1], hp, b[2], b[3])
    torch.cuda.synchronize()
    assert torch.equal(tr.gen_opt.flat_p, tr2.gen_opt.flat_p)
