"""Minimal end-to-end training loop on munit_amd: the loaders of munit_amd.data feeding MUNIT_Trainer, in the
shape of the reference's scripts/train.py:157-330 (dis_update / gen_update cadence of `ratio_disc_gen`,
update_learning_rate first, periodic save).  Control plane only -- no comet, FID or image dumps.

  python examples/train_loop.py --config configs.yaml --data-root /path/with/trainA,trainB,testA,testB [--iters N]
  torchrun --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_loop.py ...      (data parallel, RCCL)
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--data-root", default=None, help="folder with trainA/ trainB/ testA/ testB (overrides the YAML lists)")
    ap.add_argument("--file-list-a"), ap.add_argument("--file-list-b")
    ap.add_argument("--mask-list-a"), ap.add_argument("--mask-list-b")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--output", default=None, help="checkpoint directory")
    ap.add_argument("--save-every", type=int, default=0)
    args = ap.parse_args(argv)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from munit_amd.utils import get_config
    from munit_amd.trainer import MUNIT_Trainer
    from munit_amd import data as D

    config = get_config(args.config)
    torch.manual_seed(1234)                       # identical initial weights on every rank
    trainer = MUNIT_Trainer(config)
    trainer.to(dev)

    b, ns = config["batch_size"], config.get("new_size")
    h, w, nw = config["crop_image_height"], config["crop_image_width"], config.get("num_workers", 4)
    if args.mask_list_a:                          # scripts/train.py:80-100: image + mask loaders
        loader_a = D.get_data_loader_mask_and_im(args.file_list_a, args.mask_list_a, b, True, ns, h, w, nw, seed=1)
        loader_b = D.get_data_loader_mask_and_im(args.file_list_b, args.mask_list_b, b, True, ns, h, w, nw, seed=2)
    else:
        root = args.data_root or config["data_root"]
        loader_a = D.get_data_loader_folder(os.path.join(root, "trainA"), b, True, ns, h, w, nw, seed=1)
        loader_b = D.get_data_loader_folder(os.path.join(root, "trainB"), b, True, ns, h, w, nw, seed=2)

    if args.output and local_rank == 0:
        os.makedirs(args.output, exist_ok=True)   # the reference's prepare_sub_folder (utils.py:817-834)
    it, t0 = 0, time.perf_counter()
    ratio = int(config.get("ratio_disc_gen", 1))
    while it < args.iters:
        for batch_a, batch_b in zip(loader_a, loader_b):
            (x_a, m_a), (x_b, m_b) = [(t if isinstance(t, tuple) else (t, None)) for t in (batch_a, batch_b)]
            if m_a is None and config.get("recon_mask", 0) == 1:      # no mask files: everything counts
                m_a, m_b = torch.ones_like(x_a[:, :1]), torch.ones_like(x_b[:, :1])
            trainer.iterations = it
            trainer.update_learning_rate()
            trainer.dis_update(x_a, x_b, config)                        # scripts/train.py:182
            if (it + 1) % ratio == 0:
                trainer.gen_update(x_a, x_b, config, m_a, m_b)          # scripts/train.py:185-187
            it += 1
            if args.output and args.save_every and it % args.save_every == 0 and local_rank == 0:
                trainer.save(args.output, it - 1)      # file names carry iterations + 1 (trainer.py:1337-1344)
            if it >= args.iters:
                break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if local_rank == 0:
        print("iterations %d  %.1f ms/iter  loss_dis_total %.5f" % (it, 1e3 * dt / max(it, 1), float(trainer.loss_dis_total)))
    if world > 1:
        dist.destroy_process_group()
    return trainer


if __name__ == "__main__":
    main()
