"""Throughput of the input pipeline alone (GPU box): JPEG files (+ PNG masks) -> device batches (decode threads, one H2D per
batch, HIP transform).  python tools/bench_loader.py [n_images] [workers ...]"""
import os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from munit_amd import data as D
from PIL import Image

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
workers_list = [int(a) for a in sys.argv[2:]] or [8, 16]
tmp = tempfile.mkdtemp(prefix="munit_loader_")
rng = np.random.RandomState(0)
with open(os.path.join(tmp, "images.txt"), "w") as fi, open(os.path.join(tmp, "masks.txt"), "w") as fm:
    for k in range(n):
        h, w = 768 + int(rng.randint(0, 64)), 1024 + int(rng.randint(0, 64))
        base = rng.randint(0, 256, (h // 8, w // 8, 3)).astype(np.uint8)
        Image.fromarray(base).resize((w, h)).save(os.path.join(tmp, "im%04d.jpg" % k), quality=90)
        m = (rng.randint(0, 2, (h // 16, w // 16)) * 255).astype(np.uint8)
        Image.fromarray(m, mode="L").resize((w, h), Image.NEAREST).save(os.path.join(tmp, "mk%04d.png" % k))
        fi.write(os.path.join(tmp, "im%04d.jpg" % k) + "\n")
        fm.write(os.path.join(tmp, "mk%04d.png" % k) + "\n")


def rate(ld, with_mask):
    for _ in ld:            # warm-up epoch (file cache, thread pool)
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cnt = 0
    for _ in range(3):
        for x in ld:
            cnt += (x[0] if with_mask else x).shape[0]
    torch.cuda.synchronize()
    return cnt / (time.perf_counter() - t0)


for workers in workers_list:
    ld = D.get_data_loader_folder(tmp, 8, True, new_size=256, height=256, width=256, num_workers=workers, rank=0, world_size=1)
    r1 = rate(ld, False)
    lm = D.get_data_loader_mask_and_im(os.path.join(tmp, "images.txt"), os.path.join(tmp, "masks.txt"), 8, True, new_size=256,
                                       height=256, width=256, num_workers=workers, rank=0, world_size=1)
    r2 = rate(lm, True)
    print("%d decode threads: images only %.0f images/s ; image + mask pairs %.0f pairs/s   (1024x768 JPEG + PNG mask -> 256x256 crops, "
          "batches of 8; %d cores visible)" % (workers, r1, r2, os.cpu_count()))
