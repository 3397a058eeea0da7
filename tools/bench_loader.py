"""Throughput of the input pipeline alone (GPU box): JPEG files -> device batches (decode threads, one H2D per
batch, HIP transform).  python tools/bench_loader.py [n_images] [workers]"""
import os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from munit_amd import data as D
from PIL import Image

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tmp = tempfile.mkdtemp(prefix="munit_loader_")
rng = np.random.RandomState(0)
for k in range(n):
    h, w = 768 + int(rng.randint(0, 64)), 1024 + int(rng.randint(0, 64))
    base = rng.randint(0, 256, (h // 8, w // 8, 3)).astype(np.uint8)
    Image.fromarray(base).resize((w, h)).save(os.path.join(tmp, "im%04d.jpg" % k), quality=90)
ld = D.get_data_loader_folder(tmp, 8, True, new_size=256, height=256, width=256, num_workers=workers, rank=0, world_size=1)
for _ in ld:            # warm-up epoch (file cache, thread pool)
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
cnt = 0
for _ in range(3):
    for x in ld:
        cnt += x.shape[0]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d images (%dx%d JPEG -> 256x256 crop), %d decode threads: %.0f images/s" % (cnt, 1024, 768, workers, cnt / dt))
