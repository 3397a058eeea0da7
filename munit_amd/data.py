"""Input pipeline: the reference's data loaders with the per-image work moved to the GPU.

Mirrors the loader API of the reference (same function names, argument order and batch contents):
  get_all_data_loaders(conf)                       scripts/utils.py:50-156
  get_data_loader_list(root, file_list, ...)       scripts/utils.py:192-250   (ImageFilelist, scripts/data.py:25-49)
  get_data_loader_folder(input_folder, ...)        scripts/utils.py:680-740   (ImageFolder, scripts/data.py:116-153)
  get_data_loader_mask_and_im(file_list, mask_list, ...)  scripts/utils.py:638-677 (MyDataset, utils.py:270-363)
  default_txt_reader / default_flist_reader        scripts/utils.py:253-267 / scripts/data.py:13-23

Design (MI355X-first, not the reference's worker-process + CPU-transform pipeline):
  * host threads only DECODE files (PIL releases the GIL while decoding) into one pinned staging buffer
    per batch: [descriptors][image 0 bytes][image 1 bytes]... -> ONE async H2D copy per batch on a side stream;
  * flip / anti-aliased bilinear resize / crop / ToTensor / Normalize run as one HIP pass over the batch
    (munit_image_preprocess, munit_mask_preprocess: csrc/image.hip, bit-identical to PIL + torchvision),
    producing the channels_last fp32 batch the convolutions consume -- no per-sample CPU tensors, no
    collate, no NCHW->NHWC copy;
  * a producer thread keeps `prefetch` batches in flight, so decode + PCIe of step n+1 overlap step n;
  * data parallel: rank r takes every world_size-th sample of the epoch permutation (same seed on all ranks).
There is no CPU transform path: iterating a loader without a HIP device raises.
"""
import ctypes
import os
import queue
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib
from ._lib import ImageDesc

IMG_EXTENSIONS = (".jpg", ".JPG", ".jpeg", ".JPEG", ".png", ".PNG", ".ppm", ".PPM", ".bmp", ".BMP")


def default_flist_reader(flist):
    """One image path per line (scripts/data.py:13-23)."""
    with open(flist, "r") as f:
        return [line.strip() for line in f.readlines()]


def default_txt_reader(flist):
    """One whitespace-split record per line; element 0 is the path (scripts/utils.py:253-267)."""
    with open(flist, "r") as f:
        return [line.strip().split() for line in f.readlines()]


def is_image_file(filename):
    return filename.endswith(IMG_EXTENSIONS)


def make_dataset(directory):
    """All image files under `directory`, recursively (scripts/data.py:104-114)."""
    if not os.path.isdir(directory):
        raise AssertionError("%s is not a valid directory" % directory)
    images = []
    for root, _, fnames in sorted(os.walk(directory)):
        for fname in fnames:
            if is_image_file(fname):
                images.append(os.path.join(root, fname))
    return images


def resize_size(w, h, size):
    """torchvision Resize(int): shorter side -> size, other side int(size * long / short); (w, h) unchanged
    when the shorter side already matches or size is None.  Returns (rs_w, rs_h)."""
    if size is None or (w <= h and w == size) or (h <= w and h == size):
        return w, h
    if w < h:
        return size, int(size * h / w)
    return int(size * w / h), size


def shard_indices(n, batch_size, train, epoch_seed, rank=0, world_size=1):
    """Sample order of one epoch for one rank: permutation when train (shuffle=train), rank r takes
    positions r, r+world, ...; trailing samples that do not fill a batch are dropped (drop_last=True)."""
    if train:
        g = torch.Generator()
        g.manual_seed(int(epoch_seed))
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    # every rank must see the same number of batches (one gradient all-reduce per step): drop the tail that
    # does not fill a global batch, then deal the rest round-robin
    nb = n // (batch_size * world_size)
    order = order[:nb * batch_size * world_size][rank::world_size]
    return [order[b * batch_size:(b + 1) * batch_size] for b in range(nb)]


def _decode_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(arr)


def _decode_mask(path):
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("L", "P"):
            raise ValueError("munit_amd.data: mask %s has mode %r; single-band 8-bit masks (L / P) are supported"
                             % (path, im.mode))
        arr = np.frombuffer(im.tobytes(), dtype=np.uint8).reshape(im.size[1], im.size[0])
    return np.ascontiguousarray(arr)


def pack_batch(arrays, masks, draws):
    """Lay out one batch for the device.  Layout of the staging buffer:
    [B image descs][B mask descs][pad to 256][image bytes ...][mask bytes ...] (every item 16-byte aligned).
    draws[b] = (flip, rs_h, rs_w, crop_i, crop_j, out_h, out_w).  Returns (descs, offsets, mask offsets, total)."""
    B = len(arrays)
    dsz = ctypes.sizeof(ImageDesc)
    cur = (2 * B * dsz + 255) // 256 * 256
    offs, moffs = [], []
    for a in arrays:
        offs.append(cur)
        cur += (a.nbytes + 15) // 16 * 16
    if masks is not None:
        for m in masks:
            moffs.append(cur)
            cur += (m.nbytes + 15) // 16 * 16
    descs = (ImageDesc * (2 * B))()
    for b, (a, dr) in enumerate(zip(arrays, draws)):
        flip, rs_h, rs_w, i, j, _, _ = dr
        descs[b] = ImageDesc(offs[b], a.shape[0], a.shape[1], rs_h, rs_w, i, j, flip, 0)
        if masks is not None:
            m = masks[b]
            descs[B + b] = ImageDesc(moffs[b], m.shape[0], m.shape[1], 0, 0, i, j, flip, 0)
    return descs, offs, moffs, cur


class _StagingRing:
    """Pinned host buffers reused round-robin: hipHostMalloc of a ~20 MB batch costs milliseconds, and the producer thread
    paid it for every batch.  A slot is handed out again only after the H2D copy that read it has completed (its event)."""

    def __init__(self, slots):
        self.bufs = [None] * slots
        self.events = [None] * slots
        self.pos = 0

    def take(self, nbytes):
        k = self.pos
        self.pos = (self.pos + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()          # the copy out of this slot (issued `slots` batches ago) is done
        if self.bufs[k] is None or self.bufs[k].numel() < nbytes:
            self.bufs[k] = torch.empty(int(nbytes * 1.25), dtype=torch.uint8).pin_memory()
        return k, self.bufs[k]

    def mark(self, k, event):
        self.events[k] = event


def launch_transform(arrays, masks, draws, device, stream, ring=None, pool=None):
    """Stage decoded uint8 images (H,W,3) [and masks (H,W)] in pinned memory, upload them with one async copy
    and run the device transform on `stream`.  Returns (outputs, ready event, buffers to keep alive until the
    event): outputs = images (B,3,h,w) channels_last, or (images, masks (B,1,h,w)).  ring: _StagingRing to take the
    pinned buffer from (else a fresh one); pool: unused (the copies run in the calling thread, see below)."""
    lib = _lib.load()
    B = len(arrays)
    th, tw = draws[0][5], draws[0][6]
    if any((d[5], d[6]) != (th, tw) for d in draws):
        raise ValueError("crop=False needs equally sized images inside a batch")
    for a, d in zip(arrays, draws):
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("images must be uint8 (H, W, 3) arrays")
        if d[3] < 0 or d[4] < 0 or d[3] + th > d[1] or d[4] + tw > d[2]:
            raise ValueError("crop window (%d,%d,%d,%d) outside the resized image (%d,%d)" % (d[3], d[4], th, tw, d[1], d[2]))
    descs, offs, moffs, total = pack_batch(arrays, masks, draws)
    slot = None
    if ring is not None:
        slot, stage = ring.take(total)
    else:
        stage = torch.empty(total, dtype=torch.uint8).pin_memory()
    sv = stage.numpy()
    ctypes.memmove(sv.ctypes.data, ctypes.addressof(descs), ctypes.sizeof(descs))

    def put(a, o):
        sv[o:o + a.nbytes] = a.reshape(-1)     # a large contiguous copy: numpy drops the GIL for it

    # The copies run HERE, in the producer thread: the decode pool's queue already holds the files of the next `ahead`
    # batches, and copies submitted behind them made every upload wait for all of that decode work (a latency bubble per
    # batch).  2 B contiguous memcpys of a few MB each, with the GIL released, cost less than that wait.  (`pool` is kept in
    # the signature for callers that pass it.)
    jobs = list(zip(arrays, offs)) + (list(zip(masks, moffs)) if masks is not None else [])
    for a, o in jobs:
        put(a, o)
    ksize = 3
    for a, d in zip(arrays, draws):
        ksize = max(ksize, lib.munit_image_ksize(a.shape[0], d[1]), lib.munit_image_ksize(a.shape[1], d[2]))
    vp = ctypes.c_void_p
    with torch.cuda.stream(stream):
        dev = torch.empty(total, dtype=torch.uint8, device=device)
        dev.copy_(stage[:total], non_blocking=True)     # the one H2D transfer of the batch
        if slot is not None:
            copied = torch.cuda.Event()
            copied.record(stream)
            ring.mark(slot, copied)
        st = vp(stream.cuda_stream)
        base = dev.data_ptr()
        images = torch.empty((B, 3, th, tw), device=device, dtype=torch.float32, memory_format=torch.channels_last)
        nws = lib.munit_image_preprocess_workspace_bytes(B, th, tw, ksize)
        ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=device)
        rc = lib.munit_image_preprocess(vp(base), vp(base), B, th, tw, ksize, vp(images.data_ptr()),
                                        vp(ws.data_ptr()), ctypes.c_size_t(nws), st)
        if rc:
            raise RuntimeError("munit_image_preprocess: " + lib.munit_last_error().decode())
        keep = (dev, ws) if slot is not None else (stage, dev, ws)    # a ring slot outlives the batch by itself
        out = images
        if masks is not None:
            out_masks = torch.empty((B, 1, th, tw), device=device, dtype=torch.float32)
            nwm = lib.munit_mask_preprocess_workspace_bytes(B, th, tw)
            wsm = torch.empty(max(nwm, 1), dtype=torch.uint8, device=device)
            rc = lib.munit_mask_preprocess(vp(base), vp(base + B * ctypes.sizeof(ImageDesc)), B, th, tw,
                                           vp(out_masks.data_ptr()), vp(wsm.data_ptr()), ctypes.c_size_t(nwm), st)
            if rc:
                raise RuntimeError("munit_mask_preprocess: " + lib.munit_last_error().decode())
            keep = keep + (wsm,)
            out = (images, out_masks)
        ev = torch.cuda.Event()
        ev.record(stream)
    return out, ev, keep


def transform_batch(arrays, masks, draws, device=None):
    """Synchronous convenience wrapper of launch_transform on the current stream's device."""
    if not torch.cuda.is_available():
        raise RuntimeError("munit_amd.data: the transforms run on the GPU (libmunit_hip.so); no HIP device is "
                           "visible and there is no CPU fallback")
    device = device or torch.device("cuda", torch.cuda.current_device())
    out, ev, keep = launch_transform(arrays, masks, draws, device, torch.cuda.current_stream(device))
    ev.synchronize()
    return out


class _Dataset:
    """Indexable view used by scripts/train.py:133-143 (`loader.dataset[i]`): one transformed sample."""

    def __init__(self, loader):
        self._loader = loader

    def __len__(self):
        return len(self._loader.image_paths)

    def __getitem__(self, index):
        out = self._loader._run_batch([index], self._loader._rng)
        if self._loader.mask_paths is not None:
            return out[0][0], out[1][0]
        return out[0]


class DeviceBatchLoader:
    """Iterable of device batches: images (B,3,h,w) channels_last fp32 in [-1,1] -- and, with masks,
    (images, masks (B,1,h,w)) tuples, like DataLoader(MyDataset)."""

    def __init__(self, image_paths, mask_paths, batch_size, train, new_size, height, width, num_workers=4,
                 crop=True, device=None, seed=0, rank=None, world_size=None, prefetch=2, torch_flip=False):
        if mask_paths is not None and len(mask_paths) != len(image_paths):
            raise ValueError("image list and mask list differ in length: %d vs %d" % (len(image_paths), len(mask_paths)))
        self.image_paths = list(image_paths)
        self.mask_paths = None if mask_paths is None else list(mask_paths)
        self.batch_size = int(batch_size)
        self.train = bool(train)
        self.new_size = new_size
        self.height, self.width = int(height), int(width)
        self.crop = bool(crop)
        self.num_workers = max(1, int(num_workers))
        self.prefetch = max(1, int(prefetch))
        self.device = device
        self.seed = int(seed)
        if rank is None or world_size is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            rank = dist.get_rank() if on else 0
            world_size = dist.get_world_size() if on else 1
        self.rank, self.world_size = int(rank), int(world_size)
        self.epoch = 0
        self.flip_always_drawn = bool(torch_flip)   # MyDataset flips with or without `train` (utils.py:309-312)
        self._rng = random.Random(self.seed * 1000003 + self.rank)
        self._pool = None
        self._stream = None
        self.dataset = _Dataset(self)

    def __len__(self):
        return len(self.image_paths) // (self.batch_size * self.world_size)

    # ---------------------------------------------------------------- host side of one batch
    def draw(self, src_w, src_h, rng):
        """The random draws of one sample, in the reference's order: flip, then the crop corner inside the
        resized image (RandomCrop.get_params: randint(0, h - th), randint(0, w - tw))."""
        flip = 1 if ((self.train or self.flip_always_drawn) and rng.random() < 0.5) else 0
        rs_w, rs_h = resize_size(src_w, src_h, self.new_size)
        if self.crop:
            th, tw = self.height, self.width
            if rs_h < th or rs_w < tw:
                raise ValueError("crop (%d, %d) larger than the resized image (%d, %d)" % (th, tw, rs_h, rs_w))
            i = 0 if rs_h == th else rng.randint(0, rs_h - th)
            j = 0 if rs_w == tw else rng.randint(0, rs_w - tw)
        else:
            th, tw, i, j = rs_h, rs_w, 0, 0
        return flip, rs_h, rs_w, i, j, th, tw

    def _ensure_device(self):
        if not torch.cuda.is_available():
            raise RuntimeError("munit_amd.data: the transforms run on the GPU (libmunit_hip.so); no HIP device is "
                               "visible and there is no CPU fallback")
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=self.num_workers)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)

    def _submit_decodes(self, indices):
        """Hand the files of one batch to the decode threads; returns the futures (images, masks or None)."""
        futs = [self._pool.submit(_decode_rgb, self.image_paths[k]) for k in indices]
        mfuts = None
        if self.mask_paths is not None:
            mfuts = [self._pool.submit(_decode_mask, self.mask_paths[k]) for k in indices]
        return futs, mfuts

    def _finish_batch(self, futs, mfuts, rng, ring=None):
        """Collect a batch's decodes, draw its random parameters (in sample order, as the reference does per sample) and
        launch the device transform."""
        arrays = [f.result() for f in futs]
        masks = None if mfuts is None else [f.result() for f in mfuts]
        draws = [self.draw(a.shape[1], a.shape[0], rng) for a in arrays]
        return launch_transform(arrays, masks, draws, self.device, self._stream, ring, self._pool)

    def _launch_batch(self, indices, rng):
        """Decode one batch with the thread pool, draw its random parameters, launch the device transform."""
        futs, mfuts = self._submit_decodes(indices)
        return self._finish_batch(futs, mfuts, rng)

    def _hand_over(self, item):
        """Make the consumer's current stream wait for a produced batch and own its buffers."""
        out, ev, keep = item
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for x in (out if isinstance(out, tuple) else (out,)) + keep:
            if x.is_cuda:
                x.record_stream(cur)
        return out

    def _run_batch(self, indices, rng):
        """Synchronous single batch (dataset[i])."""
        self._ensure_device()
        return self._hand_over(self._launch_batch(indices, rng))

    # ---------------------------------------------------------------- iteration
    def __iter__(self):
        self._ensure_device()
        batches = shard_indices(len(self.image_paths), self.batch_size, self.train, self.seed + self.epoch, self.rank,
                                self.world_size)
        self.epoch += 1
        rng = self._rng
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        device = self.device

        # Decodes run AHEAD of the batch being assembled: the files of the next `ahead` batches are already with the decode
        # threads while the producer packs and uploads the current one (round 2 decoded one batch at a time, so at most
        # batch_size of the threads ever worked and every batch also paid a fresh pinned allocation: 170 images/s whatever
        # the thread count).  The draws stay in batch order, so the random stream is unchanged.
        ahead = max(2, (2 * self.num_workers) // max(1, self.batch_size * (2 if self.mask_paths is not None else 1)) + 1)
        ring = _StagingRing(self.prefetch + 3)

        def produce():
            try:
                torch.cuda.set_device(device)
                from collections import deque
                pending = deque()
                it = iter(batches)
                for idx in it:
                    pending.append(self._submit_decodes(idx))
                    if len(pending) >= ahead:
                        break
                while pending:
                    if stop.is_set():
                        return
                    futs, mfuts = pending.popleft()
                    nxt = next(it, None)
                    if nxt is not None:
                        pending.append(self._submit_decodes(nxt))
                    q.put(self._finish_batch(futs, mfuts, rng, ring))
                q.put(None)
            except BaseException as e:  # surface decode / launch errors in the consumer
                q.put(e)

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield self._hand_over(item)
        finally:
            stop.set()
            while t.is_alive():
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass



def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def get_data_loader_list(root, file_list, batch_size, train, new_size=None, height=256, width=256, num_workers=4,
                         crop=True, **kw):
    """List-based loader (scripts/utils.py:192-250): `file_list` holds one path per line, relative to `root`."""
    paths = [os.path.join(root, p) for p in default_flist_reader(file_list)]
    return DeviceBatchLoader(paths, None, batch_size, train, new_size, height, width, num_workers, crop, **kw)


def get_data_loader_folder(input_folder, batch_size, train, new_size=None, height=256, width=256, num_workers=4,
                           crop=True, **kw):
    """Folder-based loader (scripts/utils.py:680-740): every image under `input_folder`, sorted."""
    paths = sorted(make_dataset(input_folder))
    if not paths:
        raise RuntimeError("Found 0 images in: " + input_folder + "\nSupported image extensions are: " +
                           ",".join(IMG_EXTENSIONS))
    return DeviceBatchLoader(paths, None, batch_size, train, new_size, height, width, num_workers, crop, **kw)


def get_data_loader_mask_and_im(file_list, mask_list, batch_size, train, new_size=None, height=256, width=256,
                                num_workers=4, crop=True, **kw):
    """Image + mask loader (scripts/utils.py:638-677, MyDataset utils.py:270-363): yields (images, masks);
    with mask_list None the second element is an empty tensor, as in the reference.  MyDataset flips with
    probability 1/2 whether or not `train` is set (utils.py:309-312)."""
    paths = [rec[0] for rec in default_txt_reader(file_list)]
    mpaths = None if mask_list is None else [rec[0] for rec in default_txt_reader(mask_list)]
    loader = DeviceBatchLoader(paths, mpaths, batch_size, train, new_size, height, width, num_workers, crop,
                               torch_flip=True, **kw)
    if mpaths is None:
        return _WithEmptyMask(loader)
    return loader


class _WithEmptyMask:
    """MyDataset without a mask list returns (image, torch.tensor([])) (utils.py:357-360)."""

    def __init__(self, loader):
        self._loader = loader
        self.dataset = loader.dataset

    def __len__(self):
        return len(self._loader)

    def __iter__(self):
        for images in self._loader:
            yield images, torch.empty((images.shape[0], 0))


def get_all_data_loaders(conf, **kw):
    """train/test loaders of both domains (scripts/utils.py:50-156): `data_root` selects the folder layout
    (trainA/testA/trainB/testB), otherwise the data_folder_* / data_list_* keys are used.  Test loaders use
    new_size for the crop as well and do not shuffle or flip."""
    batch_size = conf["batch_size"]
    num_workers = conf["num_workers"]
    if "new_size" in conf:
        new_size_a = new_size_b = conf["new_size"]
    else:
        new_size_a, new_size_b = conf["new_size_a"], conf["new_size_b"]
    height, width = conf["crop_image_height"], conf["crop_image_width"]
    out = []
    for dom, ns in (("a", new_size_a), ("b", new_size_b)):
        for train in (True, False):
            h, w = (height, width) if train else (ns, ns)
            if "data_root" in conf:
                folder = os.path.join(conf["data_root"], ("train" if train else "test") + dom.upper())
                out.append(get_data_loader_folder(folder, batch_size, train, ns, h, w, num_workers, True, **kw))
            else:
                split = "train" if train else "test"
                out.append(get_data_loader_list(conf["data_folder_%s_%s" % (split, dom)],
                                                conf["data_list_%s_%s" % (split, dom)], batch_size, train, ns, h, w,
                                                num_workers, True, **kw))
    train_a, test_a, train_b, test_b = out
    return train_a, train_b, test_a, test_b
