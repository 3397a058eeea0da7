"""Host logic of the input pipeline (no GPU): list readers, Resize size rule, epoch sharding, batch packing,
and the data oracle's own invariants."""
import ctypes
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from munit_amd import data as D            # noqa: E402
from munit_amd._lib import ImageDesc       # noqa: E402
from oracle import data_oracle as O        # noqa: E402


def test_resize_size_rule_matches_oracle():
    for (w, h, s) in [(1024, 768, 256), (300, 500, 256), (256, 400, 256), (400, 256, 256), (256, 256, 256),
                      (77, 113, 64), (113, 77, 200), (500, 500, 100), (640, 480, None)]:
        rs = O.resize_size(w, h, s)
        want = (w, h) if rs is None else rs
        assert D.resize_size(w, h, s) == want


def test_list_readers(tmp_path):
    f = tmp_path / "l.txt"
    f.write_text("a/b.png\n c.jpg  extra\n")
    assert D.default_flist_reader(str(f)) == ["a/b.png", "c.jpg  extra"]
    assert D.default_txt_reader(str(f)) == [["a/b.png"], ["c.jpg", "extra"]]


def test_make_dataset_recurses_and_filters(tmp_path):
    (tmp_path / "sub").mkdir()
    for n in ("x.png", "y.txt", "sub/z.JPG", "sub/w.gif"):
        (tmp_path / n).write_bytes(b"0")
    got = sorted(os.path.relpath(p, tmp_path) for p in D.make_dataset(str(tmp_path)))
    assert got == ["sub/z.JPG", "x.png"]
    with pytest.raises(AssertionError):
        D.make_dataset(str(tmp_path / "missing"))


def test_shard_indices_partition_and_drop_last():
    full = D.shard_indices(23, 4, True, 5)
    assert len(full) == 5 and all(len(b) == 4 for b in full)
    flat = [i for b in full for i in b]
    assert len(set(flat)) == 20 and set(flat) <= set(range(23))
    assert D.shard_indices(23, 4, True, 5) == full                # same seed, same epoch order
    assert D.shard_indices(23, 4, True, 6) != full
    r0 = D.shard_indices(23, 4, True, 5, 0, 2)
    r1 = D.shard_indices(23, 4, True, 5, 1, 2)
    a = {i for b in r0 for i in b}
    b = {i for b in r1 for i in b}
    assert not (a & b) and len(r0) == len(r1) == 2                # disjoint shards of the same permutation
    assert D.shard_indices(10, 3, False, 0) == [[0, 1, 2], [3, 4, 5], [6, 7, 8]]


def test_draw_order_and_bounds():
    ld = D.DeviceBatchLoader(["a"] * 4, None, 2, True, 256, 256, 256, rank=0, world_size=1)
    rng = random.Random(3)
    ref = random.Random(3)
    flip, rs_h, rs_w, i, j, th, tw = ld.draw(1024, 768, rng)
    assert flip == (1 if ref.random() < 0.5 else 0)
    assert (rs_w, rs_h) == (341, 256) and (th, tw) == (256, 256)
    assert i == 0                                               # height already equals the crop: no draw consumed
    assert j == ref.randint(0, 341 - 256)
    test = D.DeviceBatchLoader(["a"], None, 1, False, 256, 256, 256, rank=0, world_size=1)
    assert test.draw(300, 300, random.Random(0))[0] == 0        # no flip outside training
    with pytest.raises(ValueError):
        D.DeviceBatchLoader(["a"], None, 1, True, 128, 256, 256, rank=0, world_size=1).draw(300, 300, random.Random(0))


def test_pack_batch_layout():
    arrays = [np.zeros((5, 7, 3), np.uint8), np.zeros((4, 4, 3), np.uint8)]
    masks = [np.zeros((5, 7), np.uint8), np.zeros((9, 3), np.uint8)]
    draws = [(1, 5, 7, 0, 1, 4, 4), (0, 4, 4, 0, 0, 4, 4)]
    descs, offs, moffs, total = D.pack_batch(arrays, masks, draws)
    assert ctypes.sizeof(ImageDesc) == 40
    assert offs[0] == 256 and offs[1] == 256 + 112 and all(o % 16 == 0 for o in offs + moffs)
    assert moffs[0] >= offs[1] + 48 and total >= moffs[1] + 27
    assert (descs[0].src_h, descs[0].src_w, descs[0].flip, descs[0].crop_j) == (5, 7, 1, 1)
    assert (descs[3].src_h, descs[3].src_w, descs[3].src_off) == (9, 3, moffs[1])


def test_loader_refuses_cpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ld = D.DeviceBatchLoader(["a.png"], None, 1, False, None, 4, 4, rank=0, world_size=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        iter(ld).__next__()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        D.transform_batch([np.zeros((4, 4, 3), np.uint8)], None, [(0, 4, 4, 0, 0, 4, 4)])


def test_oracle_identity_and_range():
    from PIL import Image
    rng = np.random.RandomState(0)
    arr = rng.randint(0, 256, (40, 60, 3)).astype(np.uint8)
    t = O.transform_image(Image.fromarray(arr), False, None, None)
    want = (torch.from_numpy(arr).permute(2, 0, 1).float() / 255 - 0.5) / 0.5
    assert torch.equal(t, want) and t.min() >= -1 and t.max() <= 1
    tf = O.transform_image(Image.fromarray(arr), True, None, (3, 5, 20, 30))
    assert torch.equal(tf, want.flip(2)[:, 3:23, 5:35])
    m = (rng.rand(40, 60) > 0.5).astype(np.uint8)
    mt = O.transform_mask(Image.fromarray(m), False, (0, 0, 40, 60))
    assert torch.equal(mt[0], torch.from_numpy(m).float() / 255 * 255)
