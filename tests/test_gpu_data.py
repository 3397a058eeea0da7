"""Input pipeline parity (GPU): munit_image_preprocess / munit_mask_preprocess through the C ABI against the
PIL + torch-CPU oracle (oracle/data_oracle.py).  Integer stages are bit-exact; so is the float stage, because
the kernel performs the same fp32 operations in the same order (v/255, -0.5, /0.5)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from munit_amd import data as D          # noqa: E402
from oracle import data_oracle as O      # noqa: E402


def _img(rng, h, w, kind):
    if kind == "noise":
        return rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    if kind == "extreme":                  # saturated checkerboards: exercises the clip8 rounding paths
        a = ((np.add.outer(np.arange(h), np.arange(w)) % 2) * 255).astype(np.uint8)
        return np.stack([a, 255 - a, a], -1)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    return np.stack([(yy * 3 + xx) % 256, (xx * 5) % 256, (yy * 7 + 13) % 256], -1).astype(np.uint8)


def _check(arrays, draws, new_size, masks=None):
    from PIL import Image
    out = D.transform_batch(arrays, masks, draws)
    images = out[0] if masks is not None else out
    assert images.shape == (len(arrays), 3, draws[0][5], draws[0][6])
    assert images.is_contiguous(memory_format=torch.channels_last)
    for b, (a, d) in enumerate(zip(arrays, draws)):
        flip, rs_h, rs_w, i, j, th, tw = d
        want = O.transform_image(Image.fromarray(a), bool(flip), new_size, (i, j, th, tw))
        got = images[b].cpu()
        assert torch.equal(got, want), "image %d: max diff %g" % (b, (got - want).abs().max().item())
        if masks is not None:
            wm = O.transform_mask(Image.fromarray(masks[b]), bool(flip), (i, j, th, tw))
            gm = out[1][b].cpu()
            assert torch.equal(gm, wm), "mask %d: %d pixels differ" % (b, (gm != wm).sum().item())


def test_downscale_crop_flip_bit_exact():
    rng = np.random.RandomState(1)
    pr = random.Random(1)
    sizes = [(768, 1024), (600, 450), (256, 256), (300, 257), (1080, 1920), (257, 700), (511, 513), (256, 1000)]
    arrays = [_img(rng, h, w, k) for (h, w), k in zip(sizes, ["noise", "ramp", "noise", "extreme"] * 2)]
    ld = D.DeviceBatchLoader(["x"] * 8, None, 8, True, 256, 256, 256, rank=0, world_size=1)
    draws = [ld.draw(a.shape[1], a.shape[0], pr) for a in arrays]
    assert any(d[0] for d in draws) and any(not d[0] for d in draws)
    _check(arrays, draws, 256)


def test_upscale_and_identity_bit_exact():
    rng = np.random.RandomState(2)
    pr = random.Random(2)
    arrays = [_img(rng, 90, 130, "noise"), _img(rng, 200, 120, "ramp"), _img(rng, 64, 64, "extreme"),
              _img(rng, 128, 128, "noise")]
    ld = D.DeviceBatchLoader(["x"] * 4, None, 4, True, 128, 96, 112, rank=0, world_size=1)
    draws = [ld.draw(a.shape[1], a.shape[0], pr) for a in arrays]
    _check(arrays, draws, 128)
    # no Resize at all (new_size None), crop only
    ld2 = D.DeviceBatchLoader(["x"] * 4, None, 4, True, None, 60, 60, rank=0, world_size=1)
    draws2 = [ld2.draw(a.shape[1], a.shape[0], pr) for a in arrays]
    _check(arrays, draws2, None)


def test_hd_config_sizes():
    # config_HD.yaml: new_size 600, crop 512x512
    rng = np.random.RandomState(3)
    pr = random.Random(3)
    arrays = [_img(rng, 1200, 1600, "noise"), _img(rng, 700, 650, "ramp")]
    ld = D.DeviceBatchLoader(["x"] * 2, None, 2, True, 600, 512, 512, rank=0, world_size=1)
    draws = [ld.draw(a.shape[1], a.shape[0], pr) for a in arrays]
    _check(arrays, draws, 600)


def test_masks_nearest_crop_quirk_and_scale_rule():
    rng = np.random.RandomState(4)
    pr = random.Random(7)
    sizes = [(768, 1024), (480, 640), (256, 300), (1000, 600)]
    arrays = [_img(rng, h, w, "noise") for h, w in sizes]
    masks = [(rng.rand(768, 1024) > 0.5).astype(np.uint8) * 255,      # 0/255 mask
             (rng.rand(480, 640) > 0.7).astype(np.uint8),             # 0/1 mask: the x255 rule fires
             rng.randint(0, 7, (256, 300)).astype(np.uint8),          # label map, max > 1
             np.zeros((500, 300), np.uint8)]                          # empty mask of another size than its image
    ld = D.DeviceBatchLoader(["x"] * 4, ["m"] * 4, 4, True, 256, 256, 256, rank=0, world_size=1, torch_flip=True)
    draws = [ld.draw(a.shape[1], a.shape[0], pr) for a in arrays]
    assert any(d[4] > 0 for d in draws)        # a non-zero horizontal offset: the zero-filled crop strip is exercised
    _check(arrays, draws, 256, masks)


def test_loader_end_to_end(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(5)
    paths, mpaths = [], []
    for k in range(6):
        h, w = int(rng.randint(260, 400)), int(rng.randint(260, 500))
        p = tmp_path / ("im%d.png" % k)
        Image.fromarray(_img(rng, h, w, "noise")).save(p)
        m = tmp_path / ("mask%d.png" % k)
        Image.fromarray(((rng.rand(h, w) > 0.5) * 255).astype(np.uint8)).save(m)
        paths.append(str(p))
        mpaths.append(str(m))
    fl, ml = tmp_path / "files.txt", tmp_path / "masks.txt"
    fl.write_text("\n".join(paths) + "\n")
    ml.write_text("\n".join(mpaths) + "\n")
    ld = D.get_data_loader_mask_and_im(str(fl), str(ml), 2, True, new_size=256, height=256, width=256, num_workers=2,
                                       seed=11, rank=0, world_size=1)
    assert len(ld) == 3
    # replay the loader's random stream and epoch order on the host to build the expectation
    replay = random.Random()
    replay.setstate(ld._rng.getstate())
    order = D.shard_indices(6, 2, True, ld.seed + ld.epoch)
    n = 0
    for (images, masks), idx in zip(ld, order):
        assert images.shape == (2, 3, 256, 256) and masks.shape == (2, 1, 256, 256)
        for b, k in enumerate(idx):
            im = Image.open(paths[k]).convert("RGB")
            flip, rs_h, rs_w, i, j, th, tw = ld.draw(im.size[0], im.size[1], replay)
            want = O.transform_image(im, bool(flip), 256, (i, j, th, tw))
            assert torch.equal(images[b].cpu(), want)
            wm = O.transform_mask(Image.open(mpaths[k]), bool(flip), (i, j, th, tw))
            assert torch.equal(masks[b].cpu(), wm)
        n += 1
    assert n == 3
    # folder loader + dataset[i]
    fld = D.get_data_loader_folder(str(tmp_path), 3, False, new_size=256, height=256, width=256, num_workers=2,
                                   rank=0, world_size=1)
    assert len(fld) == 4                         # 12 png files (images and masks), test mode keeps file order
    files = sorted(D.make_dataset(str(tmp_path)))
    replay = random.Random()
    replay.setstate(fld._rng.getstate())
    first = next(iter(fld))
    im0 = Image.open(files[0]).convert("RGB")
    # RandomCrop stays in the test-mode chain of the reference (utils.py:721-725): offsets are still drawn
    flip, rs_h, rs_w, i, j, th, tw = fld.draw(im0.size[0], im0.size[1], replay)
    assert flip == 0
    assert torch.equal(first[0].cpu(), O.transform_image(im0, False, 256, (i, j, th, tw)))
    replay.setstate(fld._rng.getstate())
    one = fld.dataset[0]
    flip, rs_h, rs_w, i, j, th, tw = fld.draw(im0.size[0], im0.size[1], replay)
    assert one.shape == (3, 256, 256)
    assert torch.equal(one.cpu(), O.transform_image(im0, False, 256, (i, j, th, tw)))


def test_train_loop_example_runs(tmp_path):
    """examples/train_loop.py: folder loaders -> trainer, a few iterations at 64x64 (end-to-end drop-in path)."""
    import yaml
    from PIL import Image
    import bench
    rng = np.random.RandomState(9)
    for dom in ("trainA", "trainB"):
        (tmp_path / dom).mkdir()
        for k in range(4):
            Image.fromarray(_img(rng, 80 + k, 96, "noise")).save(tmp_path / dom / ("i%d.png" % k))
    hp = bench.bench_hp(64, 2)
    hp.update(new_size=64, data_root=str(tmp_path), num_workers=2, ratio_disc_gen=1)
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(yaml.safe_dump(hp))
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_loop
    tr = train_loop.main(["--config", str(cfg), "--iters", "3", "--output", str(tmp_path / "ckpt"), "--save-every", "3"])
    assert tr.iterations == 2
    for name in ("loss_dis_total", "loss_gen_total"):
        v = float(getattr(tr, name))
        assert v == v and v > 0
    assert sorted(os.listdir(tmp_path / "ckpt"))[:2] == ["dis_00000003.pt", "gen_00000003.pt"]


def test_translate_folder_example_and_sample_fid(tmp_path):
    """examples/translate_folder.py (the reference's test.py harness) end to end, and MUNIT_Trainer.sample_fid."""
    import yaml
    from PIL import Image
    import bench
    from munit_amd.trainer import MUNIT_Trainer
    rng = np.random.RandomState(4)
    (tmp_path / "content").mkdir()
    for k in range(2):
        Image.fromarray(_img(rng, 70 + 8 * k, 90, "noise")).save(tmp_path / "content" / ("c%d.png" % k))
    Image.fromarray(_img(rng, 100, 80, "ramp")).save(tmp_path / "style.png")
    hp = bench.bench_hp(64, 1)
    hp.update(new_size=64)
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(yaml.safe_dump(hp))
    torch.manual_seed(1234)
    tr = MUNIT_Trainer(hp)
    tr.to(torch.device("cuda:0"))
    (tmp_path / "ckpt").mkdir()
    tr.save(str(tmp_path / "ckpt"), 0)
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import translate_folder
    kept = []
    outs = translate_folder.main(["--config", str(cfg), "--checkpoint", str(tmp_path / "ckpt" / "gen_00000001.pt"),
                                  "--input-folder", str(tmp_path / "content"), "--style", str(tmp_path / "style.png"),
                                  "--output-folder", str(tmp_path / "out"), "--save-input"], keep=kept)
    assert [os.path.basename(o) for o in outs] == ["output000.jpg", "output001.jpg"]
    # the same harness on the oracle (scripts/test.py:86-129): PIL / torchvision-restated transform of the files, fp64
    # GenView with the checkpoint's weights -- style from encode(., 2), content from encode(., 1), decode(., ., 2)
    from oracle import data_oracle as DO
    from oracle import munit_oracle as O
    from tests.parity import nerr
    sd = torch.load(tmp_path / "ckpt" / "gen_00000001.pt", weights_only=True)["2"]
    gen64 = {k: v.double() for k, v in sd.items() if v.dtype == torch.float32 and "running_" not in k}
    view = O.GenView(gen64, hp["gen"], True)
    load = lambda f: DO.transform_image(Image.open(f).convert("RGB"), False, 64, None)[None].double()   # noqa: E731
    with torch.no_grad():
        _, s_ref = view.encode(load(tmp_path / "style.png"), 2)
        for k in range(2):
            c_ref, _ = view.encode(load(tmp_path / "content" / ("c%d.png" % k)), 1)
            want = view.decode(c_ref, s_ref, 2)
            assert tuple(kept[k].shape) == tuple(want.shape)
            assert nerr(kept[k], want) <= 1e-4, (k, nerr(kept[k], want))
    im = Image.open(outs[0])
    rs_w = int(64 * 90 / 70)                                          # Resize(64): shorter side 64, aspect kept -> 82
    down = lambda n: (n + 2 - 4) // 2 + 1                             # the two 4x4 stride-2 convs, then two x2 up-samplings
    assert im.size == (4 * down(down(rs_w)), 64)
    assert sorted(os.listdir(tmp_path / "out")) == ["input000.jpg", "input001.jpg", "output000.jpg", "output001.jpg"]
    # sample_fid = decode(content(x_a), style(x_b)) per sample
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)).mul(2).sub(1).to("cuda:0")
    y = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(2)).mul(2).sub(1).to("cuda:0")
    got = tr.sample_fid(x, y)
    with torch.no_grad():
        tr.eval()
        c, _ = tr.gen.encode(x[1:2], 1)
        _, s = tr.gen.encode(y[1:2], 2)
        want = tr.gen.decode(c, s, 2)
        tr.train()
    assert got.shape == (2, 3, 64, 64) and torch.equal(got[1:2], want)
