"""Folder translation with a trained generator -- the reference's scripts/test.py:86-129 on munit_amd:
style image -> gen.encode(., 2) style code; every content image -> gen.encode(., 1) content; gen.decode(c, s, 2);
outputs saved as JPEG after the same (x + 1) / 2 de-normalisation and per-image min-max scaling that
torchvision.utils.save_image(normalize=True) applies.  The Resize + ToTensor + Normalize transform runs on the GPU
(munit_amd.data, bit-identical to PIL + torchvision).

  python examples/translate_folder.py --config cfg.yaml --checkpoint outputs/checkpoints/gen_00100000.pt \
         --input-folder content/ --style style.jpg --output-folder out/
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_image(path, new_size, dev):
    from PIL import Image
    from munit_amd import data as D
    arr = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8))
    rs_w, rs_h = D.resize_size(arr.shape[1], arr.shape[0], new_size)
    return D.transform_batch([arr], None, [(0, rs_h, rs_w, 0, 0, rs_h, rs_w)], dev)      # (1, 3, h, w), [-1, 1]


def save_image(x, path):
    """vutils.save_image(x, padding=0, normalize=True) for a single image: min-max to [0, 1], x255 + 0.5, clamp."""
    from PIL import Image
    x = x[0].float()
    lo, hi = float(x.min()), float(x.max())
    x = (x - lo) / max(hi - lo, 1e-5)
    arr = x.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
    Image.fromarray(arr).save(path)


def main(argv=None, keep=None):
    """keep: optional list that receives the translated tensors (x_ab per content image, before the JPEG encoding)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--checkpoint", default=None, help="gen_XXXXXXXX.pt written by MUNIT_Trainer.save (random weights if omitted)")
    ap.add_argument("--input-folder", required=True)
    ap.add_argument("--style", required=True)
    ap.add_argument("--output-folder", required=True)
    ap.add_argument("--save-input", action="store_true")
    args = ap.parse_args(argv)

    from munit_amd import data as D
    from munit_amd.trainer import MUNIT_Trainer
    from munit_amd.utils import get_config
    dev = torch.device("cuda", torch.cuda.current_device())
    config = get_config(args.config)
    new_size = config["new_size"]
    trainer = MUNIT_Trainer(config)
    if args.checkpoint:
        state = torch.load(args.checkpoint, weights_only=True)            # {"2": sd} (gen_state 1) or {"a", "b"}
        if trainer.gen_state == 1:
            trainer.gen.load_state_dict(state["2"])
        else:
            trainer.gen_a.load_state_dict(state["a"])
            trainer.gen_b.load_state_dict(state["b"])
    trainer.to(dev)
    trainer.eval()
    os.makedirs(args.output_folder, exist_ok=True)
    if trainer.gen_state == 1:
        enc = lambda x, k: trainer.gen.encode(x, k)                                     # noqa: E731
        dec = lambda c, s, k: trainer.gen.decode(c, s, k)                               # noqa: E731
    else:
        enc = lambda x, k: (trainer.gen_a if k == 1 else trainer.gen_b).encode(x)       # noqa: E731
        dec = lambda c, s, k: (trainer.gen_a if k == 1 else trainer.gen_b).decode(c, s)  # noqa: E731
    outs = []
    with torch.no_grad():
        _, s_b = enc(load_image(args.style, new_size, dev), 2)
        for j, path in enumerate(sorted(D.make_dataset(args.input_folder))):
            x_a = load_image(path, new_size, dev)
            if args.save_input:
                save_image((x_a + 1) / 2.0, os.path.join(args.output_folder, "input{:03d}.jpg".format(j)))
            c_a, _ = enc(x_a, 1)
            x_ab = dec(c_a, s_b, 2)
            if keep is not None:
                keep.append(x_ab.detach().float().cpu())
            out = os.path.join(args.output_folder, "output{:03d}.jpg".format(j))
            save_image((x_ab + 1) / 2.0, out)
            outs.append(out)
    return outs


if __name__ == "__main__":
    main()
