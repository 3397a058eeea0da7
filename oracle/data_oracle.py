"""CPU oracle of the input pipeline -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module.  It restates, on PIL + torch-CPU, the per-sample transform
chains of the reference's loaders:

  * list / folder loaders (scripts/utils.py:229-249 and :717-738):
      Compose([RandomHorizontalFlip, Resize(new_size), RandomCrop((h, w)), ToTensor, Normalize(.5, .5)])
  * MyDataset.transform (scripts/utils.py:296-345): the same chain for the image plus the mask chain
      flip -> mask.resize((image.width, image.height), NEAREST) -> crop(i, j, h, w) -> ToTensor (x255 if max == 1)

The arithmetic lives in third-party code that the reference pins in requirements.txt and that is not
part of /root/reference: torchvision==0.2.1 (transform semantics, restated below from its documented
behaviour -- torchvision is NOT installed here, so this restatement is "parity unpinned" against
torchvision itself) and Pillow==6.2.0 (Image.resize / transpose / crop).  Pillow IS installed in this
image (12.x, same resampling code path), so the pixel arithmetic of the oracle is Pillow's own and the
HIP kernels are checked against it bit for bit.  The random draws (flip, crop offset) are inputs here.
"""
import numpy as np
import torch
from PIL import Image

_FLIP = getattr(Image, "Transpose", Image).FLIP_LEFT_RIGHT
_BILINEAR = getattr(Image, "Resampling", Image).BILINEAR
_NEAREST = getattr(Image, "Resampling", Image).NEAREST


def resize_size(w, h, size):
    """torchvision.transforms.functional.resize with an int size: shorter side -> size, aspect kept
    with int() truncation; None when the image is returned untouched."""
    if size is None:
        return None
    if (w <= h and w == size) or (h <= w and h == size):
        return None
    if w < h:
        return size, int(size * h / w)
    return int(size * w / h), size


def to_tensor(pic):
    """torchvision ToTensor for 8-bit PIL images: HWC bytes -> CHW float / 255."""
    nch = len(pic.getbands())
    arr = np.frombuffer(pic.tobytes(), dtype=np.uint8).reshape(pic.size[1], pic.size[0], nch)
    t = torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous()
    return t.float().div(255)


def transform_image(img, flip, new_size, crop):
    """img: PIL RGB.  crop: (i, j, h, w) or None.  Returns (3, h, w) float32 tensor in [-1, 1]."""
    if flip:
        img = img.transpose(_FLIP)
    rs = resize_size(img.size[0], img.size[1], new_size)
    if rs is not None:
        img = img.resize(rs, _BILINEAR)
    if crop is not None:
        i, j, h, w = crop
        img = img.crop((j, i, j + w, i + h))
    t = to_tensor(img)
    return t.sub_(0.5).div_(0.5)     # Normalize((.5,.5,.5), (.5,.5,.5))


def resized_hw(w, h, new_size):
    rs = resize_size(w, h, new_size)
    return (h, w) if rs is None else (rs[1], rs[0])


def transform_mask(mask, flip, crop):
    """mask: single-band 8-bit PIL image.  crop = (i, j, h, w) of the IMAGE; the mask is resized to the crop
    size and then cropped at the same offsets (the reference's behaviour, utils.py:322-324)."""
    i, j, h, w = crop
    if flip:
        mask = mask.transpose(_FLIP)
    mask = mask.resize((w, h), _NEAREST)
    mask = mask.crop((j, i, j + w, i + h))
    if np.max(mask) == 1:
        return to_tensor(mask) * 255
    return to_tensor(mask)
