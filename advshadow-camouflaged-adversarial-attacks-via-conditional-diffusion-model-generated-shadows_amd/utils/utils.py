"""Image saving of the generation path (utils/utils.py:51-91): uint8 NCHW batch -> files."""
import os

import torch
from PIL import Image

from ..imageops import u8_nchw_to_hwc


def check_and_create_dir(path):
    os.makedirs(path, exist_ok=True)


def make_grid(images, nrow=8, padding=2, pad_value=0):
    """torchvision.utils.make_grid semantics for a uint8 [n,3,H,W] batch (plumbing: tile copies)."""
    if images.dim() == 3:
        images = images[None]
    n, c, H, W = images.shape
    if n == 1:
        return images[0]
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    h, w = H + padding, W + padding
    grid = torch.full((c, h * ymaps + padding, w * xmaps + padding), pad_value, dtype=images.dtype, device=images.device)
    for k in range(n):
        y, x = divmod(k, xmaps)
        grid[:, y * h + padding:y * h + padding + H, x * w + padding:x * w + padding + W] = images[k]
    return grid


def _to_pil(chw_u8):
    hwc = u8_nchw_to_hwc(chw_u8[None].contiguous())[0] if chw_u8.is_cuda else chw_u8.permute(1, 2, 0)
    return Image.fromarray(hwc.cpu().numpy())


def save_images(images, path, **kwargs):
    """utils/utils.py:51-62: one grid image of the whole batch."""
    _to_pil(make_grid(images, **kwargs)).save(path)


def save_one_image_in_images(images, path, generate_name, image_size=None, image_format="jpg", **kwargs):
    """utils/utils.py:65-91: ``name_{k}.fmt`` per image (+ ``name_{size}_{k}.fmt`` resized copies)."""
    for count, img in enumerate(images):
        im = _to_pil(img)
        im.save(os.path.join(path, f"{generate_name}_{count}.{image_format}"))
        if image_size is not None:
            # Image.ANTIALIAS (utils/utils.py:88) was removed in Pillow 10; LANCZOS is the same filter
            im.resize((image_size, image_size), Image.LANCZOS).save(
                os.path.join(path, f"{generate_name}_{image_size}_{count}.{image_format}"))
