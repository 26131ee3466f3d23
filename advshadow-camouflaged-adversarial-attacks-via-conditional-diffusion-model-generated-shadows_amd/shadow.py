"""Shadow composites of the attack path on the GPU.

* ``apply_shadow`` — tensor form, closed-form part of tools/train_shadow.py:224-266 (variants
  ddim2/diff_model2.py:615-654 without blur, ddim2/test.py:830-871 with intensity 0.051): circular
  shadow mask, 5x5 Gaussian softening, feature-mask intersection, darkening, clamp.  With a
  ``classifier`` the gradient attack the reference interleaves (``apply_adversarial_perturbation``) runs on the
  victim's HIP backward plan (``adversarial.py``); with ``classifier=None`` the closed form alone.
* ``add_shadow`` — add_shadow.py:35-60 as a function: triangle shadow in the bounding box of the
  mask's largest blob, ``Image.alpha_composite`` then ``Image.composite`` through the mask.
* ``add_shadow_to_mask_area`` — shadow_for_attack.py:22-93.
The per-pixel arithmetic (Pillow's fixed-point blends, the float darkening) runs in
``advs_composite_u8`` / ``advs_apply_shadow``; the three triangle vertices are rasterised on the
host with ``ImageDraw.polygon`` exactly as the reference does.
"""
import ctypes as C
import random

import numpy as np
import torch
from PIL import Image, ImageDraw

from . import _lib
from ._lib import check

# cv2.GaussianBlur(mask, (k, k), 0): for odd k <= 7 and sigma <= 0 OpenCV uses fixed binomial taps
# (small_gaussian_tab in smooth.dispatch.cpp), not the sigma = 0.3*((k-1)/2-1)+0.8 Gaussian.
_GAUSS_TAPS = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
               7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def gaussian_taps(ksize):
    if ksize in _GAUSS_TAPS:
        return list(_GAUSS_TAPS[ksize])
    raise ValueError(f"blur_kernel_size={ksize}: only odd sizes up to 7 are supported")


def apply_shadow_batch(images, centers, radii, feature_masks, shadow_intensity=0.43, blur_kernel_size=5, out=None,
                       stream=None):
    """images [B,C,H,W] f32 in [0,1] (GPU), centers [B,2] (x, y), radii [B], feature_masks [B,1|C,H,W]."""
    _lib.init_device()
    lib = _lib.load()
    B, Cc, H, W = images.shape
    dev = images.device
    images = images.contiguous().float()
    fm = feature_masks.to(dev, torch.float32).contiguous()
    ctr = torch.as_tensor(centers, dtype=torch.float32).to(dev).reshape(B, 2).contiguous()
    rad = torch.as_tensor(radii, dtype=torch.float32).to(dev).reshape(B).contiguous()
    out = torch.empty_like(images) if out is None else out
    taps = gaussian_taps(blur_kernel_size if blur_kernel_size else 1)
    arr = (C.c_float * len(taps))(*taps)
    s = (stream or torch.cuda.current_stream(dev)).cuda_stream
    check(lib.advs_apply_shadow(images.data_ptr(), fm.data_ptr(), ctr.data_ptr(), rad.data_ptr(), out.data_ptr(),
                                B, Cc, H, W, fm.shape[1], float(shadow_intensity), arr, len(taps), s), "apply_shadow")
    return out


def apply_shadow(image, shadow_center, shadow_radius, feature_mask, classifier=None, target_label=None, device=None,
                 shadow_intensity=0.43, epsilon=0.01, blur_kernel_size=5):
    """Reference signature (tools/train_shadow.py:224-225).  image [C,H,W] in [0,1]."""
    dev = torch.device(device) if device is not None else (image.device if image.is_cuda else torch.device("cuda"))
    img = image.to(dev)[None]
    fm = feature_mask.to(dev)
    fm = fm[None] if fm.dim() == 3 else fm[None, None]
    ctr = torch.as_tensor([float(shadow_center[0]), float(shadow_center[1])])[None]
    if classifier is not None:
        from .adversarial import apply_shadow_adversarial_batch
        if target_label is None:
            raise ValueError("apply_shadow: a classifier needs target_label (train_shadow.py:259)")
        return apply_shadow_adversarial_batch(classifier, img, ctr, torch.as_tensor([float(shadow_radius)]), fm,
                                              torch.as_tensor(target_label).reshape(1), shadow_intensity, epsilon,
                                              blur_kernel_size)[0]
    return apply_shadow_batch(img, ctr, torch.as_tensor([float(shadow_radius)]), fm, shadow_intensity, blur_kernel_size)[0]


def create_shadow_mask(image_size, shadow_center, shadow_radius, device="cpu"):
    """tools/train_shadow.py:156-174 (host helper, tiny)."""
    _, H, W = image_size
    Y, X = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    d = torch.sqrt((X.to(device) - shadow_center[0]) ** 2 + (Y.to(device) - shadow_center[1]) ** 2)
    return (d <= shadow_radius).float()


# --------------------------------------------------------------------------- PIL composites
def mask_blobs(mask_l):
    """Bounding boxes (x, y, w, h, area) of the 8-connected foreground blobs of an L-mode mask: the
    host stand-in for cv2.findContours(EXTERNAL) + boundingRect (cv2 is not a dependency here)."""
    m = np.asarray(mask_l) != 0
    H, W = m.shape
    lab = np.zeros((H, W), dtype=np.int32)
    boxes = []
    for y0, x0 in zip(*np.nonzero(m)):
        if lab[y0, x0]:
            continue
        idx = len(boxes) + 1
        stack = [(y0, x0)]
        lab[y0, x0] = idx
        xs0, xs1, ys0, ys1, area = x0, x0, y0, y0, 0
        while stack:
            y, x = stack.pop()
            area += 1
            xs0, xs1, ys0, ys1 = min(xs0, x), max(xs1, x), min(ys0, y), max(ys1, y)
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    yy, xx = y + dy, x + dx
                    if 0 <= yy < H and 0 <= xx < W and m[yy, xx] and not lab[yy, xx]:
                        lab[yy, xx] = idx
                        stack.append((yy, xx))
        boxes.append((int(xs0), int(ys0), int(xs1 - xs0 + 1), int(ys1 - ys0 + 1), area))
    return boxes


def _triangle_layer(size, cx, cy, t):
    layer = Image.new("RGBA", size, (255, 255, 255, 0))
    ImageDraw.Draw(layer).polygon([(cx, cy - t), (cx - t, cy + t), (cx + t, cy + t)], fill=(0, 0, 0, 128))
    return layer


def _composite(image_rgb, layer, mask_l, mode, factor=0.43):
    _lib.init_device()
    dev = torch.device("cuda", torch.cuda.current_device())
    img = torch.from_numpy(np.asarray(image_rgb, dtype=np.uint8).copy()).to(dev)
    lay = torch.from_numpy(np.asarray(layer, dtype=np.uint8).copy()).to(dev)
    msk = torch.from_numpy(np.asarray(mask_l, dtype=np.uint8).copy()).to(dev)
    out = torch.empty_like(img)
    s = torch.cuda.current_stream(dev).cuda_stream
    check(_lib.load().advs_composite_u8(img.data_ptr(), lay.data_ptr(), msk.data_ptr(), out.data_ptr(),
                                        img.shape[0] * img.shape[1], mode, float(factor), s), "composite_u8")
    return Image.fromarray(out.cpu().numpy())


def add_shadow(image, mask):
    """add_shadow.py:35-60: PIL RGB image + mask -> PIL RGB image with the triangle shadow."""
    image = image.convert("RGB")
    mask_l = mask.convert("L")
    boxes = mask_blobs(mask_l)
    if not boxes:
        raise ValueError("mask has no foreground (add_shadow.py:44 would fail on max() of no contours)")
    x, y, w, h, _ = max(boxes, key=lambda b: b[4])
    cx, cy, t = x + w // 2, y + h // 2, min(w, h) // 2
    return _composite(image, _triangle_layer(image.size, cx, cy, t), mask_l, 0)


def add_shadow_to_mask_area(image, mask, rng=random):
    """shadow_for_attack.py:22-93 (triangle in the centre part of a randomly chosen blob, then the
    0.43 darkening of every masked pixel)."""
    mask_l = mask.convert("L")
    boxes = mask_blobs(mask_l)
    if not boxes:
        return image
    x, y, w, h, _ = rng.choice(boxes)
    sx, sy, sw, sh = x + w // 4, y + h // 4, w // 2, h // 2
    cx, cy, t = sx + sw // 2, sy + sh // 2, min(sw, sh) // 3
    layer = _triangle_layer(mask.size, cx, cy, t)
    if mask_l.size != image.size:
        mask_l = mask_l.resize(image.size, Image.NEAREST)     # cv2.resize(..., INTER_NEAREST) in the reference
    return _composite(image.convert("RGB"), layer, mask_l, 1, 0.43)
