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


# --------------------------------------------------------------------------- contours (cv2.findContours stand-in)
MAX_CONTOURS = 4096


def cv_gray(mask):
    """``np.array(mask)`` + ``cv2.cvtColor(RGB2GRAY)`` for 3-channel masks (add_shadow.py:36-38, shadow_for_attack.py:26-28):
    OpenCV's 8-bit fixed point (R*4899 + G*9617 + B*1868 + 8192) >> 14, which is NOT Pillow's ``convert('L')``.
    (The 14-bit constants are those of OpenCV up to 4.x's generic 8-bit path; builds that take the 15-bit 9798 / 19235 / 3735
    form can differ by one grey level on a few pixels, which matters to the contours only for near-black mask pixels.  cv2 is
    absent from the image: parity unpinned either way)"""
    a = np.asarray(mask)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8) * 255
    if a.ndim == 3:
        r, g, b = (a[..., k].astype(np.int32) for k in range(3))
        a = ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
    return np.ascontiguousarray(a.astype(np.uint8))


def external_contours_batch(masks_u8):
    """``cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)`` for a batch of masks, on the device
    (``advs_mask_contours``): masks_u8 uint8 [n, H, W] on the GPU (nonzero = foreground) -> per mask the list of
    ``(x, y, w, h, 2 * contourArea, first_pixel)`` -- ``cv2.boundingRect`` and twice ``cv2.contourArea`` of every external
    contour -- in OpenCV's list order (the reverse of the raster order in which the border following meets them)."""
    _lib.init_device()
    lib = _lib.load()
    m = masks_u8.contiguous()
    n, H, W = m.shape
    dev = m.device
    work = torch.empty(lib.advs_mask_contours_work_bytes(n, H, W), dtype=torch.uint8, device=dev)
    cap = MAX_CONTOURS
    while True:
        out = torch.empty((n, cap, 8), dtype=torch.int32, device=dev)
        cnt = torch.empty((n,), dtype=torch.int32, device=dev)
        check(lib.advs_mask_contours(m.data_ptr(), n, H, W, work.data_ptr(), out.data_ptr(), cnt.data_ptr(), cap,
                                     torch.cuda.current_stream(dev).cuda_stream), "mask_contours")
        cnt_h = cnt.cpu().tolist()
        most = max(cnt_h) if cnt_h else 0
        if most <= cap:
            break
        cap = most              # a speckled mask (cv2.findContours just returns them all): once more with room for every contour
    res = []
    for i, c in enumerate(cnt_h):
        rows = out[i, :c].cpu().tolist()
        rows.sort(key=lambda e: -e[0])                       # last found first
        res.append([(e[1], e[2], e[3] - e[1] + 1, e[4] - e[2] + 1, e[6], e[0]) for e in rows])
    return res


def external_contours(mask):
    """One PIL mask -> its external contours (see external_contours_batch)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        raise _lib.AdvsError("no GPU visible: the AdvShadow HIP path needs an MI355X (there is no CPU fallback)")
    return external_contours_batch(torch.from_numpy(cv_gray(mask)).to(dev)[None])[0]


def cv_resize_nearest(a, width, height):
    """``cv2.resize(a, (width, height), interpolation=cv2.INTER_NEAREST)``: source index floor(dst * src / dst_size), clipped
    (not Pillow's centre-aligned NEAREST)."""
    sh, sw = a.shape[:2]
    ys = np.minimum(np.floor(np.arange(height) * (sh / height)).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(width) * (sw / width)).astype(np.int64), sw - 1)
    return a[ys][:, xs]


# --------------------------------------------------------------------------- PIL composites
def _triangle_layer(size, cx, cy, t):
    layer = Image.new("RGBA", size, (255, 255, 255, 0))
    ImageDraw.Draw(layer).polygon([(cx, cy - t), (cx - t, cy + t), (cx + t, cy + t)], fill=(0, 0, 0, 128))
    return layer


def _composite(image_rgb, layer, paste_mask, dark_mask, mode, factor=0.43):
    """image RGB, layer RGBA and the two masks all at the image's size (dark_mask [H,W] or [H,W,3]; mode 0 uses paste_mask
    only)."""
    _lib.init_device()
    dev = torch.device("cuda", torch.cuda.current_device())
    img_np = np.asarray(image_rgb, dtype=np.uint8)
    lay_np, pm_np, dm_np = (np.ascontiguousarray(np.asarray(a, dtype=np.uint8)) for a in (layer, paste_mask, dark_mask))
    H, W = img_np.shape[:2]
    if lay_np.shape != (H, W, 4) or pm_np.shape != (H, W) or dm_np.shape[:2] != (H, W) or dm_np.ndim not in (2, 3):
        raise ValueError(f"composite: layer {lay_np.shape}, masks {pm_np.shape} / {dm_np.shape} do not match the image {img_np.shape}")
    dch = 1 if dm_np.ndim == 2 else dm_np.shape[2]
    if dch not in (1, 3):
        raise ValueError(f"composite: a mask with {dch} channels cannot index an RGB image (shadow_for_attack.py:62-65)")
    img, lay, pm, dm = (torch.from_numpy(a.copy()).to(dev) for a in (img_np, lay_np, pm_np, dm_np))
    out = torch.empty_like(img)
    s = torch.cuda.current_stream(dev).cuda_stream
    check(_lib.load().advs_composite_u8_masks(img.data_ptr(), lay.data_ptr(), pm.data_ptr(), dm.data_ptr(), dch, out.data_ptr(),
                                              H * W, mode, float(factor), s), "composite_u8")
    return Image.fromarray(out.cpu().numpy())


def add_shadow(image, mask):
    """add_shadow.py:35-60: PIL RGB image + mask -> PIL RGB image with the triangle shadow in the bounding box of the external
    contour of largest ``cv2.contourArea`` (first one in OpenCV's list order on ties, as Python's ``max`` picks)."""
    image = image.convert("RGB")
    if mask.size != image.size:
        raise ValueError("images do not match")              # Image.composite's own error (add_shadow.py:58)
    contours = external_contours(mask)
    if not contours:
        raise ValueError("mask has no foreground (add_shadow.py:44 would fail on max() of no contours)")
    x, y, w, h = max(contours, key=lambda c: c[4])[:4]
    cx, cy, t = x + w // 2, y + h // 2, min(w, h) // 2
    mask_l = np.asarray(mask.convert("L"))
    return _composite(image, _triangle_layer(image.size, cx, cy, t), mask_l, mask_l, 0)


def add_shadow_to_mask_area(image, mask, rng=random):
    """shadow_for_attack.py:22-93: triangle in the centre part of ``random.choice(contours)``, pasted through
    ``L(layer) & L(mask)``, then every mask-true element scaled by 0.43.  The mask may differ from the image in size: the
    triangle layer and the paste mask live on the mask's grid and are pasted at (0, 0), cropped or padded as Pillow's
    ``paste`` does; the darkening mask is ``cv2.resize(..., INTER_NEAREST)`` of ``np.array(mask)``, per channel when the
    mask has three (shadow_for_attack.py:50-71)."""
    contours = external_contours(mask)
    if not contours:
        return image
    x, y, w, h = rng.choice(contours)[:4]
    sx, sy, sw, sh = x + w // 4, y + h // 4, w // 2, h // 2
    cx, cy, t = sx + sw // 2, sy + sh // 2, min(sw, sh) // 3
    tri = np.asarray(_triangle_layer(mask.size, cx, cy, t))
    W, H = image.size
    layer = np.empty((H, W, 4), dtype=np.uint8)
    layer[...] = (255, 255, 255, 0)
    paste = np.zeros((H, W), dtype=np.uint8)
    hh, ww = min(H, tri.shape[0]), min(W, tri.shape[1])
    layer[:hh, :ww] = tri[:hh, :ww]
    paste[:hh, :ww] = np.asarray(mask.convert("L"))[:hh, :ww]
    m = np.asarray(mask)
    if m.dtype == np.bool_:
        m = m.astype(np.uint8)
    dark = cv_resize_nearest(m, W, H) if (m.shape[1], m.shape[0]) != (W, H) else m
    return _composite(image.convert("RGB"), layer, paste, dark, 1, 0.43)
