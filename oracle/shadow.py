"""Oracle: the shadow composites (apply_shadow closed form; PIL triangle composites) on CPU.

Test infrastructure only.  The integer composites call Pillow itself -- the library the reference
calls (add_shadow.py:57-58, shadow_for_attack.py:76-93) and which IS present -- so that part of the
oracle is the reference's own arithmetic.  The float closed form restates tools/train_shadow.py
:242-256,262-266 with ``cv2.GaussianBlur(mask, (k, k), 0)`` replaced by an explicit separable
filter with OpenCV's fixed small-kernel taps and BORDER_REFLECT_101: cv2 is absent from the image,
so that piece is PARITY UNPINNED (no reference output exists to check the taps against).
"""
import numpy as np
import torch
from PIL import Image, ImageDraw

TAPS = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
        7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def gaussian_blur_reflect101(mask, ksize):
    """Separable filter, rows then columns, float32, border gfedcb|abcdefgh|gfedcba."""
    m = np.asarray(mask, dtype=np.float32)
    if ksize <= 1:
        return m
    taps = np.asarray(TAPS[ksize], dtype=np.float32)
    r = ksize // 2
    p = np.pad(m, ((0, 0), (r, r)), mode="reflect")
    rows = sum(taps[k] * p[:, k:k + m.shape[1]] for k in range(ksize))
    p = np.pad(rows, ((r, r), (0, 0)), mode="reflect")
    return sum(taps[k] * p[k:k + m.shape[0], :] for k in range(ksize)).astype(np.float32)


def apply_shadow(image, shadow_center, shadow_radius, feature_mask, shadow_intensity=0.43, blur_kernel_size=5):
    """tools/train_shadow.py:242-256,262-266 with adversarial_image == shadowed_image."""
    C, H, W = image.shape
    Y, X = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    dist = torch.sqrt((X - shadow_center[0]) ** 2 + (Y - shadow_center[1]) ** 2)
    m = (dist <= shadow_radius).float()
    mb = torch.from_numpy(gaussian_blur_reflect101(m.numpy(), blur_kernel_size))
    cm = mb * feature_mask
    shadowed = image * (1 - cm) + cm * (image * (1 - shadow_intensity))
    return torch.clamp(image * (1 - cm) + shadowed * cm, 0, 1)


def triangle_layer(size, cx, cy, t):
    layer = Image.new("RGBA", size, (255, 255, 255, 0))
    ImageDraw.Draw(layer).polygon([(cx, cy - t), (cx - t, cy + t), (cx + t, cy + t)], fill=(0, 0, 0, 128), outline=None)
    return layer


def add_shadow_with_bbox(original_image, mask_image, bbox):
    """add_shadow.py:46-58 given the bounding rect (x, y, w, h) of the largest contour."""
    x, y, w, h = bbox
    cx, cy = x + w // 2, y + h // 2
    layer = triangle_layer(original_image.size, cx, cy, min(w, h) // 2)
    combined = Image.alpha_composite(original_image.convert("RGBA"), layer)
    return Image.composite(combined, original_image, mask_image)


def adjust_shadow_brightness(image, mask, factor=0.43):
    """shadow_for_attack.py:50-73 (mask already at the image size)."""
    image_np = np.array(image)
    mask_bool = np.array(mask).astype(bool)
    image_float = image_np.astype(np.float32)
    image_float[mask_bool] *= factor
    np.clip(image_float, 0, 255, out=image_float)
    return Image.fromarray(image_float.astype(np.uint8))


def add_shadow_to_mask_area_with_bbox(image, mask, bbox):
    """shadow_for_attack.py:22-93 given the bounding rect of the chosen contour."""
    x, y, w, h = bbox
    sx, sy, sw, sh = x + w // 4, y + h // 4, w // 2, h // 2
    cx, cy = sx + sw // 2, sy + sh // 2
    tri = triangle_layer(mask.size, cx, cy, min(sw, sh) // 3)
    shadow_mask = np.array(tri.convert("L"))
    inter = Image.fromarray(np.bitwise_and(shadow_mask, np.array(mask.convert("L"))))
    layer = Image.new("RGBA", image.size, (255, 255, 255, 0))
    layer.paste(tri, mask=inter)
    combined = Image.alpha_composite(image.convert("RGBA"), layer).convert("RGB")
    return adjust_shadow_brightness(combined, mask.convert("L"))


# --------------------------------------------------------------------------- the two scripts end to end (contours included)
def cv_resize_nearest(a, width, height):
    """``cv2.resize(a, (width, height), interpolation=cv2.INTER_NEAREST)`` restated: src = min(floor(dst * scale), size - 1)."""
    sh, sw = a.shape[:2]
    ys = np.minimum(np.floor(np.arange(height) * (sh / height)).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(width) * (sw / width)).astype(np.int64), sw - 1)
    return a[ys][:, xs]


def add_shadow(original_image, mask_image):
    """add_shadow.py:35-58: largest external contour by ``contourArea`` (first maximum in OpenCV's list order)."""
    from . import contours as oc
    cs = oc.external_contours(oc.cv_gray(mask_image))
    best = max(cs, key=lambda c: c[4])
    return add_shadow_with_bbox(original_image, mask_image.convert("L"), best[:4])


def add_shadow_to_mask_area(image, mask, rng):
    """shadow_for_attack.py:22-93 with Pillow doing the integer work: ``rng.choice(contours)``; the triangle layer and
    the paste mask live at mask.size and are pasted at (0, 0) (Pillow crops / leaves the rest); the darkening mask is
    ``np.array(mask)`` after cv2's nearest resize, indexing per element (so per channel for an RGB mask)."""
    from . import contours as oc
    cs = oc.external_contours(oc.cv_gray(mask))
    if not cs:
        return image
    x, y, w, h = rng.choice(cs)[:4]
    sx, sy, sw, sh = x + w // 4, y + h // 4, w // 2, h // 2
    cx, cy = sx + sw // 2, sy + sh // 2
    tri = triangle_layer(mask.size, cx, cy, min(sw, sh) // 3)
    inter = Image.fromarray(np.bitwise_and(np.array(tri.convert("L")), np.array(mask.convert("L"))))
    layer = Image.new("RGBA", image.size, (255, 255, 255, 0))
    layer.paste(tri, mask=inter)
    combined = Image.alpha_composite(image.convert("RGBA"), layer).convert("RGB")
    image_np = np.array(combined)
    mask_np = np.array(mask)
    if mask_np.dtype == np.bool_:
        mask_np = mask_np.astype(np.uint8)
    mask_np = cv_resize_nearest(mask_np, image_np.shape[1], image_np.shape[0])
    fl = image_np.astype(np.float32)
    fl[mask_np.astype(bool)] *= 0.43
    np.clip(fl, 0, 255, out=fl)
    return Image.fromarray(fl.astype(np.uint8))
