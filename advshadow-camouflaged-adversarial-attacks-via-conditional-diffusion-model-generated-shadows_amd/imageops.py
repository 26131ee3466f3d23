"""Image hand-off between the sampler and the victim: Pillow-exact resize, ToTensor, uint8 layout.

Replaces ``transforms.Resize((224, 224)) + ToTensor`` on PIL images (ASR_fast.py:90-97,
PSNR_SSIM_fast.py:10-13) for batches that never leave the GPU.  ``transforms.Resize`` on a PIL image
is ``Image.resize(size, BILINEAR)``: Pillow's two-pass antialiased resampler with 22-bit fixed-point
coefficients.  The coefficient tables are computed on the host exactly as ``precompute_coeffs`` /
``normalize_coeffs_8bpc`` (libImaging/Resample.c) do; the passes run in ``advs_resample_u8``.
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check

_PRECISION_BITS = 22


def bilinear_coeffs(in_size, out_size):
    """(bounds int32 [out,2], coefs int32 [out,ksize], ksize) of Pillow's bilinear filter."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)], dtype=np.float64)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << _PRECISION_BITS)), np.trunc(0.5 + kk * (1 << _PRECISION_BITS)))
    return bounds, fixed.astype(np.int32), ksize


_coef_cache = {}


def _coeffs_on(dev, in_size, out_size):
    key = (str(dev), in_size, out_size)
    hit = _coef_cache.get(key)
    if hit is None:
        b, k, ks = bilinear_coeffs(in_size, out_size)
        hit = (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), ks)
        _coef_cache[key] = hit
    return hit


def resize_u8(images_hwc, out_h, out_w, stream=None):
    """uint8 [n,H,W,ch] on the GPU -> uint8 [n,out_h,out_w,ch], Pillow BILINEAR semantics."""
    _lib.init_device()
    lib = _lib.load()
    n, H, W, ch = images_hwc.shape
    dev = images_hwc.device
    s = (stream or torch.cuda.current_stream(dev)).cuda_stream
    x = images_hwc.contiguous()
    if W != out_w:                                          # horizontal pass first (ImagingResampleInner)
        b, k, ks = _coeffs_on(dev, W, out_w)
        y = torch.empty((n, H, out_w, ch), dtype=torch.uint8, device=dev)
        check(lib.advs_resample_u8(x.data_ptr(), y.data_ptr(), b.data_ptr(), k.data_ptr(), ks, n, H, W, H, out_w, ch, 1, s),
              "resample_u8(h)")
        x, W = y, out_w
    if H != out_h:
        b, k, ks = _coeffs_on(dev, H, out_h)
        y = torch.empty((n, out_h, W, ch), dtype=torch.uint8, device=dev)
        check(lib.advs_resample_u8(x.data_ptr(), y.data_ptr(), b.data_ptr(), k.data_ptr(), ks, n, H, W, out_h, W, ch, 0, s),
              "resample_u8(v)")
        x = y
    return x


def u8_nchw_to_hwc(images, stream=None):
    """uint8 [n,ch,H,W] -> [n,H,W,ch] (the permute of save_images, utils/utils.py:59-60)."""
    _lib.init_device()
    n, ch, H, W = images.shape
    out = torch.empty((n, H, W, ch), dtype=torch.uint8, device=images.device)
    s = (stream or torch.cuda.current_stream(images.device)).cuda_stream
    check(_lib.load().advs_u8_nchw_to_hwc(images.contiguous().data_ptr(), out.data_ptr(), n, ch, H, W, s), "u8_nchw_to_hwc")
    return out


def to_tensor(images_hwc, mean=None, std=None, stream=None):
    """ToTensor (+ optional Normalize): uint8 [n,H,W,ch] -> f32 [n,ch,H,W] in [0,1]."""
    _lib.init_device()
    n, H, W, ch = images_hwc.shape
    dev = images_hwc.device
    out = torch.empty((n, ch, H, W), dtype=torch.float32, device=dev)
    m = torch.as_tensor(mean, dtype=torch.float32, device=dev) if mean is not None else None
    sd = torch.as_tensor(std, dtype=torch.float32, device=dev) if std is not None else None
    s = (stream or torch.cuda.current_stream(dev)).cuda_stream
    check(_lib.load().advs_u8hwc_to_f32nchw(images_hwc.contiguous().data_ptr(), out.data_ptr(), n, H, W, ch,
                                            0 if m is None else m.data_ptr(), 0 if sd is None else sd.data_ptr(), s),
          "u8hwc_to_f32nchw")
    return out


def jpeg_roundtrip(images_hwc, quality=75, stream=None):
    """What ``Image.save(path.jpg)`` + ``Image.open(path)`` does to the pixels (utils/utils.py:51-91 ->
    ASR_fast.py:90-92), without the file: uint8 [n,H,W,3] -> uint8 [n,H,W,3], bit-exact with Pillow's baseline
    4:2:0 round trip at ``quality`` (Pillow's default 75).  H and W must be multiples of 16."""
    _lib.init_device()
    lib = _lib.load()
    n, H, W, ch = images_hwc.shape
    if ch != 3:
        raise ValueError("jpeg_roundtrip expects RGB images [n,H,W,3]")
    dev = images_hwc.device
    x = images_hwc.contiguous()
    out = torch.empty_like(x)
    scratch = torch.empty(lib.advs_jpeg_scratch_bytes(n, H, W), dtype=torch.uint8, device=dev)
    s = (stream or torch.cuda.current_stream(dev)).cuda_stream
    check(lib.advs_jpeg_roundtrip_u8(x.data_ptr(), out.data_ptr(), scratch.data_ptr(), n, H, W, int(quality), s),
          "jpeg_roundtrip_u8")
    scratch.record_stream(torch.cuda.current_stream(dev) if stream is None else stream)
    return out


def preprocess_batch(images_u8_nchw, size=224, mean=None, std=None, jpeg_quality=None):
    """The batched, on-device form of ``preprocess_image`` (ASR_fast.py:90-97): uint8 [n,3,S,S]
    sampler output -> Resize((size,size)) -> ToTensor -> f32 [n,3,size,size] (no normalisation
    unless mean/std are given, as test.py:118-122 does).  ``jpeg_quality`` (e.g. 75) inserts the lossy
    ``.jpg`` hop the reference's scripts put between the sampler and the victim."""
    hwc = u8_nchw_to_hwc(images_u8_nchw)
    if jpeg_quality is not None:
        hwc = jpeg_roundtrip(hwc, jpeg_quality)
    return to_tensor(resize_u8(hwc, size, size), mean, std)
