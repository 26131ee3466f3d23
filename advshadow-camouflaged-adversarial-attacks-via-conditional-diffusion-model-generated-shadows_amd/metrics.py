"""PSNR / SSIM (PSNR_SSIM_fast.py) and the attack-success-rate reduction (ASR_fast.py:101-126) with
the per-image work on the GPU."""
import os

import numpy as np
import torch

from . import _lib
from ._lib import check

_EXTS = ("png", "jpg", "jpeg", "bmp", "gif")


def ssim_psnr_batch(images1, images2, win_size=11, stream=None):
    """images [B,3,h,w] f32 on the GPU (h,w <= 64) -> f64 tensor [B,2] = (ssim, psnr) per pair."""
    _lib.init_device()
    B, Cc, H, W = images1.shape
    dev = images1.device
    out = torch.empty((B, 2), dtype=torch.float64, device=dev)
    s = (stream or torch.cuda.current_stream(dev)).cuda_stream
    check(_lib.load().advs_psnr_ssim(images1.contiguous().float().data_ptr(), images2.contiguous().float().data_ptr(),
                                     out.data_ptr(), B, Cc, H, W, win_size, s), "psnr_ssim")
    return out


def calculate_ssim_psnr(image1, image2, win_size=11):
    """Reference signature (PSNR_SSIM_fast.py:21): CHW float arrays in [0,1] -> (ssim, psnr)."""
    a = torch.as_tensor(np.ascontiguousarray(image1), dtype=torch.float32).cuda()[None]
    b = torch.as_tensor(np.ascontiguousarray(image2), dtype=torch.float32).cuda()[None]
    r = ssim_psnr_batch(a, b, win_size).cpu().numpy()[0]
    return float(r[0]), float(r[1])


def load_image(image_path):
    """PSNR_SSIM_fast.py:16-18: open, RGB, Resize((64,64)), ToTensor -> CHW float32 array."""
    from PIL import Image
    from .imageops import resize_u8, to_tensor
    img = np.asarray(Image.open(image_path).convert("RGB"), dtype=np.uint8)
    t = torch.from_numpy(img.copy()).cuda()[None]
    return to_tensor(resize_u8(t, 64, 64))[0].cpu().numpy()


def compare_folders(folder1, folder2, win_size=7):
    """PSNR_SSIM_fast.py:38-56 (pairs by os.listdir order, like the reference)."""
    def load(folder):
        return [load_image(os.path.join(folder, f)) for f in os.listdir(folder) if f.lower().endswith(_EXTS)]
    a, b = load(folder1), load(folder2)
    if len(a) != len(b):
        raise ValueError("Folders must contain the same number of images")
    r = ssim_psnr_batch(torch.from_numpy(np.stack(a)).cuda(), torch.from_numpy(np.stack(b)).cuda(), win_size).cpu().numpy()
    return float(np.mean(r[:, 0])), float(np.mean(r[:, 1]))


def argmax_rows(logits, stream=None):
    """torch.max(outputs, 1)[1] (ASR_fast.py:115) as int32 on the GPU."""
    _lib.init_device()
    rows, n = logits.shape
    out = torch.empty((rows,), dtype=torch.int32, device=logits.device)
    s = (stream or torch.cuda.current_stream(logits.device)).cuda_stream
    check(_lib.load().advs_argmax_rows(logits.contiguous().float().data_ptr(), out.data_ptr(), rows, n, s), "argmax_rows")
    return out


def attack_success(pred, true_idx):
    """successful_attacks / total (ASR_fast.py:118-123) from predicted and true class indices."""
    pred = torch.as_tensor(pred).to(torch.int64).cpu()
    true_idx = torch.as_tensor(true_idx).to(torch.int64).cpu()
    return float((pred != true_idx).sum().item()) / max(1, pred.numel())
