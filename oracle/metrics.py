"""Oracle: PSNR / SSIM (PSNR_SSIM_fast.py:21-26) and the ASR count (ASR_fast.py:101-126) on CPU.

Test infrastructure only.  skimage is absent from the image, so ``structural_similarity`` and
``peak_signal_noise_ratio`` are restated from their published algorithm (Wang et al. 2004 as
implemented in scikit-image 0.19-0.25: gaussian_weights=True -> sigma 1.5, truncate 3.5, filter via
scipy.ndimage.gaussian_filter mode='reflect', sample covariance, crop (win-1)//2, float64 mean).
scipy IS present and supplies the filter, so the only unpinned part is the thin formula layer:
PARITY UNPINNED against skimage itself (no skimage outputs exist in the reference tree).
"""
import numpy as np
from scipy.ndimage import gaussian_filter


def structural_similarity(im1, im2, win_size=11, data_range=1.0, K1=0.01, K2=0.03, sigma=1.5):
    """HWC float32 images, channel_axis=2, gaussian_weights=True."""
    vals = []
    for ch in range(im1.shape[2]):
        a, b = im1[..., ch].astype(np.float32), im2[..., ch].astype(np.float32)
        f = lambda v: gaussian_filter(v, sigma=sigma, truncate=3.5, mode="reflect")
        NP = win_size ** 2
        cov_norm = NP / (NP - 1)
        ux, uy = f(a), f(b)
        uxx, uyy, uxy = f(a * a), f(b * b), f(a * b)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        R = data_range
        C1, C2 = (K1 * R) ** 2, (K2 * R) ** 2
        A1, A2, B1, B2 = 2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2
        S = (A1 * A2) / (B1 * B2)
        pad = (win_size - 1) // 2
        vals.append(S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean(dtype=np.float64))
    return float(np.mean(vals))


def peak_signal_noise_ratio(im1, im2, data_range):
    err = np.mean((im1.astype(np.float32) - im2.astype(np.float32)) ** 2, dtype=np.float64)
    return float(10 * np.log10((data_range ** 2) / err))


def calculate_ssim_psnr(image1, image2, win_size=11):
    """PSNR_SSIM_fast.py:21-26: CHW inputs; data_range = max-min of image1."""
    a, b = np.transpose(image1, (1, 2, 0)), np.transpose(image2, (1, 2, 0))
    R = a.max() - a.min()
    return structural_similarity(a, b, win_size=win_size, data_range=R), peak_signal_noise_ratio(a, b, R)


def compute_asr(filenames, predicted_labels):
    """ASR_fast.py:107-123 on (filename, predicted label) pairs: true = filename.rsplit('_', 1)[0]."""
    total = succ = 0
    for fn, pred in zip(filenames, predicted_labels):
        if fn.lower().endswith(("png", "jpg", "jpeg", "bmp", "gif")):
            total += 1
            succ += pred != fn.rsplit("_", 1)[0]
    return succ / total
