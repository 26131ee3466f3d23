"""CPU checks of the oracle pieces that have something to be pinned against in this image:
Pillow (the reference's own library) for resize coefficients, scipy for the SSIM window."""
import numpy as np
import pytest
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


# ------------------------------------------------------------------------------ JPEG file hop
def _pil_jpeg_roundtrip(a, **kw):
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, "JPEG", **kw)
    buf.seek(0)
    return np.asarray(Image.open(buf).convert("RGB"))


def jpeg_cases():
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:64, 0:64]
    smooth = np.stack([(yy * 4) % 256, (xx * 3 + yy) % 256, (255 - xx * 2) % 256], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    return {"smooth": smooth, "noise": noise, "mix": (0.7 * smooth + 0.3 * noise).astype(np.uint8),
            "rect": rng.integers(0, 256, (48, 128, 3), dtype=np.uint8), "black": np.zeros((32, 32, 3), np.uint8),
            "white": np.full((32, 32, 3), 255, np.uint8),
            "checker": ((np.indices((64, 64)).sum(0) % 2) * 255).astype(np.uint8)[..., None].repeat(3, 2)}


def test_jpeg_roundtrip_oracle_is_pillow_exact():
    """oracle/jpeg.py against Pillow's own save/open (Pillow defaults = what utils/utils.py:51-91 writes and
    ASR_fast.py:90-92 reads): bit-exact, every quality."""
    from oracle import jpeg as oj
    for name, a in jpeg_cases().items():
        assert np.array_equal(oj.jpeg_roundtrip(a), _pil_jpeg_roundtrip(a)), name             # default quality
        for q in (30, 75, 90, 100):
            assert np.array_equal(oj.jpeg_roundtrip(a, q), _pil_jpeg_roundtrip(a, quality=q)), (name, q)
    with pytest.raises(ValueError):
        oj.jpeg_roundtrip(np.zeros((20, 32, 3), np.uint8))


def test_contour_oracle_against_scipy_route():
    """oracle/contours.py (Suzuki-Abe border following, Green's formula) against an independent route through scipy.ndimage
    (fill the holes, label with 8-connectivity, area from the 2x2 cells of pixel centres) on random and hand-made masks:
    same boxes, same doubled areas, same first pixels, same list order (last found first)."""
    import numpy as np
    from scipy import ndimage as ndi
    from oracle import contours as oc

    def scipy_route(m):
        X = ndi.binary_fill_holes(m != 0)
        lab, n = ndi.label(X, structure=np.ones((3, 3)))
        Xp, L = np.pad(X, ((0, 1), (0, 1))), np.pad(lab, ((0, 1), (0, 1)))
        k = Xp[:-1, :-1].astype(int) + Xp[:-1, 1:] + Xp[1:, :-1] + Xp[1:, 1:]
        cl = np.maximum(np.maximum(L[:-1, :-1], L[:-1, 1:]), np.maximum(L[1:, :-1], L[1:, 1:]))
        out = []
        for c in range(1, n + 1):
            ys, xs = np.nonzero(lab == c)
            a2 = 2 * int(((k == 4) & (cl == c)).sum()) + int(((k == 3) & (cl == c)).sum())
            out.append((xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, a2, int((ys * m.shape[1] + xs).min())))
        return [tuple(int(v) for v in e) for e in sorted(out, key=lambda e: -e[5])]

    m = np.zeros((5, 5), np.uint8)
    m[1, 1] = m[2, 1] = m[2, 2] = 1                                  # an L of three pixels: cv2.contourArea = 0.5
    assert oc.external_contours(m) == [(1, 1, 2, 2, 1, 6)]
    m = np.zeros((12, 14), np.uint8)
    m[1:10, 1:12] = 1; m[3:8, 3:10] = 0; m[5, 5:7] = 1; m[0, 13] = 1   # ring, a blob nested in its hole, a speck on the frame
    assert oc.external_contours(m) == [(1, 1, 11, 9, 160, 15), (13, 0, 1, 1, 0, 13)]
    rng = np.random.default_rng(0)
    for _ in range(200):
        h, w = rng.integers(3, 24, 2)
        m = (rng.random((h, w)) < rng.choice([0.3, 0.5, 0.65, 0.8])).astype(np.uint8) * 255
        assert oc.external_contours(m) == scipy_route(m)
    rgb = np.zeros((2, 2, 3), np.uint8)
    rgb[0, 0] = (1, 0, 1); rgb[0, 1] = (0, 1, 0); rgb[1, 0] = (0, 0, 5); rgb[1, 1] = (0, 0, 4)
    assert oc.cv_gray(rgb).tolist() == [[0, 1], [1, 0]]               # (R*4899 + G*9617 + B*1868 + 8192) >> 14
