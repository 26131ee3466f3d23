"""Oracle: the lossy JPEG round trip between ``save_images`` and ``preprocess_image``.

The reference writes every generated image as ``.jpg`` (tools/generate.py:124 default ``--image_format jpg``;
utils/utils.py:51-91 ``Image.save`` with Pillow's defaults: quality 75, 4:2:0) and the metric scripts read
the files back (ASR_fast.py:90-97, PSNR_SSIM_fast.py:21-31), so the victim never sees the sampler's pixels
but their baseline-JPEG reconstruction.  The entropy coding is lossless; what changes pixels is restated here
in integer arithmetic exactly as the JPEG library behind Pillow computes it (libjpeg / libjpeg-turbo,
published algorithm; file names below are the library's):

  RGB -> YCbCr (jccolor.c, 16-bit fixed point) -> 2x2 chroma box filter with alternating bias (jcsample.c
  h2v2_downsample) -> level shift, 8x8 forward DCT "islow" (jfdctint.c) -> quantisation, round half away
  from zero (jcdctmgr.c) -> dequantisation, inverse DCT "islow" + range limit (jidctint.c) -> triangle
  ("fancy") 2x2 chroma upsampling (jdsample.c h2v2_fancy_upsample) -> YCbCr -> RGB (jdcolor.c).

Test infrastructure only (see ``oracle/__init__.py``).  **Pinned by Pillow itself** (present in the image):
tests/test_oracle_misc.py compares this restatement with an actual ``Image.save`` / ``Image.open`` round trip,
bit for bit.  Sizes must be multiples of 16 (whole MCUs; the generator's 64/128/256 are).
"""
import numpy as np

# Annex K tables (quality 50), row-major natural order
_LUM = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64).reshape(8, 8)
_CHR = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, dtype=np.int64).reshape(8, 8)

CONST_BITS, PASS1_BITS = 13, 2
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865 = 2446, 3196, 4433, 6270
F_0_899976223, F_1_175875602, F_1_501321110, F_1_847759065 = 7373, 9633, 12299, 15137
F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 16069, 16819, 20995, 25172


def quant_tables(quality=75):
    """jpeg_quality_scaling + jpeg_add_quant_table(force_baseline) (jcparam.c)."""
    quality = min(max(int(quality), 1), 100)
    scale = 5000 // quality if quality < 50 else 200 - quality * 2
    return [np.clip((t * scale + 50) // 100, 1, 255) for t in (_LUM, _CHR)]


def _fix(x):
    return int(x * 65536 + 0.5)


def rgb_to_ycc(rgb):
    """jccolor.c rgb_ycc_convert: [H,W,3] uint8 -> three [H,W] int64 planes."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    half, off = 1 << 15, 128 << 16
    y = (_fix(0.29900) * r + _fix(0.58700) * g + _fix(0.11400) * b + half) >> 16
    cb = (-_fix(0.16874) * r - _fix(0.33126) * g + _fix(0.50000) * b + off + half - 1) >> 16
    cr = (_fix(0.50000) * r - _fix(0.41869) * g - _fix(0.08131) * b + off + half - 1) >> 16
    return y, cb, cr


def downsample_h2v2(p):
    """jcsample.c h2v2_downsample: 2x2 box, bias 1,2,1,2,... along the output row."""
    s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
    bias = np.where(np.arange(s.shape[1]) % 2 == 0, 1, 2)
    return (s + bias[None, :]) >> 2


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _fdct_1d(d, first):
    """One pass of jfdctint.c over the last axis of [..., 8]."""
    d0, d1, d2, d3, d4, d5, d6, d7 = (d[..., i] for i in range(8))
    t0, t7, t1, t6 = d0 + d7, d0 - d7, d1 + d6, d1 - d6
    t2, t5, t3, t4 = d2 + d5, d2 - d5, d3 + d4, d3 - d4
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    out = [None] * 8
    if first:
        out[0], out[4] = (t10 + t11) << PASS1_BITS, (t10 - t11) << PASS1_BITS
        n = CONST_BITS - PASS1_BITS
    else:
        out[0], out[4] = _descale(t10 + t11, PASS1_BITS), _descale(t10 - t11, PASS1_BITS)
        n = CONST_BITS + PASS1_BITS
    z1 = (t12 + t13) * F_0_541196100
    out[2] = _descale(z1 + t13 * F_0_765366865, n)
    out[6] = _descale(z1 + t12 * (-F_1_847759065), n)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * F_1_175875602
    t4, t5, t6, t7 = t4 * F_0_298631336, t5 * F_2_053119869, t6 * F_3_072711026, t7 * F_1_501321110
    z1, z2 = z1 * (-F_0_899976223), z2 * (-F_2_562915447)
    z3, z4 = z3 * (-F_1_961570560) + z5, z4 * (-F_0_390180644) + z5
    out[7], out[5] = _descale(t4 + z1 + z3, n), _descale(t5 + z2 + z4, n)
    out[3], out[1] = _descale(t6 + z2 + z3, n), _descale(t7 + z1 + z4, n)
    return np.stack(out, axis=-1)


def fdct_islow(blocks):
    """jfdctint.c jpeg_fdct_islow on [..., 8, 8] level-shifted samples (rows then columns); output x8."""
    w = _fdct_1d(blocks, True)
    return np.swapaxes(_fdct_1d(np.swapaxes(w, -1, -2), False), -1, -2)


def quantize(coef, q):
    """jcdctmgr.c: divisor = q * 8, round half away from zero."""
    d = q * 8
    a = (np.abs(coef) + (d >> 1)) // d
    return np.where(coef < 0, -a, a)


def _idct_1d(v, first):
    i0, i1, i2, i3, i4, i5, i6, i7 = (v[..., i] for i in range(8))
    z1 = (i2 + i6) * F_0_541196100
    t2 = z1 + i6 * (-F_1_847759065)
    t3 = z1 + i2 * F_0_765366865
    t0, t1 = (i0 + i4) << CONST_BITS, (i0 - i4) << CONST_BITS
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = i7, i5, i3, i1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F_1_175875602
    t0, t1, t2, t3 = t0 * F_0_298631336, t1 * F_2_053119869, t2 * F_3_072711026, t3 * F_1_501321110
    z1, z2 = z1 * (-F_0_899976223), z2 * (-F_2_562915447)
    z3, z4 = z3 * (-F_1_961570560) + z5, z4 * (-F_0_390180644) + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    n = CONST_BITS - PASS1_BITS if first else CONST_BITS + PASS1_BITS + 3
    out = [_descale(t10 + t3, n), _descale(t11 + t2, n), _descale(t12 + t1, n), _descale(t13 + t0, n),
           _descale(t13 - t0, n), _descale(t12 - t1, n), _descale(t11 - t2, n), _descale(t10 - t3, n)]
    return np.stack(out, axis=-1)


def idct_islow(coef):
    """jidctint.c jpeg_idct_islow on dequantised [..., 8, 8] (columns then rows) + range limit -> 0..255."""
    w = np.swapaxes(_idct_1d(np.swapaxes(coef, -1, -2), True), -1, -2)
    return np.clip(_idct_1d(w, False) + 128, 0, 255)


def _blocks(p):
    h, w = p.shape
    return p.reshape(h // 8, 8, w // 8, 8).swapaxes(1, 2)


def _unblocks(b):
    nby, nbx = b.shape[:2]
    return b.swapaxes(1, 2).reshape(nby * 8, nbx * 8)


def codec_plane(p, q):
    """samples [H,W] 0..255 -> reconstructed samples after FDCT/quantise/dequantise/IDCT."""
    c = quantize(fdct_islow(_blocks(p) - 128), q)
    return _unblocks(idct_islow(c * q))


def upsample_h2v2_fancy(p):
    """jdsample.c h2v2_fancy_upsample (edge rows/columns replicated)."""
    h, w = p.shape
    up = np.concatenate([p[:1], p[:-1]], 0)          # row above (first row: itself)
    dn = np.concatenate([p[1:], p[-1:]], 0)
    out = np.empty((2 * h, 2 * w), dtype=np.int64)
    for v, nb in ((0, up), (1, dn)):
        cs = p * 3 + nb                                # column sums of the vertical 3:1 blend
        last = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        even = (cs * 3 + last + 8) >> 4
        odd = (cs * 3 + nxt + 7) >> 4
        even[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2], out[v::2, 1::2] = even, odd
    return out


def ycc_to_rgb(y, cb, cr):
    """jdcolor.c ycc_rgb_convert."""
    half = 1 << 15
    cbx, crx = cb - 128, cr - 128
    r = y + ((_fix(1.40200) * crx + half) >> 16)
    g = y + ((-_fix(0.34414) * cbx + half - _fix(0.71414) * crx) >> 16)
    b = y + ((_fix(1.77200) * cbx + half) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def jpeg_roundtrip(rgb, quality=75):
    """[H,W,3] uint8 -> the pixels ``Image.open(save(rgb, 'JPEG', quality))`` yields (4:2:0, baseline)."""
    h, w, _ = rgb.shape
    if h % 16 or w % 16:
        raise ValueError("jpeg_roundtrip: height and width must be multiples of 16 (whole MCUs)")
    ql, qc = quant_tables(quality)
    y, cb, cr = rgb_to_ycc(rgb)
    y2 = codec_plane(y, ql)
    cb2 = upsample_h2v2_fancy(codec_plane(downsample_h2v2(cb), qc))
    cr2 = upsample_h2v2_fancy(codec_plane(downsample_h2v2(cr), qc))
    return ycc_to_rgb(y2, cb2, cr2)
