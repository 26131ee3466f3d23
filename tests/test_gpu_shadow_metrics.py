"""Composite / resize / metric kernels on the MI355X against Pillow (the reference's own integer
arithmetic), scipy-based SSIM restatement and the float closed form of apply_shadow."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from advshadow_amd import imageops, metrics, shadow  # noqa: E402
from oracle import metrics as om  # noqa: E402
from oracle import shadow as osh  # noqa: E402


def synth(h, w, seed):
    rng = np.random.default_rng(seed)
    img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    yy, xx = np.mgrid[0:h, 0:w]
    m = (((xx - w * 0.45) / (w * 0.3)) ** 2 + ((yy - h * 0.5) / (h * 0.35)) ** 2 <= 1).astype(np.uint8) * 255
    m[2:6, 3:9] = 255                                     # a second, small blob
    soft = np.clip(m.astype(np.int32) - rng.integers(0, 60, (h, w)), 0, 255).astype(np.uint8) * (m > 0)
    return img, Image.fromarray(m), Image.fromarray(soft.astype(np.uint8))


def contour_masks():
    """Masks that separate the contour rules: holes (contourArea counts them, a pixel count does not), a blob nested in a
    hole (not an external contour), one-pixel bridges, blobs on the frame, specks, equal-area ties."""
    ms = {}
    yy, xx = np.mgrid[0:96, 0:128]
    ring = (((xx - 40) ** 2 + (yy - 48) ** 2 <= 30 ** 2) & ((xx - 40) ** 2 + (yy - 48) ** 2 > 22 ** 2))
    disc = (xx - 98) ** 2 + (yy - 30) ** 2 <= 24 ** 2                       # more pixels than the ring, smaller contourArea
    nested = (xx - 40) ** 2 + (yy - 48) ** 2 <= 6 ** 2
    ms["ring_vs_disc"] = ring | disc | nested
    m = np.zeros((64, 80), bool)
    m[5:20, 5:20] = True; m[19:21, 19:21] = True; m[20:40, 20:45] = True     # two squares joined by a diagonal bridge
    m[0:6, 70:80] = True; m[58:64, 0:9] = True                               # on the frame
    m[50, 50] = True; m[30, 60:63] = True                                    # specks
    ms["bridges"] = m
    m = np.zeros((40, 40), bool)
    m[3:13, 3:13] = True; m[20:30, 22:32] = True                             # equal areas: the tie goes to the LAST one found
    ms["tie"] = m
    rng = np.random.default_rng(5)
    ms["noise"] = rng.random((72, 90)) < 0.55
    return {k: (v.astype(np.uint8) * 255) for k, v in ms.items()}


@pytest.mark.parametrize("name", ["ring_vs_disc", "bridges", "tie", "noise"])
def test_external_contours_vs_suzuki_oracle_and_scipy(name):
    """advs_mask_contours (flood + labelling + 2x2-cell area, no border following) against the Suzuki-Abe restatement
    (oracle/contours.py: border following, Green's formula) -- boxes, twice the contourArea, first pixels and list order --
    and the boxes against scipy.ndimage (fill holes, label with 8-connectivity), an implementation neither shares."""
    from scipy import ndimage as ndi
    from oracle import contours as oc
    m = contour_masks()[name]
    got = shadow.external_contours(Image.fromarray(m))
    assert got == [tuple(int(v) for v in c) for c in oc.external_contours(m)]
    lab, n = ndi.label(ndi.binary_fill_holes(m != 0), structure=np.ones((3, 3)))
    boxes = sorted((sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start) for sl in ndi.find_objects(lab))
    assert sorted(c[:4] for c in got) == boxes and len(got) == n
    if name == "ring_vs_disc":                          # the ring wins by contourArea although the disc has more pixels
        best = max(got, key=lambda c: c[4])
        assert best[:4] == (10, 18, 61, 61) and (m[18:79, 10:71] != 0).sum() < (m[6:55, 74:123] != 0).sum()
    batch = shadow.external_contours_batch(torch.from_numpy(np.stack([m, np.zeros_like(m), m[::-1].copy()])).cuda())
    assert batch[0] == got and batch[1] == [] and len(batch[2]) == len(got)


def test_external_contours_more_than_the_first_buffer_holds():
    """A speckled mask (JPEG noise around a threshold) has more external contours than the first output buffer (4096): the host
    runs the kernel again with room for all of them, as cv2.findContours simply returns them all.  Boxes and the
    count against scipy.ndimage; every speckle is one pixel: area 0, 1 x 1 box, raster-last first."""
    from scipy import ndimage as ndi
    m = np.zeros((200, 200), np.uint8)
    m[::2, ::2] = 255                                     # 10 000 isolated pixels (8-connectivity keeps them apart)
    m[101:140, 31:90] = 255                               # and one blob that swallows a few of them
    got = shadow.external_contours(Image.fromarray(m))
    lab, n = ndi.label(m != 0, structure=np.ones((3, 3)))
    assert n > shadow.MAX_CONTOURS and len(got) == n
    boxes = sorted((sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start) for sl in ndi.find_objects(lab))
    assert sorted(c[:4] for c in got) == boxes
    assert got[0][:5] == (198, 198, 1, 1, 0) and [c[5] for c in got] == sorted((c[5] for c in got), reverse=True)


@pytest.mark.parametrize("size", [(64, 64), (97, 131), (256, 256)])
def test_add_shadow_bit_exact_vs_pillow(size):
    img, hard, soft = synth(*size, seed=1)
    for mask in (hard, soft):                             # soft mask exercises the fractional BLEND8 path
        ref = osh.add_shadow(img, mask)
        got = shadow.add_shadow(img, mask)
        assert got.mode == "RGB" and np.array_equal(np.asarray(got), np.asarray(ref.convert("RGB")))
    ringm = Image.fromarray(contour_masks()["ring_vs_disc"])
    img2 = Image.fromarray(np.random.default_rng(3).integers(0, 256, (96, 128, 3), dtype=np.uint8))
    assert np.array_equal(np.asarray(shadow.add_shadow(img2, ringm)), np.asarray(osh.add_shadow(img2, ringm).convert("RGB")))
    with pytest.raises(ValueError):
        shadow.add_shadow(img2, hard)                     # Image.composite refuses mismatched sizes


def test_add_shadow_to_mask_area_bit_exact_vs_pillow():
    import random
    img, hard, _ = synth(120, 90, seed=2)
    for seed in (0, 1, 2):
        ref = osh.add_shadow_to_mask_area(img, hard, random.Random(seed))
        got = shadow.add_shadow_to_mask_area(img, hard, rng=random.Random(seed))
        assert np.array_equal(np.asarray(got), np.asarray(ref))
    multi = Image.fromarray(contour_masks()["bridges"])
    img2 = Image.fromarray(np.random.default_rng(4).integers(0, 256, (64, 80, 3), dtype=np.uint8))
    for seed in range(6):                                  # random.choice walks OpenCV's list order
        assert np.array_equal(np.asarray(shadow.add_shadow_to_mask_area(img2, multi, rng=random.Random(seed))),
                              np.asarray(osh.add_shadow_to_mask_area(img2, multi, random.Random(seed))))


@pytest.mark.parametrize("msize", [(60, 44), (150, 100), (90, 133)])
@pytest.mark.parametrize("mode", ["L", "RGB"])
def test_add_shadow_to_mask_area_mask_size_differs(msize, mode):
    """shadow_for_attack.py:50-93 with mask.size != image.size (smaller, larger, other aspect): the layer is pasted at (0, 0)
    on the mask's grid, the darkening goes through cv2's nearest resize -- per channel for an RGB mask whose channels differ."""
    import random
    rng = np.random.default_rng(8)
    img = Image.fromarray(rng.integers(0, 256, (100, 120, 3), dtype=np.uint8))             # 120 x 100
    w, h = msize
    yy, xx = np.mgrid[0:h, 0:w]
    m = ((((xx - w * 0.5) / (w * 0.32)) ** 2 + ((yy - h * 0.45) / (h * 0.3)) ** 2) <= 1).astype(np.uint8) * 255
    m[1:4, 2:7] = 200
    if mode == "RGB":
        m = np.stack([m, m, m], 2)
        m[5:9, 20:30, 1] = 0; m[h // 2, w // 2, 2] = 0                                      # channels that disagree
        m[h - 4:h - 1, 3:9, 0] = 90
    mask = Image.fromarray(m, mode)
    for seed in (0, 3):
        ref = osh.add_shadow_to_mask_area(img, mask, random.Random(seed))
        got = shadow.add_shadow_to_mask_area(img, mask, rng=random.Random(seed))
        assert got.size == img.size and np.array_equal(np.asarray(got), np.asarray(ref))


@pytest.mark.parametrize("k", [5, 1, 3])
def test_apply_shadow_closed_form(k):
    g = torch.Generator().manual_seed(7)
    img = torch.rand(3, 64, 80, generator=g)
    yy, xx = torch.meshgrid(torch.arange(64), torch.arange(80), indexing="ij")
    fm = (((xx - 40.0) ** 2 + (yy - 30.0) ** 2) <= 28 ** 2).float()[None]
    ref = osh.apply_shadow(img, (37.5, 2.0), 21.0, fm, 0.43, k)
    got = shadow.apply_shadow(img, (37.5, 2.0), 21.0, fm, None, None, "cuda", 0.43, 0.01, k).cpu()
    assert (got - ref).abs().max().item() < 1e-6
    fm3 = fm.expand(3, -1, -1).clone()
    got3 = shadow.apply_shadow(img, torch.tensor([37.5, 2.0]), torch.tensor(21.0), fm3, blur_kernel_size=k).cpu()
    assert torch.equal(got3, got)


def test_apply_shadow_rejects_a_classifier_without_a_hip_backward():
    """No autograd fallback: only a victim with a HIP backward plan can drive the gradient attack."""
    from advshadow_amd import AdvsError
    with pytest.raises(AdvsError, match="no HIP backward plan"):
        shadow.apply_shadow(torch.zeros(3, 16, 16), (1, 1), 2.0, torch.ones(1, 16, 16), classifier=object(),
                            target_label=torch.tensor([0]))


@pytest.mark.parametrize("shape", [(256, 256, 224), (64, 64, 224), (375, 500, 224), (256, 256, 64), (33, 77, 64)])
def test_resize_bit_exact_vs_pillow(shape):
    h, w, out = shape
    rng = np.random.default_rng(h * 1000 + w)
    arr = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    got = imageops.resize_u8(torch.from_numpy(arr).cuda(), out, out).cpu().numpy()
    for i in range(2):
        ref = np.asarray(Image.fromarray(arr[i]).resize((out, out), Image.BILINEAR))
        assert np.array_equal(got[i], ref)


def test_preprocess_batch_matches_pil_pipeline():
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (3, 3, 64, 64), dtype=np.uint8)
    got = imageops.preprocess_batch(torch.from_numpy(u8).cuda(), 224).cpu()
    for i in range(3):
        pil = Image.fromarray(u8[i].transpose(1, 2, 0)).resize((224, 224), Image.BILINEAR)
        ref = torch.from_numpy(np.asarray(pil).transpose(2, 0, 1).astype(np.float32) / 255.0)
        assert torch.equal(got[i], ref)


@pytest.mark.parametrize("win", [11, 7])
def test_psnr_ssim_vs_restatement(win):
    g = torch.Generator().manual_seed(11)
    a = torch.rand(4, 3, 64, 64, generator=g)
    b = (a + 0.05 * torch.randn(4, 3, 64, 64, generator=g)).clamp(0, 1)
    b[3] = a[3] * 0.57                                         # a darkened copy (shadow-like)
    got = metrics.ssim_psnr_batch(a.cuda(), b.cuda(), win).cpu().numpy()
    for i in range(4):
        s, p = om.calculate_ssim_psnr(a[i].numpy(), b[i].numpy(), win)
        assert abs(got[i, 0] - s) < 2e-6 and abs(got[i, 1] - p) < 1e-5, (got[i], s, p)
    s, p = metrics.calculate_ssim_psnr(a[0].numpy(), b[0].numpy(), win)
    assert abs(s - got[0, 0]) < 1e-12 and abs(p - got[0, 1]) < 1e-12


def test_argmax_and_asr():
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(9, 37, generator=g)
    logits[4, 7] = logits[4, 20] = 9.0                          # tie -> first index, like torch.max
    pred = metrics.argmax_rows(logits.cuda()).cpu()
    assert torch.equal(pred.long(), torch.max(logits, 1)[1])
    true = pred.clone().long()
    true[[1, 5, 6]] += 1
    assert metrics.attack_success(pred, true) == 3 / 9
    names = [f"Abyssinian_{i}.jpg" for i in range(3)] + ["notes.txt"]
    assert om.compute_asr(names, ["Abyssinian", "Bengal", "Abyssinian", "x"]) == 1 / 3


# ------------------------------------------------------------------------------ JPEG file hop on the device
def _pil_jpeg(a, **kw):
    import io
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, "JPEG", **kw)
    buf.seek(0)
    return np.asarray(Image.open(buf).convert("RGB"))


@pytest.mark.parametrize("quality", [None, 30, 75, 95, 100])
def test_jpeg_roundtrip_bit_exact_vs_pillow_and_oracle(quality):
    """advs_jpeg_roundtrip_u8 == Pillow's save(.jpg)/open == oracle/jpeg.py, bit for bit: noise, smooth ramps,
    saturated colours, flat images, a rectangular batch."""
    from oracle import jpeg as oj
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:64, 0:64]
    smooth = np.stack([(yy * 4) % 256, (xx * 3 + yy) % 256, (255 - xx * 2) % 256], -1).astype(np.uint8)
    batch = np.stack([smooth, rng.integers(0, 256, (64, 64, 3), dtype=np.uint8),
                      (0.6 * smooth + 0.4 * rng.integers(0, 256, (64, 64, 3))).astype(np.uint8),
                      np.zeros((64, 64, 3), np.uint8), np.full((64, 64, 3), 255, np.uint8),
                      np.tile(np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 0, 255]], np.uint8).repeat(16, 0)[None], (64, 1, 1))])
    kw = {} if quality is None else {"quality": quality}
    got = imageops.jpeg_roundtrip(torch.from_numpy(batch).cuda(), **kw).cpu().numpy()
    for i in range(batch.shape[0]):
        assert np.array_equal(got[i], _pil_jpeg(batch[i], **kw)), i
        assert np.array_equal(got[i], oj.jpeg_roundtrip(batch[i], quality or 75)), i
    rect = rng.integers(0, 256, (2, 48, 160, 3), dtype=np.uint8)
    got = imageops.jpeg_roundtrip(torch.from_numpy(rect).cuda(), **kw).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], _pil_jpeg(rect[i], **kw)), i


def test_jpeg_roundtrip_full_size_and_errors():
    """256x256 x 8 images against Pillow; sizes that are not whole MCUs are rejected."""
    from advshadow_amd import _lib
    rng = np.random.default_rng(12)
    base = rng.integers(0, 256, (8, 32, 32, 3), dtype=np.uint8)
    big = np.stack([np.asarray(Image.fromarray(b).resize((256, 256), Image.BICUBIC)) for b in base])     # photo-like content
    got = imageops.jpeg_roundtrip(torch.from_numpy(big).cuda()).cpu().numpy()
    for i in range(8):
        assert np.array_equal(got[i], _pil_jpeg(big[i])), i
    with pytest.raises(_lib.AdvsError):
        imageops.jpeg_roundtrip(torch.zeros((1, 20, 32, 3), dtype=torch.uint8, device="cuda"))


def test_preprocess_batch_with_jpeg_matches_the_file_pipeline(tmp_path):
    """sampler uint8 -> save as .jpg (utils/utils.py) -> ASR_fast.preprocess_image, against the on-device path."""
    rng = np.random.default_rng(13)
    base = rng.integers(0, 256, (3, 16, 16, 3), dtype=np.uint8)
    imgs = np.stack([np.asarray(Image.fromarray(b).resize((64, 64), Image.BICUBIC)) for b in base])
    ref = []
    for i in range(3):
        p = tmp_path / f"x_{i}.jpg"
        Image.fromarray(imgs[i]).save(p)
        im = Image.open(p).convert("RGB").resize((224, 224), Image.BILINEAR)
        ref.append(np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / 255.0)
    got = imageops.preprocess_batch(torch.from_numpy(imgs.transpose(0, 3, 1, 2).copy()).cuda(), 224, jpeg_quality=75).cpu().numpy()
    assert np.array_equal(got, np.stack(ref))
