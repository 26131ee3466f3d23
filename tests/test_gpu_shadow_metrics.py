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


@pytest.mark.parametrize("size", [(64, 64), (97, 131), (256, 256)])
def test_add_shadow_bit_exact_vs_pillow(size):
    img, hard, soft = synth(*size, seed=1)
    for mask in (hard, soft):                             # soft mask exercises the fractional BLEND8 path
        boxes = shadow.mask_blobs(mask.convert("L"))
        bbox = max(boxes, key=lambda b: b[4])[:4]
        ref = osh.add_shadow_with_bbox(img, mask, bbox)
        got = shadow.add_shadow(img, mask)
        assert got.mode == "RGB" and np.array_equal(np.asarray(got), np.asarray(ref.convert("RGB")))


def test_add_shadow_to_mask_area_bit_exact_vs_pillow():
    import random
    img, hard, _ = synth(120, 90, seed=2)
    boxes = shadow.mask_blobs(hard)
    for seed in (0, 1, 2):
        bbox = random.Random(seed).choice(boxes)[:4]
        ref = osh.add_shadow_to_mask_area_with_bbox(img, hard, bbox)
        got = shadow.add_shadow_to_mask_area(img, hard, rng=random.Random(seed))
        assert np.array_equal(np.asarray(got), np.asarray(ref))


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


def test_apply_shadow_rejects_classifier():
    with pytest.raises(NotImplementedError):
        shadow.apply_shadow(torch.zeros(3, 8, 8), (1, 1), 2.0, torch.ones(1, 8, 8), classifier=object())


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
