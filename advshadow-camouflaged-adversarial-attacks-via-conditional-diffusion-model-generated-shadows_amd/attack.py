"""The attack-evaluation loop as one on-device pipeline.

The reference joins three scripts through the file system: generate shadow images (DDIM), score
them with a victim classifier (ASR_fast.py), and measure PSNR/SSIM between clean and shadowed
images at 64x64 (PSNR_SSIM_fast.py).  Here a shard of image ids goes through the same stages without
leaving HBM; per-image results are all-gathered (parallel.py) and reduced on every rank.
"""
import torch

from . import _lib
from ._lib import check
from .asr import evaluate_batch
from .imageops import resize_u8, to_tensor, u8_nchw_to_hwc
from .metrics import ssim_psnr_batch
from .parallel import gather_results, reduce_metrics, shard_bounds
from .shadow import apply_shadow_batch


def unit_to_uint8(x):
    """[0,1] float NCHW -> uint8 NCHW (ToPILImage semantics)."""
    _lib.init_device()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(_lib.load().advs_unit_to_uint8(x.contiguous().float().data_ptr(), out.data_ptr(), x.numel(),
                                         torch.cuda.current_stream(x.device).cuda_stream), "unit_to_uint8")
    return out


def to_64(images_f32_nchw):
    """What PSNR_SSIM_fast.load_image does to a saved image: uint8 -> Resize((64,64)) -> ToTensor."""
    return to_tensor(resize_u8(u8_nchw_to_hwc(unit_to_uint8(images_f32_nchw)), 64, 64))


def attack_shard(sample_fn, victim, clean, feature_masks, centers, radii, shadow_intensity=0.43, blur_kernel_size=5,
                 win_size=7, jpeg_quality=None, gradient_attack=None):
    """One shard: returns (generated uint8 [n,3,S,S], pred int32 [n], psnr f32 [n], ssim f32 [n]).

    ``gradient_attack`` = dict(labels=[n] int64, epsilon=0.01, alpha=0.005, iterations=20) makes the composite the
    classifier branch of ``apply_shadow`` (tools/train_shadow.py:242-266: shadow + iterative-gradient perturbation inside
    the combined mask, on the victim's HIP backward plan) instead of the closed-form shadow; the victim must be a
    ``ResNet50`` / ``VGG``.  Images are independent, so the shard boundaries of ``run_attack`` apply unchanged."""
    generated = sample_fn()                                           # DDIM sampler output, uint8 on the GPU
    # [.jpg round trip, as generate()'s default image_format implies ->] resize 224 -> victim -> argmax
    pred = evaluate_batch(generated, victim, jpeg_quality=jpeg_quality)
    if gradient_attack is None:
        shadowed = apply_shadow_batch(clean, centers, radii, feature_masks, shadow_intensity, blur_kernel_size)
    else:
        from .adversarial import apply_shadow_adversarial_batch
        ga = dict(gradient_attack)
        shadowed = apply_shadow_adversarial_batch(victim, clean, centers, radii, feature_masks, ga.pop("labels"),
                                                  shadow_intensity, ga.pop("epsilon", 0.01), blur_kernel_size,
                                                  ga.pop("alpha", 0.005), ga.pop("iterations", 20))
        if ga:
            raise TypeError(f"attack_shard: unknown gradient_attack keys {sorted(ga)}")
    sp = ssim_psnr_batch(to_64(clean), to_64(shadowed), win_size)     # [n,2] f64 (ssim, psnr)
    return generated, pred, sp[:, 1].float(), sp[:, 0].float()


def run_attack(total, make_shard, labels, rank=0, world=1):
    """Shard ``total`` image ids over ``world`` ranks, run ``make_shard(lo, hi)`` (which returns
    attack_shard's tuple for ids [lo, hi)), all-gather the per-image results and reduce them."""
    lo, hi = shard_bounds(total, rank, world)
    _, pred, psnr, ssim = make_shard(lo, hi)
    pred, psnr, ssim = gather_results(pred, psnr, ssim, total)
    return reduce_metrics(pred, labels, psnr, ssim), pred
