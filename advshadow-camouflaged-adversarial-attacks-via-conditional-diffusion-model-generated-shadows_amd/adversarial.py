"""Gradient-based perturbation and shadow-position search of the training-data synthesis loop, on the GPU.

Mirrors tools/train_shadow.py:76-266 (``optimize_shadow_position``, ``apply_adversarial_perturbation``, the
classifier branch of ``apply_shadow``) and the integrated-gradient variant ddim2/test.py:647-681.  Where the
reference calls ``loss.backward()`` and reads ``image.grad``, this runs the victim's static forward + backward plan
(``ResNet50.grad_engine``: every conv's data gradient is an ``advs_conv2d`` on transposed weights, the pooling /
ReLU / stem backward are the kernels of ``csrc/victim_grad.hip``), replayed as one hipGraph per iteration; the sign
update and clamps are ``advs_iga_step`` / ``advs_perturb_clamp01``.  There is no autograd and no CPU fallback.

The classifier must be (or wrap, as ``classifier.model``) a victim with a backward plan: ``advshadow_amd.victims.ResNet50``
(the architecture of ddim2/test.py:22-36), ``victims.VGG``, ``victims.ViTVictim`` / ``victims.Dinov2Victim`` (the HF ViT / DINOv2 of ASR_fast.py:47-58: LayerNorm, GELU and
attention gradients in ``csrc/vit_grad.hip`` and ``csrc/attention_bwd.hip``) ``victims.ConvNeXtVictim`` (ASR_fast.py:21-26: ``csrc/convnext_grad.hip``), ``victims.SwinVictim`` (ASR_fast.py:27-32: the window
attention gradient with its score bias) or ``victims.EfficientNetV2S`` (ASR_fast.py:59-65: ``csrc/effnet_grad.hip``) -- every victim
family of ASR_fast.py; the fastai learner pickle of tools/train_shadow.py:50 is not loadable here.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .shadow import gaussian_taps


def _victim(classifier):
    model = getattr(classifier, "model", classifier)
    if not hasattr(model, "grad_engine"):
        raise _lib.AdvsError(f"{type(model).__name__} has no HIP backward plan: the gradient attack needs an "
                             "advshadow_amd.victims.ResNet50, VGG, ViTVictim, Dinov2Victim, ConvNeXtVictim, SwinVictim or EfficientNetV2S (there is no autograd fallback)")
    return model


def _mask4(mask, B, Cc, H, W, dev):
    m = torch.as_tensor(mask).to(dev, torch.float32)
    if m.dim() == 2:
        m = m[None, None]
    elif m.dim() == 3:
        m = m[None]
    if m.shape[0] == 1 and B > 1:
        m = m.expand(B, *m.shape[1:])
    if m.shape[1] not in (1, Cc) or tuple(m.shape[2:]) != (H, W) or m.shape[0] != B:
        raise ValueError(f"feature mask {tuple(m.shape)} does not fit images [{B},{Cc},{H},{W}]")
    return m.contiguous()


def adversarial_perturbation_batch(model, images, labels, masks, epsilon=0.05, alpha=0.005, iterations=20):
    """Batched train_shadow.py:177-221: every image runs its own 20-step iterative-gradient attack
    (perturbation -= alpha * sign(grad * mask), clamped to +-epsilon; untargeted: *towards lower* loss, exactly as the
    reference writes it).  images [B,3,H,W] f32 in [0,1] on the GPU, labels [B], masks [B|1, 1|3, H, W].
    Returns (clamp(images + perturbation, 0, 1), perturbation)."""
    model = _victim(model)
    lib = _lib.load()
    B, Cc, H, W = images.shape
    dev = images.device
    eng = model.grad_engine(B, H)
    cur = torch.cuda.current_stream(dev)
    x0 = images.contiguous().float()
    m = _mask4(masks, B, Cc, H, W, dev)
    lab = torch.as_tensor(labels).to(dev, torch.int64).reshape(B)
    eng.stream.wait_stream(cur)
    with torch.cuda.stream(eng.stream):
        s = eng.stream.cuda_stream
        pert = torch.zeros_like(x0)
        out = torch.empty_like(x0)
        eng.x.copy_(x0, non_blocking=True)
        eng.labels.copy_(lab, non_blocking=True)
        for _ in range(int(iterations)):
            eng.run()                                                     # forward + backward: eng.grad = d CE / d x
            check(lib.advs_iga_step(x0.data_ptr(), eng.grad.data_ptr(), m.data_ptr(), pert.data_ptr(), eng.x.data_ptr(),
                                    B, Cc, H * W, m.shape[1], 1, float(alpha), float(epsilon), s), "iga_step")
        check(lib.advs_perturb_clamp01(x0.data_ptr(), pert.data_ptr(), out.data_ptr(), x0.numel(), s), "perturb_clamp01")
    cur.wait_stream(eng.stream)
    for t in (x0, m, lab, pert, out):
        t.record_stream(cur)
    return out, pert


def apply_adversarial_perturbation(classifier, original_image, label, device, feature_mask, epsilon=0.05, alpha=0.005,
                                   iterations=20):
    """Reference signature (tools/train_shadow.py:177-178).  original_image [C,H,W] -> perturbed image [C,H,W]."""
    dev = torch.device(device) if device is not None else original_image.device
    out, _ = adversarial_perturbation_batch(classifier, original_image.to(dev)[None], torch.as_tensor(label).reshape(1),
                                            _mask4(feature_mask, 1, original_image.shape[0], *original_image.shape[1:], dev),
                                            epsilon, alpha, iterations)
    return out[0]


def integrated_gradient_perturbation(classifier, original_image, label, device, feature_mask, epsilon=0.5, alpha=0.005,
                                     iterations=10, steps=20, baseline=None):
    """ddim2/test.py:647-681.  The steps + 1 interpolations between a random baseline and the image are one batch of the
    backward plan; their gradients are summed inside ``advs_iga_step`` (the positive factors 1 / steps and 1 / L1-norm do
    not change the sign the update uses).  The path does not depend on the perturbation, so the reference's ``iterations``
    passes recompute the same integrated gradient: it is computed once and the decaying-step updates are applied to it.
    ``baseline`` [C,H,W] defaults to ``torch.randn`` on the device, as in the reference.
    Returns (perturbed image [C,H,W], display perturbation HWC numpy in [0,1])."""
    model = _victim(classifier)
    lib = _lib.load()
    dev = torch.device(device) if device is not None else original_image.device
    Cc, H, W = original_image.shape
    x0 = original_image.to(dev, torch.float32).contiguous()[None]
    base = (torch.randn_like(x0) if baseline is None else torch.as_tensor(baseline).to(dev, torch.float32).reshape(x0.shape)).contiguous()
    m = _mask4(feature_mask, 1, Cc, H, W, dev)
    nb = int(steps) + 1
    eng = model.grad_engine(nb, H)
    cur = torch.cuda.current_stream(dev)
    eng.stream.wait_stream(cur)
    with torch.cuda.stream(eng.stream):
        s = eng.stream.cuda_stream
        pert = torch.zeros_like(x0)
        out = torch.empty_like(x0)
        check(lib.advs_lerp_stack(base.data_ptr(), x0.data_ptr(), eng.x.data_ptr(), int(steps), x0.numel(), s), "lerp_stack")
        eng.labels.copy_(torch.as_tensor(label).to(dev, torch.int64).reshape(1).expand(nb), non_blocking=True)
        eng.run()
        for i in range(int(iterations)):
            check(lib.advs_iga_step(x0.data_ptr(), eng.grad.data_ptr(), m.data_ptr(), pert.data_ptr(), 0, 1, Cc, H * W,
                                    m.shape[1], nb, float(alpha) / (i + 1) ** 0.5, float(epsilon), s), "iga_step")
        check(lib.advs_perturb_clamp01(x0.data_ptr(), pert.data_ptr(), out.data_ptr(), x0.numel(), s), "perturb_clamp01")
    cur.wait_stream(eng.stream)
    for t in (x0, base, m, pert, out):
        t.record_stream(cur)
    shown = np.clip((pert[0].cpu().numpy().transpose(1, 2, 0) + 1) / 2, 0, 1)
    return out[0], shown


def apply_shadow_adversarial_batch(model, images, centers, radii, feature_masks, labels, shadow_intensity=0.43,
                                   epsilon=0.01, blur_kernel_size=5, alpha=0.005, iterations=20, integrated=None):
    """Batched classifier branch of ``apply_shadow`` (train_shadow.py:242-266): shadow composite, gradient attack on the
    shadowed image restricted to the combined mask, blend back through that mask, clamp.
    ``integrated`` = dict(steps=, iterations=, baseline=) switches the attack to ddim2/test.py:647-681 (one image)."""
    lib = _lib.load()
    B, Cc, H, W = images.shape
    dev = images.device
    img = images.contiguous().float()
    fm = _mask4(feature_masks, B, Cc, H, W, dev)
    ctr = torch.as_tensor(centers, dtype=torch.float32).to(dev).reshape(B, 2).contiguous()
    rad = torch.as_tensor(radii, dtype=torch.float32).to(dev).reshape(B).contiguous()
    shadowed, cm, out = torch.empty_like(img), torch.empty_like(img), torch.empty_like(img)
    taps = gaussian_taps(blur_kernel_size if blur_kernel_size else 1)
    arr = (C.c_float * len(taps))(*taps)
    s = torch.cuda.current_stream(dev).cuda_stream
    check(lib.advs_apply_shadow_parts(img.data_ptr(), fm.data_ptr(), ctr.data_ptr(), rad.data_ptr(), shadowed.data_ptr(),
                                      cm.data_ptr(), B, Cc, H, W, fm.shape[1], float(shadow_intensity), arr, len(taps), s),
          "apply_shadow_parts")
    if integrated is None:
        adv, _ = adversarial_perturbation_batch(model, shadowed, labels, cm, epsilon, alpha, iterations)
    else:
        if B != 1:
            raise ValueError("the integrated-gradient attack takes one image at a time (its batch is the interpolation path)")
        one, _ = integrated_gradient_perturbation(model, shadowed[0], labels, dev, cm[0], epsilon, alpha,
                                                  integrated.get("iterations", 10), integrated.get("steps", 20),
                                                  integrated.get("baseline"))
        adv = one[None].contiguous()
    check(lib.advs_blend_mask_clamp01(img.data_ptr(), adv.data_ptr(), cm.data_ptr(), out.data_ptr(), img.numel(), s),
          "blend_mask_clamp01")
    return out


def optimize_shadow_position(classifier, original_image, mask, target_label, device, lr=1e-1, iterations=1,
                             variant="train_shadow", baselines=None, ig_iterations=10, ig_steps=20):
    """Reference signature (tools/train_shadow.py:76-77).  Returns (shadow_center, shadow_radius, shadowed_image).

    What the reference's loop computes: the shadow centre stays at the mask's centroid -- taken as ``mean(0)[1:]`` of
    ``nonzero(mask)``, i.e. (row, column), and then used by ``apply_shadow`` as (x, y); kept as is.  The only
    differentiable path from the loss to ``shadow_radius`` is the regulariser ``0.01 * radius**2`` (the circular mask is
    a comparison, the perturbation is detached), so Adam moves the radius by that gradient alone; each iteration applies
    ``apply_shadow`` (composite + gradient attack) to the previous iteration's image.

    ``variant="ddim2"`` is ddim2/test.py:479-617 (call it with ``iterations=11`` for that file's default): radius starts at
    15, AdamW, the regulariser enters the loss with a minus sign (the radius grows), shadow intensity 0.051 and the
    integrated-gradient attack (``baselines``: optional list of per-iteration [C,H,W] baselines instead of ``torch.randn``;
    ``ig_iterations`` / ``ig_steps``: that attack's loop counts, the reference's defaults);
    the Grad-CAM / matplotlib display of that file is not reproduced."""
    dev = torch.device(device) if device is not None else original_image.device
    model = _victim(classifier)
    ddim2 = variant == "ddim2"
    if variant not in ("train_shadow", "ddim2"):
        raise ValueError(f"optimize_shadow_position: unknown variant {variant!r}")
    mask = torch.as_tensor(mask)
    idx = torch.nonzero(mask.cpu())
    mask_center = idx.float().mean(0)[1:]
    shadow_center = mask_center.clone()
    radius = torch.nn.Parameter(torch.tensor(15.0 if ddim2 else 20.0))
    # host scalar: the reference's optimiser itself
    opt = torch.optim.AdamW([radius], lr=lr) if ddim2 else torch.optim.Adam([radius], lr=lr)
    image = original_image.clone().to(dev)
    Cc, H, W = original_image.shape
    fm = _mask4(mask, 1, Cc, H, W, dev)
    label = torch.as_tensor(target_label).reshape(1)
    for it in range(int(iterations)):
        opt.zero_grad()
        ctr = torch.as_tensor([float(shadow_center[0]), float(shadow_center[1])])[None]
        rad = torch.as_tensor([float(radius.detach())])
        if ddim2:
            integ = {"iterations": ig_iterations, "steps": ig_steps, "baseline": None if baselines is None else baselines[it]}
            image = apply_shadow_adversarial_batch(model, image[None], ctr, rad, fm, label, 0.051, 0.01, 5, integrated=integ)[0]
        else:
            image = apply_shadow_adversarial_batch(model, image[None], ctr, rad, fm, label)[0]
        # d loss / d r: loss = -adv + 0.01 * (|c - c0|^2 + r^2)  |  ddim2: -100 * adv - 0.01 * (...)
        radius.grad = ((-0.02 if ddim2 else 0.02) * radius.detach()).clone()
        opt.step()
        with torch.no_grad():
            shadow_center.clamp_(min=0, max=W)
            radius.clamp_(min=0, max=min(H, W) / 2)
    return shadow_center.detach(), radius.detach(), image
