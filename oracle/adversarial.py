"""Oracle: the gradient-based perturbation of the training-data synthesis loop, with torch autograd on CPU.

Test infrastructure only (see ``oracle/__init__.py``).  Restates tools/train_shadow.py:76-266
(``apply_adversarial_perturbation``, the classifier branch of ``apply_shadow``, ``optimize_shadow_position``) and the
integrated-gradient variant ddim2/test.py:647-681 with the same torch calls (``F.cross_entropy`` + ``backward``), over
the functional ResNet-50 of ``oracle/victims.py``.  PARITY UNPINNED: neither module can be imported here to generate
vectors (tools/train_shadow.py:50 unpickles a fastai learner at import; ddim2/test.py imports pytorch_grad_cam), and
the reference ships no outputs of these functions.
"""
import torch
import torch.nn.functional as F

from .shadow import gaussian_blur_reflect101
from .victims import resnet50_logits


def input_gradient(sd, x, labels, logits_fn=None):
    """d cross_entropy(model(x_b), label_b) / d x_b for every image of the batch (train_shadow.py:204-212, batch of one).
    ``logits_fn(sd, x)``: the functional victim, ResNet-50 by default."""
    x = x.clone().requires_grad_(True)
    logits = (logits_fn or resnet50_logits)(sd, x)
    F.cross_entropy(logits, labels, reduction="sum").backward()
    return logits.detach(), x.grad.detach()


def apply_adversarial_perturbation(sd, original_image, label, feature_mask, epsilon=0.05, alpha=0.005, iterations=20,
                                   grads=None, logits_fn=None):
    """train_shadow.py:177-221.  ``grads`` (optional list) receives each iteration's masked gradient."""
    image = original_image[None]
    pert = torch.zeros_like(image)
    for _ in range(iterations):
        _, g = input_gradient(sd, image + pert, label, logits_fn)
        gm = g * feature_mask
        if grads is not None:
            grads.append(gm[0])
        pert = torch.clamp(pert - alpha * gm.sign(), min=-epsilon, max=epsilon)
    return torch.clamp(original_image + pert[0], 0, 1), pert[0]


def integrated_gradient_perturbation(sd, original_image, label, feature_mask, baseline, epsilon=0.5, alpha=0.005,
                                     iterations=10, steps=20):
    """ddim2/test.py:647-672 with the random baseline passed in."""
    x = original_image[None]
    pert = torch.zeros_like(x)
    for i in range(iterations):
        ig = torch.zeros_like(x)
        a = alpha / (i + 1) ** 0.5
        for k in range(steps + 1):
            _, g = input_gradient(sd, baseline[None] + (k / steps) * (x - baseline[None]), label)
            ig += g / steps
        ng = ig / (torch.norm(ig, p=1) + 1e-8)
        pert = torch.clamp(pert - a * (ng * feature_mask).sign(), min=-epsilon, max=epsilon)
    return torch.clamp(x + pert, 0, 1)[0], pert[0], ig[0]


def apply_shadow(sd, image, shadow_center, shadow_radius, feature_mask, target_label, shadow_intensity=0.43, epsilon=0.01,
                 blur_kernel_size=5):
    """train_shadow.py:242-266 with the classifier branch."""
    C, H, W = image.shape
    Y, X = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    dist = torch.sqrt((X - shadow_center[0]) ** 2 + (Y - shadow_center[1]) ** 2)
    m = (dist <= shadow_radius).float()
    mb = torch.from_numpy(gaussian_blur_reflect101(m.numpy(), blur_kernel_size))
    cm = mb * feature_mask
    shadowed = image * (1 - cm) + cm * (image * (1 - shadow_intensity))
    adv, _ = apply_adversarial_perturbation(sd, shadowed, target_label, cm, epsilon)
    return torch.clamp(image * (1 - cm) + adv * cm, 0, 1)


def optimize_shadow_position(sd, original_image, mask, target_label, lr=1e-1, iterations=1):
    """train_shadow.py:76-144, autograd and Adam as the reference runs them (the classifier forward for the loss
    included, although no gradient reaches the radius through it)."""
    mask_center = torch.nonzero(mask).float().mean(0)[1:]
    shadow_center = mask_center.clone()
    radius = torch.nn.Parameter(torch.tensor(20.0), requires_grad=True)
    opt = torch.optim.Adam([radius], lr=lr)
    image = original_image.clone()
    for _ in range(iterations):
        opt.zero_grad()
        upd = apply_shadow(sd, image, shadow_center, radius, mask, target_label)
        out = resnet50_logits(sd, upd[None])
        adv_loss = F.cross_entropy(out, target_label)
        reg = (shadow_center - mask_center).pow(2).sum() + radius.pow(2)
        loss = -adv_loss + reg * 0.01
        loss.backward()
        if radius.grad is not None:
            opt.step()
        with torch.no_grad():
            shadow_center.clamp_(min=0, max=original_image.size(2))
            radius.clamp_(min=0, max=min(original_image.size(1), original_image.size(2)) / 2)
        image = upd.detach()
    return shadow_center.detach(), radius.detach(), image


def apply_shadow_ig(sd, image, shadow_center, shadow_radius, feature_mask, target_label, baseline, shadow_intensity=0.051,
                    epsilon=0.01, blur_kernel_size=5, ig_iterations=10, ig_steps=20):
    """ddim2/test.py:830-871: as apply_shadow, with the integrated-gradient attack (epsilon 0.01, defaults otherwise)."""
    C, H, W = image.shape
    Y, X = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    dist = torch.sqrt((X - shadow_center[0]) ** 2 + (Y - shadow_center[1]) ** 2)
    m = (dist <= shadow_radius).float()
    mb = torch.from_numpy(gaussian_blur_reflect101(m.numpy(), blur_kernel_size))
    cm = mb * feature_mask
    shadowed = image * (1 - cm) + cm * (image * (1 - shadow_intensity))
    adv, _, _ = integrated_gradient_perturbation(sd, shadowed, target_label, cm, baseline, epsilon, 0.005, ig_iterations, ig_steps)
    return torch.clamp(image * (1 - cm) + adv * cm, 0, 1)


def optimize_shadow_position_ddim2(sd, original_image, mask, target_label, baselines, lr=1e-1, iterations=11,
                                   ig_iterations=10, ig_steps=20):
    """ddim2/test.py:479-617 without its Grad-CAM / matplotlib display: AdamW, radius from 15, loss = -100 adv - 0.01 reg."""
    mask_center = torch.nonzero(mask).float().mean(0)[1:]
    shadow_center = mask_center.clone()
    radius = torch.nn.Parameter(torch.tensor(15.0), requires_grad=True)
    opt = torch.optim.AdamW([radius], lr=lr)
    image = original_image.clone()
    for it in range(iterations):
        opt.zero_grad()
        upd = apply_shadow_ig(sd, image, shadow_center, radius, mask, target_label, baselines[it],
                              ig_iterations=ig_iterations, ig_steps=ig_steps)
        out = resnet50_logits(sd, upd[None])
        adv_loss = F.cross_entropy(out, target_label)
        reg = (shadow_center - mask_center).pow(2).sum() + radius.pow(2)
        loss = -100 * adv_loss - reg * 0.01
        loss.backward()
        if radius.grad is not None:
            opt.step()
        with torch.no_grad():
            shadow_center.clamp_(min=0, max=original_image.size(2))
            radius.clamp_(min=0, max=min(original_image.size(1), original_image.size(2)) / 2)
        image = upd.detach()
    return shadow_center.detach(), radius.detach(), image
