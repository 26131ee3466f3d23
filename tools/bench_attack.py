#!/usr/bin/env python
"""Gradient attack throughput (not the headline metric): images/s of the 20-iteration iterative-gradient
perturbation (tools/train_shadow.py:177-221) on the ResNet-50 victim, and the time of one forward + backward-to-image
replay.  GPU box only.
    python tools/bench_attack.py [--victim resnet50|vgg16|vit|dinov2|convnext|swin|effnet] [--size 224] [--batch 32] [--iters 20] [--dtype fp32|bf16] [--cpu-images 1]
``--cpu-images N`` also times the CPU autograd oracle on N images (the reference's own code path, one image at a time).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from advshadow_amd import adversarial  # noqa: E402
from advshadow_amd.victims import ResNet50  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--cpu-images", type=int, default=0)
    ap.add_argument("--victim", default="resnet50", choices=["resnet50", "vgg16", "vit", "dinov2", "convnext", "swin", "effnet"])
    a = ap.parse_args()
    torch.manual_seed(0)
    if a.victim == "vit":
        from advshadow_amd.victims import ViTVictim
        net = ViTVictim(37, compute_dtype=a.dtype).to("cuda").eval()
    elif a.victim == "dinov2":
        from advshadow_amd.victims import Dinov2Victim
        net = Dinov2Victim(37, image_size=a.size, compute_dtype=a.dtype).to("cuda").eval()
    elif a.victim == "convnext":
        from advshadow_amd.victims import ConvNeXtVictim
        net = ConvNeXtVictim(37, image_size=a.size, compute_dtype=a.dtype).to("cuda").eval()
    elif a.victim == "swin":
        from advshadow_amd.victims import SwinVictim
        net = SwinVictim(37, image_size=a.size, compute_dtype=a.dtype).to("cuda").eval()
    elif a.victim == "effnet":
        from advshadow_amd.victims import EfficientNetV2S
        net = EfficientNetV2S(37, image_size=a.size, compute_dtype=a.dtype).to("cuda").eval()
    elif a.victim == "vgg16":
        from advshadow_amd.victims import VGG
        net = VGG(16, 37, compute_dtype=a.dtype).to("cuda").eval()
    else:
        net = ResNet50(num_classes=37, compute_dtype=a.dtype).to("cuda").eval()
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(a.batch, 3, a.size, a.size, generator=g).cuda()
    labels = (torch.arange(a.batch) % 37).cuda()
    masks = torch.ones(a.batch, 1, a.size, a.size, device="cuda")
    adversarial.adversarial_perturbation_batch(net, imgs, labels, masks, 0.05, 0.005, 2)      # build + capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        adversarial.adversarial_perturbation_batch(net, imgs, labels, masks, 0.05, 0.005, a.iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    eng = net.grad_engine(a.batch, a.size)
    tot = {k: v for k, v in bench.conv_profile(eng).items() if "[" not in k}     # (bench.py also returns the conv launches split into groups)
    rec = {"workload": "iterative-gradient attack on " + {"vgg16": "VGG16", "vit": "ViT-B/16", "dinov2": "DINOv2-B/14", "convnext": "ConvNeXt-B", "swin": "Swin-B", "effnet": "EfficientNetV2-S", "resnet50": "ResNet-50"}[a.victim], "size": a.size, "batch": a.batch, "iterations": a.iters,
           "dtype": a.dtype, "images_per_s": a.batch / dt, "s_per_batch": dt,
           "fwd_bwd_ms_by_kernel": {k: round(v[1], 3) for k, v in sorted(tot.items())},
           "fwd_bwd_ms_total": round(sum(v[1] for v in tot.values()), 3)}
    if a.cpu_images > 0:
        from oracle import adversarial as oa, victims as ov
        fn = ov.vgg_logits if a.victim == "vgg16" else None
        torch.set_num_threads(bench.host_cores())
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        t0 = time.perf_counter()
        for i in range(a.cpu_images):
            oa.apply_adversarial_perturbation(sd, imgs[i].cpu(), labels[i:i + 1].cpu(), masks[i].cpu(), 0.05, 0.005, a.iters, logits_fn=fn)
        rec["cpu_oracle_images_per_s"] = a.cpu_images / (time.perf_counter() - t0)
        rec["cpu_cores"] = bench.host_cores()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
