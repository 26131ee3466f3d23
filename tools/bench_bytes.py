#!/usr/bin/env python
"""Achieved HBM GB/s of the byte-bound kernels around the sampler (composite, resize, JPEG hop, metrics, GroupNorm)
at the attack-loop shapes of BASELINE config 2 (batch 64, 256x256).  GPU box only.
    python tools/bench_bytes.py [--batch 64] [--size 256]
Algorithmic bytes = every input read once + every output written once (stated per line).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from advshadow_amd import imageops, metrics, shadow  # noqa: E402
from advshadow_amd.engine import Builder  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    a = ap.parse_args()
    B, S = a.batch, a.size
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(7)
    clean = torch.rand(B, 3, S, S, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    fmask = (((yy - S // 2) ** 2 + (xx - S // 2) ** 2) <= (0.31 * S) ** 2).float()[None, None].repeat(B, 1, 1, 1).to(dev)
    centers = torch.tensor([[S // 2, S // 2]] * B, dtype=torch.float32, device=dev)
    radii = torch.full((B,), 0.156 * S, device=dev)
    u8 = (torch.rand(B, 3, S, S, generator=g) * 255).to(torch.uint8).to(dev)
    hwc = imageops.u8_nchw_to_hwc(u8)
    out = {}

    def rec(name, seconds, nbytes, what):
        out[name] = {"us": round(seconds * 1e6, 1), "GB/s": round(nbytes / seconds / 1e9, 1), "bytes": what}

    t = timeit(lambda: shadow.apply_shadow_batch(clean, centers, radii, fmask))
    rec("apply_shadow (f32 3ch in + mask + out)", t, B * S * S * (12 + 4 + 12), "28 B/pixel")
    t = timeit(lambda: imageops.u8_nchw_to_hwc(u8))
    rec("u8 NCHW->HWC", t, B * S * S * 6, "6 B/pixel")
    t = timeit(lambda: imageops.jpeg_roundtrip(hwc))
    rec("jpeg_roundtrip q75", t, B * S * S * 9, "9 B/pixel (3 in, 3 scratch, 3 out)")
    t = timeit(lambda: imageops.resize_u8(hwc, 224, 224))
    rec("resize 256->224 (2 passes)", t, B * 3 * (S * S + 2 * S * 224 + 224 * 224), "in + 2x intermediate + out")
    r224 = imageops.resize_u8(hwc, 224, 224)
    t = timeit(lambda: imageops.to_tensor(r224))
    rec("ToTensor u8 HWC -> f32 NCHW", t, B * 224 * 224 * 3 * 5, "5 B/element")
    a64, b64 = torch.rand(B, 3, 64, 64, device=dev), torch.rand(B, 3, 64, 64, device=dev)
    t = timeit(lambda: metrics.ssim_psnr_batch(a64, b64, 7))
    rec("psnr_ssim 64x64 pairs", t, B * 3 * 64 * 64 * 8, "two f32 images")
    # GroupNorm(32)+SiLU on the headline activation (batch 32, 256x256, 128 channels, bf16), statistics given
    st = torch.cuda.Stream(dev)
    bld = Builder(dev, "bf16", st, 32)
    x = bld.buf((32, S, S, 128))
    x.copy_(torch.randn(x.shape, device=dev).to(x.dtype))
    bld.groupnorm(x, torch.ones(128, device=dev), torch.zeros(128, device=dev), 32, act="silu")
    torch.cuda.synchronize()

    def gn():
        with torch.cuda.stream(st):
            bld.plan.run_eager()
        st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        with torch.cuda.stream(st):
            e0.record(st)
            bld.plan.run_eager()
            e1.record(st)
        st.synchronize()
        best = min(best, e0.elapsed_time(e1))
    rec("GroupNorm(32)+SiLU 32x256x256x128 bf16 (stats pass + apply)", best * 1e-3, 32 * S * S * 128 * 2 * 3, "read, read, write")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
