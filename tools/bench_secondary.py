#!/usr/bin/env python
"""Secondary workloads (not the headline metric): per-kernel forward breakdown + sampler throughput of
the class-conditional UNet and CSPDarkUnet (IDDM lineage) with DDIM + classifier-free guidance.  GPU box only.
    python tools/bench_secondary.py [--net unet|cspdarkunet] [--size 64] [--batch 32] [--steps 50] [--dtype bf16]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from advshadow_amd.model.networks.cspdarkunet import CSPDarkUnet  # noqa: E402
from advshadow_amd.model.networks.unet import UNet  # noqa: E402
from advshadow_amd.model.samples.ddim import DDIMDiffusion  # noqa: E402


class _PlanView:
    def __init__(self, eng, mode):
        self.stream, self.plan = eng.stream, eng.plan(mode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="unet", choices=["unet", "cspdarkunet"])
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    torch.manual_seed(0)
    cls = UNet if a.net == "unet" else CSPDarkUnet
    net = cls(num_classes=37, image_size=a.size, compute_dtype=a.dtype).to("cuda").eval()
    diff = DDIMDiffusion(sample_steps=a.steps, img_size=a.size, device="cuda")
    labels = (torch.arange(a.batch) % 37).cuda()
    xT = torch.randn(a.batch, 3, a.size, a.size, generator=torch.Generator().manual_seed(1234))
    diff.sample(net, a.batch, labels=labels, cfg_scale=3, x_T=xT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        diff.sample(net, a.batch, labels=labels, cfg_scale=3, x_T=xT)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    tot = {k: v for k, v in bench.conv_profile(_PlanView(net.engine(a.batch), "cfg")).items() if "[" not in k}     # (bench.py also returns the conv launches split into groups)
    print(json.dumps({"net": a.net, "size": a.size, "batch": a.batch, "ddim_steps": a.steps, "dtype": a.dtype, "cfg": True,
                      "images_per_s": a.batch / dt, "s_per_pass": dt,
                      "cfg_forward_ms_by_kernel": {k: round(v[1], 3) for k, v in sorted(tot.items())},
                      "cfg_forward_ms_total": round(sum(v[1] for v in tot.values()), 3)}))


if __name__ == "__main__":
    main()
