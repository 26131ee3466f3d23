#!/usr/bin/env python
"""The headline workload through the HANDLE-LEVEL C entry points (advs_unet_plan + advs_ddim_run; csrc/unet_handle.hip) beside the
Python plan, same weights, same box: the two hosts replay the same graph, so their rates must agree.
    python tools/bench_handle.py [--batch 32] [--size 256] [--steps 50] [--reps 3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from advshadow_amd.diff_model import GaussianDiffusion, UNetModel  # noqa: E402
from advshadow_amd.handle import CUNet, ddim_tables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    torch.manual_seed(0)
    net = UNetModel(compute_dtype=a.dtype).to("cuda").eval()
    diff = GaussianDiffusion(timesteps=1000, beta_schedule="cosine")
    xT = torch.randn(a.batch, 3, a.size, a.size, device="cuda")

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps, out

    tp, xp = timed(lambda: diff.ddim_sample(net, a.size, batch_size=a.batch, ddim_timesteps=a.steps, x_T=xT, return_tensor=True))
    del diff._loops
    diff._loops = {}
    net._engines.clear()                     # the Python plan's activations go back before the C host takes its own
    torch.cuda.empty_cache()
    c = CUNet(compute_dtype=a.dtype)
    c.load_state_dict(net.state_dict())
    half = net.model_channels // 2
    c.set_param("freqs", net.packed_weights(1 if a.dtype == "bf16" else (2 if a.dtype == "fp16" else 0))["freqs"])
    c.plan(a.batch, a.size, uniform_t=True)
    coef, tseq = ddim_tables(1000, a.steps, "cosine", "uniform", 0.0)
    tc, xc = timed(lambda: c.ddim_run(xT, coef, tseq))
    print(json.dumps({"workload": f"DDIM-{a.steps}, batch {a.batch}, {a.size}x{a.size}, {a.dtype}, default UNetModel",
                      "python_plan_img_per_s": a.batch / tp, "c_handle_img_per_s": a.batch / tc,
                      "max_abs_diff_of_samples": (xp - xc).abs().max().item()}))
    c.close()


if __name__ == "__main__":
    main()
