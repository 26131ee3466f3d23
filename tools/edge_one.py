#!/usr/bin/env python
"""Time the thin convolutions at the ends of the eps-predictor (advs_conv3x3_first / advs_conv_last) in
isolation, with the input either cold (a 1 GiB buffer is streamed in between) or just written.  GPU box only.
    python tools/edge_one.py [--batch 32] [--size 256] [--ch 128] [--dtype bf16]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from advshadow_amd.engine import Builder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--ch", type=int, default=128)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    B, S, C = a.batch, a.size, a.ch
    last = Builder(dev, a.dtype, st, B)
    h = last.buf((B, S, S, C))
    h.copy_(torch.randn(h.shape, device=dev).to(h.dtype))
    out = torch.empty(B, 3, S, S, device=dev)
    last.conv_last(h, torch.randn(3, C, 3, 3, device=dev) * 0.05, torch.zeros(3, device=dev), 3, 3, out)
    first = Builder(dev, a.dtype, st, B)
    x = torch.randn(B, 3, S, S, device=dev)
    first.conv_first(x, torch.randn(C, 3, 3, 3, device=dev) * 0.2, torch.zeros(C, device=dev), C)
    flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, bld in (("conv_last", last), ("conv3x3_first", first)):
        for cold in (True, False):
            ts = []
            for _ in range(a.reps):
                with torch.cuda.stream(st):
                    if cold:
                        flush.fill_(1)
                    e0.record(st)
                    bld.plan.run_eager()
                    e1.record(st)
                st.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            print(f"{name:14s} {'cold' if cold else 'warm'}: " + " ".join(f"{t:.0f}" for t in ts) + " us", flush=True)


if __name__ == "__main__":
    main()
