#!/usr/bin/env python
"""Run ONE attention shape repeatedly (for rocprofv3 --pmc / --kernel-trace).  GPU box only.
    python tools/attn_one.py [--batch 32] [--n 1024] [--heads 4] [--d 64] [--dtype bf16] [--reps 5]
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
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    bld = Builder(dev, a.dtype, st, a.batch)
    C = a.heads * a.d
    qkv = bld.buf((a.batch, 1, a.n, 3 * C))
    qkv.copy_(torch.randn(qkv.shape, device=dev).to(qkv.dtype))
    bld.attention(qkv, a.heads, a.d, 0, a.d, 2 * a.d, 3 * a.d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fl = 4.0 * a.batch * a.heads * a.n * a.n * a.d
    for r in range(a.reps):
        with torch.cuda.stream(st):
            e0.record(st)
            bld.plan.run_eager()
            e1.record(st)
        st.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"rep {r}: {ms * 1e3:.0f} us  {fl / ms / 1e9:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
