#!/usr/bin/env python
"""Run ONE conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace).  GPU box only.
    python tools/conv_one.py --h 256 --c1 128 --cout 128 --k 3 --tile 1 --reps 5
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from advshadow_amd.engine import Builder, pack_conv_weight, dtype_code  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--c1", type=int, default=128)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    bld = Builder(dev, a.dtype, st, a.batch)
    x = bld.buf((a.batch, a.h, a.h, a.c1))
    x.copy_(torch.randn(x.shape, device=dev).to(x.dtype))
    w = pack_conv_weight(torch.randn(a.cout, a.c1, a.k, a.k, device=dev) / math.sqrt(a.c1 * a.k * a.k), dtype_code(a.dtype))
    bias = torch.zeros(a.cout, device=dev)
    bld.conv(x, w, a.cout, bias=bias, ksize=a.k, pad=a.k // 2, tile=a.tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(a.reps):
        with torch.cuda.stream(st):
            e0.record(st)
            bld.plan.run_eager()
            e1.record(st)
        st.synchronize()
        ms = e0.elapsed_time(e1)
        fl = 2.0 * a.batch * a.h * a.h * a.cout * a.k * a.k * a.c1
        print(f"rep {r}: {ms * 1e3:.0f} us  {fl / ms / 1e9:.0f} TF")


if __name__ == "__main__":
    main()
