#!/usr/bin/env python
"""Feasibility probe: does running the two halves of a batch as two forwards on two streams (GroupNorm of one half beside the
convs of the other) beat one forward of the whole batch?  GPU box only.
    python tools/dual_stream_probe.py [--batch 32] [--size 256]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from advshadow_amd import diff_model as dm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--priority", action="store_true", help="first half on a high-priority stream, second on a low-priority one")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--split", default="", help="comma-separated sub-batch sizes, one stream each (default: two halves)")
    ap.add_argument("--no-fuse", action="store_true", help="GroupNorm as its own pass everywhere (round 3: can it hide under the other half's convs?)")
    a = ap.parse_args()
    torch.manual_seed(0)
    net = dm.UNetModel(compute_dtype="bf16", use_graph=not a.eager).to("cuda").eval()
    net.fuse_norm = not a.no_fuse
    dt = dm.dtype_code("bf16")
    W = net.packed_weights(dt)
    full = dm._ForwardEngine(net, W, a.batch, a.size, dt)
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    s1 = torch.cuda.Stream(priority=hi) if a.priority else None
    s2 = torch.cuda.Stream(priority=lo) if a.priority else None
    h1 = dm._ForwardEngine(net, W, a.batch // 2, a.size, dt, stream=s1)
    h2 = dm._ForwardEngine(net, W, a.batch // 2, a.size, dt, stream=s2)
    parts = []
    if a.split:
        sizes = [int(v) for v in a.split.split(",")]
        assert sum(sizes) == a.batch
        parts = [dm._ForwardEngine(net, W, n, a.size, dt) for n in sizes]
    for e in [full, h1, h2] + parts:
        e.x.normal_()
        e.t.fill_(501)
        e.run(); e.run()
        e.stream.synchronize()

    def timeit(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps * 1e3

    def both_serial():
        h1.run(); h1.stream.synchronize(); h2.run(); h2.stream.synchronize()

    def both_concurrent():
        h1.run(); h2.run()

    print(f"one forward, batch {a.batch}:                 {timeit(full.run):8.3f} ms")
    print(f"two forwards of {a.batch // 2}, one after the other: {timeit(both_serial):8.3f} ms")
    print(f"two forwards of {a.batch // 2}, two streams:         {timeit(both_concurrent):8.3f} ms")
    if parts:
        def split_concurrent():
            for e in parts:
                e.run()
        print(f"forwards of {a.split} on {len(parts)} streams:      {timeit(split_concurrent):8.3f} ms")


if __name__ == "__main__":
    main()
