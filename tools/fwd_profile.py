#!/usr/bin/env python
"""Per-launch HIP-event timing of ONE UNetModel forward (eager replay of the plan), grouped by kernel and, for the convs, by shape.
A/B two builds on one box:  ADVS_LIB_PATH=tools/_diag/libbase.so python tools/fwd_profile.py ; python tools/fwd_profile.py
    python tools/fwd_profile.py [--batch 32] [--size 256] [--dtype bf16] [--reps 3] [--shapes]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from advshadow_amd import _lib  # noqa: E402
from advshadow_amd.diff_model import UNetModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--shapes", action="store_true")
    ap.add_argument("--tile", type=int, default=0, help="advs_conv_set_tile override for every conv that can take it (tuning)")
    a = ap.parse_args()
    lib = _lib.load()
    if a.tile:
        lib.advs_conv_set_tile(a.tile)
    torch.manual_seed(0)
    net = UNetModel(compute_dtype=a.dtype, use_graph=False).to("cuda").eval()
    eng = net.engine(a.batch, a.size, uniform_t=True)       # the sampler's plan (bench.py profiles the same one)
    eng.x.copy_(torch.randn(eng.x.shape, generator=torch.Generator().manual_seed(1)).cuda())
    eng.t.fill_(501)
    s = eng.stream.cuda_stream
    ops = eng.plan.ops
    evs = []
    for _ in range(len(ops) + 1):
        e = C.c_void_p()
        lib.advs_event_create(C.byref(e))
        evs.append(e)
    best = [1e9] * len(ops)
    for rep in range(a.reps + 1):
        eng.stream.synchronize()
        lib.advs_event_record(evs[0], s)
        for i, (fn, args) in enumerate(ops):
            assert fn(*args, s) == 0
            lib.advs_event_record(evs[i + 1], s)
        eng.stream.synchronize()
        if rep == 0:
            continue
        for i in range(len(ops)):
            ms = C.c_float()
            lib.advs_event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms))
            best[i] = min(best[i], ms.value)
    by_kernel, by_shape = {}, {}
    for (fn, args), ms in zip(ops, best):
        k = by_kernel.setdefault(fn.__name__, [0, 0.0])
        k[0] += 1
        k[1] += ms
        if fn.__name__ == "advs_conv2d":
            c = args[0]._obj
            key = (c.h, c.c1 + c.c2, c.cout, c.ksize, c.stride, c.upsample, c.ce1 + c.ce2, int(bool(c.residual)), int(bool(c.temb)))
            e = by_shape.setdefault(key, [0, 0.0])
            e[0] += 1
            e[1] += ms
    print("lib:", _lib.LIB_PATH)
    for k, (n, ms) in sorted(by_kernel.items()):
        print(f"{k:28s} x{n:3d} {ms:8.3f} ms")
    print(f"{'forward total':28s}      {sum(best):8.3f} ms")
    if a.shapes:
        print("   h  cin cout k s u extra res temb   n   ms(total)")
        for key, (n, ms) in sorted(by_shape.items(), key=lambda kv: -kv[1][1]):
            print("%4d %4d %4d %d %d %d %5d %3d %4d %3d %9.3f" % (key + (n, ms)))


if __name__ == "__main__":
    main()
