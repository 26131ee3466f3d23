#!/usr/bin/env python
"""Time every distinct conv launch of a UNetModel plan under each tile configuration
(advs_conv_set_tile) with HIP events on the engine's stream.  GPU box only.

    python tools/tune_conv.py [--batch 32] [--size 256] [--dtype bf16]
"""
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
    ap.add_argument("--tiles", default="1,2,3,4")
    args = ap.parse_args()
    lib = _lib.load()
    torch.manual_seed(0)
    net = UNetModel(compute_dtype=args.dtype, use_graph=False).to("cuda").eval()
    eng = net.engine(args.batch, args.size)
    s = eng.stream.cuda_stream
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.advs_event_create(C.byref(e0)); lib.advs_event_create(C.byref(e1))
    eng.plan.run_eager(); eng.stream.synchronize()
    shapes = {}
    for fn, a in eng.plan.ops:
        if fn.__name__ != "advs_conv2d":
            continue
        c = a[0]._obj
        key = (c.h, c.w_, c.c1, c.c2, c.cout, c.ksize, c.stride, c.upsample, bool(c.residual), bool(c.temb))
        shapes.setdefault(key, []).append((fn, a))
    tiles = [int(t) for t in args.tiles.split(",")]
    print(f"{'h':>4} {'c1':>4} {'c2':>4} {'cout':>4} k s u res  n " + " ".join(f"{'t%d us (TF)' % t:>16}" for t in tiles))
    tot = {t: 0.0 for t in tiles}
    for key, ops in shapes.items():
        h, w, c1, c2, cout, k, st, up, res, temb = key
        hl = h * 2 if up else h
        ho = (hl + 2 * (k // 2) - k) // st + 1
        flops = 2.0 * args.batch * ho * ho * cout * k * k * (c1 + c2)
        fn, a = ops[0]
        cells = []
        for t in tiles:
            lib.advs_conv_set_tile(t)
            best = 1e9
            ok = True
            for rep in range(4):
                lib.advs_event_record(e0, s)
                if fn(*a, s) != 0:
                    ok = False                      # tile not applicable to this shape
                    break
                lib.advs_event_record(e1, s)
                ms = C.c_float()
                lib.advs_event_elapsed_ms(e0, e1, C.byref(ms))
                if rep:
                    best = min(best, ms.value)
            if not ok:
                cells.append("n/a")
                tot[t] += float("nan")
                continue
            tot[t] += best * len(ops)
            cells.append(f"{best * 1e3:8.0f} ({flops / best / 1e9:5.0f})")
        print(f"{h:4d} {c1:4d} {c2:4d} {cout:4d} {k} {st} {up} {int(res)}{int(temb)} {len(ops):2d} " + " ".join(f"{c:>16}" for c in cells))
    lib.advs_conv_set_tile(0)
    print("total conv ms per forward by tile:", {t: round(v, 2) for t, v in tot.items()})


if __name__ == "__main__":
    main()
