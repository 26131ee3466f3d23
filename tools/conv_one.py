#!/usr/bin/env python
"""Run ONE conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace and for A/B builds of the kernels).  GPU box only.
    python tools/conv_one.py --h 256 --c1 128 --cout 128 --k 3 --tile 0 --reps 5 [--temb] [--residual] [--extra 384] [--stats]
    python tools/conv_one.py --suite            # the headline forward's distinct halo shapes, one line each
``--lib PATH`` loads another build of libadvshadow_hip.so (tuning variants under tools/_diag/)."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SUITE = [  # (h, c1, cout, extra, temb, residual, upsample)  -- level 0..3 of the default UNetModel at 256x256
    (256, 128, 128, 0, 1, 0, 0), (256, 128, 128, 0, 0, 1, 0), (256, 128, 128, 384, 0, 0, 0), (256, 128, 128, 256, 0, 0, 0),
    (256, 256, 128, 0, 1, 0, 0), (256, 384, 128, 0, 1, 0, 0),
    (128, 128, 256, 0, 1, 0, 0), (128, 256, 256, 0, 1, 0, 0), (128, 256, 256, 0, 0, 1, 0), (128, 256, 256, 512, 0, 0, 0),
    (128, 512, 256, 0, 1, 0, 0), (128, 384, 256, 0, 1, 0, 0), (128, 256, 256, 0, 0, 0, 2),
    (64, 256, 256, 0, 1, 0, 0), (64, 512, 256, 0, 1, 0, 0), (64, 256, 256, 512, 0, 0, 0), (64, 256, 256, 0, 0, 0, 2),
    (32, 256, 256, 0, 1, 0, 0), (32, 512, 256, 0, 1, 0, 0),
]


def one(a, h, c1, cout, extra, temb, residual, ups, quiet=False):
    from advshadow_amd.engine import Builder, pack_conv_weight, pack_subpixel_upsample_weight, dtype_code
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    bld = Builder(dev, a.dtype, st, a.batch)
    dt = dtype_code(a.dtype)
    x = bld.buf((a.batch, h, h, c1))
    x.copy_(torch.randn(x.shape, device=dev).to(x.dtype))
    w_oihw = torch.randn(cout, c1, a.k, a.k, device=dev) / math.sqrt(c1 * a.k * a.k)
    kw = {}
    if ups == 2:
        w = pack_subpixel_upsample_weight(w_oihw, dt)
        kw["upsample"] = "subpixel"
    else:
        w = pack_conv_weight(w_oihw, dt)
    bias = torch.zeros(cout, device=dev)
    if extra:
        e = bld.buf((a.batch, h, h, extra))
        e.copy_(torch.randn(e.shape, device=dev).to(e.dtype))
        we = pack_conv_weight(torch.randn(cout, extra, 1, 1, device=dev) / math.sqrt(extra), dt)
        w = torch.cat([w.reshape(cout, -1), we.reshape(cout, -1)], 1).contiguous()
        kw["extra"] = (e, None)
    if temb:
        kw["temb"] = torch.randn(a.batch, cout, device=dev)
        kw["temb_stride"] = cout
    if residual:
        r = bld.buf((a.batch, h, h, cout))
        r.copy_(torch.randn(r.shape, device=dev).to(r.dtype))
        kw["residual"] = r
    bld.conv(x, w, cout, bias=bias, ksize=a.k, pad=a.k // 2, tile=a.tile, want_stats=a.stats or a.suite, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    ho = h * 2 if ups else h
    fl = 2.0 * a.batch * ho * ho * cout * (a.k * a.k * c1 + extra)
    for r in range(a.reps):
        with torch.cuda.stream(st):
            e0.record(st)
            bld.plan.run_eager()
            e1.record(st)
        st.synchronize()
        ms = e0.elapsed_time(e1)
        if r:
            best = min(best, ms)
        if not quiet:
            print(f"rep {r}: {ms * 1e3:.0f} us  {fl / ms / 1e9:.0f} TF")
    return best, fl


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
    ap.add_argument("--temb", action="store_true")
    ap.add_argument("--residual", action="store_true")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--extra", type=int, default=0)
    ap.add_argument("--upsample", type=int, default=0)
    ap.add_argument("--suite", action="store_true")
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    if a.lib:
        from advshadow_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    if not a.suite:
        one(a, a.h, a.c1, a.cout, a.extra, a.temb, a.residual, a.upsample)
        return
    tot = 0.0
    for h, c1, cout, extra, temb, res, ups in SUITE:
        ms, fl = one(a, h, c1, cout, extra, temb, res, ups, quiet=True)
        tot += ms
        print(f"h {h:3d} c1 {c1:3d} cout {cout:3d} extra {extra:3d} temb {temb} res {res} ups {ups}: {ms * 1e3:7.0f} us  {fl / ms / 1e9:6.0f} TF (algorithmic)")
    print(f"suite total {tot:.3f} ms")


if __name__ == "__main__":
    main()
