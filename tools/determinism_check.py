#!/usr/bin/env python
"""Run every launch of the UNetModel forward plan twice on the same inputs and report launches whose output bytes
differ between the two runs (a kernel that is not deterministic, or that reads memory it did not define).  GPU box only.
    python tools/determinism_check.py [--batch 32] [--size 256] [--dtype bf16]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from advshadow_amd.diff_model import UNetModel  # noqa: E402


class _Dev:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def view(ptr, nbytes):
    return torch.as_tensor(_Dev(ptr, nbytes), device="cuda")


def outputs(fn, args, esz):
    n = fn.__name__
    if n == "advs_conv2d":
        a = args[0]._obj
        ups = 1 if a.upsample else 0
        ho = ((a.h << ups) + 2 * a.pad - a.ksize) // a.stride + 1
        wo = ((a.w_ << ups) + 2 * a.pad - a.ksize) // a.stride + 1
        outs = [(a.y, a.b * ho * wo * a.cout * esz)]
        if a.stats:
            outs.append((a.stats, (a.b * ho * wo // a.stats_rows) * a.cout * 8))
        return outs
    if n == "advs_groupnorm_stats":
        return [(args[11], args[13] * args[14] * (args[15] + args[16]) * esz)]
    if n == "advs_groupnorm":
        return [(args[7], args[9] * args[10] * (args[11] + args[12]) * esz)]
    if n == "advs_attention_masked":
        return [(args[1], args[2] * args[3] * args[5] * args[6] * esz)]
    if n == "advs_conv3x3_first_stats":
        return [(args[3], args[5] * args[7] * args[8] * args[9] * esz)]
    if n == "advs_conv_last":
        return [(args[3], args[4] * args[8] * args[6] * args[7] * 4)]
    return []


def main(argv=None):
    """Returns the number of launches whose rerun differed (0 = every launch bit-reproducible)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=1, help="reruns of every launch compared with its first run")
    ap.add_argument("--convs-only", action="store_true")
    ap.add_argument("--lib", default=None, help="load this build of libadvshadow_hip.so instead of the in-tree one (A/B diagnostics)")
    a = ap.parse_args(argv)
    if a.lib:
        from advshadow_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    torch.manual_seed(0)
    net = UNetModel(compute_dtype=a.dtype, use_graph=False).to("cuda").eval()
    eng = net.engine(a.batch, a.size)
    eng.x.copy_(torch.randn(eng.x.shape, generator=torch.Generator().manual_seed(1)).cuda())
    eng.t.fill_(501)
    s = eng.stream.cuda_stream
    esz = 4 if a.dtype == "fp32" else 2
    bad = 0
    for i, (fn, args) in enumerate(eng.plan.ops):
        outs = outputs(fn, args, esz)
        if a.convs_only and fn.__name__ != "advs_conv2d":
            assert fn(*args, s) == 0
            continue
        assert fn(*args, s) == 0
        eng.stream.synchronize()
        first = [view(p, n).clone() for p, n in outs]
        for rep in range(a.reps):
            assert fn(*args, s) == 0
            eng.stream.synchronize()
            for k, (p, n) in enumerate(outs):
                cur = view(p, n)
                if not torch.equal(cur, first[k]):
                    d = (cur != first[k]).sum().item()
                    extra = ""
                    if fn.__name__ == "advs_conv2d":
                        c = args[0]._obj
                        extra = f" h={c.h} c1={c.c1} c2={c.c2} cout={c.cout} k={c.ksize} s={c.stride} ups={c.upsample} e1={bool(c.e1)} res={bool(c.residual)} tile={c.tile}"
                    print(f"op {i} {fn.__name__} output {k}: {d} of {n} bytes differ on rerun {rep}{extra}", flush=True)
                    if k == 1 and fn.__name__ == "advs_conv2d":
                        c = args[0]._obj
                        f1, f2 = first[k].view(torch.float32).view(-1, c.cout, 2), cur.view(torch.float32).view(-1, c.cout, 2)
                        idx = (f1 != f2).nonzero()
                        print("   entries differing:", idx.shape[0], "sum:", int((idx[:, 2] == 0).sum()), "sumsq:", int((idx[:, 2] == 1).sum()),
                              "channels mod 8:", sorted(set((idx[:, 1] % 8).tolist())), "row-blocks mod 4:", sorted(set((idx[:, 0] % 4).tolist())))
                        for r in idx[:8].tolist():
                            print("   row-block", r[0], "channel", r[1], "sum/sq", r[2], f1[tuple(r)].item(), f2[tuple(r)].item(),
                                  "delta", f2[tuple(r)].item() - f1[tuple(r)].item())
                    bad += 1
                    break
    print("nondeterministic launches:", bad, "of", len(eng.plan.ops))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
