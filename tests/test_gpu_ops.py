"""Per-kernel parity on the MI355X: every C-ABI op against the same op restated with torch CPU
fp32 functional calls (the op-level oracle).  f32 mode must agree to fp32 rounding
noise; the 16-bit modes (bf16, fp16) are compared with the reference evaluated on operands rounded to that type."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_helpers import OneOp, bf16_round, dev, lp_round, nchw, nhwc, tdt  # noqa: E402
from advshadow_amd import _lib  # noqa: E402
from advshadow_amd.engine import pack_conv_weight, pack_subpixel_upsample_weight, dtype_code, ptr  # noqa: E402

DTS = ["fp32", "bf16", "fp16"]


def tol(dt, f32, bf16):
    return f32 if dt == "fp32" else bf16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------ conv
CONV_CASES = [
    # B, H, W, C1, C2, Cout, k, stride, ups, bias, temb, res, act
    (2, 12, 12, 64, 0, 128, 3, 1, 0, 1, 0, 0, None),       # M tail (288 rows)
    (1, 16, 16, 128, 0, 128, 3, 1, 0, 1, 1, 1, None),      # full epilogue
    (2, 16, 16, 64, 0, 64, 3, 2, 0, 1, 0, 0, None),        # stride 2, N tail (64 < 128)
    (3, 8, 8, 128, 0, 192, 1, 1, 0, 1, 0, 1, None),        # 1x1, N tail 192, batch 3
    (1, 8, 8, 64, 0, 128, 3, 1, 1, 1, 0, 0, None),         # nearest x2 on load
    (2, 8, 8, 64, 128, 128, 3, 1, 0, 1, 1, 0, None),       # concat on load
    (2, 8, 8, 128, 64, 64, 1, 1, 0, 1, 0, 0, "relu"),      # concat 1x1 + relu
    (1, 32, 32, 64, 0, 256, 3, 1, 0, 0, 0, 0, "silu"),     # no bias, 2 N tiles, 8 M tiles
    (1, 14, 14, 256, 0, 64, 1, 2, 0, 1, 0, 0, None),       # 1x1 stride 2 (victim downsample path)
    (2, 16, 16, 64, 128, 128, 3, 1, 0, 1, 1, 1, None),     # halo-eligible: concat, full epilogue
    (1, 48, 32, 128, 0, 256, 3, 1, 0, 1, 0, 0, "silu"),    # halo-eligible: 3x2 pixel tiles, 2 channel tiles
    (3, 16, 32, 64, 0, 64, 3, 1, 0, 0, 1, 0, None),        # halo-eligible: batch 3, N tail (64 < 128), 1 slab
    (2, 16, 32, 64, 128, 128, 3, 1, 0, 1, 1, 1, None),     # 512-pixel halo tile: concat (2 + 4 half slabs), full epilogue
    (1, 32, 64, 128, 0, 128, 3, 1, 0, 1, 1, 0, None),      # 512-pixel halo tile: 2x2 pixel tiles, the fast epilogue's shape
    (2, 32, 32, 64, 0, 192, 3, 1, 0, 1, 0, 0, None),       # second-generation halo tiles: one unit pair, N tail (192), statistics-free fast epilogue
]

HALO2_TILES = (17, 18, 19)          # conv_halo2.hip: 16 x 32 pixels / 8 waves, 16 x 16 / 8 waves, 16 x 16 / 4 waves


def halo2_ok(tile, dt, H, W):
    return dt != "fp32" and H % 16 == 0 and W % (32 if tile == 17 else 16) == 0


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15, 16, 17, 18, 19])
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, dt, tile):
    B, H, W, C1, C2, Cout, k, stride, ups, has_b, has_t, has_r, act = case
    if tile == 10 and not (k == 3 and stride == 1 and not ups and H % 16 == 0 and W % 16 == 0):
        pytest.skip("halo kernel: 3x3 stride 1, H and W multiples of 16")
    if tile in HALO2_TILES and not (k == 3 and stride == 1 and not ups and halo2_ok(tile, dt, H, W)):
        pytest.skip("second-generation halo kernels: 16-bit, 3x3 stride 1, H a multiple of 16, W of 16 (tile 17: 32)")
    pad = 1 if k == 3 else 0
    x1 = rnd(B, C1, H, W, seed=1)
    x2 = rnd(B, C2, H, W, seed=2) if C2 else None
    w = rnd(Cout, C1 + C2, k, k, seed=3, scale=1.0 / math.sqrt((C1 + C2) * k * k))
    bias = rnd(Cout, seed=4) if has_b else None
    temb = rnd(B, Cout + 5, seed=5) if has_t else None          # strided rows
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    if dt != "fp32":
        xin, wr = lp_round(dt, xin), lp_round(dt, w)
    else:
        wr = w
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, wr, bias, stride=stride, padding=pad)
    res = rnd(*ref.shape, seed=6) if has_r else None
    if has_t:
        ref = ref + temb[:, :Cout, None, None]
    if has_r:
        ref = ref + (lp_round(dt, res) if dt != "fp32" else res)
    if act == "relu":
        ref = F.relu(ref)
    elif act == "silu":
        ref = F.silu(ref)

    op = OneOp(dt, B)
    wp = pack_conv_weight(w.to(dev()), dtype_code(dt))
    tb = temb.to(dev()) if has_t else None
    y = op.b.conv(nhwc(x1, dt), wp, Cout, x2=nhwc(x2, dt) if C2 else None,
                  bias=bias.to(dev()) if has_b else None,
                  temb=tb[:, :Cout] if has_t else None, temb_stride=Cout + 5 if has_t else 0,
                  residual=nhwc(res, dt) if has_r else None,
                  ksize=k, stride=stride, pad=pad, upsample=bool(ups), act=act, tile=tile)
    op.go()
    got = nchw(y)
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err < tol(dt, 2e-5, 4e-2), err


@pytest.mark.parametrize("dt", DTS)
def test_pack_conv_weight(dt):
    w = rnd(6, 5, 3, 3, seed=9)
    p = pack_conv_weight(w.to(dev()), dtype_code(dt)).float().cpu()
    ref = w.permute(0, 2, 3, 1)
    ref = lp_round(dt, ref) if dt != "fp32" else ref
    slab = 64 if dt != "fp32" else 32                      # input channels are zero-padded to whole 128-byte slabs
    assert p.shape == (6, 3, 3, slab)
    assert torch.equal(p[..., :5], ref.contiguous()) and not p[..., 5:].any()
    w2 = rnd(4, 2 * slab, 1, 1, seed=10)
    p2 = pack_conv_weight(w2.to(dev()), dtype_code(dt)).float().cpu()
    assert torch.equal(p2, (lp_round(dt, w2) if dt != "fp32" else w2).permute(0, 2, 3, 1).contiguous())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 3, 16, 16, 64), (1, 3, 9, 13, 128), (1, 1, 8, 8, 32), (2, 3, 40, 48, 128), (1, 2, 16, 32, 64),
                                   (1, 3, 24, 16, 32)])
def test_conv_first(shape, dt):
    B, Cin, H, W, Cout = shape
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=0.2), rnd(Cout, seed=3)
    ref = F.conv2d(x, w, b, padding=1)
    op = OneOp(dt, B)
    y = op.b.conv_first(x.to(dev()), w.to(dev()), b.to(dev()), Cout)
    op.go()
    err = (nchw(y) - ref).abs().max().item()
    assert err < tol(dt, 1e-5, 3e-2), err


@pytest.mark.parametrize("dt", DTS)
def test_conv_first_statistics_feed_groupnorm(dt):
    """The MFMA first conv leaves per-channel statistics of its rounded outputs: the GroupNorm that follows
    (diff_model.py:70) must equal GroupNorm of the stored tensor, alone and as the second half of a concat."""
    B, Cin, H, W, Cout = 2, 3, 64, 32, 128
    x, w, b = rnd(B, Cin, H, W, seed=81), rnd(Cout, Cin, 3, 3, seed=82, scale=0.2), rnd(Cout, seed=83)
    g, be = rnd(2 * Cout, seed=84) + 1, rnd(2 * Cout, seed=85)
    other = rnd(B, Cout, H, W, seed=86)
    op = OneOp(dt, B)
    y = op.b.conv_first(x.to(dev()), w.to(dev()), b.to(dev()), Cout, want_stats=True)
    has = y.data_ptr() in op.b.stats
    assert has == (dt != "fp32")                                        # f32 keeps the direct kernel (no statistics)
    n1 = op.b.groupnorm(y, g[:Cout].to(dev()), be[:Cout].to(dev()), 32, act="silu")
    o = op.b.conv(nhwc(other, dt), pack_conv_weight(torch.eye(Cout).reshape(Cout, Cout, 1, 1).to(dev()), dtype_code(dt)),
                  Cout, ksize=1, pad=0, want_stats=True)                # identity 1x1: a conv-produced tensor with statistics
    n2 = op.b.groupnorm(o, g.to(dev()), be.to(dev()), 32, x2=y)         # norm of cat([o, y])
    op.go()
    ys = nchw(y)                                                        # stored (rounded) conv output
    err = (ys - F.conv2d(x, w, b, padding=1)).abs().max().item()
    assert err < tol(dt, 1e-5, 3e-2), err
    ref1 = F.silu(F.group_norm(ys, 32, g[:Cout], be[:Cout], eps=1e-5))
    assert (nchw(n1) - ref1).abs().max().item() < tol(dt, 2e-5, 4e-2)
    ref2 = F.group_norm(torch.cat([nchw(o), ys], 1), 32, g, be, eps=1e-5)
    assert (nchw(n2) - ref2).abs().max().item() < tol(dt, 2e-5, 4e-2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 64, 16, 16, 3, 3), (1, 128, 9, 13, 3, 3), (2, 64, 8, 8, 3, 1), (1, 32, 8, 8, 1, 3),
                                   (1, 256, 8, 24, 3, 3), (2, 128, 5, 7, 2, 3), (1, 64, 6, 6, 4, 3), (2, 128, 70, 40, 3, 3),
                                   (1, 64, 33, 18, 3, 3), (1, 128, 16, 16, 3, 1)])
def test_conv_last(shape, dt):
    B, Cin, H, W, Cout, k = shape
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, k, k, seed=2, scale=0.1), rnd(Cout, seed=3)
    xr = lp_round(dt, x) if dt != "fp32" else x
    # 16-bit modes with <= 3 outputs round the weights to the storage type like every other conv (dot2 kernel)
    wr = lp_round(dt, w) if (dt != "fp32" and Cout <= 3) else w
    ref = F.conv2d(xr, wr, b, padding=k // 2)
    op = OneOp(dt, B)
    out = torch.empty(B, Cout, H, W, device=dev())
    op.b.conv_last(nhwc(x, dt), w.to(dev()), b.to(dev()), Cout, k, out)
    op.go()
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-5, err


# ------------------------------------------------------------------------------ groupnorm
GN_CASES = [
    # B, H, W, C1, C2, G, act, residual
    (2, 8, 8, 64, 0, 32, "silu", 0),
    (1, 32, 32, 128, 0, 32, "silu", 0),
    (2, 16, 16, 128, 64, 32, "silu", 0),      # 192 ch: groups of 6 straddle the two sources
    (3, 16, 16, 256, 128, 32, None, 0),
    (2, 8, 8, 64, 0, 1, "gelu", 0),           # GroupNorm(1, C) of DoubleConv
    (1, 16, 16, 128, 0, 1, "silu", 1),        # residual DoubleConv: act(x + GN(.))
    (1, 64, 64, 64, 0, 32, None, 0),
    (1, 8, 8, 512, 0, 32, "silu", 0),
]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", GN_CASES)
def test_groupnorm(case, dt):
    B, H, W, C1, C2, G, act, has_r = case
    C = C1 + C2
    x1, x2 = rnd(B, C1, H, W, seed=1) * 2 + 0.5, (rnd(B, C2, H, W, seed=2) - 1.0) if C2 else None
    gamma, beta = rnd(C, seed=3) * 0.5 + 1, rnd(C, seed=4) * 0.2
    res = rnd(B, C, H, W, seed=5) if has_r else None
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    if dt != "fp32":
        xin = lp_round(dt, xin)
    ref = F.group_norm(xin, G, gamma, beta, eps=1e-5)
    if has_r:
        ref = ref + (lp_round(dt, res) if dt != "fp32" else res)
    ref = {"silu": F.silu, "gelu": F.gelu, None: lambda v: v}[act](ref)
    op = OneOp(dt, B)
    y = op.b.groupnorm(nhwc(x1, dt), gamma.to(dev()), beta.to(dev()), G, act=act,
                       x2=nhwc(x2, dt) if C2 else None, residual=nhwc(res, dt) if has_r else None)
    op.go()
    err = (nchw(y) - ref).abs().max().item()
    assert err < tol(dt, 2e-5, 3e-2), err


def test_groupnorm_is_batch_invariant():
    """A sample's result must not depend on its batch (shard equality across GPUs)."""
    x = rnd(4, 128, 16, 16, seed=7)
    gamma, beta = torch.ones(128), torch.zeros(128)
    outs = []
    for sl in (slice(0, 4), slice(2, 3)):
        op = OneOp("fp32", x[sl].shape[0])
        y = op.b.groupnorm(nhwc(x[sl], "fp32"), gamma.to(dev()), beta.to(dev()), 32, act="silu")
        op.go()
        outs.append(nchw(y))
    assert torch.equal(outs[0][2:3], outs[1])


# ------------------------------------------------------------------------------ attention
ATT_CASES = [
    # B, N, heads, d, layout
    (2, 64, 4, 64, "b"), (1, 256, 4, 32, "b"), (1, 1024, 4, 64, "b"), (2, 64, 2, 96, "b"),
    (1, 256, 2, 128, "b"), (1, 1024, 2, 32, "b"), (2, 256, 4, 16, "a"), (1, 64, 4, 64, "a"),
    # head widths that are not multiples of 32: zero-padded K chunk + ones row (24), ones row in the SECOND V^T tile (40, 48, 56)
    (1, 256, 2, 24, "b"), (2, 196, 3, 48, "a"), (1, 144, 2, 40, "b"), (1, 324, 1, 56, "a"),
]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", ATT_CASES)
def test_attention(case, dt):
    B, N, heads, d, lay = case
    C = heads * d
    qkv = rnd(B, N, 3 * C, seed=11)
    if dt != "fp32":
        qkv = lp_round(dt, qkv)
    if lay == "b":      # per head [q|k|v] interleave (diff_model.py:120)
        t = qkv.view(B, N, heads, 3, d)
        q, k, v = t[:, :, :, 0], t[:, :, :, 1], t[:, :, :, 2]
        offs = (0, d, 2 * d, 3 * d)
    else:               # in_proj layout: [Q(C) | K(C) | V(C)], head h at h*d
        t = qkv.view(B, N, 3, heads, d)
        q, k, v = t[:, :, 0], t[:, :, 1], t[:, :, 2]
        offs = (0, C, 2 * C, d)
    q, k, v = (u.permute(0, 2, 1, 3) for u in (q, k, v))          # [B, heads, N, d]
    w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1)
    ref = (w @ v).permute(0, 2, 1, 3).reshape(B, N, C)
    op = OneOp(dt, B)
    side = int(math.isqrt(N))
    x = qkv.view(B, side, side, 3 * C).to(dev(), tdt(dt))
    y = op.b.attention(x, heads, d, *offs)
    op.go()
    got = y.float().cpu().view(B, N, C)
    err = (got - ref).abs().max().item()
    assert err < tol(dt, 2e-5, 3e-2), err


# ------------------------------------------------------------------------------ small ops
def test_linear_and_embedding():
    lib = _lib.load()
    _lib.init_device()
    s = torch.cuda.current_stream().cuda_stream
    x, w, b = rnd(3, 128, seed=1), rnd(512, 128, seed=2, scale=0.1), rnd(512, seed=3)
    y = torch.empty(3, 512, device=dev())
    xd, wd, bd = x.to(dev()), w.to(dev()), b.to(dev())
    _lib.check(lib.advs_linear_f32(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), 3, 128, 512, 2, 2, s))
    ref = F.silu(F.linear(F.silu(x), w, b))
    assert (y.cpu() - ref).abs().max().item() < 1e-5
    # sinusoidal embeddings, both orders, with a label table
    t = torch.tensor([1, 501, 981], dtype=torch.int64)
    half = 64
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    table, labels = rnd(37, 128, seed=4), torch.tensor([0, 36, 17])
    out = torch.empty(3, 128, device=dev())
    td, fd, tabd, lbd = t.to(dev()), freqs.to(dev()), table.to(dev()), labels.to(dev())
    _lib.check(lib.advs_timestep_embedding(td.data_ptr(), fd.data_ptr(), half, 1, 0, 0, out.data_ptr(), 3, s))
    args = t[:, None].float() * freqs[None]
    assert (out.cpu() - torch.cat([args.cos(), args.sin()], -1)).abs().max().item() < 2e-6
    _lib.check(lib.advs_timestep_embedding(td.data_ptr(), fd.data_ptr(), half, 0, tabd.data_ptr(), lbd.data_ptr(),
                                           out.data_ptr(), 3, s))
    assert (out.cpu() - (torch.cat([args.sin(), args.cos()], -1) + table[labels])).abs().max().item() < 2e-6


@pytest.mark.parametrize("cfg", [None, 3.0, 0.3])
def test_ddim_step_bit_exact(cfg):
    """The fused update must round exactly like the reference's chain of torch ops."""
    lib = _lib.load()
    _lib.init_device()
    s = torch.cuda.current_stream().cuda_stream
    B, per = 2, 3 * 16 * 16
    x, eps, eu, nz = rnd(B, per, seed=1), rnd(B, per, seed=2), rnd(B, per, seed=3), rnd(B, per, seed=4)
    a_t, a_p = torch.tensor(0.3712, dtype=torch.float32), torch.tensor(0.5521, dtype=torch.float32)
    sig = 0.7 * torch.sqrt((1 - a_p) / (1 - a_t) * (1 - a_t / a_p))
    coef = torch.stack([torch.tensor([0.9, 0.95, 0.0]), torch.stack([a_t, a_p, sig])]).float()
    tseq = torch.tensor([981, 961], dtype=torch.int64)
    e = eps if cfg is None else torch.lerp(eu, eps, cfg)
    x0 = torch.clamp((x - torch.sqrt(1.0 - a_t) * e) / torch.sqrt(a_t), -1.0, 1.0)
    ref = torch.sqrt(a_p) * x0 + torch.sqrt(1 - a_p - sig ** 2) * e + sig * nz
    xd, ed, ud, nd = x.to(dev()), eps.to(dev()), eu.to(dev()), nz.to(dev())
    cd, td = coef.to(dev()), tseq.to(dev())
    counter = torch.ones(1, dtype=torch.int32, device=dev())       # row 1
    tout = torch.zeros(B, dtype=torch.int64, device=dev())
    _lib.check(lib.advs_ddim_step(xd.data_ptr(), ed.data_ptr(), ud.data_ptr() if cfg is not None else 0,
                                  float(cfg or 0.0), nd.data_ptr(), cd.data_ptr(), td.data_ptr(), 2,
                                  counter.data_ptr(), tout.data_ptr(), B, per, 1, s))
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), ref)
    assert counter.item() == 2 and tout.tolist() == [961, 961]


def test_to_uint8_wraps_like_torch_cpu():
    lib = _lib.load()
    _lib.init_device()
    x = torch.tensor([-1.02, -1.0, -0.999, 0.0, 0.5, 0.999, 1.0, 1.004, 1.03, -1.3], dtype=torch.float32)
    ref = (((x + 1) * 0.5) * 255).type(torch.uint8)
    xd = x.to(dev())
    y = torch.empty(x.numel(), dtype=torch.uint8, device=dev())
    _lib.check(lib.advs_to_uint8(xd.data_ptr(), y.data_ptr(), x.numel(), 0, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(y.cpu(), ref)
    _lib.check(lib.advs_to_uint8(xd.data_ptr(), y.data_ptr(), x.numel(), 1, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(y.cpu(), (((x + 1) * 0.5) * 255).clamp(0, 255).type(torch.uint8))


@pytest.mark.parametrize("dt", DTS)
def test_layout_roundtrip(dt):
    lib = _lib.load()
    _lib.init_device()
    s = torch.cuda.current_stream().cuda_stream
    x = rnd(2, 37, 9, 11, seed=3)
    xd = x.to(dev())
    y = torch.empty(2, 9, 11, 37, dtype=tdt(dt), device=dev())
    z = torch.empty_like(xd)
    code = dtype_code(dt)
    _lib.check(lib.advs_nchw_f32_to_nhwc(xd.data_ptr(), y.data_ptr(), 2, 37, 9, 11, code, s))
    _lib.check(lib.advs_nhwc_to_nchw_f32(y.data_ptr(), z.data_ptr(), 2, 37, 9, 11, code, s))
    ref = lp_round(dt, x) if dt != "fp32" else x
    assert torch.equal(y.float().cpu(), ref.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(z.cpu(), ref)


# ------------------------------------------------------------------------------ lineage-A resampling ops
@pytest.mark.parametrize("dt", DTS)
def test_maxpool2(dt):
    x = rnd(2, 64, 12, 16, seed=21)
    xr = lp_round(dt, x) if dt != "fp32" else x
    op = OneOp(dt, 2)
    y = op.b.maxpool2(nhwc(x, dt))
    op.go()
    assert torch.equal(nchw(y), F.max_pool2d(xr, 2))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 8, 8, 64, 128), (1, 4, 6, 128, 64), (1, 1, 1, 64, 64)])
def test_concat_upsample2x(shape, dt):
    B, h, w, C1, C2 = shape
    skip, x = rnd(B, C1, 2 * h, 2 * w, seed=22), rnd(B, C2, h, w, seed=23)
    if dt != "fp32":
        skip, x = lp_round(dt, skip), lp_round(dt, x)
    ref = torch.cat([skip, F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)], 1)
    op = OneOp(dt, B)
    y = op.b.concat_upsample2x(nhwc(skip, dt), nhwc(x, dt))
    op.go()
    err = (nchw(y) - ref).abs().max().item()
    assert err < tol(dt, 2e-6, 2e-2), err


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (1, 8, 8, 256), (3, 5, 7, 128)])
def test_layernorm(shape, dt):
    B, H, W, Cc = shape
    x = rnd(B, H, W, Cc, seed=24) * 3 + 1
    g, b = rnd(Cc, seed=25) + 1, rnd(Cc, seed=26)
    xr = lp_round(dt, x) if dt != "fp32" else x
    ref = F.layer_norm(xr, (Cc,), g, b, eps=1e-5)
    op = OneOp(dt, B)
    y = op.b.layernorm(x.to(dev(), tdt(dt)), g.to(dev()), b.to(dev()))
    op.go()
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < tol(dt, 1e-5, 4e-2), err


@pytest.mark.parametrize("dt", DTS)
def test_groupnorm_chan_add(dt):
    B, C, H, W = 2, 128, 8, 8
    x, emb = rnd(B, C, H, W, seed=27), rnd(B, C + 64, seed=28)
    g, b = rnd(C, seed=29) + 1, rnd(C, seed=30)
    xr = lp_round(dt, x) if dt != "fp32" else x
    ref = F.group_norm(xr, 1, g, b, eps=1e-5) + emb[:, 32:32 + C, None, None]
    op = OneOp(dt, B)
    e = emb.to(dev())
    y = op.b.groupnorm(nhwc(x, dt), g.to(dev()), b.to(dev()), 1, chan_add=e[:, 32:32 + C], chan_add_stride=C + 64)
    op.go()
    err = (nchw(y) - ref).abs().max().item()
    assert err < tol(dt, 2e-5, 3e-2), err


# ------------------------------------------------------------------------------ GN statistics from the conv epilogue
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tiles", [(1, 1), (4, 1), (1, 4), (3, 5), (10, 1), (17, 1), (18, 1), (19, 1)])
def test_groupnorm_with_epilogue_stats(dt, tiles):
    """conv(+stats) x2 -> GroupNorm(32) over their concat (groups of 12 straddle the sources) must equal
    the unfused path; statistics come from [row block][channel] partials written by the epilogues."""
    if tiles[0] in HALO2_TILES and dt == "fp32":
        pytest.skip("second-generation halo kernels: 16-bit only")
    B, H, W = 2, 16, 32 if tiles[0] == 17 else 16
    xa, xb = rnd(B, 64, H, W, seed=41), rnd(B, 64, H, W, seed=42)
    wa, wb = rnd(256, 64, 3, 3, seed=43, scale=0.05), rnd(128, 64, 1, 1, seed=44, scale=0.2)
    gamma, beta = rnd(384, seed=45) * 0.3 + 1, rnd(384, seed=46) * 0.1
    r = (lambda t: lp_round(dt, t)) if dt != "fp32" else (lambda t: t)
    ya = r(F.conv2d(r(xa), r(wa), padding=1))
    yb = r(F.relu(F.conv2d(r(xb), r(wb))))
    ref = F.silu(F.group_norm(torch.cat([ya, yb], 1), 32, gamma, beta, eps=1e-5))
    op = OneOp(dt, B)
    code = dtype_code(dt)
    ca = op.b.conv(nhwc(xa, dt), pack_conv_weight(wa.to(dev()), code), 256, tile=tiles[0], want_stats=True)
    cb = op.b.conv(nhwc(xb, dt), pack_conv_weight(wb.to(dev()), code), 128, ksize=1, pad=0, act="relu", tile=tiles[1],
                   want_stats=True)
    assert ca.data_ptr() in op.b.stats and cb.data_ptr() in op.b.stats
    y = op.b.groupnorm(ca, gamma.to(dev()), beta.to(dev()), 32, act="silu", x2=cb)
    assert op.b.plan.ops[-1][0].__name__ == "advs_groupnorm_stats"
    op.go()
    err = (nchw(y) - ref).abs().max().item()
    assert err < tol(dt, 3e-5, 4e-2), err


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile", [0, 1, 4])
@pytest.mark.parametrize("two", [False, True])
def test_conv2d_with_extra_1x1_operand(dt, tile, two):
    """y = conv3x3(a) + conv1x1(cat[e1, e2]) + bias in one launch (residual block with a shortcut conv)."""
    B, H, W, Ca, Cout, E1, E2 = 2, 12, 12, 128, 128, 64, (128 if two else 0)
    a, e1 = rnd(B, Ca, H, W, seed=51), rnd(B, E1, H, W, seed=52)
    e2 = rnd(B, E2, H, W, seed=53) if two else None
    w3 = rnd(Cout, Ca, 3, 3, seed=54, scale=1 / math.sqrt(9 * Ca))
    w1 = rnd(Cout, E1 + E2, 1, 1, seed=55, scale=1 / math.sqrt(E1 + E2))
    bias = rnd(Cout, seed=56)
    r = (lambda t: lp_round(dt, t)) if dt != "fp32" else (lambda t: t)
    e = e1 if e2 is None else torch.cat([e1, e2], 1)
    ref = F.conv2d(r(a), r(w3), bias, padding=1) + F.conv2d(r(e), r(w1))
    code = dtype_code(dt)
    wcat = torch.cat([pack_conv_weight(w3.to(dev()), code).reshape(Cout, -1),
                      pack_conv_weight(w1.to(dev()), code).reshape(Cout, -1)], 1).contiguous()
    op = OneOp(dt, B)
    y = op.b.conv(nhwc(a, dt), wcat, Cout, bias=bias.to(dev()), extra=(nhwc(e1, dt), nhwc(e2, dt) if two else None), tile=tile)
    op.go()
    err = (nchw(y) - ref).abs().max().item()
    assert err < tol(dt, 3e-5, 5e-2), err


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [
    # B, H, W, Ca (one or two sources), Cout, E1, E2, residual-free full epilogue with temb + stats
    (2, 16, 16, (128, 0), 128, 64, 0), (1, 32, 16, (128, 0), 128, 256, 128), (2, 16, 16, (64, 64), 256, 128, 128),
    (1, 16, 48, (64, 0), 64, 64, 64), (3, 16, 16, (128, 0), 128, 192, 0),
    (1, 16, 32, (64, 64), 128, 128, 64), (2, 32, 64, (128, 0), 256, 64, 0),      # widths the 16 x 32-pixel tile takes
])
@pytest.mark.parametrize("tile", [10, 17, 18, 19])
def test_conv2d_extra_operand_on_the_halo_kernel(case, dt, tile):
    """The fused shortcut operand as one-tap units behind the 3x3 slabs of the halo kernels (tile 10 -> 13, and tiles 17-19): one
    to six extra slabs, e1 only and e1 + e2, two 3x3 sources, temb and epilogue statistics."""
    B, H, W, (C1, C2), Cout, E1, E2 = case
    if tile in HALO2_TILES and not halo2_ok(tile, dt, H, W):
        pytest.skip("second-generation halo kernels: 16-bit, W a multiple of 32 for tile 17")
    x1 = rnd(B, C1, H, W, seed=111)
    x2 = rnd(B, C2, H, W, seed=112) if C2 else None
    e1 = rnd(B, E1, H, W, seed=113)
    e2 = rnd(B, E2, H, W, seed=114) if E2 else None
    w3 = rnd(Cout, C1 + C2, 3, 3, seed=115, scale=1 / math.sqrt(9 * (C1 + C2)))
    w1 = rnd(Cout, E1 + E2, 1, 1, seed=116, scale=1 / math.sqrt(E1 + E2))
    bias, temb = rnd(Cout, seed=117), rnd(B, Cout, seed=118)
    r = lambda t: lp_round(dt, t)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    e = e1 if e2 is None else torch.cat([e1, e2], 1)
    ref = F.conv2d(r(xin), r(w3), bias, padding=1) + F.conv2d(r(e), r(w1)) + temb[:, :, None, None]
    code = dtype_code(dt)
    wcat = torch.cat([pack_conv_weight(w3.to(dev()), code).reshape(Cout, -1),
                      pack_conv_weight(w1.to(dev()), code).reshape(Cout, -1)], 1).contiguous()
    op = OneOp(dt, B)
    y = op.b.conv(nhwc(x1, dt), wcat, Cout, x2=nhwc(x2, dt) if C2 else None, bias=bias.to(dev()), temb=temb.to(dev()),
                  temb_stride=Cout, extra=(nhwc(e1, dt), nhwc(e2, dt) if E2 else None), tile=tile, want_stats=True)
    assert y.data_ptr() in op.b.stats
    g, be = rnd(Cout, seed=119) + 1, rnd(Cout, seed=120)
    n = op.b.groupnorm(y, g.to(dev()), be.to(dev()), 32, act="silu")
    op.go()
    got = nchw(y)
    err = (got - ref).abs().max().item()
    assert err < tol(dt, 3e-5, 5e-2), err
    refn = F.silu(F.group_norm(got, 32, g, be, eps=1e-5))
    rel = ((nchw(n) - refn).abs() / refn.abs().clamp(min=1.0)).max().item()
    assert rel < tol(dt, 2e-5, 1e-2), rel


# ------------------------------------------------------------------------------ GroupNorm + SiLU inside the conv (advs_conv_args.norm)
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [
    # B, H, W, (C1, C2), Cout, temb, residual, (E1, E2)
    (2, 32, 32, (128, 0), 128, 1, 0, (0, 0)),          # ResidualBlock.conv1 at level 0: fast epilogue
    (1, 32, 48, (128, 256), 128, 1, 0, (0, 0)),        # up path: norm over the concat [h, skip], 12 units
    (2, 16, 32, (128, 0), 128, 0, 1, (0, 0)),          # conv2 with the identity shortcut: generic epilogue
    (1, 32, 32, (128, 0), 128, 0, 0, (128, 256)),      # conv2 with the fused 1x1 shortcut: extra units stay un-normalised
    (3, 16, 16, (64, 0), 64, 1, 0, (0, 0)),            # two units, N tail, image edges everywhere
    (1, 16, 48, (192, 192), 96, 1, 0, (0, 0)),         # the widest table the kernel takes (384 channels), 12 channels a group, N tail
    (3, 32, 16, (256, 0), 40, 0, 1, (0, 0)),           # residual epilogue with a 40-channel tail
    (2, 16, 16, (64, 64), 128, 1, 0, (64, 0)),         # two normalised sources and ONE un-normalised extra source
])
def test_conv2d_norm_on_load(case, dt):
    """conv3x3(SiLU(GroupNorm32(x))) with the norm applied while the halo is staged must equal, BIT FOR BIT, the two-pass form
    (advs_groupnorm_stats, then advs_conv2d on its output): same statistics, same coefficient arithmetic, same rounding of the
    normalised activation.  Also against the torch oracle (norm_layer + SiLU + Conv2d, diff_model.py:70-73)."""
    B, H, W, (C1, C2), Cout, has_t, has_r, (E1, E2) = case
    code = dtype_code(dt)
    r = lambda t: lp_round(dt, t)
    src = [rnd(B, c, H, W, seed=131 + i) * (1.0 + 0.5 * i) + 0.3 * i for i, c in enumerate((C1, C2)) if c]
    prods = []
    op = OneOp(dt, B)
    for i, xs in enumerate(src):           # each source is a conv output, so that its statistics come from an epilogue
        c = xs.shape[1]
        wi = rnd(c, 64, 1, 1, seed=141 + i, scale=0.2)
        xi = rnd(B, 64, H, W, seed=151 + i)
        prods.append(op.b.conv(nhwc(xi, dt), pack_conv_weight(wi.to(dev()), code), c, ksize=1, pad=0, want_stats=True))
    x1, x2 = prods[0], (prods[1] if C2 else None)
    C = C1 + C2
    gamma, beta = rnd(C, seed=161) * 0.3 + 1, rnd(C, seed=162) * 0.2
    w3 = rnd(Cout, C, 3, 3, seed=163, scale=1 / math.sqrt(9 * C))
    bias = rnd(Cout, seed=164)
    temb = rnd(B, Cout, seed=165) if has_t else None
    res = rnd(B, Cout, H, W, seed=166) if has_r else None
    e = [rnd(B, c, H, W, seed=171 + i) for i, c in enumerate((E1, E2)) if c]
    wp = pack_conv_weight(w3.to(dev()), code)
    if e:
        w1 = rnd(Cout, E1 + E2, 1, 1, seed=173, scale=1 / math.sqrt(E1 + E2))
        wp = torch.cat([wp.reshape(Cout, -1), pack_conv_weight(w1.to(dev()), code).reshape(Cout, -1)], 1).contiguous()
    kw = dict(bias=bias.to(dev()), temb=temb.to(dev()) if has_t else None, temb_stride=Cout if has_t else 0,
              residual=nhwc(res, dt) if has_r else None,
              extra=(nhwc(e[0], dt), nhwc(e[1], dt) if len(e) > 1 else None) if e else None)
    g, be = gamma.to(dev()), beta.to(dev())
    two = op.b.conv(op.b.groupnorm(x1, g, be, 32, act="silu", x2=x2), wp, Cout, tile=19, want_stats=True, **kw)
    tab = op.b.groupnorm_affine(x1, g, be, 32, x2=x2)
    one = op.b.conv(x1, wp, Cout, x2=x2, norm=tab, want_stats=True, **kw)
    assert op.b.lib.advs_conv_resolve_tile(op.b.plan.ops[-1][1][0]) == 19
    op.go()
    assert torch.equal(one.view(torch.int16), two.view(torch.int16)), (one.float() - two.float()).abs().max().item()
    s1, s2 = op.b.stats[one.data_ptr()][0], op.b.stats[two.data_ptr()][0]
    assert torch.equal(s1, s2)
    # and the torch restatement on the stored producers
    xin = torch.cat([nchw(t) for t in (x1, x2) if t is not None], 1)
    a = r(F.silu(F.group_norm(xin, 32, gamma, beta, eps=1e-5)))
    ref = F.conv2d(a, r(w3), bias, padding=1)
    if e:
        ref = ref + F.conv2d(r(torch.cat(e, 1)), r(w1))
    if has_t:
        ref = ref + temb[:, :, None, None]
    if has_r:
        ref = ref + r(res)
    err = (nchw(one) - ref).abs().max().item()
    assert err < 6e-2, err


def test_conv2d_norm_on_load_rejects_what_it_cannot_take():
    op = OneOp("fp32", 1)
    x = torch.zeros(1, 16, 16, 32, device=dev())
    w = pack_conv_weight(torch.zeros(32, 32, 3, 3, device=dev()), dtype_code("fp32"))
    tab = torch.zeros(1, 32, 2, device=dev())
    op.b.conv(x, w, 32, norm=tab)
    fn, args = op.b.plan.ops[-1]
    assert fn(*args, op.stream.cuda_stream) != 0 and "norm" in op.b.lib.advs_last_error().decode()


# ------------------------------------------------------------------------------ CSPDarkUnet additions
@pytest.mark.parametrize("tile", [0, 1, 4, 10])
@pytest.mark.parametrize("case", [
    # B, H, W, C1, C2, Cout, k, stride : 32-channel bf16 sources = half a 128-byte slab (ld < c)
    (2, 16, 16, 32, 0, 32, 3, 1), (2, 16, 16, 32, 0, 64, 3, 2), (1, 8, 8, 32, 0, 96, 1, 1),
    (2, 8, 8, 32, 32, 64, 1, 1), (1, 16, 32, 96, 32, 64, 3, 1), (1, 5, 7, 32, 0, 32, 1, 1),
])
def test_conv2d_half_slab_sources_bf16(case, tile):
    B, H, W, C1, C2, Cout, k, stride = case
    if tile == 10 and not (k == 3 and stride == 1 and H % 16 == 0 and W % 16 == 0):
        pytest.skip("halo kernel: 3x3 stride 1, H and W multiples of 16")
    pad = 1 if k == 3 else 0
    x1 = bf16_round(rnd(B, C1, H, W, seed=61))
    x2 = bf16_round(rnd(B, C2, H, W, seed=62)) if C2 else None
    w = rnd(Cout, C1 + C2, k, k, seed=63, scale=1.0 / math.sqrt((C1 + C2) * k * k))
    ref = F.conv2d(x1 if x2 is None else torch.cat([x1, x2], 1), bf16_round(w), None, stride=stride, padding=pad)
    op = OneOp("bf16", B)
    wp = pack_conv_weight(w.to(dev()), dtype_code("bf16"), sources=(C1, C2) if C2 else None)
    y = op.b.conv(nhwc(x1, "bf16"), wp, Cout, x2=nhwc(x2, "bf16") if C2 else None, ksize=k, stride=stride, pad=pad, tile=tile)
    op.go()
    err = (nchw(y) - ref).abs().max().item()
    assert err < 4e-2, err


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [(2, 16, 4, 8, "a"), (1, 4, 4, 16, "a"), (1, 100, 2, 8, "b"), (2, 4096, 4, 8, "a"), (1, 16, 4, 128, "a")])
def test_attention_short_sequences_and_narrow_heads(case, dt):
    """CSPDarkUnet: sa4 has 16 tokens at 64x64 input, sa8 has 8-channel heads (cspdarkunet.py:50-78)."""
    B, N, heads, d, lay = case
    C = heads * d
    qkv = rnd(B, N, 3 * C, seed=64)
    if dt != "fp32":
        qkv = lp_round(dt, qkv)
    if lay == "b":
        t = qkv.view(B, N, heads, 3, d)
        q, k, v = t[:, :, :, 0], t[:, :, :, 1], t[:, :, :, 2]
        offs = (0, d, 2 * d, 3 * d)
    else:
        t = qkv.view(B, N, 3, heads, d)
        q, k, v = t[:, :, 0], t[:, :, 1], t[:, :, 2]
        offs = (0, C, 2 * C, d)
    q, k, v = (u.permute(0, 2, 1, 3) for u in (q, k, v))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, N, C)
    op = OneOp(dt, B)
    x = qkv.view(B, 1, N, 3 * C).to(dev(), tdt(dt))
    y = op.b.attention(x, heads, d, *offs)
    op.go()
    err = (y.float().cpu().view(B, N, C) - ref).abs().max().item()
    assert err < tol(dt, 2e-5, 3e-2), err


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 8, 8, 32, 32), (1, 4, 6, 128, 64), (1, 1, 1, 64, 64)])
def test_concat_nearest2x(shape, dt):
    B, h, w, C1, C2 = shape
    skip, x = rnd(B, C1, 2 * h, 2 * w, seed=65), rnd(B, C2, h, w, seed=66)
    if dt != "fp32":
        skip, x = lp_round(dt, skip), lp_round(dt, x)
    ref = torch.cat([skip, F.interpolate(x, scale_factor=2, mode="nearest")], 1)
    op = OneOp(dt, B)
    y = op.b.concat_nearest2x(nhwc(skip, dt), nhwc(x, dt))
    op.go()
    assert torch.equal(nchw(y), ref)


@pytest.mark.parametrize("dt", DTS)
def test_groupnorm_residual_after_activation(dt):
    """Bottleneck: act(gn(conv)) + x (module.py:42-47), with the block's emb add in the same pass."""
    B, C, H, W = 2, 64, 8, 8
    x, r, emb = rnd(B, C, H, W, seed=67), rnd(B, C, H, W, seed=68), rnd(B, C, seed=69)
    g, b = rnd(C, seed=70) + 1, rnd(C, seed=71)
    xr, rr = (lp_round(dt, x), lp_round(dt, r)) if dt != "fp32" else (x, r)
    ref = F.silu(F.group_norm(xr, 1, g, b, eps=1e-5)) + emb[:, :, None, None] + rr
    op = OneOp(dt, B)
    e = emb.to(dev())
    y = op.b.groupnorm(nhwc(x, dt), g.to(dev()), b.to(dev()), 1, act="silu", residual=nhwc(r, dt),
                       residual_after_act=True, chan_add=e, chan_add_stride=C)
    op.go()
    err = (nchw(y) - ref).abs().max().item()
    assert err < tol(dt, 2e-5, 6e-2), err


@pytest.mark.parametrize("dt", DTS)
def test_groupnorm_with_epilogue_stats_narrow_channels(dt):
    """32- and 64-channel producers (CSPDarkUnet's stem level): the fold kernel packs several row lanes per pass."""
    for C, groups in ((32, 1), (64, 32), (96, 1)):
        B, H, W = 2, 16, 16
        x, w = rnd(B, 64, H, W, seed=91), rnd(C, 64, 1, 1, seed=92, scale=0.2)
        g, be = rnd(C, seed=93) + 1, rnd(C, seed=94)
        op = OneOp(dt, B)
        y = op.b.conv(nhwc(x, dt), pack_conv_weight(w.to(dev()), dtype_code(dt)), C, ksize=1, pad=0, want_stats=True)
        assert y.data_ptr() in op.b.stats
        n = op.b.groupnorm(y, g.to(dev()), be.to(dev()), groups, act="silu")
        op.go()
        ref = F.silu(F.group_norm(nchw(y), groups, g, be, eps=1e-5))
        assert (nchw(n) - ref).abs().max().item() < tol(dt, 2e-5, 4e-2), (C, groups)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [
    # B, H, W, C1, C2, Cout, bias, temb, res, act, stats
    (1, 16, 16, 64, 0, 128, 1, 0, 0, None, 0), (2, 32, 16, 128, 0, 256, 1, 1, 1, "silu", 1), (1, 16, 48, 64, 64, 64, 0, 0, 0, None, 1),
    (3, 16, 16, 256, 0, 256, 1, 0, 0, None, 0),
    (1, 16, 32, 64, 0, 128, 1, 1, 0, None, 1), (2, 16, 64, 128, 64, 256, 1, 0, 1, None, 0),      # widths the 16 x 32-pixel tile takes
])
@pytest.mark.parametrize("tile", [0, 17, 18, 19])
def test_conv2d_upsample_subpixel(case, dt, tile):
    """nearest x2 + 3x3 (Upsample, diff_model.py:129-140) as four 2x2 convolutions of the low-res input."""
    B, H, W, C1, C2, Cout, has_b, has_t, has_r, act, stats = case
    if tile in HALO2_TILES and not halo2_ok(tile, dt, H, W):
        pytest.skip("second-generation halo kernels: 16-bit, W a multiple of 32 for tile 17")
    x1 = rnd(B, C1, H, W, seed=101)
    x2 = rnd(B, C2, H, W, seed=102) if C2 else None
    w = rnd(Cout, C1 + C2, 3, 3, seed=103, scale=1.0 / math.sqrt((C1 + C2) * 9))
    bias = rnd(Cout, seed=104) if has_b else None
    temb = rnd(B, Cout, seed=105) if has_t else None
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    xin = lp_round(dt, xin)
    ref = F.conv2d(F.interpolate(xin, scale_factor=2, mode="nearest"), lp_round(dt, w), bias, padding=1)
    res = rnd(*ref.shape, seed=106) if has_r else None
    if has_t:
        ref = ref + temb[:, :, None, None]
    if has_r:
        ref = ref + lp_round(dt, res)
    if act == "silu":
        ref = F.silu(ref)
    op = OneOp(dt, B)
    w4 = pack_subpixel_upsample_weight(w.to(dev()), dtype_code(dt))
    assert w4.shape == (4, Cout, 2, 2, C1 + C2)
    y = op.b.conv(nhwc(x1, dt), w4, Cout, x2=nhwc(x2, dt) if C2 else None, bias=bias.to(dev()) if has_b else None,
                  temb=temb.to(dev()) if has_t else None, temb_stride=Cout if has_t else 0,
                  residual=nhwc(res, dt) if has_r else None, upsample="subpixel", act=act, want_stats=bool(stats), tile=tile)
    n = None
    if stats:
        assert y.data_ptr() in op.b.stats
        g, be = rnd(Cout, seed=107) + 1, rnd(Cout, seed=108)
        n = op.b.groupnorm(y, g.to(dev()), be.to(dev()), 32, act="silu")
    op.go()
    got = nchw(y)
    assert got.shape == ref.shape
    # 16-bit modes: the summed taps are rounded once, the reference rounds each tap: allow that on top of the usual bound
    err = (got - ref).abs().max().item()
    assert err < tol(dt, 3e-5, 6e-2), err
    if stats:
        refn = F.silu(F.group_norm(got, 32, g, be, eps=1e-5))
        rel = ((nchw(n) - refn).abs() / refn.abs().clamp(min=1.0)).max().item()     # outputs reach |8|: relative to the value
        assert rel < tol(dt, 2e-5, 1e-2), rel


def test_unknown_dtype_code_is_rejected():
    """Every entry point that takes a dtype code refuses one it does not know (no silent f32 interpretation)."""
    lib = _lib.load()
    _lib.init_device()
    x = torch.zeros(1, 8, 8, 64, device=dev())
    y = torch.zeros_like(x)
    rc = lib.advs_maxpool2(x.data_ptr(), y.data_ptr(), 1, 8, 8, 64, 7, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and "unknown dtype" in lib.advs_last_error().decode()
    op = OneOp("fp32", 1)
    w = pack_conv_weight(torch.zeros(64, 64, 1, 1, device=dev()), dtype_code("fp32"))
    op.b.conv(x, w, 64, ksize=1, pad=0)
    fn, args = op.b.plan.ops[-1]
    args[0]._obj.dtype = 9
    assert fn(*args, op.stream.cuda_stream) != 0 and "unknown dtype" in lib.advs_last_error().decode()


# ------------------------------------------------------------------------------ second-generation halo tiles: seeded random shapes
def _halo2_random_cases(n=30):
    rs = np.random.RandomState(20251005)
    cases = []
    for _ in range(n):
        tile = int(rs.choice([17, 18, 19]))
        B = int(rs.randint(1, 4))
        H = 16 * int(rs.randint(1, 4))
        W = (32 if tile == 17 else 16) * int(rs.randint(1, 4))
        C1 = 64 * int(rs.randint(1, 4))
        C2 = 64 * int(rs.randint(0, 3))
        Cout = 8 * int(rs.choice([4, 8, 16, 20, 24, 32, 40]))
        kind = int(rs.randint(0, 4))                   # 0 plain fast epilogue, 1 residual (generic), 2 fused 1x1 operand, 3 sub-pixel upsample
        E = (64 * int(rs.randint(1, 3)), 64 * int(rs.randint(0, 2))) if kind == 2 else (0, 0)
        cases.append((tile, B, H, W, C1, C2, Cout, kind, E, bool(rs.randint(0, 2)), rs.choice(["bf16", "fp16"])))
    return cases


@pytest.mark.parametrize("case", _halo2_random_cases(), ids=lambda c: "t%d-%dx%dx%d-c%d+%d-o%d-k%d" % c[:8])
def test_conv2d_halo2_random_shapes(case):
    """Tiles 17-19 on seeded random shapes (batch 1-3, 1-3 tiles each way, one to five units per source, output-channel tails,
    all three kernel kinds, both epilogues) against the per-tap implicit-GEMM tile 1 -- another kernel, the same arithmetic in
    another order -- and the epilogue statistics against the sums of what was stored."""
    tile, B, H, W, C1, C2, Cout, kind, (E1, E2), has_t, dt = case
    code = dtype_code(dt)
    x1 = rnd(B, C1, H, W, seed=201)
    x2 = rnd(B, C2, H, W, seed=202) if C2 else None
    w = rnd(Cout, C1 + C2, 3, 3, seed=203, scale=1.0 / math.sqrt(9 * (C1 + C2)))
    bias = rnd(Cout, seed=204)
    ups = kind == 3
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    temb = rnd(B, Cout, seed=205) if has_t and kind != 1 else None
    res = rnd(B, Cout, Ho, Wo, seed=206) if kind == 1 else None
    e1 = rnd(B, E1, H, W, seed=207) if E1 else None
    e2 = rnd(B, E2, H, W, seed=208) if E2 else None

    def run(t):
        op = OneOp(dt, B)
        wp = pack_subpixel_upsample_weight(w.to(dev()), code) if ups else pack_conv_weight(w.to(dev()), code)
        kw = {}
        if E1:
            w1 = rnd(Cout, E1 + E2, 1, 1, seed=209, scale=1 / math.sqrt(E1 + E2))
            wp = torch.cat([wp.reshape(Cout, -1), pack_conv_weight(w1.to(dev()), code).reshape(Cout, -1)], 1).contiguous()
            kw["extra"] = (nhwc(e1, dt), nhwc(e2, dt) if E2 else None)
        y = op.b.conv(nhwc(x1, dt), wp, Cout, x2=nhwc(x2, dt) if C2 else None, bias=bias.to(dev()),
                      temb=temb.to(dev()) if temb is not None else None, temb_stride=Cout if temb is not None else 0,
                      residual=nhwc(res, dt) if res is not None else None, upsample="subpixel" if ups else False, tile=t,
                      want_stats=True, **kw)
        st = op.b.stats.get(y.data_ptr())
        op.go()
        return y.float().cpu(), (st[0].cpu() if st is not None else None)

    got, st = run(tile)
    ref, _ = run(0 if ups else 1)        # sub-pixel weights exist on the halo kernels only: the default rule (another tile) is the reference
    if ups and Ho * Wo >= 64 * 64 * 4:
        pytest.skip("the default rule picks the same family here")
    err = (got - ref).abs().max().item()
    assert err < 4e-2, err
    assert st is not None and st.shape[1] == Cout
    tot = st.view(B, -1, Cout, 2).sum(1).double()
    yy = got.double().view(B, -1, Cout)
    assert torch.allclose(tot[..., 0], yy.sum(1), rtol=1e-3, atol=5e-2) and torch.allclose(tot[..., 1], (yy * yy).sum(1), rtol=1e-3, atol=5e-2)
