"""Host-side plan builder for the HIP kernels.

A *plan* is the static list of C-ABI calls one network forward (or one sampler step) needs for a
fixed (batch, image size, dtype): buffers are carved from an arena once, every launch argument
is frozen, and the list is either replayed call by call or captured into a hipGraph and replayed
as one launch.  PyTorch supplies device memory (the arena's backing tensors) and the stream; all
arithmetic runs in ``libadvshadow_hip.so``.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT, BF16, F16, F32, GN_RESIDUAL_AFTER_ACT, ConvArgs, check

TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}
ESZ = {F32: 4, BF16: 2, F16: 2}
SLAB_ELEMS = {F32: 32, BF16: 64, F16: 64}       # conv K-slab: 128 bytes of channels


def dtype_code(name):
    if name in (F32, "fp32", "f32", "float32", torch.float32):
        return F32
    if name in (BF16, "bf16", "bfloat16", torch.bfloat16):
        return BF16
    if name in (F16, "fp16", "f16", "float16", "half", torch.float16):
        return F16
    raise ValueError(f"unsupported compute dtype {name!r} (use 'fp32', 'bf16' or 'fp16')")


def ptr(t):
    return 0 if t is None else t.data_ptr()


class Arena:
    """Best-fit pool of device buffers with explicit release (plans are static, single stream)."""

    def __init__(self, device):
        self.device = device
        self.free_list = []           # list of uint8 tensors
        self.total = 0

    def alloc(self, nbytes):
        nbytes = (int(nbytes) + 255) // 256 * 256
        best = None
        for i, t in enumerate(self.free_list):
            n = t.numel()
            if n >= nbytes and n <= 2 * nbytes + 4096 and (best is None or n < self.free_list[best].numel()):
                best = i
        if best is not None:
            return self.free_list.pop(best)
        self.total += nbytes
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def release(self, base):
        self.free_list.append(base)


class Plan:
    """Frozen launch list + optional hipGraph."""

    def __init__(self, stream):
        self.ops = []                 # (cfunc, args)
        self.keep = []                # objects that must outlive the plan (arg structs, tensors)
        self.stream = stream          # torch.cuda.Stream
        self.graph = None

    def add(self, fn, *args, keep=()):
        self.ops.append((fn, args))
        self.keep.extend(keep)

    def run_eager(self):
        s = self.stream.cuda_stream
        for fn, args in self.ops:
            rc = fn(*args, s)
            if rc != 0:
                check(rc, fn.__name__)

    def capture(self):
        lib = _lib.load()
        s = self.stream.cuda_stream
        check(lib.advs_graph_begin(s), "graph_begin")
        try:
            self.run_eager()
        finally:
            g = C.c_void_p()
            rc = lib.advs_graph_end(s, C.byref(g))
        check(rc, "graph_end")
        self.graph = g

    def run(self):
        if self.graph is not None:
            check(_lib.load().advs_graph_launch(self.graph, self.stream.cuda_stream), "graph_launch")
        else:
            self.run_eager()

    def __del__(self):
        try:
            if self.graph is not None:
                _lib.load().advs_graph_destroy(self.graph)
        except Exception:
            pass


class Builder:
    """Emits kernel calls into a Plan; owns the arena the activations live in."""

    def __init__(self, device, dtype, stream, batch):
        _lib.init_device()
        self.lib = _lib.load()
        self.device = device
        self.dt = dtype_code(dtype)
        self.tdt = TORCH_DT[self.dt]
        self.B = batch
        self.arena = Arena(device)
        self.plan = Plan(stream)
        self._base = {}
        self.stats = {}               # data_ptr of a conv output -> (stats tensor, row blocks per image)
        nb = self.lib.advs_groupnorm_scratch_bytes(batch, 64)
        self.gn_scratch = torch.empty(nb, dtype=torch.uint8, device=device)
        self.plan.keep.append(self.gn_scratch)   # the plan outlives the Builder: every launch refers to this scratch

    # ---- buffers
    def buf(self, shape, dtype=None):
        dtype = dtype or self.tdt
        n = 1
        for s in shape:
            n *= int(s)
        base = self.arena.alloc(n * torch.empty((), dtype=dtype).element_size())
        t = base.view(dtype)[:n].view(*shape)
        self._base[t.data_ptr()] = base
        self.plan.keep.append(base)
        return t

    def free(self, t):
        base = self._base.pop(t.data_ptr(), None)
        if base is not None:
            self.arena.release(base)
        st = self.stats.pop(t.data_ptr(), None)
        if st is not None:
            self.free(st[0])

    # ---- ops (activations are NHWC tensors [B,H,W,C] of the compute dtype)
    def conv(self, x1, w, cout, *, x2=None, bias=None, temb=None, temb_stride=0, residual=None,
             ksize=3, stride=1, pad=1, upsample=False, act=None, out=None, tile=0, want_stats=False, extra=None,
             residual_after_act=False, relu_mask=None, norm=None):
        """relu_mask: forward activation whose ReLU backward is fused into this (data-gradient) conv.
        norm: (scale, shift) table from ``groupnorm_affine`` -- the conv reads SiLU(GroupNorm(cat[x1, x2])) computed while the
        halo is staged (include/advshadow.h: advs_conv_args.norm).
        extra = (e1, e2_or_None): NHWC tensors at the OUTPUT resolution whose 1x1 conv is summed in;
        ``w`` then holds [taps * (C1 + C2) | E1 + E2] per output channel."""
        B, H, W, L1 = x1.shape
        L2 = 0 if x2 is None else x2.shape[3]
        # a source whose channel count is not a whole number of 128-byte slabs is read with pixel stride L and
        # K extent C (rounded up); pack_conv_weight zero-fills the weights of the surplus channels
        slab = SLAB_ELEMS[self.dt]
        C1, C2 = -(-L1 // slab) * slab, -(-L2 // slab) * slab
        subpixel = upsample == "subpixel"                 # weights from pack_subpixel_upsample_weight
        HL, WL = (H * 2, W * 2) if upsample else (H, W)
        Ho = (HL + 2 * pad - ksize) // stride + 1
        Wo = (WL + 2 * pad - ksize) // stride + 1
        y = out if out is not None else self.buf((B, Ho, Wo, cout))
        a = ConvArgs(ptr(x1), ptr(x2), ptr(w), ptr(bias), ptr(temb), ptr(residual), ptr(y),
                     B, H, W, C1, C2, cout, ksize, stride, pad, 2 if subpixel else (1 if upsample else 0),
                     ACT[act] | (GN_RESIDUAL_AFTER_ACT if residual_after_act else 0), self.dt, temb_stride, tile, 0, 0,
                     ptr(extra[0]) if extra else 0, ptr(extra[1]) if extra and extra[1] is not None else 0,
                     extra[0].shape[3] if extra else 0, extra[1].shape[3] if extra and extra[1] is not None else 0,
                     L1 if L1 != C1 else 0, L2 if L2 != C2 else 0, ptr(relu_mask), ptr(norm))
        kw = 16 * (C1 + C2) if subpixel else ksize * ksize * (C1 + C2) + a.ce1 + a.ce2
        if w.numel() != cout * kw:
            raise ValueError(f"conv: packed weight has {w.numel()} elements, expected {cout} x {kw} "
                             f"(sources {L1}+{L2} channels, slab {slab})")
        stats = None
        if want_stats:
            # per-channel (sum, sumsq) per row block from the epilogue, for the GroupNorm that reads y
            a.tile = self.lib.advs_conv_resolve_tile(C.byref(a))
            rows = self.lib.advs_conv_tile_rows(a.tile)
            if rows > 0 and (Ho * Wo) % rows == 0:
                stats = self.buf((B * Ho * Wo // rows, cout, 2), torch.float32)
                self.stats[y.data_ptr()] = (stats, Ho * Wo // rows)
                a.stats, a.stats_rows = ptr(stats), rows
        self.plan.add(self.lib.advs_conv2d, C.byref(a), keep=(a, x1, x2, w, bias, temb, residual, y, stats, extra, relu_mask, norm))
        return y

    def conv_first(self, x_nchw, w, bias, cout, want_stats=False):
        B, Cin, H, W = x_nchw.shape
        y = self.buf((B, H, W, cout))
        stats = None
        rows = self.lib.advs_conv_first_stats_rows(Cin, H, W, cout, self.dt) if want_stats else 0
        if rows > 0:
            stats = self.buf((B * H * W // rows, cout, 2), torch.float32)
            self.stats[y.data_ptr()] = (stats, H * W // rows)
        self.plan.add(self.lib.advs_conv3x3_first_stats, ptr(x_nchw), ptr(w), ptr(bias), ptr(y), ptr(stats), B, Cin, H, W,
                      cout, self.dt, keep=(x_nchw, w, bias, y, stats))
        return y

    def conv_last(self, x, w, bias, cout, ksize, out_nchw):
        B, H, W, Cin = x.shape
        self.plan.add(self.lib.advs_conv_last, ptr(x), ptr(w), ptr(bias), ptr(out_nchw), B, Cin, H, W, cout, ksize, self.dt,
                      keep=(x, w, bias, out_nchw))
        return out_nchw

    def groupnorm(self, x, gamma, beta, groups, act=None, x2=None, residual=None, chan_add=None, chan_add_stride=0,
                  residual_after_act=False):
        B, H, W, C1 = x.shape
        actc = ACT[act] | (GN_RESIDUAL_AFTER_ACT if residual_after_act else 0)
        C2 = 0 if x2 is None else x2.shape[3]
        y = self.buf((B, H, W, C1 + C2))
        s1 = self.stats.get(x.data_ptr())
        s2 = self.stats.get(x2.data_ptr()) if x2 is not None else None
        if s1 is not None and (x2 is None or s2 is not None):
            self.plan.add(self.lib.advs_groupnorm_stats, ptr(x), ptr(x2), ptr(s1[0]), s1[1],
                          ptr(s2[0]) if s2 else 0, s2[1] if s2 else 0, ptr(gamma), ptr(beta), ptr(residual),
                          ptr(chan_add), chan_add_stride, ptr(y), ptr(self.gn_scratch), B, H * W, C1, C2, groups,
                          actc, self.dt, keep=(x, x2, s1, s2, gamma, beta, residual, chan_add, y))
            return y
        self.plan.add(self.lib.advs_groupnorm, ptr(x), ptr(x2), ptr(gamma), ptr(beta), ptr(residual), ptr(chan_add),
                      chan_add_stride, ptr(y), ptr(self.gn_scratch), B, H * W, C1, C2, groups, actc, self.dt,
                      keep=(x, x2, gamma, beta, residual, chan_add, y))
        return y

    NORM_MAXC = 384              # csrc/conv_halo2.hip: H2Geom::NORM_MAXC

    def can_fuse_norm(self, x, x2, cout, ksize=3):
        """GroupNorm + SiLU applied inside the conv that reads it (advs_conv_args.norm)?  16-bit storage, a 3x3 conv on a map the
        4-wave halo tile takes (>= 128 x 128, multiples of 16), one 128-channel output tile -- with two the transform would run
        twice per element: measured at level 1 (128 x 128, 256 outputs; round 3, one box) the convs grow by 1.10 ms for 1.07 ms of
        norm passes saved, and with only the two 512-channel ones by 0.42 for 0.36 -- at most NORM_MAXC input channels, statistics
        of every source available from its producer."""
        B, H, W, C1 = x.shape
        C2 = 0 if x2 is None else x2.shape[3]
        if self.dt == F32 or ksize != 3 or cout > 128 or C1 + C2 > self.NORM_MAXC or C1 % 64 or C2 % 64:
            return False
        if H % 16 or W % 16 or H * W < 128 * 128:
            return False
        return x.data_ptr() in self.stats and (x2 is None or x2.data_ptr() in self.stats)

    def groupnorm_affine(self, x, gamma, beta, groups, x2=None):
        """The [B][C][2] (scale, shift) table of GroupNorm(groups)(cat[x, x2]) from the producers' epilogue statistics."""
        B, H, W, C1 = x.shape
        C2 = 0 if x2 is None else x2.shape[3]
        s1 = self.stats[x.data_ptr()]
        s2 = self.stats[x2.data_ptr()] if x2 is not None else None
        table = self.buf((B, C1 + C2, 2), torch.float32)
        self.plan.add(self.lib.advs_groupnorm_affine_stats, ptr(s1[0]), s1[1], ptr(s2[0]) if s2 else 0, s2[1] if s2 else 0,
                      ptr(gamma), ptr(beta), ptr(self.gn_scratch), ptr(table), B, H * W, C1, C2, groups,
                      keep=(x, x2, s1, s2, gamma, beta, table))
        return table

    def maxpool2(self, x):
        B, H, W, Cc = x.shape
        y = self.buf((B, H // 2, W // 2, Cc))
        self.plan.add(self.lib.advs_maxpool2, ptr(x), ptr(y), B, H, W, Cc, self.dt, keep=(x, y))
        return y

    def concat_upsample2x(self, skip, x):
        B, h, w, C2 = x.shape
        C1 = skip.shape[3]
        y = self.buf((B, 2 * h, 2 * w, C1 + C2))
        self.plan.add(self.lib.advs_concat_upsample2x, ptr(skip), ptr(x), ptr(y), B, h, w, C1, C2, self.dt,
                      keep=(skip, x, y))
        return y

    def concat_nearest2x(self, skip, x):
        B, h, w, C2 = x.shape
        C1 = skip.shape[3]
        y = self.buf((B, 2 * h, 2 * w, C1 + C2))
        self.plan.add(self.lib.advs_concat_nearest2x, ptr(skip), ptr(x), ptr(y), B, h, w, C1, C2, self.dt,
                      keep=(skip, x, y))
        return y

    def layernorm(self, x, gamma, beta, eps=1e-5):
        rows = x.numel() // x.shape[-1]
        y = self.buf(tuple(x.shape))
        self.plan.add(self.lib.advs_layernorm, ptr(x), ptr(gamma), ptr(beta), ptr(y), rows, x.shape[-1], float(eps),
                      self.dt, keep=(x, gamma, beta, y))
        return y

    def attention(self, qkv, heads, d, q_off, k_off, v_off, head_stride, n_valid=None):
        B, H, W, LD = qkv.shape
        y = self.buf((B, H, W, heads * d))
        self.plan.add(self.lib.advs_attention_masked, ptr(qkv), ptr(y), B, H * W, n_valid or H * W, heads, d, LD,
                      q_off, k_off, v_off, head_stride, self.dt, keep=(qkv, y))
        return y

    def attention_bias(self, qkv, heads, d, q_off, k_off, v_off, head_stride, bias_log2e, bias_mod):
        """qkv [Bseq, 1, N, LD]; bias_log2e f32 [bias_mod][heads][N][N] (already times log2 e)."""
        Bs, H, W, LD = qkv.shape
        y = self.buf((Bs, H, W, heads * d))
        self.plan.add(self.lib.advs_attention_bias, ptr(qkv), ptr(y), ptr(bias_log2e), bias_mod, Bs, H * W, heads, d, LD,
                      q_off, k_off, v_off, head_stride, self.dt, keep=(qkv, y, bias_log2e))
        return y

    def window_shift(self, x, window, shift, inverse=False, residual=None, image_hw=None):
        """Swin window partition (+ cyclic shift) and its inverse (+ residual); see advs_window_shift."""
        if not inverse:
            B, H, W, Cc = x.shape
            y = self.buf((B * (H // window) * (W // window), 1, window * window, Cc))
        else:
            H, W = image_hw
            Cc = x.shape[-1]
            B = x.shape[0] // ((H // window) * (W // window))
            y = self.buf((B, H, W, Cc))
        self.plan.add(self.lib.advs_window_shift, ptr(x), ptr(residual), ptr(y), B, H, W, Cc, window, shift, 1 if inverse else 0,
                      self.dt, keep=(x, residual, y))
        return y

    def linear(self, x, w, bias, act_in=None, act_out=None):
        Bn, K = x.shape
        N = w.shape[0]
        y = self.buf((Bn, N), torch.float32)
        self.plan.add(self.lib.advs_linear_f32, ptr(x), ptr(w), ptr(bias), ptr(y), Bn, K, N, ACT[act_in], ACT[act_out],
                      keep=(x, w, bias, y))
        return y

    def timestep_embedding(self, t, freqs, cos_first, table=None, labels=None, rows=None):
        """rows: embed only the first ``rows`` timesteps (1 in a sampler step, where every sample sits at the same timestep)."""
        half = freqs.numel()
        n = self.B if rows is None else rows
        y = self.buf((n, 2 * half), torch.float32)
        self.plan.add(self.lib.advs_timestep_embedding, ptr(t), ptr(freqs), half, 1 if cos_first else 0, ptr(table),
                      ptr(labels), ptr(y), n, keep=(t, freqs, table, labels, y))
        return y


def subpixel_upsample_eligible(h, w, ksize=3):
    """Can "nearest x2 then 3x3" on an h x w input run as the four-parity 2x2 form (advs_conv_args.upsample = 2)?"""
    return ksize == 3 and h % 16 == 0 and w % 16 == 0


def pack_subpixel_upsample_weight(w, dt):
    """OIHW 3x3 weight of a conv that follows a nearest x2 upsample -> [4][O][2][2][I] in the compute dtype:
    for output parity (a, b) the taps that read the same low-res pixel are summed (in f32, before rounding):
    rows {w0, w1 + w2} for a = 0 and {w0 + w1, w2} for a = 1, columns likewise (include/advshadow.h)."""
    w = w.detach().float()
    rows = (torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2]], 2), torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2]], 2))
    parts = []
    for a in (0, 1):
        r = rows[a]                                            # [O, I, 2, 3]
        for b in (0, 1):
            c = torch.stack([r[..., 0], r[..., 1] + r[..., 2]], 3) if b == 0 else torch.stack([r[..., 0] + r[..., 1], r[..., 2]], 3)
            parts.append(pack_conv_weight(c, dt))              # [O][2][2][I]
    return torch.stack(parts, 0).contiguous()


def pack_conv_weight(w, dt, stream=None, sources=None):
    """torch OIHW f32 (device) -> [O][R][S][I] in the compute dtype, on the current stream.  ``sources`` =
    channel counts of the concatenated inputs (default: one source); each is zero-padded to a whole number of
    128-byte slabs, matching how Builder.conv presents such a source to the kernel."""
    lib = _lib.load()
    w = w.detach().float()
    slab = SLAB_ELEMS[dt]
    sources = tuple(sources) if sources else (w.shape[1],)
    if sum(sources) != w.shape[1]:
        raise ValueError(f"pack_conv_weight: sources {sources} do not add up to {w.shape[1]} input channels")
    if any(c % slab for c in sources):
        parts, o = [], 0
        for c in sources:
            parts.append(w[:, o:o + c])
            if c % slab:
                parts.append(torch.zeros((w.shape[0], slab - c % slab) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device))
            o += c
        w = torch.cat(parts, dim=1)
    w = w.contiguous()
    O, I, R, S = w.shape
    out = torch.empty((O, R, S, I), dtype=TORCH_DT[dt], device=w.device)
    s = (stream or torch.cuda.current_stream(w.device)).cuda_stream
    check(lib.advs_pack_conv_weight(w.data_ptr(), out.data_ptr(), O, I, R, S, dt, s), "pack_conv_weight")
    # w must stay alive until the kernel ran; same-stream ordering + sync keeps it simple
    torch.cuda.current_stream(w.device).synchronize() if stream is None else stream.synchronize()
    return out
