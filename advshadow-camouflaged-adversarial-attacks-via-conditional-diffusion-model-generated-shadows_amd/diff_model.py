"""Drop-in for the reference's ``diff_model.py`` hot path on MI355X.

Same public names and call signatures as the reference (``UNetModel``, ``GaussianDiffusion``,
``timestep_embedding``, the beta schedules) and the same ``state_dict`` key names, so checkpoints
interchange.  ``UNetModel`` is an ``nn.Module`` only as a *parameter container*: its forward never
calls a torch op on activations, it replays a plan of HIP kernels (``engine.py``), optionally as
one hipGraph.  ``GaussianDiffusion.ddim_sample`` runs the whole T-step loop on the device with
the per-step DDIM update fused into one kernel.

Reference: diff_model.py:16-33 (embedding), :157-267 (UNetModel), :269-285 (schedules),
:286-338 (GaussianDiffusion tables), :416-474 (ddim_sample).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .engine import (BF16, F32, Builder, dtype_code, pack_conv_weight, pack_subpixel_upsample_weight, ptr, SLAB_ELEMS,
                     subpixel_upsample_eligible)
from ._lib import check


# --------------------------------------------------------------------------- layout of the network
def unet_layout(model_channels, channel_mult, num_res_blocks, attention_resolutions, in_channels):
    """Flatten the constructor's loops (diff_model.py:190-237) into three stage lists.

    A stage is the list of layers one ``TimestepEmbedSequential`` holds; a layer is
    ``(kind, key_prefix, cin, cout)`` with kind in {"stem", "res", "attn", "down", "up"}.
    """
    stages_down = [[("stem", "down_blocks.0.0", in_channels, model_channels)]]
    skip_ch = [model_channels]
    ch, ds = model_channels, 1
    last = len(channel_mult) - 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            idx = len(stages_down)
            stage = [("res", f"down_blocks.{idx}.0", ch, mult * model_channels)]
            ch = mult * model_channels
            if ds in attention_resolutions:
                stage.append(("attn", f"down_blocks.{idx}.1", ch, ch))
            stages_down.append(stage)
            skip_ch.append(ch)
        if level != last:
            stages_down.append([("down", f"down_blocks.{len(stages_down)}.0", ch, ch)])
            skip_ch.append(ch)
            ds *= 2
    middle = [("res", "middle_block.0", ch, ch), ("attn", "middle_block.1", ch, ch), ("res", "middle_block.2", ch, ch)]
    stages_up = []
    for level in range(last, -1, -1):
        mult = channel_mult[level]
        for i in range(num_res_blocks + 1):
            idx = len(stages_up)
            stage = [("res", f"up_blocks.{idx}.0", ch + skip_ch.pop(), model_channels * mult)]
            ch = model_channels * mult
            if ds in attention_resolutions:
                stage.append(("attn", f"up_blocks.{idx}.{len(stage)}", ch, ch))
            if level and i == num_res_blocks:
                stage.append(("up", f"up_blocks.{idx}.{len(stage)}", ch, ch))
                ds //= 2
            stages_up.append(stage)
    return stages_down, middle, stages_up


def _attach(root, dotted, module):
    """Register ``module`` under a dotted state_dict prefix, creating plain containers on the way."""
    parts = dotted.split(".")
    cur = root
    for p in parts[:-1]:
        nxt = cur._modules.get(p)
        if nxt is None:
            nxt = nn.Module()
            cur.add_module(p, nxt)
        cur = nxt
    cur.add_module(parts[-1], module)


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal embedding [cos | sin] (diff_model.py:16-33), computed by the HIP kernel."""
    assert dim % 2 == 0, "odd embedding widths are not used by the reference networks"
    _lib.init_device()
    lib = _lib.load()
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)
    t = timesteps.to(torch.int64).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    s = torch.cuda.current_stream(t.device).cuda_stream
    check(lib.advs_timestep_embedding(t.data_ptr(), freqs.data_ptr(), half, 1, 0, 0, out.data_ptr(), t.shape[0], s),
          "timestep_embedding")
    return out


# --------------------------------------------------------------------------- the eps-predictor
class UNetModel(nn.Module):
    """HIP-backed eps-predictor with the reference's constructor and ``forward(x, timesteps)``.

    Extra keyword (not in the reference): ``compute_dtype`` = "fp32" (exact-f32 MFMA, parity mode,
    default) or "bf16" (bf16 storage + MFMA, f32 accumulate).  ``use_graph`` replays each forward
    as one hipGraph.
    """

    def __init__(self, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2,
                 attention_resolutions=(8, 16), dropout=0, channel_mult=(1, 2, 2, 2), conv_resample=True,
                 num_heads=4, compute_dtype="fp32", use_graph=True):
        super().__init__()
        if not conv_resample:
            raise NotImplementedError("conv_resample=False is malformed in the reference (diff_model.py:150)")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, tuple(attention_resolutions)
        self.dropout, self.channel_mult, self.conv_resample, self.num_heads = dropout, tuple(channel_mult), True, num_heads
        self.compute_dtype, self.use_graph = compute_dtype, use_graph
        self.fuse_norm = True          # 16-bit modes: GroupNorm + SiLU inside the conv that reads it where the shape allows (engine.Builder.can_fuse_norm)
        ted = 4 * model_channels
        self.layout = unet_layout(model_channels, self.channel_mult, num_res_blocks, self.attention_resolutions, in_channels)
        # parameter holders, created in the reference's order so a seeded construction matches it
        _attach(self, "time_embed.0", nn.Linear(model_channels, ted))
        _attach(self, "time_embed.2", nn.Linear(ted, ted))
        down, middle, up = self.layout
        for stage in down + [middle] + up:
            for kind, p, cin, cout in stage:
                if kind == "stem":
                    _attach(self, p, nn.Conv2d(cin, cout, 3, padding=1))
                elif kind == "res":
                    _attach(self, p + ".conv1.0", nn.GroupNorm(32, cin))
                    _attach(self, p + ".conv1.2", nn.Conv2d(cin, cout, 3, padding=1))
                    _attach(self, p + ".time_emb.1", nn.Linear(ted, cout))
                    _attach(self, p + ".conv2.0", nn.GroupNorm(32, cout))
                    _attach(self, p + ".conv2.3", nn.Conv2d(cout, cout, 3, padding=1))
                    if cin != cout:
                        _attach(self, p + ".shortcut", nn.Conv2d(cin, cout, 1))
                elif kind == "attn":
                    _attach(self, p + ".norm", nn.GroupNorm(32, cin))
                    _attach(self, p + ".qkv", nn.Conv2d(cin, 3 * cin, 1, bias=False))
                    _attach(self, p + ".proj", nn.Conv2d(cin, cin, 1))
                elif kind == "down":
                    _attach(self, p + ".op", nn.Conv2d(cin, cin, 3, stride=2, padding=1))
                elif kind == "up":
                    _attach(self, p + ".conv", nn.Conv2d(cin, cin, 3, padding=1))
        _attach(self, "out.0", nn.GroupNorm(32, model_channels))
        _attach(self, "out.2", nn.Conv2d(model_channels, out_channels, 3, padding=1))
        self._packed = {}          # dtype code -> (version, dict)
        self._engines = {}         # (B, S, dtype code) -> _ForwardEngine

    # ---- packed weights ---------------------------------------------------------------------
    def _version(self):
        dev = next(self.parameters()).device
        return (str(dev), sum(p._version for p in self.parameters()))

    def packed_weights(self, dt):
        ver = self._version()
        hit = self._packed.get(dt)
        if hit is not None and hit[0] == ver:
            return hit[1]
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AdvsError("UNetModel parameters are on %s: move the model to the GPU (model.to('cuda')); "
                                 "the HIP path has no CPU fallback" % dev)
        W = {}
        slab = SLAB_ELEMS[dt]
        down, middle, up = self.layout
        temb_w, temb_b, off = [], [], 0
        W["temb_off"] = {}
        for stage in down + [middle] + up:
            for kind, p, cin, cout in stage:
                if kind == "stem":
                    W[p + ".w"] = sd[p + ".weight"].float().contiguous()
                    W[p + ".b"] = sd[p + ".bias"].float().contiguous()
                elif kind == "res":
                    if cin % slab or cout % slab:
                        raise _lib.AdvsError(f"{p}: channels ({cin}->{cout}) must be multiples of {slab} for this dtype")
                    for n in (".conv1.0", ".conv2.0"):
                        W[p + n + ".g"] = sd[p + n + ".weight"].float().contiguous()
                        W[p + n + ".b"] = sd[p + n + ".bias"].float().contiguous()
                    for n in (".conv1.2", ".conv2.3"):
                        W[p + n + ".w"] = pack_conv_weight(sd[p + n + ".weight"], dt)
                        W[p + n + ".b"] = sd[p + n + ".bias"].float().contiguous()
                    if cin != cout:
                        # shortcut fused into conv2: K = [9*cout | cin], one bias (diff_model.py:102-103)
                        ws = pack_conv_weight(sd[p + ".shortcut.weight"], dt)
                        W[p + ".conv2.3.w"] = torch.cat([W[p + ".conv2.3.w"].reshape(cout, -1), ws.reshape(cout, -1)], 1).contiguous()
                        W[p + ".conv2.3.b"] = (W[p + ".conv2.3.b"] + sd[p + ".shortcut.bias"].float()).contiguous()
                    temb_w.append(sd[p + ".time_emb.1.weight"].float())
                    temb_b.append(sd[p + ".time_emb.1.bias"].float())
                    W["temb_off"][p] = off
                    off += cout
                elif kind == "attn":
                    W[p + ".norm.g"] = sd[p + ".norm.weight"].float().contiguous()
                    W[p + ".norm.b"] = sd[p + ".norm.bias"].float().contiguous()
                    W[p + ".qkv.w"] = pack_conv_weight(sd[p + ".qkv.weight"], dt)
                    W[p + ".proj.w"] = pack_conv_weight(sd[p + ".proj.weight"], dt)
                    W[p + ".proj.b"] = sd[p + ".proj.bias"].float().contiguous()
                elif kind == "down":
                    W[p + ".op.w"] = pack_conv_weight(sd[p + ".op.weight"], dt)
                    W[p + ".op.b"] = sd[p + ".op.bias"].float().contiguous()
                elif kind == "up":
                    W[p + ".conv.w"] = pack_conv_weight(sd[p + ".conv.weight"], dt)
                    W[p + ".conv.w4"] = pack_subpixel_upsample_weight(sd[p + ".conv.weight"], dt)   # four-parity 2x2 form
                    W[p + ".conv.b"] = sd[p + ".conv.bias"].float().contiguous()
        W["temb_w"] = torch.cat(temb_w, 0).contiguous()       # every block's Linear(512 -> cout), stacked
        W["temb_b"] = torch.cat(temb_b, 0).contiguous()
        W["temb_total"] = off
        for k in ("time_embed.0", "time_embed.2"):
            W[k + ".w"] = sd[k + ".weight"].float().contiguous()
            W[k + ".b"] = sd[k + ".bias"].float().contiguous()
        W["out.0.g"] = sd["out.0.weight"].float().contiguous()
        W["out.0.b"] = sd["out.0.bias"].float().contiguous()
        W["out.2.w"] = sd["out.2.weight"].float().contiguous()
        W["out.2.b"] = sd["out.2.bias"].float().contiguous()
        half = self.model_channels // 2
        # host-side table, computed with the same torch CPU ops as the reference (diff_model.py:26-28)
        W["freqs"] = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[2] == dt]:
            del self._engines[key]                          # plans hold pointers into the old weights
        return W

    # ---- plan -------------------------------------------------------------------------------------
    def engine(self, batch, size, dtype=None, uniform_t=False):
        """uniform_t: the plan of a SAMPLER step -- every image of the batch sits at the same timestep (diff_model.py:447: one ``t`` is
        broadcast over the batch), so the time-embedding MLP and the stacked per-block Linear run for ONE row that every conv's
        epilogue reads (timestep_embedding / time_embed hoisting); ``forward(x, t)`` keeps the per-image plan."""
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        key = (batch, size, dt, bool(uniform_t))
        eng = self._engines.get(key)
        if eng is None:
            eng = _ForwardEngine(self, W, batch, size, dt, uniform_t=bool(uniform_t))
            self._engines[key] = eng
        return eng

    def forward(self, x, timesteps):
        """x: [N,C,H,W] f32 on the GPU, timesteps: [N] -> eps [N,C,H,W] f32 (diff_model.py:245-267)."""
        B, Cc, H, Wd = x.shape
        assert H == Wd and Cc == self.in_channels
        eng = self.engine(B, H)
        cur = torch.cuda.current_stream(x.device)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x.to(torch.float32), non_blocking=True)
            eng.t.copy_(timesteps.to(torch.int64), non_blocking=True)
            eng.run()
            out = eng.eps.clone()
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        return out


class _ForwardEngine:
    """The frozen plan of one UNetModel forward for (batch, size, dtype)."""

    def __init__(self, model, W, batch, size, dt, stream=None, uniform_t=False):
        dev = next(model.parameters()).device
        self.uniform_t = uniform_t
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.model, self.B, self.S, self.dt = model, batch, size, dt
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            self.bld = bld
            self.x = torch.zeros((batch, model.in_channels, size, size), dtype=torch.float32, device=dev)
            self.t = torch.zeros((batch,), dtype=torch.int64, device=dev)
            self.eps = torch.zeros((batch, model.out_channels, size, size), dtype=torch.float32, device=dev)
            emit_unet_forward(bld, model, W, self.x, self.t, self.eps, uniform_t=uniform_t)
            self.plan = bld.plan
            self.captured = False
            torch.cuda.synchronize(dev)

    def run(self):
        """Enqueue one forward on self.stream (first call eager, then captured if use_graph)."""
        if self.model.use_graph and not self.captured:
            self.plan.run_eager()              # warm-up outside capture (module load, attribute opt-ins)
            self.stream.synchronize()
            self.plan.capture()
            self.captured = True
        self.plan.run()


def emit_unet_forward(bld, model, W, x_nchw, t_dev, eps_out, uniform_t=False):
    """Append the kernels of one forward (diff_model.py:245-267) to ``bld.plan``.  uniform_t: t_dev[0] is every image's timestep."""
    heads = model.num_heads
    down, middle, up = model.layout
    # time embedding MLP (diff_model.py:254) and every block's time_emb Linear, hoisted in front
    e0 = bld.timestep_embedding(t_dev, W["freqs"], cos_first=True, rows=1 if uniform_t else None)
    e1 = bld.linear(e0, W["time_embed.0.w"], W["time_embed.0.b"], act_out="silu")
    emb = bld.linear(e1, W["time_embed.2.w"], W["time_embed.2.b"])
    temb = bld.linear(emb, W["temb_w"], W["temb_b"], act_in="silu")          # [B, sum cout]
    tstride = -1 if uniform_t else W["temb_total"]           # -1: one embedding row for the whole batch (advs_conv_args.temb_stride)

    fuse = getattr(model, "fuse_norm", True)

    def norm_silu_conv(x1, x2, gk, wk, cout, **kw):
        """conv3x3(SiLU(GroupNorm32(cat[x1, x2]))) (diff_model.py:70-73, 83-86): on the large single-tile maps the norm is applied
        inside the conv while its halo is staged (bit-identical to the two passes), elsewhere as its own pass."""
        if fuse and bld.can_fuse_norm(x1, x2, cout):
            tab = bld.groupnorm_affine(x1, W[gk + ".g"], W[gk + ".b"], 32, x2=x2)
            y = bld.conv(x1, W[wk + ".w"], cout, x2=x2, bias=W[wk + ".b"], norm=tab, want_stats=True, **kw)
            bld.free(tab)
            return y
        a = bld.groupnorm(x1, W[gk + ".g"], W[gk + ".b"], 32, act="silu", x2=x2)
        y = bld.conv(a, W[wk + ".w"], cout, bias=W[wk + ".b"], want_stats=True, **kw)
        bld.free(a)
        return y

    def res_block(p, cin, cout, x1, x2):
        o = W["temb_off"][p]
        h1 = norm_silu_conv(x1, x2, p + ".conv1.0", p + ".conv1.2", cout, temb=temb[:, o:o + cout], temb_stride=tstride)
        if cin != cout:     # conv2(h) + shortcut(cat[x1, x2]) as one implicit GEMM over K = [9*cout | cin]
            y = norm_silu_conv(h1, None, p + ".conv2.0", p + ".conv2.3", cout, extra=(x1, x2))
        else:
            assert x2 is None
            y = norm_silu_conv(h1, None, p + ".conv2.0", p + ".conv2.3", cout, residual=x1)
        bld.free(h1)
        return y

    def attn_block(p, ch, x):
        n = bld.groupnorm(x, W[p + ".norm.g"], W[p + ".norm.b"], 32)
        qkv = bld.conv(n, W[p + ".qkv.w"], 3 * ch, ksize=1, pad=0)
        bld.free(n)
        d = ch // heads
        # per head the 3d output channels are [q | k | v] (reshape + chunk, diff_model.py:120)
        o = bld.attention(qkv, heads, d, 0, d, 2 * d, 3 * d)
        bld.free(qkv)
        y = bld.conv(o, W[p + ".proj.w"], ch, bias=W[p + ".proj.b"], residual=x, ksize=1, pad=0, want_stats=True)
        bld.free(o)
        return y

    def run_stage(stage, h, skip):
        for kind, p, cin, cout in stage:
            if kind == "stem":
                new = bld.conv_first(x_nchw, W[p + ".w"], W[p + ".b"], cout, want_stats=True)
            elif kind == "res":
                new = res_block(p, cin, cout, h, skip)
                skip = None
            elif kind == "attn":
                new = attn_block(p, cin, h)
            elif kind == "down":
                new = bld.conv(h, W[p + ".op.w"], cout, bias=W[p + ".op.b"], stride=2, want_stats=True)
            elif kind == "up":
                # Upsample (diff_model.py:129-140): nearest x2 + 3x3, computed on the low-res grid where the shape allows
                if subpixel_upsample_eligible(h.shape[1], h.shape[2]):
                    new = bld.conv(h, W[p + ".conv.w4"], cout, bias=W[p + ".conv.b"], upsample="subpixel", want_stats=True)
                else:
                    new = bld.conv(h, W[p + ".conv.w"], cout, bias=W[p + ".conv.b"], upsample=True, want_stats=True)
            if h is not None and not any(h is s for s in hs):
                bld.free(h)
            h = new
        return h

    hs = []
    h = None
    for stage in down:
        h = run_stage(stage, h, None)
        hs.append(h)
    h = run_stage(middle, h, None)
    for stage in up:
        skip = hs.pop()
        h2 = run_stage(stage, h, skip)      # first layer is the res block reading cat([h, skip])
        bld.free(skip)
        h = h2
    a = bld.groupnorm(h, W["out.0.g"], W["out.0.b"], 32, act="silu")
    bld.free(h)
    bld.conv_last(a, W["out.2.w"], W["out.2.b"], model.out_channels, 3, eps_out)
    bld.free(a)


# --------------------------------------------------------------------------- schedules
def linear_beta_schedule(timesteps):
    """diff_model.py:269-273."""
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps, s=0.008):
    """diff_model.py:275-285."""
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


class GaussianDiffusion:
    """Sampler tables + the DDIM loop (diff_model.py:286-338, 416-474)."""

    def __init__(self, timesteps=1000, beta_schedule="cosine"):
        self.timesteps = timesteps
        if beta_schedule == "linear":
            betas = linear_beta_schedule(timesteps)
        elif beta_schedule == "cosine":
            betas = cosine_beta_schedule(timesteps)
        else:
            raise ValueError(f"unknown beta schedule {beta_schedule}")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        # tables of the forward process and of the posterior q(x_{t-1} | x_t, x_0) (diff_model.py:304-331)
        self.alphas_cumprod_prev = F.pad(self.alphas_cumprod[:-1], (1, 0), value=1.)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = torch.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = torch.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = torch.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = self.betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = torch.log(torch.cat([self.posterior_variance[1:2], self.posterior_variance[1:]]))
        self.posterior_mean_coef1 = self.betas * torch.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * torch.sqrt(self.alphas) / (1.0 - self.alphas_cumprod)
        self._loops = {}
        self._ancestral = {}

    def _extract(self, a, t, x_shape):
        out = a.to(t.device).gather(0, t).float()
        return out.reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))

    # ---- forward process / posterior helpers: tiny elementwise torch ops on the caller's tensors, as in the reference
    def q_sample(self, x_start, t, noise=None):
        """diff_model.py:340-347."""
        if noise is None:
            noise = torch.randn_like(x_start)
        return (self._extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + self._extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def q_mean_variance(self, x_start, t):
        """diff_model.py:349-353."""
        return (self._extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start,
                self._extract(1.0 - self.alphas_cumprod, t, x_start.shape),
                self._extract(self.log_one_minus_alphas_cumprod, t, x_start.shape))

    def q_posterior_mean_variance(self, x_start, x_t, t):
        """diff_model.py:356-363."""
        mean = (self._extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + self._extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return (mean, self._extract(self.posterior_variance, t, x_t.shape),
                self._extract(self.posterior_log_variance_clipped, t, x_t.shape))

    def predict_start_from_noise(self, x_t, t, noise):
        """diff_model.py:366-370."""
        return (self._extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - self._extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def p_mean_variance(self, model, x_t, t, clip_denoised=True):
        """diff_model.py:373-383: (posterior mean, variance, clipped log-variance) of one step.  The mean comes from the same
        fused kernel as the sampling loop (``advs_ddpm_posterior_step`` with zero noise); the two table look-ups are the
        reference's ``_extract``."""
        mean = self._posterior_step(model, x_t, t, clip_denoised, None)
        return (mean, self._extract(self.posterior_variance, t, x_t.shape),
                self._extract(self.posterior_log_variance_clipped, t, x_t.shape))

    def _posterior_step(self, model, x_t, t, clip_denoised, noise):
        """x_{t-1} = mean(x_t, eps) + [t != 0] * exp(0.5 * logvar) * noise on the device: eps-predictor forward, then the
        fused update kernel per run of equal timesteps (the reference's loop passes one t for the whole batch)."""
        lib = _lib.load()
        dev = x_t.device
        eps = model(x_t, t)
        x = x_t.to(torch.float32).contiguous().clone()
        z = torch.zeros_like(x) if noise is None else noise.to(dev, torch.float32).contiguous()
        coef, tseq = self._posterior_tables(dev)
        counter = torch.zeros((1,), dtype=torch.int32, device=dev)
        t_out = torch.zeros((x.shape[0],), dtype=torch.int64, device=dev)
        per = x[0].numel()
        s = torch.cuda.current_stream(dev).cuda_stream
        th = t.to("cpu", torch.int64).tolist()
        if len(th) != x.shape[0]:
            raise ValueError(f"t has {len(th)} entries for a batch of {x.shape[0]}")
        if any(not 0 <= v < self.timesteps for v in th):
            # the reference's a.gather(0, t) (diff_model.py:334-338) raises here; the kernel would read its tables out of bounds
            raise IndexError(f"timestep out of range [0, {self.timesteps}): {[v for v in th if not 0 <= v < self.timesteps][:4]}")
        if per % 4:
            raise ValueError(f"one image has {per} elements: the fused update works on 16-byte vectors (a multiple of 4 floats)")
        i = 0
        while i < len(th):
            j = i
            while j < len(th) and th[j] == th[i]:
                j += 1
            counter.fill_(self.timesteps - 1 - th[i])                    # the tables are in loop order t = T-1 .. 0
            check(lib.advs_ddpm_posterior_step(x[i:j].data_ptr(), eps[i:j].data_ptr(), z[i:j].data_ptr(), ptr(coef), ptr(tseq),
                                               self.timesteps, ptr(counter), ptr(t_out), j - i, per, 1 if clip_denoised else 0, s),
                  "ddpm_posterior_step")
            i = j
        return x

    @torch.no_grad()
    def p_sample(self, model, x_t, t, clip_denoised=True, noise=None):
        """One ancestral step on caller tensors (diff_model.py:386-396); ``noise`` injects the randn_like."""
        if noise is None:
            noise = torch.randn_like(x_t)
        return self._posterior_step(model, x_t, t, clip_denoised, noise)

    def _posterior_tables(self, device):
        """[T][5] f32 per STEP in loop order (t = T-1 .. 0), rounded exactly where the reference rounds: f64 table ->
        gather -> .float() (_extract), then f32 ops; the last column is mask * exp(0.5 * logvar)."""
        order = torch.arange(self.timesteps - 1, -1, -1)
        f = lambda a: a[order].float()
        sg = (0.5 * f(self.posterior_log_variance_clipped)).exp() * (order != 0).float()
        coef = torch.stack([f(self.sqrt_recip_alphas_cumprod), f(self.sqrt_recipm1_alphas_cumprod),
                            f(self.posterior_mean_coef1), f(self.posterior_mean_coef2), sg], dim=1).contiguous()
        return coef.to(device), order.to(torch.int64).to(device)

    @torch.no_grad()
    def p_sample_loop(self, model, shape, x_T=None, noise_fn=None, keep="all", clip_denoised=True):
        """The full-length ancestral loop (diff_model.py:398-408): T x (eps-predictor forward + fused posterior step,
        one captured graph replayed per step).  Returns the reference's list of per-step numpy images; extra
        keywords (not in the reference): ``x_T`` / ``noise_fn(i, shape)`` inject the random stream, ``keep="last"``
        skips the T-1 intermediate device->host copies and returns a one-element list."""
        batch_size, channels, image_size = shape[0], shape[1], shape[2]
        dev = next(model.parameters()).device
        eng = model.engine(batch_size, image_size, uniform_t=True)
        key = (id(eng), bool(clip_denoised))
        loop = self._ancestral.get(key)
        if loop is None:
            coef, tseq = self._posterior_tables(dev)
            loop = _AncestralLoop(eng, coef, tseq, clip_denoised)
            self._ancestral = {key: loop}
        cur = torch.cuda.current_stream(dev)
        if x_T is None:
            x_T = torch.randn(shape, device=dev)
        eng.stream.wait_stream(cur)
        imgs = []
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x_T.to(dev, torch.float32), non_blocking=True)
            loop.start()
            for i in reversed(range(self.timesteps)):
                if noise_fn is not None:
                    loop.noise.copy_(noise_fn(i, tuple(shape)).to(dev, torch.float32), non_blocking=True)
                else:
                    loop.noise.normal_()
                loop.step()
                if keep == "all" or i == 0:
                    imgs.append(eng.x.to("cpu", non_blocking=True))
        eng.stream.synchronize()
        cur.wait_stream(eng.stream)
        return [im.numpy() for im in imgs]

    @torch.no_grad()
    def sample(self, model, image_size, batch_size=8, channels=3, **kw):
        """diff_model.py:411-413."""
        return self.p_sample_loop(model, shape=(batch_size, channels, image_size, image_size), **kw)

    @staticmethod
    def ddim_sequences(timesteps, ddim_timesteps, method="uniform"):
        """diff_model.py:428-440."""
        if method == "uniform":
            c = timesteps // ddim_timesteps
            seq = np.asarray(list(range(0, timesteps, c)))
        elif method == "quad":
            seq = ((np.linspace(0, np.sqrt(timesteps * .8), ddim_timesteps)) ** 2).astype(int)
        else:
            raise NotImplementedError(f'There is no ddim discretization method called "{method}"')
        seq = seq + 1
        return seq, np.append(np.array([0]), seq[:-1])

    def _tables(self, ddim_timesteps, method, eta, device):
        """Per-step (a_t, a_prev, sigma) in f32, rounded exactly where the reference rounds
        (f64 gather -> .float(), then f32 arithmetic; diff_model.py:450-464), in loop order."""
        seq, prev = self.ddim_sequences(self.timesteps, ddim_timesteps, method)
        order = list(reversed(range(len(seq))))
        a_t = self.alphas_cumprod[torch.as_tensor(seq[order])].float()
        a_p = self.alphas_cumprod[torch.as_tensor(prev[order])].float()
        sigma = eta * torch.sqrt((1 - a_p) / (1 - a_t) * (1 - a_t / a_p))
        coef = torch.stack([a_t, a_p, sigma.float()], dim=1).contiguous().to(device)
        tseq = torch.as_tensor(seq[order], dtype=torch.int64).to(device)
        return coef, tseq

    @torch.no_grad()
    def ddim_sample(self, model, image_size, batch_size=8, channels=3, ddim_timesteps=50,
                    ddim_discr_method="uniform", ddim_eta=0.0, clip_denoised=True, x_T=None,
                    return_tensor=False):
        """DDIM reverse loop, all on the GPU.  Extra keywords (not in the reference): ``x_T``
        injects the starting noise (otherwise drawn on the model's device as the reference does,
        diff_model.py:444); ``return_tensor`` skips the final ``.cpu().numpy()``."""
        dev = next(model.parameters()).device
        eng = model.engine(batch_size, image_size, uniform_t=True)
        lib = _lib.load()
        coef, tseq = self._tables(ddim_timesteps, ddim_discr_method, ddim_eta, dev)
        nsteps = tseq.numel()
        key = (id(eng), nsteps, ddim_discr_method, float(ddim_eta), bool(clip_denoised))
        loop = self._loops.get(key)
        if loop is None:
            loop = _DDIMLoop(eng, coef, tseq, clip_denoised, ddim_eta)
            self._loops = {key: loop}
        cur = torch.cuda.current_stream(dev)
        if x_T is None:
            x_T = torch.randn((batch_size, channels, image_size, image_size), device=dev)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x_T.to(dev, torch.float32), non_blocking=True)
            loop.run(seed_noise=ddim_eta != 0.0)
            out = eng.x.clone()
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        if return_tensor:
            return out
        return out.cpu().numpy()


class _AncestralLoop:
    """One captured step of GaussianDiffusion.p_sample: UNet forward + advs_ddpm_posterior_step."""

    def __init__(self, eng, coef, tseq, clip):
        self.eng, self.coef, self.tseq, self.nsteps = eng, coef, tseq, tseq.numel()
        dev = eng.x.device
        self.counter = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.noise = torch.zeros_like(eng.x)
        from .engine import Plan
        self.plan = Plan(eng.stream)
        self.plan.ops = list(eng.plan.ops)
        self.plan.keep = [eng.plan.keep, coef, tseq, self.counter, self.noise]
        self.plan.add(_lib.load().advs_ddpm_posterior_step, ptr(eng.x), ptr(eng.eps), ptr(self.noise), ptr(coef), ptr(tseq),
                      self.nsteps, ptr(self.counter), ptr(eng.t), eng.B, eng.x[0].numel(), 1 if clip else 0)
        self.captured = False

    def start(self):
        eng = self.eng
        self.counter.zero_()
        eng.t.fill_(int(self.tseq[0].item()))
        if eng.model.use_graph and not self.captured:
            x0 = eng.x.clone()
            self.plan.run_eager()
            eng.stream.synchronize()
            self.plan.capture()
            self.captured = True
            eng.x.copy_(x0)
            self.counter.zero_()
            eng.t.fill_(int(self.tseq[0].item()))

    def step(self):
        self.plan.run()


class _DDIMLoop:
    """One captured sampler step = UNet forward + fused DDIM update; replayed T times."""

    def __init__(self, eng, coef, tseq, clip, eta):
        self.eng, self.coef, self.tseq, self.nsteps = eng, coef, tseq, tseq.numel()
        dev = eng.x.device
        self.counter = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.noise = torch.zeros_like(eng.x) if eta != 0.0 else None
        lib = _lib.load()
        from .engine import Plan
        self.plan = Plan(eng.stream)
        self.plan.ops = list(eng.plan.ops)
        self.plan.keep = [eng.plan.keep, coef, tseq, self.counter, self.noise]
        per = eng.x[0].numel()
        self.plan.add(lib.advs_ddim_step, ptr(eng.x), ptr(eng.eps), 0, 0.0, ptr(self.noise), ptr(coef), ptr(tseq),
                      self.nsteps, ptr(self.counter), ptr(eng.t), eng.B, per, 1 if clip else 0)
        self.captured = False

    def run(self, seed_noise=False):
        eng = self.eng
        self.counter.zero_()
        eng.t.fill_(int(self.tseq[0].item()))
        if eng.model.use_graph and not self.captured and not seed_noise:
            x0 = eng.x.clone()
            self.plan.run_eager()                # warm-up (also validates every launch outside capture)
            eng.stream.synchronize()
            self.plan.capture()
            self.captured = True
            eng.x.copy_(x0)
            self.counter.zero_()
            eng.t.fill_(int(self.tseq[0].item()))
        for _ in range(self.nsteps):
            if seed_noise:
                self.noise.normal_()
                self.plan.run_eager()
            else:
                self.plan.run()
