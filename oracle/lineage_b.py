"""Oracle: the single-file DDPM lineage (``diff_model.py``) restated on CPU.

Test infrastructure only (see ``oracle/__init__.py``).  Pinned by
``tests/golden/lineage_b_*.npz`` (generated from the imported reference).

The network is evaluated functionally over a ``state_dict`` whose key names are
the reference's (``diff_model.py:157-267``), so the same dictionary drives the
oracle and the HIP engine.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_HP = dict(in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2,
                  attention_resolutions=(8, 16), channel_mult=(1, 2, 2, 2), num_heads=4)


def hparams(**over):
    hp = dict(DEFAULT_HP)
    hp.update(over)
    return hp


# --------------------------------------------------------------------------- topology
def topology(hp):
    """Layer list of UNetModel.__init__ (diff_model.py:190-243).

    Returns (down, middle, up): each a list of stages; a stage is a list of
    (kind, key_prefix, cin, cout) with kind in conv/res/attn/down/up.
    """
    mc, mult, nrb = hp["model_channels"], hp["channel_mult"], hp["num_res_blocks"]
    att = hp["attention_resolutions"]
    down = [[("conv", "down_blocks.0.0", hp["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            i = len(down)
            st = [("res", f"down_blocks.{i}.0", ch, m * mc)]
            ch = m * mc
            if ds in att:
                st.append(("attn", f"down_blocks.{i}.1", ch, ch))
            down.append(st)
            chans.append(ch)
        if level != len(mult) - 1:
            i = len(down)
            down.append([("down", f"down_blocks.{i}.0", ch, ch)])
            chans.append(ch)
            ds *= 2
    middle = [("res", "middle_block.0", ch, ch), ("attn", "middle_block.1", ch, ch),
              ("res", "middle_block.2", ch, ch)]
    up = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            k = len(up)
            st = [("res", f"up_blocks.{k}.0", ch + chans.pop(), mc * m)]
            ch = mc * m
            j = 1
            if ds in att:
                st.append(("attn", f"up_blocks.{k}.{j}", ch, ch))
                j += 1
            if level and i == nrb:
                st.append(("up", f"up_blocks.{k}.{j}", ch, ch))
                ds //= 2
            up.append(st)
    return down, middle, up


# --------------------------------------------------------------------------- seeded init
def init_state_dict(seed, hp):
    """Rebuild the state_dict ``UNetModel(**hp)`` gets under ``torch.manual_seed(seed)``.

    Layers are created with the stock torch constructors in the order the
    reference constructor creates them (diff_model.py:183-243, and inside a
    ResidualBlock :70-92 conv1, time_emb, conv2, shortcut; AttentionBlock
    :113-115 norm, qkv, proj), so the RNG stream is consumed identically.
    The golden fixtures store a digest of the reference's state_dict to prove it.
    """
    torch.manual_seed(seed)
    sd = {}
    mc = hp["model_channels"]
    ted = 4 * mc

    def put(prefix, mod):
        for k, v in mod.state_dict().items():
            sd[f"{prefix}.{k}"] = v.detach().clone()

    def conv(prefix, cin, cout, k, **kw):
        put(prefix, torch.nn.Conv2d(cin, cout, k, **kw))

    def gn(prefix, c):
        put(prefix, torch.nn.GroupNorm(32, c))

    def layer(kind, p, cin, cout):
        if kind == "conv":
            conv(p, cin, cout, 3, padding=1)
        elif kind == "res":
            gn(p + ".conv1.0", cin)
            conv(p + ".conv1.2", cin, cout, 3, padding=1)
            put(p + ".time_emb.1", torch.nn.Linear(ted, cout))
            gn(p + ".conv2.0", cout)
            conv(p + ".conv2.3", cout, cout, 3, padding=1)
            if cin != cout:
                conv(p + ".shortcut", cin, cout, 1)
        elif kind == "attn":
            gn(p + ".norm", cin)
            conv(p + ".qkv", cin, 3 * cin, 1, bias=False)
            conv(p + ".proj", cin, cin, 1)
        elif kind == "down":
            conv(p + ".op", cin, cin, 3, stride=2, padding=1)
        elif kind == "up":
            conv(p + ".conv", cin, cin, 3, padding=1)

    put("time_embed.0", torch.nn.Linear(mc, ted))
    put("time_embed.2", torch.nn.Linear(ted, ted))
    down, middle, up = topology(hp)
    for st in down:
        for l in st:
            layer(*l)
    for l in middle:
        layer(*l)
    for st in up:
        for l in st:
            layer(*l)
    gn("out.0", mc)
    conv("out.2", mc, hp["out_channels"], 3, padding=1)
    return sd


def state_dict_digest(sd):
    """Order-independent fingerprint: per-key float64 sum and abs-sum."""
    return {k: (float(v.double().sum()), float(v.double().abs().sum())) for k, v in sd.items()}


# --------------------------------------------------------------------------- forward
def timestep_embedding(t, dim, max_period=10000):
    """diff_model.py:16-33 — cos block first, then sin."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn_silu(sd, p, x):
    return F.silu(F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5))


def _res(sd, p, x, emb):
    """ResidualBlock.forward, diff_model.py:94-103 (dropout inactive in eval)."""
    h = F.conv2d(_gn_silu(sd, p + ".conv1.0", x), sd[p + ".conv1.2.weight"], sd[p + ".conv1.2.bias"], padding=1)
    h = h + F.linear(F.silu(emb), sd[p + ".time_emb.1.weight"], sd[p + ".time_emb.1.bias"])[:, :, None, None]
    h = F.conv2d(_gn_silu(sd, p + ".conv2.0", h), sd[p + ".conv2.3.weight"], sd[p + ".conv2.3.bias"], padding=1)
    if p + ".shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".shortcut.weight"], sd[p + ".shortcut.bias"])
    return h + x


def _attn(sd, p, x, heads):
    """AttentionBlock.forward, diff_model.py:117-127."""
    B, C, H, W = x.shape
    qkv = F.conv2d(F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5), sd[p + ".qkv.weight"])
    q, k, v = qkv.reshape(B * heads, -1, H * W).chunk(3, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(C // heads))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale).softmax(dim=-1)
    h = torch.einsum("bts,bcs->bct", w, v).reshape(B, -1, H, W)
    return F.conv2d(h, sd[p + ".proj.weight"], sd[p + ".proj.bias"]) + x


def _layer(sd, hp, l, h, emb):
    kind, p, _, _ = l
    if kind == "conv":
        return F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
    if kind == "res":
        return _res(sd, p, h, emb)
    if kind == "attn":
        return _attn(sd, p, h, hp["num_heads"])
    if kind == "down":
        return F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
    if kind == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        return F.conv2d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
    raise ValueError(kind)


@torch.no_grad()
def unet_forward(sd, hp, x, t, taps=None):
    """UNetModel.forward, diff_model.py:245-267.  ``taps`` (dict) collects block outputs."""
    emb = timestep_embedding(t, hp["model_channels"])
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    down, middle, up = topology(hp)
    hs = []
    h = x
    for st in down:
        for l in st:
            h = _layer(sd, hp, l, h, emb)
            if taps is not None:
                taps[l[1]] = h
        hs.append(h)
    for l in middle:
        h = _layer(sd, hp, l, h, emb)
        if taps is not None:
            taps[l[1]] = h
    for st in up:
        h = torch.cat([h, hs.pop()], dim=1)
        for l in st:
            h = _layer(sd, hp, l, h, emb)
            if taps is not None:
                taps[l[1]] = h
    h = _gn_silu(sd, "out.0", h)
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


# --------------------------------------------------------------------------- forward with 16-bit STORAGE (round 3)
def _stored_gn(sd, p, x, R, act):
    """GroupNorm(32) [+ SiLU] of a STORED tensor, stored again: statistics of the rounded values (the conv epilogues sum what
    they store), coefficient arithmetic of csrc/groupnorm.hip (scale = rstd * gamma, shift = beta - mean * scale, y = x * scale +
    shift in f32), one rounding of the result."""
    B, C, H, W = x.shape
    xg = x.double().reshape(B, 32, -1)
    mean = xg.mean(-1)
    var = ((xg * xg).mean(-1) - mean * mean).clamp_min(0.0)
    rstd = (1.0 / torch.sqrt(var + 1e-5)).float()
    cpg = C // 32
    scale = rstd.repeat_interleave(cpg, 1) * sd[p + ".weight"][None]
    shift = sd[p + ".bias"][None] - mean.float().repeat_interleave(cpg, 1) * scale
    v = x * scale[:, :, None, None] + shift[:, :, None, None]
    return R(v * torch.sigmoid(v)) if act else R(v)


def _stored_subpixel_upsample(x, w, bias, R, conv2d=F.conv2d):
    """Upsample (nearest x2, then 3x3; diff_model.py:129-140) the way the plan runs it on maps that are multiples of 16: per
    output parity (a, b) a 2x2 conv of the LOW-resolution input whose weights are the coinciding taps summed in f32 and rounded
    ONCE (engine.pack_subpixel_upsample_weight)."""
    B, C, H, W = x.shape
    rows = (torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2]], 2), torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2]], 2))
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.empty(B, w.shape[0], 2 * H, 2 * W)
    for a in (0, 1):
        r = rows[a]
        for b in (0, 1):
            k = torch.stack([r[..., 0], r[..., 1] + r[..., 2]], 3) if b == 0 else torch.stack([r[..., 0] + r[..., 1], r[..., 2]], 3)
            out[:, :, a::2, b::2] = conv2d(xp[:, :, a:a + H + 1, b:b + W + 1], R(k))
    return out + bias[None, :, None, None]


@torch.no_grad()
def unet_forward_stored(sd, hp, x, t, storage="bf16", taps=None, exact_sums=False):
    """UNetModel.forward (diff_model.py:245-267) with fp32 ARITHMETIC and 16-bit STORAGE: every activation and weight is rounded
    to ``storage`` exactly where the HIP plan stores it (advshadow_amd/diff_model.py: emit_unet_forward) and nowhere else --
      * conv / Linear-as-conv weights, and the network input of the first conv; the sub-pixel Upsample weights after the tap sums;
      * each conv's output after bias, time embedding, fused 1x1 shortcut or identity residual were added in f32;
      * each GroupNorm (+ SiLU) output, computed from the statistics of the STORED input;
      * attention: q * log2(e)/sqrt(d) once more (csrc/attention.hip: attn2_kernel), the probabilities as the P.V operand
        (relative to the row maximum; the kernel's lagged maximum moves those roundings, not their size), the output;
    time embedding, accumulations, softmax sums, the last conv's output stay f32.  What this oracle differs by from the HIP 16-bit
    modes is therefore summation order and v_exp / v_rcp rounding only: a comparison against it separates 16-bit ROUNDING (shared)
    from a kernel bug (not shared), which the comparison against the fp32 oracle cannot.
    ``exact_sums``: the convolutions and the attention products accumulate in f64 (then one rounding to f32, then the storage
    rounding) -- same rounding POINTS, another summation: the distance between the two settings is the floor below which no
    16-bit evaluation of this network can be pinned (a random-init net amplifies single flipped roundings)."""
    st = {"bf16": torch.bfloat16, "fp16": torch.float16}[storage]
    R = lambda a: a.to(st).float()
    if exact_sums:
        def conv2d(a, w, b=None, **kw):
            return torch.nn.functional.conv2d(a.double(), w.double(), None if b is None else b.double(), **kw).float()

        def einsum(eq, a, b):
            return torch.einsum(eq, a.double(), b.double()).float()
    else:
        conv2d, einsum = torch.nn.functional.conv2d, torch.einsum
    heads = hp["num_heads"]
    emb = timestep_embedding(t, hp["model_channels"])
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])

    def res(p, x):
        a1 = _stored_gn(sd, p + ".conv1.0", x, R, True)
        te = F.linear(F.silu(emb), sd[p + ".time_emb.1.weight"], sd[p + ".time_emb.1.bias"])
        h = R(conv2d(a1, R(sd[p + ".conv1.2.weight"]), sd[p + ".conv1.2.bias"], padding=1) + te[:, :, None, None])
        a2 = _stored_gn(sd, p + ".conv2.0", h, R, True)
        y = conv2d(a2, R(sd[p + ".conv2.3.weight"]), sd[p + ".conv2.3.bias"], padding=1)
        if p + ".shortcut.weight" in sd:
            return R(y + conv2d(x, R(sd[p + ".shortcut.weight"]), sd[p + ".shortcut.bias"]))
        return R(y + x)

    def attn(p, x):
        B, C, H, W = x.shape
        n = _stored_gn(sd, p + ".norm", x, R, False)
        qkv = R(conv2d(n, R(sd[p + ".qkv.weight"])))
        q, k, v = qkv.reshape(B * heads, -1, H * W).chunk(3, dim=1)
        d = C // heads
        qs = R(q * (math.log2(math.e) / math.sqrt(d)))
        s = einsum("bct,bcs->bts", qs, k)                             # log2-domain scores
        pr = torch.exp2(s - s.max(dim=-1, keepdim=True).values)
        o = einsum("bts,bcs->bct", R(pr), v) / pr.sum(-1)[:, None, :]
        o = R(o).reshape(B, -1, H, W)
        return R(conv2d(o, R(sd[p + ".proj.weight"]), sd[p + ".proj.bias"]) + x)

    def layer(l, h):
        kind, p, _, _ = l
        if kind == "conv":
            return R(conv2d(R(h), R(sd[p + ".weight"]), sd[p + ".bias"], padding=1))
        if kind == "res":
            return res(p, h)
        if kind == "attn":
            return attn(p, h)
        if kind == "down":
            return R(conv2d(h, R(sd[p + ".op.weight"]), sd[p + ".op.bias"], stride=2, padding=1))
        if kind == "up":
            if h.shape[2] % 16 == 0 and h.shape[3] % 16 == 0:
                return R(_stored_subpixel_upsample(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], R, conv2d))
            return R(conv2d(F.interpolate(h, scale_factor=2, mode="nearest"), R(sd[p + ".conv.weight"]), sd[p + ".conv.bias"], padding=1))
        raise ValueError(kind)

    down, middle, up = topology(hp)
    hs, h = [], x
    for stg in down:
        for l in stg:
            h = layer(l, h)
            if taps is not None:
                taps[l[1]] = h
        hs.append(h)
    for l in middle:
        h = layer(l, h)
        if taps is not None:
            taps[l[1]] = h
    for stg in up:
        h = torch.cat([h, hs.pop()], dim=1)
        for l in stg:
            h = layer(l, h)
            if taps is not None:
                taps[l[1]] = h
    a = _stored_gn(sd, "out.0", h, R, True)
    return conv2d(a, R(sd["out.2.weight"]), sd["out.2.bias"], padding=1)


# --------------------------------------------------------------------------- schedules / sampler
def betas_linear(T):
    """diff_model.py:269-273 (float64)."""
    s = 1000 / T
    return torch.linspace(s * 0.0001, s * 0.02, T, dtype=torch.float64)


def betas_cosine(T, s=0.008):
    """diff_model.py:275-285 (float64)."""
    x = torch.linspace(0, T, T + 1, dtype=torch.float64)
    ac = torch.cos(((x / T) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def alphas_cumprod(T=1000, schedule="cosine"):
    """GaussianDiffusion.__init__, diff_model.py:294-303 (float64 table)."""
    b = betas_linear(T) if schedule == "linear" else betas_cosine(T)
    return torch.cumprod(1.0 - b, dim=0)


def ddim_sequences(T, steps, method="uniform"):
    """diff_model.py:428-440: (seq, prev_seq) as int arrays."""
    if method == "uniform":
        c = T // steps
        seq = np.asarray(list(range(0, T, c)))
    elif method == "quad":
        seq = ((np.linspace(0, np.sqrt(T * .8), steps)) ** 2).astype(int)
    else:
        raise NotImplementedError(method)
    seq = seq + 1
    prev = np.append(np.array([0]), seq[:-1])
    return seq, prev


@torch.no_grad()
def ddim_sample(model_fn, x_T, T=1000, schedule="cosine", steps=50, method="uniform", eta=0.0,
                clip_denoised=True, noises=None, trace=None):
    """GaussianDiffusion.ddim_sample, diff_model.py:416-474, with x_T injected.

    ``model_fn(x, t_long[B]) -> eps``.  With eta=0 the per-step randn is
    multiplied by zero (:463-470), so the result is a function of x_T alone.
    ``noises`` optionally supplies the per-step randn for eta>0.
    Returns float32 ndarray like the reference (:474).
    """
    ac = alphas_cumprod(T, schedule)
    seq, prev = ddim_sequences(T, steps, method)
    x = x_T.clone()
    B = x.shape[0]
    n_loop = len(seq)
    for i in reversed(range(n_loop)):
        t = torch.full((B,), int(seq[i]), dtype=torch.long)
        a_t = ac[int(seq[i])].float().reshape(1, 1, 1, 1)       # _extract: f64 gather -> .float()
        a_p = ac[int(prev[i])].float().reshape(1, 1, 1, 1)
        eps = model_fn(x, t)
        x0 = (x - torch.sqrt(1.0 - a_t) * eps) / torch.sqrt(a_t)
        if clip_denoised:
            x0 = torch.clamp(x0, -1.0, 1.0)
        sig = eta * torch.sqrt((1 - a_p) / (1 - a_t) * (1 - a_t / a_p))
        dirx = torch.sqrt(1 - a_p - sig ** 2) * eps
        nz = noises[i] if noises is not None else torch.zeros_like(x)
        x = torch.sqrt(a_p) * x0 + dirx + sig * nz
        if trace is not None:
            trace.append((int(seq[i]), eps.clone(), x.clone()))
    return x.numpy()


# --------------------------------------------------------------------------- full-length ancestral sampler
def posterior_tables(T=1000, schedule="cosine"):
    """The tables of GaussianDiffusion.__init__ that p_sample reads (diff_model.py:296-331), f64 like the reference."""
    betas = betas_cosine(T) if schedule == "cosine" else betas_linear(T)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, 0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    return dict(sqrt_recip=torch.sqrt(1.0 / ac), sqrt_recipm1=torch.sqrt(1.0 / ac - 1),
                logvar=torch.log(torch.cat([pv[1:2], pv[1:]])),
                c1=betas * torch.sqrt(ac_prev) / (1.0 - ac), c2=(1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac))


@torch.no_grad()
def p_sample_loop(model_fn, x_T, noises, T=1000, schedule="cosine", clip_denoised=True):
    """GaussianDiffusion.p_sample_loop / p_sample / p_mean_variance (diff_model.py:361-408).  ``noises[i]`` is the
    ``randn_like`` of step i (drawn for every step, multiplied by 0 at i == 0).  Returns the list of T images."""
    tb = posterior_tables(T, schedule)
    ex = lambda a, t: a.gather(0, t).float().reshape(-1, 1, 1, 1)           # _extract (diff_model.py:333-338)
    img, out = x_T.clone(), []
    B = img.shape[0]
    for i in reversed(range(T)):
        t = torch.full((B,), i, dtype=torch.long)
        eps = model_fn(img, t)
        x_recon = ex(tb["sqrt_recip"], t) * img - ex(tb["sqrt_recipm1"], t) * eps
        if clip_denoised:
            x_recon = torch.clamp(x_recon, min=-1.0, max=1.0)
        mean = ex(tb["c1"], t) * x_recon + ex(tb["c2"], t) * img
        mask = (t != 0).float().view(-1, 1, 1, 1)
        img = mean + mask * (0.5 * ex(tb["logvar"], t)).exp() * noises[i]
        out.append(img.clone())
    return out
