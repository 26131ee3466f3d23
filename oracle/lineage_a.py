"""Oracle: the IDDM lineage (``model/networks/unet.py`` + ``model/samples/ddim.py``) on CPU.

Test infrastructure only (see ``oracle/__init__.py``).  Pinned by ``tests/golden/lineage_a_*.npz``
(generated from the imported reference).  Evaluated functionally over a ``state_dict`` with the
reference's key names.
"""
import math

import torch
import torch.nn.functional as F

CHANNEL = [32, 64, 128, 256, 512, 1024]      # BaseNet.init_channel default (base.py:48-51)


def _act(name):
    """model/modules/activation.py:16-36 (module form)."""
    return {"relu": F.relu, "relu6": F.relu6, "silu": F.silu, "gelu": F.gelu,
            "lrelu": lambda v: F.leaky_relu(v, 0.1)}.get(name, F.silu)


def _res_act(name):
    """DoubleConv.forward's residual branch (conv.py:53-66): note F.leaky_relu default slope 0.01."""
    return {"relu": F.relu, "relu6": F.relu6, "silu": F.silu, "gelu": F.gelu,
            "lrelu": lambda v: F.leaky_relu(v)}.get(name, F.silu)


# --------------------------------------------------------------------------- seeded init
def init_state_dict(seed, num_classes=37, act="silu", in_channel=3, out_channel=3, time_channel=256):
    """Same parameters as ``UNet(num_classes=..., act=...)`` built under ``torch.manual_seed(seed)``:
    stock torch modules created in the reference's order (base.py:39-40, unet.py:35-92, block.py,
    conv.py:37-43, attention.py:24-32)."""
    torch.manual_seed(seed)
    sd = {}
    ch = CHANNEL

    def put(prefix, mod):
        for k, v in mod.state_dict().items():
            sd[f"{prefix}.{k}"] = v.detach().clone()

    def double_conv(p, cin, cout, mid=None):
        mid = mid or cout
        put(p + ".double_conv.0", torch.nn.Conv2d(cin, mid, 3, padding=1, bias=False))
        put(p + ".double_conv.1", torch.nn.GroupNorm(1, mid))
        put(p + ".double_conv.3", torch.nn.Conv2d(mid, cout, 3, padding=1, bias=False))
        put(p + ".double_conv.4", torch.nn.GroupNorm(1, cout))

    def down(p, cin, cout):
        double_conv(p + ".maxpool_conv.1", cin, cin)
        double_conv(p + ".maxpool_conv.2", cin, cout)
        put(p + ".emb_layer.1", torch.nn.Linear(time_channel, cout))

    def up(p, cin, cout):
        double_conv(p + ".conv.0", cin, cin)
        double_conv(p + ".conv.1", cin, cout, cin // 2)
        put(p + ".emb_layer.1", torch.nn.Linear(time_channel, cout))

    def sa(p, c):
        put(p + ".mha", torch.nn.MultiheadAttention(c, 4, batch_first=True))
        put(p + ".ln", torch.nn.LayerNorm([c]))
        put(p + ".ff_self.0", torch.nn.LayerNorm([c]))
        put(p + ".ff_self.1", torch.nn.Linear(c, c))
        put(p + ".ff_self.3", torch.nn.Linear(c, c))

    if num_classes is not None:
        put("label_emb", torch.nn.Embedding(num_classes, time_channel))
    double_conv("inc", in_channel, ch[1])
    down("down1", ch[1], ch[2]); sa("sa1", ch[2])
    down("down2", ch[2], ch[3]); sa("sa2", ch[3])
    down("down3", ch[3], ch[3]); sa("sa3", ch[3])
    double_conv("bot1", ch[3], ch[4]); double_conv("bot2", ch[4], ch[4]); double_conv("bot3", ch[4], ch[3])
    up("up1", ch[4], ch[2]); sa("sa4", ch[2])
    up("up2", ch[3], ch[1]); sa("sa5", ch[1])
    up("up3", ch[2], ch[1]); sa("sa6", ch[1])
    put("outc", torch.nn.Conv2d(ch[1], out_channel, 1))
    return sd


# --------------------------------------------------------------------------- forward
def pos_encoding(t, channels):
    """BaseNet.pos_encoding (base.py:56-68): [sin | cos], t is [B,1] float."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))
    v = t.repeat(1, channels // 2) * inv_freq
    return torch.cat([torch.sin(v), torch.cos(v)], dim=-1)


def _double_conv(sd, p, x, act, residual=False):
    """DoubleConv.forward (conv.py:46-69)."""
    h = F.conv2d(x, sd[p + ".double_conv.0.weight"], padding=1)
    h = _act(act)(F.group_norm(h, 1, sd[p + ".double_conv.1.weight"], sd[p + ".double_conv.1.bias"], eps=1e-5))
    h = F.conv2d(h, sd[p + ".double_conv.3.weight"], padding=1)
    h = F.group_norm(h, 1, sd[p + ".double_conv.4.weight"], sd[p + ".double_conv.4.bias"], eps=1e-5)
    return _res_act(act)(x + h) if residual else h


def _emb(sd, p, t):
    return F.linear(F.silu(t), sd[p + ".emb_layer.1.weight"], sd[p + ".emb_layer.1.bias"])[:, :, None, None]


def _down(sd, p, x, t, act):
    """DownBlock.forward (block.py:39-49)."""
    x = F.max_pool2d(x, 2)
    x = _double_conv(sd, p + ".maxpool_conv.1", x, act, residual=True)
    x = _double_conv(sd, p + ".maxpool_conv.2", x, act)
    return x + _emb(sd, p, t)


def _up(sd, p, x, skip, t, act):
    """UpBlock.forward (block.py:78-90): cat([skip, x])."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = torch.cat([skip, x], dim=1)
    x = _double_conv(sd, p + ".conv.0", x, act, residual=True)
    x = _double_conv(sd, p + ".conv.1", x, act)
    return x + _emb(sd, p, t)


def _sa(sd, p, x, act, heads=4):
    """SelfAttention.forward (attention.py:40-53) with nn.MultiheadAttention written out."""
    B, C, H, W = x.shape
    tok = x.view(B, C, H * W).swapaxes(1, 2)
    ln = F.layer_norm(tok, (C,), sd[p + ".ln.weight"], sd[p + ".ln.bias"], eps=1e-5)
    qkv = F.linear(ln, sd[p + ".mha.in_proj_weight"], sd[p + ".mha.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    d = C // heads
    q, k, v = (u.reshape(B, H * W, heads, d).transpose(1, 2) for u in (q, k, v))
    N = H * W
    if B * heads * N * N <= (1 << 28):
        w = torch.softmax((q / math.sqrt(d)) @ k.transpose(-1, -2), dim=-1)
        a = w @ v
    else:
        # the same row-wise softmax, a block of queries at a time: at 256x256 the [4B, N, N] weights the reference
        # materialises (attention.py:46-47) are 64 GiB per image for sa6 (N = 65 536)
        a = torch.empty_like(q)
        qs = q / math.sqrt(d)
        step = max(1, (1 << 26) // N)
        for i in range(0, N, step):
            a[:, :, i:i + step] = torch.softmax(qs[:, :, i:i + step] @ k.transpose(-1, -2), dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, H * W, C)
    a = F.linear(a, sd[p + ".mha.out_proj.weight"], sd[p + ".mha.out_proj.bias"]) + tok
    f = F.layer_norm(a, (C,), sd[p + ".ff_self.0.weight"], sd[p + ".ff_self.0.bias"], eps=1e-5)
    f = _act(act)(F.linear(f, sd[p + ".ff_self.1.weight"], sd[p + ".ff_self.1.bias"]))
    f = F.linear(f, sd[p + ".ff_self.3.weight"], sd[p + ".ff_self.3.bias"])
    return (f + a).swapaxes(2, 1).reshape(B, C, H, W)


@torch.no_grad()
def unet_forward(sd, x, time, y=None, act="silu", time_channel=256, taps=None):
    """UNet.forward (unet.py:95-128)."""
    t = pos_encoding(time.unsqueeze(-1).float(), time_channel)
    if y is not None:
        t = t + sd["label_emb.weight"][y]
    x1 = _double_conv(sd, "inc", x, act)
    x2 = _sa(sd, "sa1", _down(sd, "down1", x1, t, act), act)
    x3 = _sa(sd, "sa2", _down(sd, "down2", x2, t, act), act)
    x4 = _sa(sd, "sa3", _down(sd, "down3", x3, t, act), act)
    b = _double_conv(sd, "bot1", x4, act)
    b = _double_conv(sd, "bot2", b, act)
    b = _double_conv(sd, "bot3", b, act)
    u = _sa(sd, "sa4", _up(sd, "up1", b, x3, t, act), act)
    u = _sa(sd, "sa5", _up(sd, "up2", u, x2, t, act), act)
    u = _sa(sd, "sa6", _up(sd, "up3", u, x1, t, act), act)
    if taps is not None:
        taps.update(x1=x1, x2=x2, x3=x3, x4=x4, bot=b, last=u)
    return F.conv2d(u, sd["outc.weight"], sd["outc.bias"])


# --------------------------------------------------------------------------- DDIM (model/samples)
def alpha_hat(noise_steps=1000, beta_start=1e-4, beta_end=2e-2):
    """BaseDiffusion.__init__ with the linear schedule (base.py:29-45): all float32."""
    beta = torch.linspace(beta_start, beta_end, noise_steps)
    return torch.cumprod(1.0 - beta, dim=0)


def time_pairs(noise_steps=1000, sample_steps=500):
    """DDIMDiffusion.__init__ (ddim.py:44-46) as python ints."""
    ts = torch.arange(0, noise_steps, noise_steps // sample_steps).long() + 1
    ts = reversed(torch.cat((torch.tensor([0], dtype=torch.long), ts)))
    return [(int(a), int(b)) for a, b in zip(ts[:-1], ts[1:])]


@torch.no_grad()
def ddim_sample(model_fn, x_T, labels=None, cfg_scale=None, noise_steps=1000, sample_steps=500, eta=0.0,
                to_uint8=True, trace=None):
    """DDIMDiffusion.sample (ddim.py:48-100) with x_T injected.  ``model_fn(x, t, y_or_None)``.
    eta is 0 in the reference (ddim.py:42) so the per-step randn (ddim.py:72-75) never reaches x."""
    ah = alpha_hat(noise_steps)
    x = x_T.clone()
    n = x.shape[0]
    for i, p_i in time_pairs(noise_steps, sample_steps):
        t = (torch.ones(n) * i).long()
        a_t = ah[t][:, None, None, None]
        a_p = ah[(torch.ones(n) * p_i).long()][:, None, None, None]
        if labels is None and cfg_scale is None:
            eps = model_fn(x, t, None)
        else:
            eps = model_fn(x, t, labels)
            if cfg_scale > 0:
                eps = torch.lerp(model_fn(x, t, None), eps, cfg_scale)
        x0 = torch.clamp((x - (eps * torch.sqrt((1 - a_t)))) / torch.sqrt(a_t), -1, 1)
        c1 = eta * torch.sqrt((1 - a_t / a_p) * (1 - a_p) / (1 - a_t))
        c2 = torch.sqrt((1 - a_p) - c1 ** 2)
        x = torch.sqrt(a_p) * x0 + c2 * eps + c1 * torch.zeros_like(x)
        if trace is not None:
            trace.append((i, eps.clone(), x.clone()))
    if not to_uint8:
        return x
    x = (x + 1) * 0.5
    return (x * 255).type(torch.uint8)          # wraps mod 256, no clamp (ddim.py:97-99)


# --------------------------------------------------------------------------- DDPM / PLMS (model/samples)
@torch.no_grad()
def ddpm_sample(model_fn, x_T, noises, labels=None, cfg_scale=None, noise_steps=1000, beta_start=1e-4, beta_end=2e-2,
                steps=None):
    """DDPMDiffusion.sample (ddpm.py:62-98) with x_T and the per-step noise injected (``noises[i]`` for i > 1)."""
    beta = torch.linspace(beta_start, beta_end, noise_steps)
    alpha = 1.0 - beta
    ahat = torch.cumprod(alpha, dim=0)
    x = x_T.clone()
    n = x.shape[0]
    ts = list(reversed(range(1, noise_steps)))
    for i in (ts if steps is None else ts[:steps]):
        t = (torch.ones(n) * i).long()
        if labels is None and cfg_scale is None:
            eps = model_fn(x, t, None)
        else:
            eps = model_fn(x, t, labels)
            if cfg_scale > 0:
                eps = torch.lerp(model_fn(x, t, None), eps, cfg_scale)
        a, ah, b = alpha[t][:, None, None, None], ahat[t][:, None, None, None], beta[t][:, None, None, None]
        noise = noises[i] if i > 1 else torch.zeros_like(x)
        x = 1 / torch.sqrt(a) * (x - ((1 - a) / (torch.sqrt(1 - ah))) * eps) + torch.sqrt(b) * noise
    x = (x.clamp(-1, 1) + 1) / 2
    return (x * 255).type(torch.uint8)


@torch.no_grad()
def plms_sample(model_fn, x_T, labels=None, cfg_scale=None, noise_steps=1000, sample_steps=500, to_uint8=True):
    """PLMSDiffusion.sample (plms.py:63-121), eta = 0."""
    ah = alpha_hat(noise_steps)
    x = x_T.clone()
    n = x.shape[0]
    old = []
    for i, p_i in time_pairs(noise_steps, sample_steps):
        t, p_t = (torch.ones(n) * i).long(), (torch.ones(n) * p_i).long()
        a_t, a_p = ah[t][:, None, None, None], ah[p_t][:, None, None, None]
        if labels is None and cfg_scale is None:
            eps = model_fn(x, t, None)
        else:
            eps = model_fn(x, t, labels)
            if cfg_scale > 0:
                eps = torch.lerp(model_fn(x, t, None), eps, cfg_scale)
        c2 = torch.sqrt((1 - a_p))
        if len(old) == 0:
            x0 = torch.clamp((x - (eps * torch.sqrt((1 - a_t)))) / torch.sqrt(a_t), -1, 1)
            p_x = torch.sqrt(a_p) * x0 + c2 * eps
            nxt = model_fn(p_x, p_t, None if (labels is None and cfg_scale is None) else labels)
            prime = (eps + nxt) / 2
        elif len(old) == 1:
            prime = (3 * eps - old[-1]) / 2
        elif len(old) == 2:
            prime = (23 * eps - 16 * old[-1] + 5 * old[-2]) / 12
        else:
            prime = (55 * eps - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
        x0 = torch.clamp((x - (prime * torch.sqrt((1 - a_t)))) / torch.sqrt(a_t), -1, 1)
        x = torch.sqrt(a_p) * x0 + c2 * prime
        old.append(eps)
    if not to_uint8:
        return x
    x = (x + 1) * 0.5
    return (x * 255).type(torch.uint8)
