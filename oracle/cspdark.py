"""Oracle: ``model/networks/cspdarkunet.py::CSPDarkUnet`` (the second ``--network`` of generate()) on CPU.

Test infrastructure only (see ``oracle/__init__.py``).  Pinned by ``tests/golden/cspdark_*.npz``
(generated from the imported reference).  Evaluated functionally over a ``state_dict`` with the
reference's key names; the samplers are shared with ``oracle/lineage_a.py`` (same BaseDiffusion).
"""
import torch
import torch.nn.functional as F

from .lineage_a import CHANNEL, _act, _sa, pos_encoding

# n of the CSPLayer inside each block (cspdarkunet.py:29-74)
DOWN_N = {"down1": 1, "down2": 3, "down3": 3, "down4": 1}
UP_N = {"up1": 3, "up2": 3, "up3": 3, "up4": 3}


# --------------------------------------------------------------------------- seeded init
def init_state_dict(seed, num_classes=37, in_channel=3, out_channel=3, time_channel=256, channel=None):
    """Same parameters as ``CSPDarkUnet(num_classes=...)`` built under ``torch.manual_seed(seed)``: stock torch
    modules created in the reference's order (base.py:39-40, cspdarkunet.py:24-79, block.py:94-123,
    module.py:29-38,100-109, conv.py:90-94).  ``up*.csp`` exists in the state_dict but is never used by
    forward (block.py:125-131 calls ``self.conv`` twice)."""
    torch.manual_seed(seed)
    sd = {}
    ch = channel or CHANNEL

    def put(prefix, mod):
        for k, v in mod.state_dict().items():
            sd[f"{prefix}.{k}"] = v.detach().clone()

    def base_conv(p, cin, cout, k, stride=1):
        put(p + ".conv", torch.nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False))
        put(p + ".gn", torch.nn.GroupNorm(1, cout))

    def csp_layer(p, cin, cout, n):
        mid = int(cout * 0.5)
        base_conv(p + ".conv1", cin, mid, 1)
        base_conv(p + ".conv2", cin, mid, 1)
        base_conv(p + ".conv3", 2 * mid, cout, 1)
        for i in range(n):                                        # Bottleneck(mid, mid, expansion=1.0)
            base_conv(f"{p}.m.{i}.conv1", mid, mid, 1)
            base_conv(f"{p}.m.{i}.conv2", mid, mid, 3)

    def sa(p, c):
        put(p + ".mha", torch.nn.MultiheadAttention(c, 4, batch_first=True))
        put(p + ".ln", torch.nn.LayerNorm([c]))
        put(p + ".ff_self.0", torch.nn.LayerNorm([c]))
        put(p + ".ff_self.1", torch.nn.Linear(c, c))
        put(p + ".ff_self.3", torch.nn.Linear(c, c))

    def down(p, cin, cout, n):
        base_conv(p + ".conv_csp.0", cin, cout, 3, stride=2)
        csp_layer(p + ".conv_csp.1", cout, cout, n)
        put(p + ".emb_layer.1", torch.nn.Linear(time_channel, cout))

    def up(p, cin, cout, n):
        base_conv(p + ".conv", cin, cout, 1)
        csp_layer(p + ".csp", cin, cout, n)
        put(p + ".emb_layer.1", torch.nn.Linear(time_channel, cout))

    if num_classes is not None:
        put("label_emb", torch.nn.Embedding(num_classes, time_channel))
    base_conv("inc", in_channel, ch[0], 1)
    for i in range(1, 5):
        down(f"down{i}", ch[i - 1], ch[i], DOWN_N[f"down{i}"]); sa(f"sa{i}", ch[i])
    for i in range(1, 5):
        up(f"up{i}", ch[5 - i], ch[4 - i], UP_N[f"up{i}"]); sa(f"sa{4 + i}", ch[4 - i])
    put("outc", torch.nn.Conv2d(ch[0], out_channel, 1))
    return sd


# --------------------------------------------------------------------------- forward
def _base_conv(sd, p, x, act, stride=1):
    """BaseConv.forward (conv.py:96-97): act(gn(conv(x))), 'same' padding (conv.py:89)."""
    w = sd[p + ".conv.weight"]
    h = F.conv2d(x, w, stride=stride, padding=(w.shape[-1] - 1) // 2)
    return _act(act)(F.group_norm(h, 1, sd[p + ".gn.weight"], sd[p + ".gn.bias"], eps=1e-5))


def _csp_layer(sd, p, x, n, act):
    """CSPLayer.forward (module.py:111-116) with Bottleneck (module.py:42-47; in == out so use_add)."""
    x1 = _base_conv(sd, p + ".conv1", x, act)
    x2 = _base_conv(sd, p + ".conv2", x, act)
    for i in range(n):
        y = _base_conv(sd, f"{p}.m.{i}.conv2", _base_conv(sd, f"{p}.m.{i}.conv1", x1, act), act)
        x1 = y + x1
    return _base_conv(sd, p + ".conv3", torch.cat([x1, x2], dim=1), act)


def _emb(sd, p, t):
    return F.linear(F.silu(t), sd[p + ".emb_layer.1.weight"], sd[p + ".emb_layer.1.bias"])[:, :, None, None]


def _down(sd, p, x, t, act):
    """CSPDarkDownBlock.forward (block.py:106-109)."""
    x = _base_conv(sd, p + ".conv_csp.0", x, act, stride=2)
    x = _csp_layer(sd, p + ".conv_csp.1", x, DOWN_N[p], act)
    return x + _emb(sd, p, t)


def _up(sd, p, x, skip, t, act):
    """CSPDarkUpBlock.forward (block.py:125-131): the SAME 1x1 BaseConv before and after the concat."""
    x = _base_conv(sd, p + ".conv", x, act)
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = torch.cat([skip, x], dim=1)
    x = _base_conv(sd, p + ".conv", x, act)
    return x + _emb(sd, p, t)


@torch.no_grad()
def cspdarkunet_forward(sd, x, time, y=None, act="silu", time_channel=256, taps=None):
    """CSPDarkUnet.forward (cspdarkunet.py:81-115)."""
    t = pos_encoding(time.unsqueeze(-1).float(), time_channel)
    if y is not None:
        t = t + sd["label_emb.weight"][y]
    x1 = _base_conv(sd, "inc", x, act)
    x2 = _sa(sd, "sa1", _down(sd, "down1", x1, t, act), act)
    x3 = _sa(sd, "sa2", _down(sd, "down2", x2, t, act), act)
    x4 = _sa(sd, "sa3", _down(sd, "down3", x3, t, act), act)
    x5 = _sa(sd, "sa4", _down(sd, "down4", x4, t, act), act)
    u = _sa(sd, "sa5", _up(sd, "up1", x5, x4, t, act), act)
    u = _sa(sd, "sa6", _up(sd, "up2", u, x3, t, act), act)
    u = _sa(sd, "sa7", _up(sd, "up3", u, x2, t, act), act)
    u = _sa(sd, "sa8", _up(sd, "up4", u, x1, t, act), act)
    if taps is not None:
        taps.update(x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, last=u)
    return F.conv2d(u, sd["outc.weight"], sd["outc.bias"])
