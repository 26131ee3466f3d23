"""Drop-in for ``model/networks/cspdarkunet.py::CSPDarkUnet`` (generate()'s second ``--network``) on MI355X.

Same constructor, ``forward(x, time, y=None)`` and ``state_dict`` key names as the reference
(model/networks/base.py:17-68, cspdarkunet.py:17-115, modules/block.py:93-131, modules/module.py:20-116,
modules/conv.py:72-97).  The module only holds parameters; the forward replays a plan of HIP kernels on
NHWC activations:

* BaseConv = implicit-GEMM conv (1x1, 3x3, 3x3 stride 2) -> GroupNorm(1)+act; the Bottleneck's ``y + x``
  and the blocks' ``x + emb`` ride in that GroupNorm pass (residual / per-channel add AFTER the activation);
* CSPLayer's ``cat([x_1, x_2])`` is never materialised (the 1x1 conv reads two sources);
* CSPDarkUpBlock applies the SAME 1x1 BaseConv before the nearest x2 upsample and after the concat
  (block.py:125-131); ``up*.csp`` is constructed (it is in the state_dict) but unused, as in the reference;
* SelfAttention is the one of the UNet (heads of 8..128 channels).

The 32-channel layers of the default configuration are half a 128-byte K-slab in bf16: the conv kernels read
them with pixel stride 32 and K extent 64 against zero-padded weights (``advs_conv_args.ld1``).
"""
import torch
import torch.nn as nn

from ... import _lib
from ...diff_model import _attach
from ...engine import Builder, dtype_code, pack_conv_weight
from .unet import _KNOWN_ACTS, _UNetEngine

_DOWN_N = (1, 3, 3, 1)                     # CSPLayer depth of down1..down4 (cspdarkunet.py:29,36,43,50)
_UP_N = (3, 3, 3, 3)                       # ... of the (unused) up1..up4 CSP layers (cspdarkunet.py:57-78)


class CSPDarkUnet(nn.Module):
    def __init__(self, in_channel=3, out_channel=3, channel=None, time_channel=256, num_classes=None, image_size=64,
                 device="cpu", act="silu", compute_dtype="fp32", use_graph=True):
        super().__init__()
        self.in_channel, self.out_channel = in_channel, out_channel
        self.channel = channel if channel is not None else [32, 64, 128, 256, 512, 1024]
        self.time_channel, self.num_classes, self.image_size = time_channel, num_classes, image_size
        self.device, self.act = device, act
        self.compute_dtype, self.use_graph = compute_dtype, use_graph
        ch = self.channel
        if num_classes is not None:
            _attach(self, "label_emb", nn.Embedding(num_classes, time_channel))
        self._base_conv("inc", in_channel, ch[0], 1)
        # (kind, prefix, cin, cout, n) in the reference's construction order
        self.blocks = []
        for i in range(1, 5):
            self.blocks += [("down", f"down{i}", ch[i - 1], ch[i], _DOWN_N[i - 1]), ("sa", f"sa{i}", ch[i], image_size >> i, 0)]
        for i in range(1, 5):
            self.blocks += [("up", f"up{i}", ch[5 - i], ch[4 - i], _UP_N[i - 1]),
                            ("sa", f"sa{4 + i}", ch[4 - i], image_size >> (4 - i), 0)]
        for kind, p, a, b, n in self.blocks:
            if kind == "down":
                self._base_conv(p + ".conv_csp.0", a, b, 3, 2)
                self._csp_layer(p + ".conv_csp.1", b, b, n)
                _attach(self, p + ".emb_layer.1", nn.Linear(time_channel, b))
            elif kind == "up":
                self._base_conv(p + ".conv", a, b, 1)
                self._csp_layer(p + ".csp", a, b, n)
                _attach(self, p + ".emb_layer.1", nn.Linear(time_channel, b))
            else:
                _attach(self, p + ".mha", nn.MultiheadAttention(a, 4, batch_first=True))
                _attach(self, p + ".ln", nn.LayerNorm([a]))
                _attach(self, p + ".ff_self.0", nn.LayerNorm([a]))
                _attach(self, p + ".ff_self.1", nn.Linear(a, a))
                _attach(self, p + ".ff_self.3", nn.Linear(a, a))
        _attach(self, "outc", nn.Conv2d(ch[0], out_channel, 1))
        self._packed, self._engines = {}, {}

    def _base_conv(self, p, cin, cout, k, stride=1):
        _attach(self, p + ".conv", nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False))
        _attach(self, p + ".gn", nn.GroupNorm(1, cout))

    def _csp_layer(self, p, cin, cout, n):
        mid = int(cout * 0.5)
        self._base_conv(p + ".conv1", cin, mid, 1)
        self._base_conv(p + ".conv2", cin, mid, 1)
        self._base_conv(p + ".conv3", 2 * mid, cout, 1)
        for i in range(n):
            self._base_conv(f"{p}.m.{i}.conv1", mid, mid, 1)
            self._base_conv(f"{p}.m.{i}.conv2", mid, mid, 3)

    # ---- packed weights ---------------------------------------------------------------------------
    def _version(self):
        dev = next(self.parameters()).device
        return (str(dev), sum(p._version for p in self.parameters()))

    def packed_weights(self, dt):
        ver = self._version()
        hit = self._packed.get(dt)
        if hit is not None and hit[0] == ver:
            return hit[1]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AdvsError(f"CSPDarkUnet parameters are on {dev}: move the model to the GPU; the HIP path has no CPU fallback")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        f32 = lambda k: sd[k].float().contiguous()
        W = {}

        def bc(p, sources=None):
            W[p + ".w"] = pack_conv_weight(sd[p + ".conv.weight"], dt, sources=sources)
            W[p + ".g"], W[p + ".b"] = f32(p + ".gn.weight"), f32(p + ".gn.bias")

        # stem: 1x1 from the NCHW f32 input, run by the 3x3 first-conv kernel with the tap at the centre
        w1 = f32("inc.conv.weight")
        w3 = torch.zeros((w1.shape[0], w1.shape[1], 3, 3), dtype=torch.float32, device=dev)
        w3[:, :, 1, 1] = w1[:, :, 0, 0]
        W["inc.w"], W["inc.g"], W["inc.b"] = w3.contiguous(), f32("inc.gn.weight"), f32("inc.gn.bias")
        emb_w, emb_b, off = [], [], 0
        W["emb_off"] = {}
        for kind, p, a, b, n in self.blocks:
            if kind == "down":
                bc(p + ".conv_csp.0")
                q = p + ".conv_csp.1"
                bc(q + ".conv1"); bc(q + ".conv2")
                bc(q + ".conv3", sources=(b // 2, b // 2))                # reads cat([x_1, x_2]) as two sources
                for i in range(n):
                    bc(f"{q}.m.{i}.conv1"); bc(f"{q}.m.{i}.conv2")
            elif kind == "up":
                bc(p + ".conv")
            if kind in ("down", "up"):
                emb_w.append(f32(p + ".emb_layer.1.weight")); emb_b.append(f32(p + ".emb_layer.1.bias"))
                W["emb_off"][p] = off
                off += b
            if kind == "sa":
                W[p + ".in_w"] = pack_conv_weight(sd[p + ".mha.in_proj_weight"].reshape(3 * a, a, 1, 1), dt)
                W[p + ".in_b"] = f32(p + ".mha.in_proj_bias")
                W[p + ".out_w"] = pack_conv_weight(sd[p + ".mha.out_proj.weight"].reshape(a, a, 1, 1), dt)
                W[p + ".out_b"] = f32(p + ".mha.out_proj.bias")
                for nm in ("ln", "ff_self.0"):
                    W[p + "." + nm + ".g"], W[p + "." + nm + ".b"] = f32(p + "." + nm + ".weight"), f32(p + "." + nm + ".bias")
                for nm in ("ff_self.1", "ff_self.3"):
                    W[p + "." + nm + ".w"] = pack_conv_weight(sd[p + "." + nm + ".weight"].reshape(a, a, 1, 1), dt)
                    W[p + "." + nm + ".b"] = f32(p + "." + nm + ".bias")
        W["emb_w"], W["emb_b"], W["emb_total"] = torch.cat(emb_w, 0).contiguous(), torch.cat(emb_b, 0).contiguous(), off
        W["outc.w"], W["outc.b"] = f32("outc.weight"), f32("outc.bias")
        if self.num_classes is not None:
            W["label_emb"] = f32("label_emb.weight")
        tc = self.time_channel
        W["inv_freq"] = (1.0 / (10000 ** (torch.arange(0, tc, 2).float() / tc))).to(dev)      # base.py:63
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[1] == dt]:
            del self._engines[key]
        return W

    def engine(self, batch, dtype=None):
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        eng = self._engines.get((batch, dt))
        if eng is None:
            eng = _UNetEngine(self, W, batch, dt, emit=emit_cspdarkunet_forward)
            self._engines[(batch, dt)] = eng
        return eng

    def forward(self, x, time, y=None):
        """x [B,3,S,S] f32, time [B] long, y [B] long or None -> eps [B,3,S,S] (cspdarkunet.py:81-115)."""
        B, _, H, Wd = x.shape
        if H != self.image_size or Wd != self.image_size:
            raise ValueError(f"input is {H}x{Wd} but the network was built for image_size={self.image_size} "
                             f"(SelfAttention.size is baked in, attention.py:23,46)")
        eng = self.engine(B)
        cur = torch.cuda.current_stream(x.device)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x.to(torch.float32), non_blocking=True)
            eng.t.copy_(time.to(torch.int64), non_blocking=True)
            if y is not None:
                eng.labels.copy_(y.to(torch.int64), non_blocking=True)
            eng.run("cond" if y is not None else "uncond")
            out = (eng.eps_c if y is not None else eng.eps_u).clone()
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        return out


def emit_cspdarkunet_forward(bld, net, W, x_nchw, t_dev, labels_dev, eps_out):
    """Append one CSPDarkUnet forward (cspdarkunet.py:81-115) to ``bld.plan``."""
    act = net.act if net.act in _KNOWN_ACTS else "silu"
    table = W["label_emb"] if labels_dev is not None else None
    temb = bld.timestep_embedding(t_dev, W["inv_freq"], cos_first=False, table=table, labels=labels_dev)
    emb = bld.linear(temb, W["emb_w"], W["emb_b"], act_in="silu")            # every emb_layer, stacked
    estride = W["emb_total"]

    def base_conv(p, x, k=1, stride=1, x2=None, residual=None, emb_slice=None):
        """BaseConv.forward (conv.py:96-97) [+ residual] [+ emb], both after the activation."""
        cout = W[p + ".g"].numel()
        h = bld.conv(x, W[p + ".w"], cout, x2=x2, ksize=k, stride=stride, pad=(k - 1) // 2)
        y = bld.groupnorm(h, W[p + ".g"], W[p + ".b"], 1, act=act, residual=residual, residual_after_act=True,
                          chan_add=emb_slice, chan_add_stride=estride if emb_slice is not None else 0)
        bld.free(h)
        return y

    def csp_layer(p, x, n, emb_slice):
        """CSPLayer.forward (module.py:111-116); Bottleneck.forward (module.py:42-47)."""
        x1 = base_conv(p + ".conv1", x)
        x2 = base_conv(p + ".conv2", x)
        for i in range(n):
            m = base_conv(f"{p}.m.{i}.conv1", x1)
            nx = base_conv(f"{p}.m.{i}.conv2", m, k=3, residual=x1)
            bld.free(m); bld.free(x1)
            x1 = nx
        y = base_conv(p + ".conv3", x1, x2=x2, emb_slice=emb_slice)          # cat([x_1, x_2]) on load
        bld.free(x1); bld.free(x2)
        return y

    def self_attention(p, x, C):
        d = C // 4
        ln = bld.layernorm(x, W[p + ".ln.g"], W[p + ".ln.b"])
        qkv = bld.conv(ln, W[p + ".in_w"], 3 * C, bias=W[p + ".in_b"], ksize=1, pad=0)
        bld.free(ln)
        o = bld.attention(qkv, 4, d, 0, C, 2 * C, d)
        bld.free(qkv)
        a = bld.conv(o, W[p + ".out_w"], C, bias=W[p + ".out_b"], residual=x, ksize=1, pad=0)
        bld.free(o)
        f = bld.layernorm(a, W[p + ".ff_self.0.g"], W[p + ".ff_self.0.b"])
        g = bld.conv(f, W[p + ".ff_self.1.w"], C, bias=W[p + ".ff_self.1.b"], act=act, ksize=1, pad=0)
        bld.free(f)
        y = bld.conv(g, W[p + ".ff_self.3.w"], C, bias=W[p + ".ff_self.3.b"], residual=a, ksize=1, pad=0)
        bld.free(g); bld.free(a)
        return y

    # inc: 1x1 BaseConv straight from the NCHW input
    h0 = bld.conv_first(x_nchw, W["inc.w"], None, W["inc.g"].numel())
    x1 = bld.groupnorm(h0, W["inc.g"], W["inc.b"], 1, act=act)
    bld.free(h0)
    skips = [x1]                                                              # x1, x2_sa, x3_sa, x4_sa
    h = x1
    for kind, p, a, b, n in net.blocks:
        if kind == "down":
            o = W["emb_off"][p]
            c = base_conv(p + ".conv_csp.0", h, k=3, stride=2)
            new = csp_layer(p + ".conv_csp.1", c, n, emb[:, o:o + b])
            bld.free(c)
        elif kind == "up":
            o = W["emb_off"][p]
            skip = skips.pop()
            if skip.shape[3] + b != a:
                raise ValueError(f"{p}: cat([skip, x]) has {skip.shape[3]}+{b} channels but {p}.conv takes {a} (block.py:128-129)")
            low = base_conv(p + ".conv", h)
            cat = bld.concat_nearest2x(skip, low)
            bld.free(low); bld.free(skip)
            new = base_conv(p + ".conv", cat, emb_slice=emb[:, o:o + b])
            bld.free(cat)
        else:
            new = self_attention(p, h, a)
        if not any(h is s for s in skips):
            bld.free(h)
        if kind == "sa" and p in ("sa1", "sa2", "sa3"):
            skips.append(new)
        h = new
    bld.conv_last(h, W["outc.w"], W["outc.b"], net.out_channel, 1, eps_out)
    bld.free(h)
