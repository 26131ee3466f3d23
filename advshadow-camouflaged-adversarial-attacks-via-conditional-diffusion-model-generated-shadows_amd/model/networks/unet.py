"""Drop-in for ``model/networks/unet.py::UNet`` (class-conditional eps-predictor) on MI355X.

Same constructor, ``forward(x, time, y=None)`` and ``state_dict`` key names as the reference
(model/networks/base.py:17-68, unet.py:17-128, modules/{conv,block,attention}.py).  The module
only holds parameters; the forward replays a plan of HIP kernels (NHWC activations):
DoubleConv = implicit-GEMM conv -> GroupNorm(1)+act -> conv -> GroupNorm(1) [+residual+act]
[+time/label embedding]; SelfAttention = LayerNorm -> in_proj (1x1 GEMM) -> flash attention ->
out_proj(+x) -> LayerNorm -> Linear+act -> Linear(+residual), all on [B, N, C] tokens, which IS the
NHWC activation (no transposes).
"""
import torch
import torch.nn as nn

from ... import _lib
from ...diff_model import _attach
from ...engine import Builder, dtype_code, pack_conv_weight, SLAB_ELEMS

_RES_ACT = {"lrelu": "lrelu001"}          # F.leaky_relu default slope in DoubleConv's residual branch (conv.py:61-62)
_KNOWN_ACTS = ("relu", "relu6", "silu", "lrelu", "gelu")


class UNet(nn.Module):
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
        # (kind, prefix, cin, cout) in the reference's construction order (unet.py:35-92)
        self.blocks = [("dc", "inc", in_channel, ch[1]),
                       ("down", "down1", ch[1], ch[2]), ("sa", "sa1", ch[2], image_size // 2),
                       ("down", "down2", ch[2], ch[3]), ("sa", "sa2", ch[3], image_size // 4),
                       ("down", "down3", ch[3], ch[3]), ("sa", "sa3", ch[3], image_size // 8),
                       ("dc", "bot1", ch[3], ch[4]), ("dc", "bot2", ch[4], ch[4]), ("dc", "bot3", ch[4], ch[3]),
                       ("up", "up1", ch[4], ch[2]), ("sa", "sa4", ch[2], image_size // 4),
                       ("up", "up2", ch[3], ch[1]), ("sa", "sa5", ch[1], image_size // 2),
                       ("up", "up3", ch[2], ch[1]), ("sa", "sa6", ch[1], image_size)]
        for kind, p, a, b in self.blocks:
            if kind == "dc":
                self._double_conv(p, a, b)
            elif kind == "down":
                self._double_conv(p + ".maxpool_conv.1", a, a)
                self._double_conv(p + ".maxpool_conv.2", a, b)
                _attach(self, p + ".emb_layer.1", nn.Linear(time_channel, b))
            elif kind == "up":
                self._double_conv(p + ".conv.0", a, a)
                self._double_conv(p + ".conv.1", a, b, a // 2)
                _attach(self, p + ".emb_layer.1", nn.Linear(time_channel, b))
            elif kind == "sa":
                _attach(self, p + ".mha", nn.MultiheadAttention(a, 4, batch_first=True))
                _attach(self, p + ".ln", nn.LayerNorm([a]))
                _attach(self, p + ".ff_self.0", nn.LayerNorm([a]))
                _attach(self, p + ".ff_self.1", nn.Linear(a, a))
                _attach(self, p + ".ff_self.3", nn.Linear(a, a))
        _attach(self, "outc", nn.Conv2d(ch[1], out_channel, 1))
        self._packed, self._engines = {}, {}

    def _double_conv(self, p, cin, cout, mid=None):
        mid = mid or cout
        _attach(self, p + ".double_conv.0", nn.Conv2d(cin, mid, 3, padding=1, bias=False))
        _attach(self, p + ".double_conv.1", nn.GroupNorm(1, mid))
        _attach(self, p + ".double_conv.3", nn.Conv2d(mid, cout, 3, padding=1, bias=False))
        _attach(self, p + ".double_conv.4", nn.GroupNorm(1, cout))

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
            raise _lib.AdvsError(f"UNet parameters are on {dev}: move the model to the GPU; the HIP path has no CPU fallback")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        f32 = lambda k: sd[k].float().contiguous()
        W = {}

        def dc(p, first=False):
            for i in ("1", "4"):
                W[p + ".gn" + i + ".g"] = f32(p + ".double_conv." + i + ".weight")
                W[p + ".gn" + i + ".b"] = f32(p + ".double_conv." + i + ".bias")
            W[p + ".c0"] = f32(p + ".double_conv.0.weight") if first else pack_conv_weight(sd[p + ".double_conv.0.weight"], dt)
            W[p + ".c3"] = pack_conv_weight(sd[p + ".double_conv.3.weight"], dt)

        emb_w, emb_b, off = [], [], 0
        W["emb_off"] = {}
        for kind, p, a, b in self.blocks:
            if kind == "dc":
                dc(p, first=(p == "inc"))
            elif kind in ("down", "up"):
                sub = (".maxpool_conv.1", ".maxpool_conv.2") if kind == "down" else (".conv.0", ".conv.1")
                dc(p + sub[0]); dc(p + sub[1])
                emb_w.append(f32(p + ".emb_layer.1.weight")); emb_b.append(f32(p + ".emb_layer.1.bias"))
                W["emb_off"][p] = off
                off += b
            elif kind == "sa":
                W[p + ".in_w"] = pack_conv_weight(sd[p + ".mha.in_proj_weight"].reshape(3 * a, a, 1, 1), dt)
                W[p + ".in_b"] = f32(p + ".mha.in_proj_bias")
                W[p + ".out_w"] = pack_conv_weight(sd[p + ".mha.out_proj.weight"].reshape(a, a, 1, 1), dt)
                W[p + ".out_b"] = f32(p + ".mha.out_proj.bias")
                for n in ("ln", "ff_self.0"):
                    W[p + "." + n + ".g"], W[p + "." + n + ".b"] = f32(p + "." + n + ".weight"), f32(p + "." + n + ".bias")
                for n in ("ff_self.1", "ff_self.3"):
                    W[p + "." + n + ".w"] = pack_conv_weight(sd[p + "." + n + ".weight"].reshape(a, a, 1, 1), dt)
                    W[p + "." + n + ".b"] = f32(p + "." + n + ".bias")
        W["emb_w"], W["emb_b"], W["emb_total"] = torch.cat(emb_w, 0).contiguous(), torch.cat(emb_b, 0).contiguous(), off
        W["outc.w"], W["outc.b"] = f32("outc.weight"), f32("outc.bias")
        if self.num_classes is not None:
            W["label_emb"] = f32("label_emb.weight")
        tc = self.time_channel
        # host table with the reference's own torch ops (base.py:63): 1 / 10000^(2i/channels)
        W["inv_freq"] = (1.0 / (10000 ** (torch.arange(0, tc, 2).float() / tc))).to(dev)
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[1] == dt]:
            del self._engines[key]
        return W

    def engine(self, batch, dtype=None):
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        eng = self._engines.get((batch, dt))
        if eng is None:
            eng = _UNetEngine(self, W, batch, dt)
            self._engines[(batch, dt)] = eng
        return eng

    def forward(self, x, time, y=None):
        """x [B,3,S,S] f32, time [B] long, y [B] long or None -> eps [B,3,S,S] (unet.py:95-128)."""
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


class _UNetEngine:
    """Frozen plans of one UNet forward for (batch, dtype): 'cond', 'uncond' and 'cfg' (both)."""

    def __init__(self, model, W, batch, dt, emit=None):
        self.emit = emit or emit_unet_forward                  # appends one forward to a Builder's plan
        dev = next(model.parameters()).device
        self.model, self.W, self.B, self.dt, self.dev = model, W, batch, dt, dev
        self.stream = torch.cuda.Stream(device=dev)
        S = model.image_size
        self.x = torch.zeros((batch, model.in_channel, S, S), dtype=torch.float32, device=dev)
        self.t = torch.zeros((batch,), dtype=torch.int64, device=dev)
        self.labels = torch.zeros((batch,), dtype=torch.int64, device=dev)
        self.eps_c = torch.zeros((batch, model.out_channel, S, S), dtype=torch.float32, device=dev)
        self.eps_u = torch.zeros_like(self.eps_c)
        self.plans, self.captured = {}, set()

    def plan(self, mode):
        pl = self.plans.get(mode)
        if pl is None:
            with torch.cuda.device(self.dev):
                bld = Builder(self.dev, self.dt, self.stream, self.B)
                if mode in ("cond", "cfg"):
                    if self.model.num_classes is None:
                        raise ValueError("labels were given but the network has no label_emb (num_classes=None)")
                    self.emit(bld, self.model, self.W, self.x, self.t, self.labels, self.eps_c)
                if mode in ("uncond", "cfg"):
                    self.emit(bld, self.model, self.W, self.x, self.t, None, self.eps_u)
                torch.cuda.synchronize(self.dev)
            pl = bld.plan
            self.plans[mode] = pl
        return pl

    def run(self, mode):
        pl = self.plan(mode)
        if self.model.use_graph and mode not in self.captured:
            pl.run_eager()
            self.stream.synchronize()
            pl.capture()
            self.captured.add(mode)
        pl.run()


def emit_unet_forward(bld, net, W, x_nchw, t_dev, labels_dev, eps_out):
    """Append one UNet forward (unet.py:95-128) to ``bld.plan``."""
    act = net.act if net.act in _KNOWN_ACTS else "silu"
    res_act = _RES_ACT.get(act, act)
    table = W["label_emb"] if labels_dev is not None else None
    temb = bld.timestep_embedding(t_dev, W["inv_freq"], cos_first=False, table=table, labels=labels_dev)
    emb = bld.linear(temb, W["emb_w"], W["emb_b"], act_in="silu")            # every emb_layer, stacked
    estride = W["emb_total"]

    def double_conv(p, x, cout_mid, cout, residual=False, emb_slice=None, first=False):
        if first:
            h = bld.conv_first(x_nchw, W[p + ".c0"], None, cout_mid)
        else:
            h = bld.conv(x, W[p + ".c0"], cout_mid)
        a = bld.groupnorm(h, W[p + ".gn1.g"], W[p + ".gn1.b"], 1, act=act)
        bld.free(h)
        h = bld.conv(a, W[p + ".c3"], cout)
        bld.free(a)
        y = bld.groupnorm(h, W[p + ".gn4.g"], W[p + ".gn4.b"], 1, act=res_act if residual else None,
                          residual=x if residual else None, chan_add=emb_slice,
                          chan_add_stride=estride if emb_slice is not None else 0)
        bld.free(h)
        return y

    def self_attention(p, x, C):
        d = C // 4
        ln = bld.layernorm(x, W[p + ".ln.g"], W[p + ".ln.b"])
        qkv = bld.conv(ln, W[p + ".in_w"], 3 * C, bias=W[p + ".in_b"], ksize=1, pad=0)
        bld.free(ln)
        o = bld.attention(qkv, 4, d, 0, C, 2 * C, d)          # in_proj rows: [Wq | Wk | Wv], head h at h*d
        bld.free(qkv)
        a = bld.conv(o, W[p + ".out_w"], C, bias=W[p + ".out_b"], residual=x, ksize=1, pad=0)
        bld.free(o)
        f = bld.layernorm(a, W[p + ".ff_self.0.g"], W[p + ".ff_self.0.b"])
        g = bld.conv(f, W[p + ".ff_self.1.w"], C, bias=W[p + ".ff_self.1.b"], act=act, ksize=1, pad=0)
        bld.free(f)
        y = bld.conv(g, W[p + ".ff_self.3.w"], C, bias=W[p + ".ff_self.3.b"], residual=a, ksize=1, pad=0)
        bld.free(g)
        bld.free(a)
        return y

    skips = {}
    h = None
    for kind, p, a, b in net.blocks:
        if kind == "dc":
            new = double_conv(p, h, b, b, first=(p == "inc"))
        elif kind == "down":
            o = W["emb_off"][p]
            pooled = bld.maxpool2(h)
            r = double_conv(p + ".maxpool_conv.1", pooled, a, a, residual=True)
            bld.free(pooled)
            new = double_conv(p + ".maxpool_conv.2", r, b, b, emb_slice=emb[:, o:o + b])
            bld.free(r)
        elif kind == "up":
            o = W["emb_off"][p]
            skip = skips[{"up1": "sa2", "up2": "sa1", "up3": "inc"}[p]]
            cat = bld.concat_upsample2x(skip, h)               # cat([skip, up(x)]) (block.py:86-87)
            r = double_conv(p + ".conv.0", cat, a, a, residual=True)
            bld.free(cat)
            new = double_conv(p + ".conv.1", r, a // 2, b, emb_slice=emb[:, o:o + b])
            bld.free(r)
        elif kind == "sa":
            new = self_attention(p, h, a)
        if h is not None and not any(h is s for s in skips.values()):
            bld.free(h)
        if p in ("inc", "sa1", "sa2"):
            skips[p] = new
        h = new
    for s in skips.values():
        bld.free(s)
    bld.conv_last(h, W["outc.w"], W["outc.b"], net.out_channel, 1, eps_out)
    bld.free(h)
