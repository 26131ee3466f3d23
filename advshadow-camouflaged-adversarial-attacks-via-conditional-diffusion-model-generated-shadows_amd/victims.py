"""Victim classifiers of the attack-success evaluation, forward only, on the HIP kernels.

``ResNet50`` carries the torchvision / timm ``resnet50`` parameter names (ASR_fast.py:16-20 loads
``timm.create_model("resnet50", num_classes=37)``; ddim2/diff_model2.py:28 torchvision's), so a
``pytorch_model.bin`` of either loads with ``load_state_dict``.  BatchNorm is evaluated in eval mode
(running statistics folded into the conv weights): the reference's ``load_resnet50_model`` forgets
``.eval()`` and so runs batch statistics on a batch of one -- that accident is not
reproduced.
"""
import torch
import torch.nn as nn

from . import _lib
from .diff_model import _attach
from .engine import Builder, dtype_code, pack_conv_weight, ptr

_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))     # (width, blocks, stride of first block)


class ResNet50(nn.Module):
    def __init__(self, num_classes=37, compute_dtype="fp32", use_graph=True):
        super().__init__()
        self.num_classes, self.compute_dtype, self.use_graph = num_classes, compute_dtype, use_graph
        _attach(self, "conv1", nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False))
        _attach(self, "bn1", nn.BatchNorm2d(64))
        self.blocks = []                                  # (prefix, cin, width, cout, stride, has_downsample)
        cin = 64
        for li, (width, n, stride) in enumerate(_LAYERS, start=1):
            for bi in range(n):
                p = f"layer{li}.{bi}"
                s = stride if bi == 0 else 1
                cout = width * 4
                ds = s != 1 or cin != cout
                _attach(self, p + ".conv1", nn.Conv2d(cin, width, 1, bias=False))
                _attach(self, p + ".bn1", nn.BatchNorm2d(width))
                _attach(self, p + ".conv2", nn.Conv2d(width, width, 3, stride=s, padding=1, bias=False))
                _attach(self, p + ".bn2", nn.BatchNorm2d(width))
                _attach(self, p + ".conv3", nn.Conv2d(width, cout, 1, bias=False))
                _attach(self, p + ".bn3", nn.BatchNorm2d(cout))
                if ds:
                    _attach(self, p + ".downsample.0", nn.Conv2d(cin, cout, 1, stride=s, bias=False))
                    _attach(self, p + ".downsample.1", nn.BatchNorm2d(cout))
                self.blocks.append((p, cin, width, cout, s, ds))
                cin = cout
        _attach(self, "fc", nn.Linear(2048, num_classes))
        for m in self.modules():                          # torchvision's init (resnet.py): kaiming fan_out
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._packed, self._engines = {}, {}

    def _version(self):
        dev = next(self.parameters()).device
        return (str(dev), sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers()))

    @staticmethod
    def _fold(sd, conv, bn):
        w = sd[conv + ".weight"].float()
        scale = sd[bn + ".weight"].float() / torch.sqrt(sd[bn + ".running_var"].float() + 1e-5)
        return w * scale[:, None, None, None], (sd[bn + ".bias"].float() - sd[bn + ".running_mean"].float() * scale).contiguous()

    def packed_weights(self, dt):
        ver = self._version()
        hit = self._packed.get(dt)
        if hit is not None and hit[0] == ver:
            return hit[1]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AdvsError(f"ResNet50 parameters are on {dev}: move the model to the GPU; there is no CPU fallback")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        W = {}
        w, b = self._fold(sd, "conv1", "bn1")
        W["stem.w"], W["stem.b"] = w.contiguous(), b
        for p, cin, width, cout, s, ds in self.blocks:
            for i in ("1", "2", "3"):
                w, b = self._fold(sd, f"{p}.conv{i}", f"{p}.bn{i}")
                W[f"{p}.w{i}"], W[f"{p}.b{i}"] = pack_conv_weight(w, dt), b
            if ds:
                w, b = self._fold(sd, p + ".downsample.0", p + ".downsample.1")
                W[p + ".wd"], W[p + ".bd"] = pack_conv_weight(w, dt), b
        W["fc.w"], W["fc.b"] = sd["fc.weight"].float().contiguous(), sd["fc.bias"].float().contiguous()
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[2] == dt]:
            del self._engines[key]
        return W

    def engine(self, batch, size, dtype=None):
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        eng = self._engines.get((batch, size, dt))
        if eng is None:
            eng = _ResNetEngine(self, W, batch, size, dt)
            self._engines[(batch, size, dt)] = eng
        return eng

    def forward(self, x):
        """x [B,3,H,W] f32 on the GPU -> logits [B,num_classes] f32."""
        B, _, H, Wd = x.shape
        assert H == Wd, "square inputs only"
        eng = self.engine(B, H)
        cur = torch.cuda.current_stream(x.device)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x.to(torch.float32), non_blocking=True)
            eng.run()
            out = eng.logits.clone()
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        return out


class _ResNetEngine:
    def __init__(self, model, W, batch, size, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            self.x = torch.zeros((batch, 3, size, size), dtype=torch.float32, device=dev)
            lib = bld.lib
            ho = (size + 6 - 7) // 2 + 1
            h = bld.buf((batch, ho, ho, 64))
            bld.plan.add(lib.advs_conv_stem, ptr(self.x), ptr(W["stem.w"]), ptr(W["stem.b"]), ptr(h), batch, 3, size, size,
                         64, 7, 2, 3, _lib.ACT["relu"], dt, keep=(self.x, h))
            hp = (ho + 2 - 3) // 2 + 1
            pooled = bld.buf((batch, hp, hp, 64))
            bld.plan.add(lib.advs_maxpool3x3s2, ptr(h), ptr(pooled), batch, ho, ho, 64, dt, keep=(h, pooled))
            bld.free(h)
            h = pooled
            for p, cin, width, cout, s, ds in model.blocks:
                y1 = bld.conv(h, W[p + ".w1"], width, bias=W[p + ".b1"], ksize=1, pad=0, act="relu")
                y2 = bld.conv(y1, W[p + ".w2"], width, bias=W[p + ".b2"], ksize=3, stride=s, pad=1, act="relu")
                bld.free(y1)
                idn = bld.conv(h, W[p + ".wd"], cout, bias=W[p + ".bd"], ksize=1, stride=s, pad=0) if ds else h
                y3 = bld.conv(y2, W[p + ".w3"], cout, bias=W[p + ".b3"], residual=idn, ksize=1, pad=0, act="relu")
                bld.free(y2)
                if idn is not h:
                    bld.free(idn)
                bld.free(h)
                h = y3
            B_, hh, ww, cc = h.shape
            pooled = bld.buf((batch, cc), torch.float32)
            bld.plan.add(lib.advs_global_avgpool, ptr(h), ptr(pooled), batch, hh * ww, cc, dt, keep=(h, pooled))
            self.logits = bld.linear(pooled, W["fc.w"], W["fc.b"])
            self.plan, self.captured = bld.plan, False
            torch.cuda.synchronize(dev)

    def run(self):
        if self.model.use_graph and not self.captured:
            self.plan.run_eager()
            self.stream.synchronize()
            self.plan.capture()
            self.captured = True
        self.plan.run()
