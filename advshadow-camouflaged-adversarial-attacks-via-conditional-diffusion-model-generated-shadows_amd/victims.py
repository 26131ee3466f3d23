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
from .engine import SLAB_ELEMS, Builder, dtype_code, pack_conv_weight, ptr

_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))     # (width, blocks, stride of first block)


class _InputGradient:
    """Victims with a HIP backward-to-the-image plan (``grad_engine``)."""

    def input_gradient(self, x, labels):
        """x [B,3,H,W] f32, labels [B] int64 (GPU) -> (logits [B,K], d cross_entropy(logits_b, label_b) / d x_b [B,3,H,W]):
        what ``loss.backward(); image.grad`` yields for a batch of one in tools/train_shadow.py:204-212, per image."""
        B, _, H, Wd = x.shape
        assert H == Wd, "square inputs only"
        eng = self.grad_engine(B, H)
        cur = torch.cuda.current_stream(x.device)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x.to(torch.float32), non_blocking=True)
            eng.labels.copy_(labels.to(torch.int64).reshape(B), non_blocking=True)
            eng.run()
            out, grad = eng.logits.clone(), eng.grad.clone()
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        grad.record_stream(cur)
        return out, grad


class ResNet50(_InputGradient, nn.Module):
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
        # the stem as a GEMM over im2col columns (advs_im2col_nchw): [64][3*7*7 -> padded to whole slabs]
        W["stem.wg"] = pack_conv_weight(w.reshape(64, 147, 1, 1), dt)
        for p, cin, width, cout, s, ds in self.blocks:
            for i in ("1", "2", "3"):
                w, b = self._fold(sd, f"{p}.conv{i}", f"{p}.bn{i}")
                W[f"{p}.w{i}"], W[f"{p}.b{i}"] = pack_conv_weight(w, dt), b
            if ds:
                w, b = self._fold(sd, p + ".downsample.0", p + ".downsample.1")
                W[p + ".wd"], W[p + ".bd"] = pack_conv_weight(w, dt), b
        W["fc.w"], W["fc.b"] = sd["fc.weight"].float().contiguous(), sd["fc.bias"].float().contiguous()
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[-1] == dt]:
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

    def packed_grad_weights(self, dt):
        """Weights of the data-gradient convs: 1x1 transposed, 3x3 transposed and flipped (BatchNorm folded as in the
        forward), the classifier weight transposed."""
        ver = self._version()
        hit = self._packed.get(("grad", dt))
        if hit is not None and hit[0] == ver:
            return hit[1]
        self.packed_weights(dt)                              # device check + cache invalidation of the engines
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        G = {}
        for p, cin, width, cout, s, ds in self.blocks:
            for i in ("1", "2", "3"):
                w, _ = self._fold(sd, f"{p}.conv{i}", f"{p}.bn{i}")
                G[f"{p}.w{i}T"] = pack_conv_weight(w.permute(1, 0, 2, 3).flip(2, 3).contiguous(), dt)
            if s == 2:
                # the stride-2 3x3 data gradient as one 1x1 GEMM over the gathered output quadruple (csrc/victim_grad.hip): rows =
                # (input parity p = 2a + b, input channel), columns = (output offset q = 2 di + dj, output channel);
                # (a, di) -> ky: (0, 0) -> 1, (1, 0) -> 2, (1, 1) -> 0, (0, 1) never meets
                w2, _ = self._fold(sd, f"{p}.conv2", f"{p}.bn2")                  # [c'][c][3][3]
                tap = {(0, 0): 1, (1, 0): 2, (1, 1): 0}
                cw = w2.shape[0]
                wp = torch.zeros((4, w2.shape[1], 4, cw), dtype=torch.float32, device=w2.device)
                for (a, di), ky in tap.items():
                    for (b2, dj), kx in tap.items():
                        wp[2 * a + b2, :, 2 * di + dj, :] = w2[:, :, ky, kx].t()
                G[p + ".w2P"] = pack_conv_weight(wp.reshape(4 * w2.shape[1], 4 * cw, 1, 1), dt)
            if ds:
                w, _ = self._fold(sd, p + ".downsample.0", p + ".downsample.1")
                G[p + ".wdT"] = pack_conv_weight(w.permute(1, 0, 2, 3).contiguous(), dt)
        w, _ = self._fold(sd, "conv1", "bn1")
        kp = -(-147 // SLAB_ELEMS[dt]) * SLAB_ELEMS[dt]
        wt = torch.zeros((kp, 64), dtype=torch.float32, device=w.device)
        wt[:147] = w.reshape(64, 147).t()
        G["stem.wT"] = pack_conv_weight(wt.reshape(kp, 64, 1, 1), dt)     # column gradients = g_stem @ W, then advs_col2im_nchw
        G["fc.wT"] = sd["fc.weight"].float().t().contiguous()
        self._packed[("grad", dt)] = (ver, G)
        return G

    def grad_engine(self, batch, size, dtype=None):
        """Static plan of forward + backward-to-the-image (d cross_entropy / d x) for [batch,3,size,size] inputs."""
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W, G = self.packed_weights(dt), self.packed_grad_weights(dt)
        eng = self._engines.get(("grad", batch, size, dt))
        if eng is None:
            eng = _ResNetGradEngine(self, W, G, batch, size, dt)
            self._engines[("grad", batch, size, dt)] = eng
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


def _resnet_stem(bld, x, W, batch, size, ho, dt):
    """conv1 + bn1 + ReLU as im2col + a 1x1 conv on the MFMA kernels (the 7x7 stride-2 stem on 3 channels)."""
    kp = W["stem.wg"].numel() // 64
    cols = bld.buf((batch, ho, ho, kp))
    bld.plan.add(bld.lib.advs_im2col_nchw, ptr(x), ptr(cols), batch, 3, size, size, 7, 2, 3, kp, dt, keep=(x, cols))
    h = bld.conv(cols, W["stem.wg"], 64, bias=W["stem.b"], ksize=1, pad=0, act="relu")
    bld.free(cols)
    return h


class _ResNetEngine:
    def __init__(self, model, W, batch, size, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            self.x = torch.zeros((batch, 3, size, size), dtype=torch.float32, device=dev)
            lib = bld.lib
            ho = (size + 6 - 7) // 2 + 1
            h = _resnet_stem(bld, self.x, W, batch, size, ho, dt)
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


class _ResNetGradEngine:
    """Forward with every post-ReLU activation retained, then the network backwards down to the image.

    Backward of one bottleneck, with G = d loss / d (pre-ReLU block output), already masked by the block's ReLU:
      g2 = conv1x1(G, w3^T) * [y2 > 0];  g1 = conv3x3(zero_insert(g2), flip(w2)^T) * [y1 > 0];
      d h = conv1x1(g1, w1^T) + (downsample ? zero_insert(conv1x1(G, wd^T)) : G);  next G = d h * [h > 0]
    (each mask is the ``relu_mask`` operand of the conv that produces the masked tensor).
    BatchNorm is folded into the conv weights exactly as in the forward, so its backward is the folded conv's."""

    def __init__(self, model, W, G, batch, size, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib, plan = bld.lib, bld.plan
            self.x = torch.zeros((batch, 3, size, size), dtype=torch.float32, device=dev)
            self.labels = torch.zeros((batch,), dtype=torch.int64, device=dev)
            self.grad = torch.zeros((batch, 3, size, size), dtype=torch.float32, device=dev)
            # ---- forward (as _ResNetEngine, nothing released)
            ho = (size + 6 - 7) // 2 + 1
            stem = _resnet_stem(bld, self.x, W, batch, size, ho, dt)
            hp = (ho + 2 - 3) // 2 + 1
            h = bld.buf((batch, hp, hp, 64))
            plan.add(lib.advs_maxpool3x3s2, ptr(stem), ptr(h), batch, ho, ho, 64, dt, keep=(stem, h))
            acts = []
            for p, cin, width, cout, s, ds in model.blocks:
                y1 = bld.conv(h, W[p + ".w1"], width, bias=W[p + ".b1"], ksize=1, pad=0, act="relu")
                y2 = bld.conv(y1, W[p + ".w2"], width, bias=W[p + ".b2"], ksize=3, stride=s, pad=1, act="relu")
                idn = bld.conv(h, W[p + ".wd"], cout, bias=W[p + ".bd"], ksize=1, stride=s, pad=0) if ds else h
                y3 = bld.conv(y2, W[p + ".w3"], cout, bias=W[p + ".b3"], residual=idn, ksize=1, pad=0, act="relu")
                if idn is not h:
                    bld.free(idn)
                acts.append((h, y1, y2, y3))
                h = y3
            _, hh, ww, cc = h.shape
            pooled = bld.buf((batch, cc), torch.float32)
            plan.add(lib.advs_global_avgpool, ptr(h), ptr(pooled), batch, hh * ww, cc, dt, keep=(h, pooled))
            self.logits = bld.linear(pooled, W["fc.w"], W["fc.b"])
            # ---- backward
            K = self.logits.shape[1]
            gl = bld.buf((batch, K), torch.float32)
            plan.add(lib.advs_softmax_ce_grad, ptr(self.logits), ptr(self.labels), ptr(gl), batch, K, 1.0,
                     keep=(self.logits, self.labels, gl))
            gp = bld.linear(gl, G["fc.wT"], None)
            g = bld.buf(tuple(h.shape))
            plan.add(lib.advs_avgpool_bwd_relu, ptr(gp), ptr(h), ptr(g), batch, hh * ww, cc, dt, keep=(gp, h, g))

            def zero_insert(t, like):
                z = bld.buf((batch, like.shape[1], like.shape[2], t.shape[3]))
                plan.add(lib.advs_zero_insert2x, ptr(t), ptr(z), batch, t.shape[1], t.shape[2], t.shape[3], like.shape[1],
                         like.shape[2], dt, keep=(t, z))
                bld.free(t)
                return z

            # the ReLU masks ride the conv epilogues (advs_conv_args.relu_mask) except behind a zero insertion
            for (p, cin, width, cout, s, ds), (hin, y1, y2, y3) in zip(reversed(model.blocks), reversed(acts)):
                g2 = bld.conv(g, G[p + ".w3T"], width, ksize=1, pad=0, relu_mask=y2)
                if s == 2 and y1.shape[1] % 2 == 0 and y1.shape[2] % 2 == 0:
                    # parity form: gather the 2x2 output neighbourhood, one 1x1 GEMM for the four input parities, interleave + ReLU mask
                    gh2, gw2 = g2.shape[1], g2.shape[2]
                    g4 = bld.buf((batch, gh2, gw2, 4 * width))
                    plan.add(lib.advs_gather2x2, ptr(g2), ptr(g4), batch, gh2, gw2, width, dt, keep=(g2, g4))
                    gq = bld.conv(g4, G[p + ".w2P"], 4 * width, ksize=1, pad=0)
                    bld.free(g4)
                    g1 = bld.buf(tuple(y1.shape))
                    plan.add(lib.advs_depth_to_space2_relu, ptr(gq), ptr(y1), ptr(g1), batch, y1.shape[1], y1.shape[2], width, dt,
                             keep=(gq, y1, g1))
                    bld.free(gq)
                else:
                    if s == 2:
                        g2 = zero_insert(g2, y1)
                    g1 = bld.conv(g2, G[p + ".w2T"], width, ksize=3, stride=1, pad=1, relu_mask=y1)
                bld.free(g2)
                if ds:
                    res = bld.conv(g, G[p + ".wdT"], cin, ksize=1, pad=0)
                    if s == 2:
                        res = zero_insert(res, hin)
                    bld.free(g)
                else:
                    res = g
                # the first block reads the max-pooled stem: its ReLU is applied by advs_maxpool3x3s2_bwd_relu below
                gh = bld.conv(g1, G[p + ".w1T"], cin, residual=res, ksize=1, pad=0,
                              relu_mask=None if hin is acts[0][0] else hin)
                bld.free(g1)
                bld.free(res)
                for t in (y1, y2, y3):
                    bld.free(t)
                g = gh
            gs = bld.buf(tuple(stem.shape))
            plan.add(lib.advs_maxpool3x3s2_bwd_relu, ptr(g), ptr(stem), ptr(gs), batch, ho, ho, 64, dt, keep=(g, stem, gs))
            kp = G["stem.wT"].numel() // 64
            gcol = bld.conv(gs, G["stem.wT"], kp, ksize=1, pad=0)          # gradient of the im2col columns
            plan.add(lib.advs_col2im_nchw, ptr(gcol), ptr(self.grad), batch, 3, size, size, 7, 2, 3, kp, dt,
                     keep=(gcol, self.grad))
            self.plan, self.captured = plan, False
            torch.cuda.synchronize(dev)

    run = _ResNetEngine.run


# ============================================================================ ViT (HF ViTForImageClassification)
class _Logits:
    """What the reference reads from a HF model: ``outputs.logits`` (ASR_fast.py:114)."""

    def __init__(self, logits):
        self.logits = logits


# transformers >= 5 renamed the encoder parameters; the authors' checkpoints (4.x) use the left-hand names
_VIT_NEW2OLD = (("vit.layers.", "vit.encoder.layer."), (".attention.q_proj.", ".attention.attention.query."),
                (".attention.k_proj.", ".attention.attention.key."), (".attention.v_proj.", ".attention.attention.value."),
                (".attention.o_proj.", ".attention.output.dense."), (".mlp.fc1.", ".intermediate.dense."),
                (".mlp.fc2.", ".output.dense."))


def vit_canonical_state_dict(sd):
    """Map a transformers-5 style ViT state_dict onto the 4.x key names this module holds."""
    out = {}
    for k, v in sd.items():
        for new, old in _VIT_NEW2OLD:
            k = k.replace(new, old)
        out[k] = v
    return out


def _patch_proj(w, dt):
    """Patch-embedding Conv2d weight [C, 3, ps, ps] as a GEMM weight whose K is padded with zeros to the row length
    advs_patchify_padded produces (a multiple of 64 elements in every dtype)."""
    w2 = w.detach().float().reshape(w.shape[0], -1)
    k = w2.shape[1]
    kpad = -(-k // 64) * 64
    if kpad != k:
        w2 = torch.nn.functional.pad(w2, (0, kpad - k))
    return pack_conv_weight(w2.reshape(w2.shape[0], kpad, 1, 1), dt)


class ViTVictim(_InputGradient, nn.Module):
    """ViT-B/16-style classifier with HF's parameter names (``AutoModelForImageClassification`` of
    ASR_fast.py:47-51; config C4 of BASELINE.json), forward on the HIP kernels: patch projection and every
    Linear as implicit-GEMM 1x1 convs (bias / GELU / residual in the epilogue), LayerNorm, flash attention
    with the 197 tokens padded to 208 rows and the padding masked."""

    def __init__(self, num_labels=37, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, patch_size=16, image_size=224, layer_norm_eps=1e-12,
                 compute_dtype="fp32", use_graph=True):
        super().__init__()
        self.cfg = dict(num_labels=num_labels, hidden=hidden_size, layers=num_hidden_layers, heads=num_attention_heads,
                        mlp=intermediate_size, patch=patch_size, image=image_size, eps=layer_norm_eps)
        self.compute_dtype, self.use_graph = compute_dtype, use_graph
        C, npatch = hidden_size, (image_size // patch_size) ** 2

        class _P(nn.Module):
            def __init__(self, *shape):
                super().__init__()

        emb = nn.Module()
        emb.cls_token = nn.Parameter(torch.randn(1, 1, C) * 0.02)
        emb.position_embeddings = nn.Parameter(torch.randn(1, npatch + 1, C) * 0.02)
        _attach(self, "vit.embeddings", emb)
        _attach(self, "vit.embeddings.patch_embeddings.projection", nn.Conv2d(3, C, patch_size, stride=patch_size))
        for i in range(num_hidden_layers):
            p = f"vit.encoder.layer.{i}"
            for n in ("query", "key", "value"):
                _attach(self, f"{p}.attention.attention.{n}", nn.Linear(C, C))
            _attach(self, p + ".attention.output.dense", nn.Linear(C, C))
            _attach(self, p + ".intermediate.dense", nn.Linear(C, intermediate_size))
            _attach(self, p + ".output.dense", nn.Linear(intermediate_size, C))
            _attach(self, p + ".layernorm_before", nn.LayerNorm(C, eps=layer_norm_eps))
            _attach(self, p + ".layernorm_after", nn.LayerNorm(C, eps=layer_norm_eps))
        _attach(self, "vit.layernorm", nn.LayerNorm(C, eps=layer_norm_eps))
        _attach(self, "classifier", nn.Linear(C, num_labels))
        self._packed, self._engines = {}, {}

    def load_state_dict(self, state_dict, strict=True, **kw):
        return super().load_state_dict(vit_canonical_state_dict(state_dict), strict=strict, **kw)

    def _version(self):
        return (str(next(self.parameters()).device), sum(p._version for p in self.parameters()))

    def packed_weights(self, dt):
        ver = self._version()
        hit = self._packed.get(dt)
        if hit is not None and hit[0] == ver:
            return hit[1]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AdvsError(f"ViTVictim parameters are on {dev}: move the model to the GPU; there is no CPU fallback")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        f32 = lambda k: sd[k].float().contiguous()
        lin = lambda k: pack_conv_weight(sd[k].float().reshape(sd[k].shape[0], -1, 1, 1), dt)
        W = {"cls": f32("vit.embeddings.cls_token").reshape(-1), "pos": f32("vit.embeddings.position_embeddings")[0].contiguous(),
             "proj.w": _patch_proj(sd["vit.embeddings.patch_embeddings.projection.weight"], dt),
             "proj.b": f32("vit.embeddings.patch_embeddings.projection.bias")}
        for i in range(self.cfg["layers"]):
            p = f"vit.encoder.layer.{i}"
            a = p + ".attention.attention."
            W[p + ".qkv.w"] = pack_conv_weight(torch.cat([sd[a + n + ".weight"].float() for n in ("query", "key", "value")], 0)
                                               .reshape(3 * self.cfg["hidden"], -1, 1, 1), dt)
            W[p + ".qkv.b"] = torch.cat([sd[a + n + ".bias"].float() for n in ("query", "key", "value")], 0).contiguous()
            for src, dst in ((".attention.output.dense", ".o"), (".intermediate.dense", ".fc1"), (".output.dense", ".fc2")):
                W[p + dst + ".w"], W[p + dst + ".b"] = lin(p + src + ".weight"), f32(p + src + ".bias")
            for n in (".layernorm_before", ".layernorm_after"):
                W[p + n + ".g"], W[p + n + ".b"] = f32(p + n + ".weight"), f32(p + n + ".bias")
        W["ln.g"], W["ln.b"] = f32("vit.layernorm.weight"), f32("vit.layernorm.bias")
        W["cls.w"], W["cls.b"] = f32("classifier.weight"), f32("classifier.bias")
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[-1] == dt]:      # (batch, dt) and ("grad", batch, dt)
            del self._engines[key]
        return W

    def engine(self, batch, dtype=None):
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        eng = self._engines.get((batch, dt))
        if eng is None:
            eng = _ViTEngine(self, W, batch, dt)
            self._engines[(batch, dt)] = eng
        return eng

    def forward(self, pixel_values):
        B = pixel_values.shape[0]
        eng = self.engine(B)
        cur = torch.cuda.current_stream(pixel_values.device)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(pixel_values.to(torch.float32), non_blocking=True)
            eng.run()
            out = eng.logits.clone()
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        return _Logits(out)

    # ---- backward to the image (the gradient attack of tools/train_shadow.py:177-221 with this victim) ------------------
    def packed_grad_weights(self, dt):
        """Every Linear's weight transposed (its data gradient is a GEMM against W: rows of d_y times W[out][in])."""
        ver = self._version()
        hit = self._packed.get(("grad", dt))
        if hit is not None and hit[0] == ver:
            return hit[1]
        self.packed_weights(dt)
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        C = self.cfg["hidden"]
        linT = lambda w: pack_conv_weight(w.float().t().contiguous().reshape(w.shape[1], w.shape[0], 1, 1), dt)
        G = {}
        for i in range(self.cfg["layers"]):
            p = f"vit.encoder.layer.{i}"
            a = p + ".attention.attention."
            wqkv = torch.cat([sd[a + n + ".weight"].float() for n in ("query", "key", "value")], 0)      # [3C, C]
            G[p + ".qkvT"] = linT(wqkv)
            for src, dst in ((".attention.output.dense", ".oT"), (".intermediate.dense", ".fc1T"), (".output.dense", ".fc2T")):
                G[p + dst] = linT(sd[p + src + ".weight"])
        wp = sd["vit.embeddings.patch_embeddings.projection.weight"].float().reshape(C, -1)           # [C, 3*ps*ps]
        k = wp.shape[1]
        kpad = -(-k // 64) * 64
        wt = torch.zeros((kpad, C), dtype=torch.float32, device=wp.device)
        wt[:k] = wp.t()
        G["projT"] = pack_conv_weight(wt.reshape(kpad, C, 1, 1), dt)
        G["cls.wT"] = sd["classifier.weight"].float().t().contiguous()
        self._packed[("grad", dt)] = (ver, G)
        return G

    def grad_engine(self, batch, size=None, dtype=None):
        """Static plan of forward + backward-to-the-image (d cross_entropy / d pixel_values) for [batch,3,S,S] inputs."""
        if size is not None and size != self.cfg["image"]:
            raise ValueError(f"ViTVictim was built for {self.cfg['image']}x{self.cfg['image']} inputs, not {size}")
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W, G = self.packed_weights(dt), self.packed_grad_weights(dt)
        eng = self._engines.get(("grad", batch, dt))
        if eng is None:
            eng = _ViTGradEngine(self, W, G, batch, dt)
            self._engines[("grad", batch, dt)] = eng
        return eng


class _ViTGradEngine:
    """ViT forward with what the reverse sweep needs retained (the residual stream before each LayerNorm, qkv, the attention output,
    the MLP pre-activation), then HF's ViTLayer backwards (pre-norm blocks):
        d f = dT' W_fc2;  d pre = d f * gelu'(pre);  d h = dT' + LN2'(d pre W_fc1; h)
        d att = d h W_o;  d qkv = attention'(d att; qkv, att);  dT = d h + LN1'(d qkv W_qkv; T)
    down to the patch columns (d emb W_proj) and the image (advs_unpatchify_padded).  Only the CLS row of the last LayerNorm
    receives gradient from the head; padding rows carry zeros throughout."""

    def __init__(self, model, W, G, batch, dt):
        cfg = model.cfg
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        C, heads, S, ps, eps = cfg["hidden"], cfg["heads"], cfg["image"], cfg["patch"], cfg["eps"]
        g = S // ps
        npatch, n_tok = g * g, g * g + 1
        n_pad = (n_tok + 15) // 16 * 16          # rows per image: the GEMMs run over every row, the attention kernels mask keys >= n_tok
        d = C // heads
        rows = batch * n_pad
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib, plan = bld.lib, bld.plan
            self.x = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            self.labels = torch.zeros((batch,), dtype=torch.int64, device=dev)
            self.grad = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            kpad = -(-3 * ps * ps // 64) * 64
            # ---- forward (as _ViTEngine, GELU as its own pass, activations retained)
            patches = bld.buf((batch, g, g, kpad))
            plan.add(lib.advs_patchify_padded, ptr(self.x), ptr(patches), batch, 3, S, S, ps, kpad, dt, keep=(self.x, patches))
            emb = bld.conv(patches, W["proj.w"], C, bias=W["proj.b"], ksize=1, pad=0)
            bld.free(patches)
            tok = bld.buf((batch, 1, n_pad, C))
            plan.add(lib.advs_vit_assemble, ptr(emb), ptr(W["cls"]), ptr(W["pos"]), ptr(tok), batch, npatch, n_pad, C, dt, keep=(emb, tok))
            bld.free(emb)
            saved = []
            for i in range(cfg["layers"]):
                p = f"vit.encoder.layer.{i}"
                ln = bld.layernorm(tok, W[p + ".layernorm_before.g"], W[p + ".layernorm_before.b"], eps)
                qkv = bld.conv(ln, W[p + ".qkv.w"], 3 * C, bias=W[p + ".qkv.b"], ksize=1, pad=0)
                bld.free(ln)
                att = bld.attention(qkv, heads, d, 0, C, 2 * C, d, n_valid=n_tok)
                h = bld.conv(att, W[p + ".o.w"], C, bias=W[p + ".o.b"], residual=tok, ksize=1, pad=0)
                ln = bld.layernorm(h, W[p + ".layernorm_after.g"], W[p + ".layernorm_after.b"], eps)
                pre = bld.conv(ln, W[p + ".fc1.w"], cfg["mlp"], bias=W[p + ".fc1.b"], ksize=1, pad=0)
                bld.free(ln)
                f = bld.buf(tuple(pre.shape))
                plan.add(lib.advs_gelu, ptr(pre), ptr(f), pre.numel(), dt, keep=(pre, f))
                nxt = bld.conv(f, W[p + ".fc2.w"], C, bias=W[p + ".fc2.b"], residual=h, ksize=1, pad=0)
                bld.free(f)
                saved.append((p, tok, qkv, att, h, pre))
                tok = nxt
            ln = bld.layernorm(tok, W["ln.g"], W["ln.b"], eps)
            cls_mean = cfg.get("head") == "cls_mean"                       # DINOv2: Linear on [cls | mean of the patch tokens]
            cls = bld.buf((batch, 2 * C if cls_mean else C), torch.float32)
            if cls_mean:
                plan.add(lib.advs_cls_mean_rows_f32, ptr(ln), ptr(cls), batch, n_pad, npatch, C, dt, keep=(ln, cls))
            else:
                plan.add(lib.advs_gather_rows_f32, ptr(ln), ptr(cls), batch, n_pad, C, dt, keep=(ln, cls))
            bld.free(ln)
            self.logits = bld.linear(cls, W["cls.w"], W["cls.b"])
            # ---- backward
            K = self.logits.shape[1]
            gl = bld.buf((batch, K), torch.float32)
            plan.add(lib.advs_softmax_ce_grad, ptr(self.logits), ptr(self.labels), ptr(gl), batch, K, 1.0, keep=(self.logits, self.labels, gl))
            gcls = bld.linear(gl, G["cls.wT"], None)                       # [B, C] (or [B, 2C]) f32
            dln = torch.zeros((batch, 1, n_pad, C), dtype=bld.tdt, device=dev)    # rows the head does not read stay zero for good
            if cls_mean:
                plan.add(lib.advs_scatter_cls_mean, ptr(gcls), ptr(dln), batch, n_pad, npatch, C, dt, keep=(gcls, dln))
            else:
                plan.add(lib.advs_scatter_row0, ptr(gcls), ptr(dln), batch, n_pad, C, dt, keep=(gcls, dln))
            scratch = torch.empty(lib.advs_attention_bwd_scratch_bytes(batch, n_pad, heads), dtype=torch.uint8, device=dev)

            def ln_bwd(dy, x, gamma, add):
                dx = bld.buf(tuple(x.shape))
                plan.add(lib.advs_layernorm_bwd, ptr(dy), ptr(x), ptr(gamma), ptr(add), ptr(dx), rows, C, float(eps), dt,
                         keep=(dy, x, gamma, add, dx))
                return dx

            dT = ln_bwd(dln, tok, W["ln.g"], None)
            bld.free(tok)
            for p, t_in, qkv, att, h, pre in reversed(saved):
                df = bld.conv(dT, G[p + ".fc2T"], cfg["mlp"], ksize=1, pad=0)
                dpre = bld.buf(tuple(pre.shape))
                plan.add(lib.advs_gelu_bwd, ptr(pre), ptr(df), ptr(dpre), pre.numel(), dt, keep=(pre, df, dpre))
                bld.free(df); bld.free(pre)
                dl2 = bld.conv(dpre, G[p + ".fc1T"], C, ksize=1, pad=0)
                bld.free(dpre)
                dh = ln_bwd(dl2, h, W[p + ".layernorm_after.g"], dT)
                bld.free(dl2); bld.free(dT); bld.free(h)
                datt = bld.conv(dh, G[p + ".oT"], C, ksize=1, pad=0)
                dqkv = bld.buf(tuple(qkv.shape))
                plan.add(lib.advs_attention_bwd, ptr(qkv), ptr(att), ptr(datt), ptr(dqkv), ptr(scratch), batch, n_pad, n_tok, heads, d,
                         3 * C, 0, C, 2 * C, d, dt, keep=(qkv, att, datt, dqkv, scratch))
                bld.free(datt); bld.free(qkv); bld.free(att)
                dl1 = bld.conv(dqkv, G[p + ".qkvT"], C, ksize=1, pad=0)
                bld.free(dqkv)
                dT = ln_bwd(dl1, t_in, W[p + ".layernorm_before.g"], dh)
                bld.free(dl1); bld.free(dh); bld.free(t_in)
            # token gradients -> patch columns (every row; the CLS and padding rows are never read) -> image
            dcols = bld.conv(dT, G["projT"], kpad, ksize=1, pad=0)
            bld.free(dT)
            plan.add(lib.advs_unpatchify_padded, ptr(dcols), ptr(self.grad), batch, 3, S, S, ps, kpad, n_pad, 1, dt, keep=(dcols, self.grad))
            self.plan, self.captured = plan, False
            torch.cuda.synchronize(dev)

    def run(self):
        _ViTEngine.run(self)


class _ViTEngine:
    def __init__(self, model, W, batch, dt):
        cfg = model.cfg
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        C, heads, S, ps, eps = cfg["hidden"], cfg["heads"], cfg["image"], cfg["patch"], cfg["eps"]
        g = S // ps
        npatch, n_tok = g * g, g * g + 1
        n_pad = (n_tok + 15) // 16 * 16          # rows per image: the GEMMs run over every row, the attention kernels mask keys >= n_tok
        d = C // heads
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib = bld.lib
            self.x = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            kpad = -(-3 * ps * ps // 64) * 64              # whole 128-byte slabs in every dtype (14x14 patches: 588 -> 640)
            patches = bld.buf((batch, g, g, kpad))
            bld.plan.add(lib.advs_patchify_padded, ptr(self.x), ptr(patches), batch, 3, S, S, ps, kpad, dt, keep=(self.x, patches))
            emb = bld.conv(patches, W["proj.w"], C, bias=W["proj.b"], ksize=1, pad=0)
            bld.free(patches)
            tok = bld.buf((batch, 1, n_pad, C))
            bld.plan.add(lib.advs_vit_assemble, ptr(emb), ptr(W["cls"]), ptr(W["pos"]), ptr(tok), batch, npatch, n_pad, C, dt,
                         keep=(emb, tok))
            bld.free(emb)
            for i in range(cfg["layers"]):
                p = f"vit.encoder.layer.{i}"
                ln = bld.layernorm(tok, W[p + ".layernorm_before.g"], W[p + ".layernorm_before.b"], eps)
                qkv = bld.conv(ln, W[p + ".qkv.w"], 3 * C, bias=W[p + ".qkv.b"], ksize=1, pad=0)
                bld.free(ln)
                att = bld.attention(qkv, heads, d, 0, C, 2 * C, d, n_valid=n_tok)
                bld.free(qkv)
                h = bld.conv(att, W[p + ".o.w"], C, bias=W[p + ".o.b"], residual=tok, ksize=1, pad=0)
                bld.free(att)
                bld.free(tok)
                ln = bld.layernorm(h, W[p + ".layernorm_after.g"], W[p + ".layernorm_after.b"], eps)
                f = bld.conv(ln, W[p + ".fc1.w"], cfg["mlp"], bias=W[p + ".fc1.b"], act="gelu", ksize=1, pad=0)
                bld.free(ln)
                tok = bld.conv(f, W[p + ".fc2.w"], C, bias=W[p + ".fc2.b"], residual=h, ksize=1, pad=0)
                bld.free(f)
                bld.free(h)
            ln = bld.layernorm(tok, W["ln.g"], W["ln.b"], eps)
            bld.free(tok)
            if cfg.get("head") == "cls_mean":          # DINOv2: Linear on [cls | mean of the patch tokens]
                cls = bld.buf((batch, 2 * C), torch.float32)
                bld.plan.add(lib.advs_cls_mean_rows_f32, ptr(ln), ptr(cls), batch, n_pad, npatch, C, dt, keep=(ln, cls))
            else:
                cls = bld.buf((batch, C), torch.float32)
                bld.plan.add(lib.advs_gather_rows_f32, ptr(ln), ptr(cls), batch, n_pad, C, dt, keep=(ln, cls))
            self.logits = bld.linear(cls, W["cls.w"], W["cls.b"])
            self.plan, self.captured = bld.plan, False
            torch.cuda.synchronize(dev)

    def run(self):
        if self.model.use_graph and not self.captured:
            self.plan.run_eager()
            self.stream.synchronize()
            self.plan.capture()
            self.captured = True
        self.plan.run()


# ============================================================================ VGG-16 / VGG-19 (torchvision names)
_VGG_CFG = {16: [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
            19: [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]}


class VGG(_InputGradient, nn.Module):
    """torchvision ``vgg16()`` / ``vgg19()`` with ``classifier[6] = Linear(4096, 37)`` (ASR_fast.py:33-46):
    ``features.{i}`` convs (bias + ReLU in the conv epilogue), MaxPool2d(2), three Linear layers."""

    def __init__(self, depth=16, num_classes=37, compute_dtype="fp32", use_graph=True):
        super().__init__()
        self.depth, self.num_classes, self.compute_dtype, self.use_graph = depth, num_classes, compute_dtype, use_graph
        self.layers, cin, idx = [], 3, 0
        for v in _VGG_CFG[depth]:
            if v == "M":
                self.layers.append(("pool", None, 0, 0))
                idx += 1
            else:
                _attach(self, f"features.{idx}", nn.Conv2d(cin, v, 3, padding=1))
                self.layers.append(("conv", f"features.{idx}", cin, v))
                cin, idx = v, idx + 2                     # conv, ReLU
        for i, (a, b) in zip((0, 3, 6), ((512 * 7 * 7, 4096), (4096, 4096), (4096, num_classes))):
            _attach(self, f"classifier.{i}", nn.Linear(a, b))
        self._packed, self._engines = {}, {}

    def _version(self):
        return (str(next(self.parameters()).device), sum(p._version for p in self.parameters()))

    def packed_weights(self, dt):
        ver = self._version()
        hit = self._packed.get(dt)
        if hit is not None and hit[0] == ver:
            return hit[1]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AdvsError(f"VGG parameters are on {dev}: move the model to the GPU; there is no CPU fallback")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        W = {}
        for kind, p, cin, cout in self.layers:
            if kind == "conv":
                W[p + ".w"] = sd[p + ".weight"].float().contiguous() if cin == 3 else pack_conv_weight(sd[p + ".weight"], dt)
                W[p + ".b"] = sd[p + ".bias"].float().contiguous()
        for i in (0, 3, 6):
            W[f"fc{i}.w"], W[f"fc{i}.b"] = sd[f"classifier.{i}.weight"].float().contiguous(), sd[f"classifier.{i}.bias"].float().contiguous()
        # classifier[0] / [3] as 1x1 convs on the MFMA kernels (the wave-per-output advs_linear_f32 re-reads the 411 MB
        # matrix once per image).  x.view(B, -1) flattens NCHW; the pooled map is NHWC, so the columns are permuted once here.
        w0 = W["fc0.w"].reshape(4096, 512, 7, 7).permute(0, 2, 3, 1).reshape(4096, 512 * 49)
        W["fc0.wp"] = pack_conv_weight(w0.reshape(4096, -1, 1, 1), dt)
        W["fc3.wp"] = pack_conv_weight(W["fc3.w"].reshape(4096, 4096, 1, 1), dt)
        del W["fc0.w"], W["fc3.w"]
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[-1] == dt]:
            del self._engines[key]
        return W

    def engine(self, batch, size, dtype=None):
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        eng = self._engines.get((batch, size, dt))
        if eng is None:
            eng = _VGGEngine(self, W, batch, size, dt)
            self._engines[(batch, size, dt)] = eng
        return eng

    def packed_grad_weights(self, dt):
        """Data-gradient weights: 3x3 convs transposed and flipped, classifier matrices transposed."""
        ver = self._version()
        hit = self._packed.get(("grad", dt))
        if hit is not None and hit[0] == ver:
            return hit[1]
        self.packed_weights(dt)
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        G = {}
        for kind, p, cin, cout in self.layers:
            if kind == "conv" and cin != 3:
                G[p + ".wT"] = pack_conv_weight(sd[p + ".weight"].float().permute(1, 0, 2, 3).flip(2, 3).contiguous(), dt)
        w0 = sd["classifier.0.weight"].float().reshape(4096, 512, 7, 7).permute(0, 2, 3, 1).reshape(4096, 512 * 49)
        G["fc0.wTp"] = pack_conv_weight(w0.t().contiguous().reshape(512 * 49, 4096, 1, 1), dt)     # -> gradient of the NHWC pooled map
        G["fc3.wTp"] = pack_conv_weight(sd["classifier.3.weight"].float().t().contiguous().reshape(4096, 4096, 1, 1), dt)
        G["fc6.wT"] = sd["classifier.6.weight"].float().t().contiguous()
        first = self.layers[0][1]                              # the 3-channel conv: column gradients, then advs_col2im_nchw
        w = sd[first + ".weight"].float()
        kp = -(-27 // SLAB_ELEMS[dt]) * SLAB_ELEMS[dt]
        wt = torch.zeros((kp, w.shape[0]), dtype=torch.float32, device=w.device)
        wt[:27] = w.reshape(w.shape[0], 27).t()
        G["stem.wT"] = pack_conv_weight(wt.reshape(kp, w.shape[0], 1, 1), dt)
        self._packed[("grad", dt)] = (ver, G)
        return G

    def grad_engine(self, batch, size, dtype=None):
        if size != 224:
            raise ValueError("VGG victim expects 224x224 inputs (AdaptiveAvgPool2d(7) is the identity there)")
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W, G = self.packed_weights(dt), self.packed_grad_weights(dt)
        eng = self._engines.get(("grad", batch, size, dt))
        if eng is None:
            eng = _VGGGradEngine(self, W, G, batch, size, dt)
            self._engines[("grad", batch, size, dt)] = eng
        return eng

    def forward(self, x):
        B, _, H, Wd = x.shape
        if H != 224 or Wd != 224:
            raise ValueError("VGG victim expects 224x224 inputs (AdaptiveAvgPool2d(7) is the identity there)")
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


def _vgg_classifier(bld, h, W, batch, dt):
    """classifier[0..6] on the pooled NHWC map h [B,7,7,512]: two 1x1-conv GEMMs (bias + ReLU in the epilogue; the 128x128
    tile gives the weight stream twice the workgroups of the 256x256 one at M = batch), then the small f32 linear."""
    lib = bld.lib
    f0 = bld.conv(h.view(batch, 1, 1, -1), W["fc0.wp"], 4096, bias=W["fc0.b"], ksize=1, pad=0, act="relu", tile=1)
    f3 = bld.conv(f0, W["fc3.wp"], 4096, bias=W["fc3.b"], ksize=1, pad=0, act="relu", tile=1)
    f3f = bld.buf((batch, 4096), torch.float32)
    bld.plan.add(lib.advs_nhwc_to_nchw_f32, ptr(f3), ptr(f3f), batch, 4096, 1, 1, dt, keep=(f3, f3f))
    return f0, f3, bld.linear(f3f, W["fc6.w"], W["fc6.b"])


class _VGGEngine(_ResNetEngine):
    def __init__(self, model, W, batch, size, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib = bld.lib
            self.x = torch.zeros((batch, 3, size, size), dtype=torch.float32, device=dev)
            h, hw = None, size
            for kind, p, cin, cout in model.layers:
                if kind == "conv" and cin == 3:
                    new = bld.buf((batch, hw, hw, cout))
                    bld.plan.add(lib.advs_conv_stem, ptr(self.x), ptr(W[p + ".w"]), ptr(W[p + ".b"]), ptr(new), batch, 3, hw, hw,
                                 cout, 3, 1, 1, _lib.ACT["relu"], dt, keep=(self.x, new))
                elif kind == "conv":
                    new = bld.conv(h, W[p + ".w"], cout, bias=W[p + ".b"], act="relu")
                else:
                    new = bld.maxpool2(h)
                    hw //= 2
                if h is not None:
                    bld.free(h)
                h = new
            _, _, self.logits = _vgg_classifier(bld, h, W, batch, dt)
            self.plan, self.captured = bld.plan, False
            torch.cuda.synchronize(dev)


class _VGGGradEngine:
    """VGG forward with every activation retained, then backwards to the image.  ``g`` entering a conv layer is the
    gradient of its pre-ReLU output: the ReLU mask is applied by the layer behind it -- ``advs_maxpool2_bwd_relu`` when
    that is a pool, otherwise the next conv's data-gradient conv (epilogue ``relu_mask``, or ``advs_relu_bwd`` after the
    halo kernel where the map is a multiple of 16)."""

    def __init__(self, model, W, G, batch, size, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib, plan = bld.lib, bld.plan
            self.x = torch.zeros((batch, 3, size, size), dtype=torch.float32, device=dev)
            self.labels = torch.zeros((batch,), dtype=torch.int64, device=dev)
            self.grad = torch.zeros((batch, 3, size, size), dtype=torch.float32, device=dev)
            h, hw, acts = None, size, []
            for kind, p, cin, cout in model.layers:
                if kind == "conv" and cin == 3:
                    new = bld.buf((batch, hw, hw, cout))
                    plan.add(lib.advs_conv_stem, ptr(self.x), ptr(W[p + ".w"]), ptr(W[p + ".b"]), ptr(new), batch, 3, hw, hw,
                             cout, 3, 1, 1, _lib.ACT["relu"], dt, keep=(self.x, new))
                elif kind == "conv":
                    new = bld.conv(h, W[p + ".w"], cout, bias=W[p + ".b"], act="relu")
                else:
                    new = bld.maxpool2(h)
                acts.append((kind, p, cin, cout, h, new, hw))
                if kind == "pool":
                    hw //= 2
                h = new
            f0, f3, self.logits = _vgg_classifier(bld, h, W, batch, dt)
            # ---- backward: classifier
            K = self.logits.shape[1]
            gl = bld.buf((batch, K), torch.float32)
            plan.add(lib.advs_softmax_ce_grad, ptr(self.logits), ptr(self.labels), ptr(gl), batch, K, 1.0,
                     keep=(self.logits, self.labels, gl))

            def relu_bwd(t, y, code):
                plan.add(lib.advs_relu_bwd, ptr(t), 0, ptr(y), ptr(t), t.numel(), code, keep=(t, y))

            g3f = bld.linear(gl, G["fc6.wT"], None)
            g3 = bld.buf((batch, 1, 1, 4096))
            plan.add(lib.advs_nchw_f32_to_nhwc, ptr(g3f), ptr(g3), batch, 4096, 1, 1, dt, keep=(g3f, g3))
            relu_bwd(g3, f3, dt)
            g0 = bld.conv(g3, G["fc3.wTp"], 4096, ksize=1, pad=0, relu_mask=f0, tile=1)
            g = bld.conv(g0, G["fc0.wTp"], 512 * hw * hw, ksize=1, pad=0, tile=1).view(batch, hw, hw, 512)
            # ---- backward: features
            for i in range(len(acts) - 1, -1, -1):
                kind, p, cin, cout, hin, hout, hwi = acts[i]
                if kind == "pool":
                    gi = bld.buf(tuple(hin.shape))
                    plan.add(lib.advs_maxpool2_bwd_relu, ptr(g), ptr(hin), ptr(gi), batch, hwi, hwi, hin.shape[3], dt,
                             keep=(g, hin, gi))
                elif cin == 3:
                    kp = G["stem.wT"].numel() // cout
                    gcol = bld.conv(g, G["stem.wT"], kp, ksize=1, pad=0)
                    plan.add(lib.advs_col2im_nchw, ptr(gcol), ptr(self.grad), batch, 3, hwi, hwi, 3, 1, 1, kp, dt,
                             keep=(gcol, self.grad))
                    break
                else:
                    masked = acts[i - 1][0] == "conv"             # hin is a conv output: its ReLU is applied here
                    if masked and hwi % 16 == 0:
                        gi = bld.conv(g, G[p + ".wT"], cin)
                        relu_bwd(gi, hin, dt)
                    else:
                        gi = bld.conv(g, G[p + ".wT"], cin, relu_mask=hin if masked else None)
                bld.free(g)
                bld.free(hout)
                g = gi
            self.plan, self.captured = plan, False
            torch.cuda.synchronize(dev)

    run = _ResNetEngine.run


# ============================================================================ ConvNeXt (timm names)
_CONVNEXT_BASE = dict(depths=(3, 3, 27, 3), dims=(128, 256, 512, 1024))


def convnext_canonical_state_dict(sd):
    """Accept HF ``ConvNextForImageClassification`` keys as well as timm's (the reference's loader,
    ASR_fast.py:21-26, is timm): returns timm-style names."""
    if not any(k.startswith("convnext.") or k.startswith("classifier.") for k in sd):
        return dict(sd)
    out = {}
    for k, v in sd.items():
        n = k.replace("convnext.layernorm.", "head.norm.").replace("classifier.", "head.fc.")
        n = n.replace("convnext.embeddings.patch_embeddings.", "stem.0.").replace("convnext.embeddings.layernorm.", "stem.1.")
        n = n.replace("convnext.encoder.stages.", "stages.").replace(".downsampling_layer.", ".downsample.")
        n = n.replace(".layers.", ".blocks.").replace(".dwconv.", ".conv_dw.").replace(".layernorm.", ".norm.")
        n = n.replace(".pwconv1.", ".mlp.fc1.").replace(".pwconv2.", ".mlp.fc2.").replace(".layer_scale_parameter", ".gamma")
        out[n] = v
    return out


class ConvNeXtVictim(_InputGradient, nn.Module):
    """``timm.create_model('convnext_base.fb_in1k', num_classes=37)`` (ASR_fast.py:21-26) on the HIP kernels.
    Parameter names follow timm (``stem.{0,1}``, ``stages.i.downsample.{0,1}``, ``stages.i.blocks.j.{conv_dw,norm,
    mlp.fc1,mlp.fc2,gamma}``, ``head.norm``, ``head.fc``); HF ConvNext names are accepted by load_state_dict.
    Block: depthwise 7x7 -> LayerNorm -> Linear(C,4C)+GELU -> Linear(4C,C) with the layer scale folded in
    (+ residual in the GEMM epilogue).  The 4x4/s4 stem and the 2x2/s2 downsampling convs are patch gathers
    followed by 1x1 GEMMs.  ``head_norm_eps``: 1e-6 (timm) / 1e-12 (HF's config.layer_norm_eps)."""

    def __init__(self, num_classes=37, depths=_CONVNEXT_BASE["depths"], dims=_CONVNEXT_BASE["dims"], image_size=224,
                 head_norm_eps=1e-6, compute_dtype="fp32", use_graph=True):
        super().__init__()
        self.num_classes, self.depths, self.dims, self.image_size = num_classes, tuple(depths), tuple(dims), image_size
        self.head_norm_eps, self.compute_dtype, self.use_graph = head_norm_eps, compute_dtype, use_graph
        _attach(self, "stem.0", nn.Conv2d(3, dims[0], 4, stride=4))
        _attach(self, "stem.1", nn.LayerNorm(dims[0], eps=1e-6))
        for i, (n, c) in enumerate(zip(depths, dims)):
            if i > 0:
                _attach(self, f"stages.{i}.downsample.0", nn.LayerNorm(dims[i - 1], eps=1e-6))
                _attach(self, f"stages.{i}.downsample.1", nn.Conv2d(dims[i - 1], c, 2, stride=2))
            for j in range(n):
                p = f"stages.{i}.blocks.{j}"
                _attach(self, p + ".conv_dw", nn.Conv2d(c, c, 7, padding=3, groups=c))
                _attach(self, p + ".norm", nn.LayerNorm(c, eps=1e-6))
                _attach(self, p + ".mlp.fc1", nn.Linear(c, 4 * c))
                _attach(self, p + ".mlp.fc2", nn.Linear(4 * c, c))
                self.register_parameter(p.replace(".", "__") + "__gamma", nn.Parameter(torch.full((c,), 1e-6)))
        _attach(self, "head.norm", nn.LayerNorm(dims[-1], eps=1e-6))
        _attach(self, "head.fc", nn.Linear(dims[-1], num_classes))
        self._packed, self._engines = {}, {}

    # gamma is a bare Parameter of the block in timm ("stages.i.blocks.j.gamma"): expose it under that name
    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        return type(sd)((kk.replace("__gamma", ".gamma").replace("__", ".") if kk.endswith("__gamma") else kk, v) for kk, v in sd.items())

    def load_state_dict(self, sd, strict=True, **k):
        sd = convnext_canonical_state_dict(sd)
        sd = {(kk[:-len(".gamma")].replace(".", "__") + "__gamma" if kk.endswith(".gamma") else kk): v for kk, v in sd.items()}
        return super().load_state_dict(sd, strict=strict, **k)

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
            raise _lib.AdvsError(f"ConvNeXtVictim parameters are on {dev}: move the model to the GPU; there is no CPU fallback")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        f32 = lambda k: sd[k].float().contiguous()
        W = {"stem.w": pack_conv_weight(sd["stem.0.weight"].float().reshape(self.dims[0], -1, 1, 1), dt), "stem.b": f32("stem.0.bias"),
             "stem.g": f32("stem.1.weight"), "stem.beta": f32("stem.1.bias")}
        for i, (n, c) in enumerate(zip(self.depths, self.dims)):
            if i > 0:
                p = f"stages.{i}.downsample"
                W[p + ".g"], W[p + ".beta"] = f32(p + ".0.weight"), f32(p + ".0.bias")
                # [cout][cin][2][2] -> [cout][dy][dx][cin] = the channel order of advs_space_to_depth2
                W[p + ".w"] = pack_conv_weight(sd[p + ".1.weight"], dt).reshape(c, 1, 1, -1)
                W[p + ".b"] = f32(p + ".1.bias")
            for j in range(n):
                p = f"stages.{i}.blocks.{j}"
                W[p + ".dw.w"] = sd[p + ".conv_dw.weight"].float().reshape(c, 49).t().contiguous()      # [49][c]
                W[p + ".dw.b"] = f32(p + ".conv_dw.bias")
                W[p + ".g"], W[p + ".beta"] = f32(p + ".norm.weight"), f32(p + ".norm.bias")
                W[p + ".fc1.w"] = pack_conv_weight(sd[p + ".mlp.fc1.weight"].float().reshape(4 * c, c, 1, 1), dt)
                W[p + ".fc1.b"] = f32(p + ".mlp.fc1.bias")
                gamma = sd[p + ".gamma"].float()
                W[p + ".fc2.w"] = pack_conv_weight((sd[p + ".mlp.fc2.weight"].float() * gamma[:, None]).reshape(c, 4 * c, 1, 1), dt)
                W[p + ".fc2.b"] = (sd[p + ".mlp.fc2.bias"].float() * gamma).contiguous()
        W["head.g"], W["head.beta"] = f32("head.norm.weight"), f32("head.norm.bias")
        W["head.w"], W["head.b"] = f32("head.fc.weight"), f32("head.fc.bias")
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[-1] == dt]:      # (batch, dt) and ("grad", batch, dt)
            del self._engines[key]
        return W

    def engine(self, batch, dtype=None):
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        eng = self._engines.get((batch, dt))
        if eng is None:
            eng = _ConvNeXtEngine(self, W, batch, dt)
            self._engines[(batch, dt)] = eng
        return eng

    # ---- backward to the image (the gradient attack of tools/train_shadow.py:177-221 with this victim) ------------------
    def packed_grad_weights(self, dt):
        """Every Linear / patch GEMM weight transposed (its data gradient is a GEMM against W), layer scale folded as forwards."""
        ver = self._version()
        hit = self._packed.get(("grad", dt))
        if hit is not None and hit[0] == ver:
            return hit[1]
        self.packed_weights(dt)
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        linT = lambda w: pack_conv_weight(w.float().t().contiguous().reshape(w.shape[1], w.shape[0], 1, 1), dt)
        G = {"stemT": linT(sd["stem.0.weight"].float().reshape(self.dims[0], -1)),                  # [48][C0]
             "head.wT": sd["head.fc.weight"].float().t().contiguous()}
        for i, (n, c) in enumerate(zip(self.depths, self.dims)):
            if i > 0:
                p = f"stages.{i}.downsample"
                # forward weight [cout][dy][dx][cin] (the channel order of advs_space_to_depth2) -> [4 cin][cout]
                G[p + ".wT"] = linT(sd[p + ".1.weight"].float().permute(0, 2, 3, 1).reshape(c, -1))
            for j in range(n):
                p = f"stages.{i}.blocks.{j}"
                G[p + ".fc1T"] = linT(sd[p + ".mlp.fc1.weight"])
                G[p + ".fc2T"] = linT(sd[p + ".mlp.fc2.weight"].float() * sd[p + ".gamma"].float()[:, None])
        self._packed[("grad", dt)] = (ver, G)
        return G

    def grad_engine(self, batch, size=None, dtype=None):
        """Static plan of forward + backward-to-the-image (d cross_entropy / d input) for [batch,3,S,S] inputs."""
        if size is not None and size != self.image_size:
            raise ValueError(f"ConvNeXtVictim was built for {self.image_size}x{self.image_size} inputs, not {size}")
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W, G = self.packed_weights(dt), self.packed_grad_weights(dt)
        eng = self._engines.get(("grad", batch, dt))
        if eng is None:
            eng = _ConvNeXtGradEngine(self, W, G, batch, dt)
            self._engines[("grad", batch, dt)] = eng
        return eng

    def forward(self, x):
        B = x.shape[0]
        if x.shape[2] != self.image_size or x.shape[3] != self.image_size:
            raise ValueError(f"ConvNeXtVictim was built for {self.image_size}x{self.image_size} inputs, got {tuple(x.shape[2:])}")
        eng = self.engine(B)
        cur = torch.cuda.current_stream(x.device)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x.to(torch.float32), non_blocking=True)
            eng.run()
            out = eng.logits.clone()
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        return out


class _ConvNeXtGradEngine:
    """ConvNeXt forward with what the reverse sweep needs retained (each block's input, depthwise output and MLP pre-activation; the
    inputs of the stem / downsampling LayerNorms), then the blocks backwards:
        d f = d out W_fc2';  d pre = d f * gelu'(pre);  d d = LN'(d pre W_fc1; d);  d in = d out + dwconv'(d d)
    (layer scale folded into W_fc2 as in the forward), the downsampling layers as GEMM' -> depth_to_space -> LN', the stem as
    LN' -> GEMM' -> unpatchify, the head as Linear' -> LN' (f32) -> the average pool's broadcast."""

    def __init__(self, model, W, G, batch, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        S, dims = model.image_size, model.dims
        if S % 32:
            raise ValueError("ConvNeXtVictim: image_size must be a multiple of 32 (4x stem, three 2x downsamplings)")
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib, plan = bld.lib, bld.plan
            self.x = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            self.labels = torch.zeros((batch,), dtype=torch.int64, device=dev)
            self.grad = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            g = S // 4
            # ---- forward
            patches = bld.buf((batch, g, g, 48))
            plan.add(lib.advs_patchify, ptr(self.x), ptr(patches), batch, 3, S, S, 4, dt, keep=(self.x, patches))
            e = bld.conv(patches, W["stem.w"], dims[0], bias=W["stem.b"], ksize=1, pad=0)
            bld.free(patches)
            h = bld.layernorm(e, W["stem.g"], W["stem.beta"], 1e-6)
            side = g
            tape = [("stem", e)]                                    # reverse sweep reads this backwards
            for i, (n, c) in enumerate(zip(model.depths, dims)):
                if i > 0:
                    p = f"stages.{i}.downsample"
                    ln = bld.layernorm(h, W[p + ".g"], W[p + ".beta"], 1e-6)
                    s2d = bld.buf((batch, side // 2, side // 2, 4 * dims[i - 1]))
                    plan.add(lib.advs_space_to_depth2, ptr(ln), ptr(s2d), batch, side, side, dims[i - 1], dt, keep=(ln, s2d))
                    bld.free(ln)
                    tape.append(("down", p, h, side, dims[i - 1], c))
                    side //= 2
                    h = bld.conv(s2d, W[p + ".w"], c, bias=W[p + ".b"], ksize=1, pad=0)
                    bld.free(s2d)
                for j in range(n):
                    p = f"stages.{i}.blocks.{j}"
                    d = bld.buf((batch, side, side, c))
                    plan.add(lib.advs_dwconv2d, ptr(h), ptr(W[p + ".dw.w"]), ptr(W[p + ".dw.b"]), ptr(d), batch, side, side, c,
                             7, 1, dt, keep=(h, d))
                    ln = bld.layernorm(d, W[p + ".g"], W[p + ".beta"], 1e-6)
                    pre = bld.conv(ln, W[p + ".fc1.w"], 4 * c, bias=W[p + ".fc1.b"], ksize=1, pad=0)
                    bld.free(ln)
                    f = bld.buf(tuple(pre.shape))
                    plan.add(lib.advs_gelu, ptr(pre), ptr(f), pre.numel(), dt, keep=(pre, f))
                    new = bld.conv(f, W[p + ".fc2.w"], c, bias=W[p + ".fc2.b"], residual=h, ksize=1, pad=0)
                    bld.free(f)
                    bld.free(h)                                     # the block input itself is not needed backwards
                    tape.append(("block", p, d, pre, side, c))
                    h = new
            C = dims[-1]
            pooled = bld.buf((batch, C), torch.float32)
            plan.add(lib.advs_global_avgpool, ptr(h), ptr(pooled), batch, side * side, C, dt, keep=(h, pooled))
            bld.free(h)
            normed = bld.buf((batch, C), torch.float32)
            plan.add(lib.advs_layernorm, ptr(pooled), ptr(W["head.g"]), ptr(W["head.beta"]), ptr(normed), batch, C,
                     float(model.head_norm_eps), _lib.F32, keep=(pooled, normed))
            self.logits = bld.linear(normed, W["head.w"], W["head.b"])
            # ---- backward
            K = self.logits.shape[1]
            gl = bld.buf((batch, K), torch.float32)
            plan.add(lib.advs_softmax_ce_grad, ptr(self.logits), ptr(self.labels), ptr(gl), batch, K, 1.0, keep=(self.logits, self.labels, gl))
            gn = bld.linear(gl, G["head.wT"], None)                                  # [B, C] f32
            gp = bld.buf((batch, C), torch.float32)
            plan.add(lib.advs_layernorm_bwd, ptr(gn), ptr(pooled), ptr(W["head.g"]), 0, ptr(gp), batch, C, float(model.head_norm_eps),
                     _lib.F32, keep=(gn, pooled, gp))
            dh = bld.buf((batch, side, side, C))
            plan.add(lib.advs_avgpool_bwd, ptr(gp), ptr(dh), batch, side * side, C, dt, keep=(gp, dh))

            def ln_bwd(dy, x, gamma, rows, c):
                dx = bld.buf(tuple(x.shape))
                plan.add(lib.advs_layernorm_bwd, ptr(dy), ptr(x), ptr(gamma), 0, ptr(dx), rows, c, 1e-6, dt, keep=(dy, x, gamma, dx))
                return dx

            for rec in reversed(tape):
                if rec[0] == "block":
                    _, p, d, pre, sd, c = rec
                    df = bld.conv(dh, G[p + ".fc2T"], 4 * c, ksize=1, pad=0)
                    dpre = bld.buf(tuple(pre.shape))
                    plan.add(lib.advs_gelu_bwd, ptr(pre), ptr(df), ptr(dpre), pre.numel(), dt, keep=(pre, df, dpre))
                    bld.free(df); bld.free(pre)
                    dln = bld.conv(dpre, G[p + ".fc1T"], c, ksize=1, pad=0)
                    bld.free(dpre)
                    dd = ln_bwd(dln, d, W[p + ".g"], batch * sd * sd, c)
                    bld.free(dln); bld.free(d)
                    din = bld.buf((batch, sd, sd, c))
                    plan.add(lib.advs_dwconv2d_bwd, ptr(dd), ptr(W[p + ".dw.w"]), ptr(dh), ptr(din), batch, sd, sd, c, 7, dt, keep=(dd, dh, din))
                    bld.free(dd); bld.free(dh)
                    dh = din
                elif rec[0] == "down":
                    _, p, hin, sd, cin, c = rec                                   # sd: the side BEFORE the downsampling
                    ds2d = bld.conv(dh, G[p + ".wT"], 4 * cin, ksize=1, pad=0)
                    bld.free(dh)
                    dln = bld.buf((batch, sd, sd, cin))
                    plan.add(lib.advs_depth_to_space2, ptr(ds2d), ptr(dln), batch, sd, sd, cin, dt, keep=(ds2d, dln))
                    bld.free(ds2d)
                    dh = ln_bwd(dln, hin, W[p + ".g"], batch * sd * sd, cin)
                    bld.free(dln); bld.free(hin)
                else:
                    _, e = rec
                    de = ln_bwd(dh, e, W["stem.g"], batch * g * g, dims[0])
                    bld.free(dh); bld.free(e)
                    dcols = bld.conv(de, G["stemT"], 48, ksize=1, pad=0)
                    bld.free(de)
                    plan.add(lib.advs_unpatchify_padded, ptr(dcols), ptr(self.grad), batch, 3, S, S, 4, 48, g * g, 0, dt, keep=(dcols, self.grad))
            self.plan, self.captured = plan, False
            torch.cuda.synchronize(dev)

    def run(self):
        if self.model.use_graph and not self.captured:
            self.plan.run_eager()
            self.stream.synchronize()
            self.plan.capture()
            self.captured = True
        self.plan.run()


class _ConvNeXtEngine:
    def __init__(self, model, W, batch, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        S, dims = model.image_size, model.dims
        if S % 32:
            raise ValueError("ConvNeXtVictim: image_size must be a multiple of 32 (4x stem, three 2x downsamplings)")
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib = bld.lib
            self.x = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            g = S // 4
            patches = bld.buf((batch, g, g, 48))
            bld.plan.add(lib.advs_patchify, ptr(self.x), ptr(patches), batch, 3, S, S, 4, dt, keep=(self.x, patches))
            e = bld.conv(patches, W["stem.w"], dims[0], bias=W["stem.b"], ksize=1, pad=0)
            bld.free(patches)
            h = bld.layernorm(e, W["stem.g"], W["stem.beta"], 1e-6)
            bld.free(e)
            side = g
            for i, (n, c) in enumerate(zip(model.depths, dims)):
                if i > 0:
                    p = f"stages.{i}.downsample"
                    ln = bld.layernorm(h, W[p + ".g"], W[p + ".beta"], 1e-6)
                    bld.free(h)
                    s2d = bld.buf((batch, side // 2, side // 2, 4 * dims[i - 1]))
                    bld.plan.add(lib.advs_space_to_depth2, ptr(ln), ptr(s2d), batch, side, side, dims[i - 1], dt, keep=(ln, s2d))
                    bld.free(ln)
                    side //= 2
                    h = bld.conv(s2d, W[p + ".w"], c, bias=W[p + ".b"], ksize=1, pad=0)
                    bld.free(s2d)
                for j in range(n):
                    p = f"stages.{i}.blocks.{j}"
                    d = bld.buf((batch, side, side, c))
                    bld.plan.add(lib.advs_dwconv2d, ptr(h), ptr(W[p + ".dw.w"]), ptr(W[p + ".dw.b"]), ptr(d), batch, side, side, c,
                                 7, 1, dt, keep=(h, d))
                    ln = bld.layernorm(d, W[p + ".g"], W[p + ".beta"], 1e-6)
                    bld.free(d)
                    f = bld.conv(ln, W[p + ".fc1.w"], 4 * c, bias=W[p + ".fc1.b"], act="gelu", ksize=1, pad=0)
                    bld.free(ln)
                    new = bld.conv(f, W[p + ".fc2.w"], c, bias=W[p + ".fc2.b"], residual=h, ksize=1, pad=0)
                    bld.free(f)
                    bld.free(h)
                    h = new
            pooled = bld.buf((batch, dims[-1]), torch.float32)
            bld.plan.add(lib.advs_global_avgpool, ptr(h), ptr(pooled), batch, side * side, dims[-1], dt, keep=(h, pooled))
            bld.free(h)
            normed = bld.buf((batch, dims[-1]), torch.float32)
            bld.plan.add(lib.advs_layernorm, ptr(pooled), ptr(W["head.g"]), ptr(W["head.beta"]), ptr(normed), batch, dims[-1],
                         float(model.head_norm_eps), _lib.F32, keep=(pooled, normed))
            self.logits = bld.linear(normed, W["head.w"], W["head.b"])
            self.plan, self.captured = bld.plan, False
            torch.cuda.synchronize(dev)

    def run(self):
        if self.model.use_graph and not self.captured:
            self.plan.run_eager()
            self.stream.synchronize()
            self.plan.capture()
            self.captured = True
        self.plan.run()


# ============================================================================ Swin (timm names)
_SWIN_BASE = dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=7)


def swin_canonical_state_dict(sd, depths):
    """timm >= 0.9 names are canonical (``patch_embed.{proj,norm}``, ``layers.i.downsample.{norm,reduction}`` in front of
    stage i >= 1, ``layers.i.blocks.j.{norm1,attn.{qkv,proj,relative_position_bias_table},norm2,mlp.fc1,mlp.fc2}``,
    ``norm``, ``head.fc``).  Accepted as well: HF SwinForImageClassification names (separate q/k/v projections,
    downsampling at the END of a stage) and the older timm layout (``layers.i.downsample`` after stage i, ``head.*``)."""
    sd = dict(sd)
    out = {}
    if any(k.startswith("swin.") for k in sd):
        for k, v in sd.items():
            n = k.replace("swin.embeddings.patch_embeddings.projection.", "patch_embed.proj.").replace("swin.embeddings.norm.", "patch_embed.norm.")
            n = n.replace("swin.layernorm.", "norm.").replace("classifier.", "head.fc.").replace("swin.encoder.layers.", "layers.")
            n = n.replace(".layernorm_before.", ".norm1.").replace(".layernorm_after.", ".norm2.")
            n = n.replace(".attention.relative_position_bias.relative_position_bias_table", ".attn.relative_position_bias_table")
            n = n.replace(".attention.o_proj.", ".attn.proj.")
            out[n] = v
        for i, nb in enumerate(depths):
            for j in range(nb):
                p = f"layers.{i}.blocks.{j}"
                for t in ("weight", "bias"):
                    out[f"{p}.attn.qkv.{t}"] = torch.cat([out.pop(f"{p}.attention.{q}_proj.{t}") for q in "qkv"], 0)
        sd, out = out, {}
        old_layout = True                                   # HF merges at the end of stage i, like old timm
    else:
        old_layout = "layers.0.downsample.reduction.weight" in sd
    for k, v in sd.items():
        n = k
        if old_layout and ".downsample." in k:
            i = int(k.split(".")[1])
            n = k.replace(f"layers.{i}.downsample.", f"layers.{i + 1}.downsample.")
        if n in ("head.weight", "head.bias"):
            n = n.replace("head.", "head.fc.")
        if n.endswith("relative_position_index") or n.endswith("attn_mask"):
            continue                                        # buffers that are functions of the window size
        out[n] = v
    return out


class SwinVictim(_InputGradient, nn.Module):
    """``timm.create_model('swin_base_patch4_window7_224', num_classes=37)`` (ASR_fast.py:27-32) on the HIP kernels.
    Per block: LayerNorm -> cyclic shift + window partition (one gather) -> qkv GEMM -> window attention with the
    relative position bias and the shifted-window mask as an additive score bias -> projection GEMM -> inverse gather
    (+ residual) -> LayerNorm -> MLP (GELU in the GEMM epilogue, residual in the second GEMM's).  Patch merging is a
    space-to-depth gather + LayerNorm + GEMM (the 4C channel blocks are reordered on the host to the gather's order)."""

    def __init__(self, num_classes=37, embed_dim=_SWIN_BASE["embed_dim"], depths=_SWIN_BASE["depths"],
                 num_heads=_SWIN_BASE["num_heads"], window_size=_SWIN_BASE["window_size"], image_size=224, patch_size=4,
                 mlp_ratio=4.0, compute_dtype="fp32", use_graph=True):
        super().__init__()
        self.num_classes, self.embed_dim, self.depths, self.num_heads = num_classes, embed_dim, tuple(depths), tuple(num_heads)
        self.window_size, self.image_size, self.patch_size, self.mlp_ratio = window_size, image_size, patch_size, mlp_ratio
        self.compute_dtype, self.use_graph = compute_dtype, use_graph
        _attach(self, "patch_embed.proj", nn.Conv2d(3, embed_dim, patch_size, stride=patch_size))
        _attach(self, "patch_embed.norm", nn.LayerNorm(embed_dim))
        self._tables = []
        for i, nb in enumerate(self.depths):
            c = embed_dim << i
            if i > 0:
                _attach(self, f"layers.{i}.downsample.norm", nn.LayerNorm(2 * c))        # 4 * (c / 2)
                _attach(self, f"layers.{i}.downsample.reduction", nn.Linear(2 * c, c, bias=False))
            for j in range(nb):
                p = f"layers.{i}.blocks.{j}"
                _attach(self, p + ".norm1", nn.LayerNorm(c))
                _attach(self, p + ".attn.qkv", nn.Linear(c, 3 * c))
                _attach(self, p + ".attn.proj", nn.Linear(c, c))
                _attach(self, p + ".norm2", nn.LayerNorm(c))
                _attach(self, p + ".mlp.fc1", nn.Linear(c, int(c * mlp_ratio)))
                _attach(self, p + ".mlp.fc2", nn.Linear(int(c * mlp_ratio), c))
                name = p.replace(".", "__") + "__attn__rpbt"
                self.register_parameter(name, nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, self.num_heads[i])))
                self._tables.append((name, p + ".attn.relative_position_bias_table"))
        _attach(self, "norm", nn.LayerNorm(embed_dim << (len(self.depths) - 1)))
        _attach(self, "head.fc", nn.Linear(embed_dim << (len(self.depths) - 1), num_classes))
        self._packed, self._engines = {}, {}

    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        ren = dict(self._tables)
        return type(sd)((ren.get(kk, kk), v) for kk, v in sd.items())

    def load_state_dict(self, sd, strict=True, **k):
        sd = swin_canonical_state_dict(sd, self.depths)
        back = {pub: priv for priv, pub in self._tables}
        return super().load_state_dict({back.get(kk, kk): v for kk, v in sd.items()}, strict=strict, **k)

    def _version(self):
        dev = next(self.parameters()).device
        return (str(dev), sum(p._version for p in self.parameters()))

    def _stage_geometry(self):
        """(resolution, window, shift) per stage: the window shrinks to the map and the shift vanishes when the map is
        not larger than the window (SwinTransformerBlock / HF set_shift_and_window_size)."""
        res = self.image_size // self.patch_size
        out = []
        for i in range(len(self.depths)):
            win = min(self.window_size, res)
            out.append((res, win, 0 if res <= self.window_size else self.window_size // 2))
            res //= 2
        return out

    def _bias(self, table, heads, res, win, shift, dev):
        """[nW or 1][heads][win^2][win^2] f32 in units of log2(e): relative position bias (+ the mask of each window
        position for shifted blocks: pixels of different wrap-around regions must not attend to each other)."""
        if table.shape[0] != (2 * win - 1) ** 2:
            raise ValueError(f"relative_position_bias_table has {table.shape[0]} rows, window {win} needs {(2 * win - 1) ** 2}")
        co = torch.stack(torch.meshgrid(torch.arange(win), torch.arange(win), indexing="ij")).flatten(1)      # 2, win^2
        rel = (co[:, :, None] - co[:, None, :]).permute(1, 2, 0) + (win - 1)
        idx = rel[..., 0] * (2 * win - 1) + rel[..., 1]
        bias = table.float().cpu()[idx.reshape(-1)].reshape(win * win, win * win, heads).permute(2, 0, 1)[None]   # 1, heads, n, n
        if shift > 0:
            reg = lambda n: (torch.arange(n) >= n - win).long() + (torch.arange(n) >= n - shift).long()
            img = reg(res)[:, None] * 3 + reg(res)[None, :]                                                   # res x res region ids
            mw = img.reshape(res // win, win, res // win, win).permute(0, 2, 1, 3).reshape(-1, win * win)   # nW, n
            mask = (mw[:, None, :] != mw[:, :, None]).float() * -100.0                                        # nW, n, n
            bias = bias + mask[:, None]
        return (bias * 1.4426950408889634).contiguous().to(dev)

    def packed_weights(self, dt):
        ver = self._version()
        hit = self._packed.get(dt)
        if hit is not None and hit[0] == ver:
            return hit[1]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AdvsError(f"SwinVictim parameters are on {dev}: move the model to the GPU; there is no CPU fallback")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        f32 = lambda k: sd[k].float().contiguous()
        lin = lambda k: pack_conv_weight(sd[k].float().reshape(sd[k].shape[0], -1, 1, 1), dt)
        W = {"pe.w": lin("patch_embed.proj.weight"), "pe.b": f32("patch_embed.proj.bias"),
             "pe.g": f32("patch_embed.norm.weight"), "pe.beta": f32("patch_embed.norm.bias")}
        geo = self._stage_geometry()
        for i, nb in enumerate(self.depths):
            c = self.embed_dim << i
            res, win, shift = geo[i]
            if i > 0:
                p = f"layers.{i}.downsample"
                cp = c // 2                                          # channels before merging
                # timm / HF concatenate [x(0,0), x(1,0), x(0,1), x(1,1)] (dy, dx); advs_space_to_depth2 emits (0,0),(0,1),(1,0),(1,1)
                perm = torch.cat([torch.arange(cp) + blk * cp for blk in (0, 2, 1, 3)]).to(dev)
                W[p + ".g"], W[p + ".beta"] = f32(p + ".norm.weight")[perm].contiguous(), f32(p + ".norm.bias")[perm].contiguous()
                W[p + ".w"] = pack_conv_weight(sd[p + ".reduction.weight"].float()[:, perm].reshape(c, 4 * cp, 1, 1), dt)
            for j in range(nb):
                p = f"layers.{i}.blocks.{j}"
                for n in ("norm1", "norm2"):
                    W[f"{p}.{n}.g"], W[f"{p}.{n}.beta"] = f32(f"{p}.{n}.weight"), f32(f"{p}.{n}.bias")
                for src, dst in ((".attn.qkv", ".qkv"), (".attn.proj", ".proj"), (".mlp.fc1", ".fc1"), (".mlp.fc2", ".fc2")):
                    W[p + dst + ".w"], W[p + dst + ".b"] = lin(p + src + ".weight"), f32(p + src + ".bias")
                W[p + ".bias"] = self._bias(sd[p + ".attn.relative_position_bias_table"], self.num_heads[i], res, win,
                                            shift if j % 2 else 0, dev)
        W["norm.g"], W["norm.beta"] = f32("norm.weight"), f32("norm.bias")
        W["head.w"], W["head.b"] = f32("head.fc.weight"), f32("head.fc.bias")
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[-1] == dt]:      # (batch, dt) and ("grad", batch, dt)
            del self._engines[key]
        return W

    def engine(self, batch, dtype=None):
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        eng = self._engines.get((batch, dt))
        if eng is None:
            eng = _SwinEngine(self, W, batch, dt)
            self._engines[(batch, dt)] = eng
        return eng

    # ---- backward to the image (the gradient attack of tools/train_shadow.py:177-221 with this victim) ------------------
    def packed_grad_weights(self, dt):
        """Every Linear's weight transposed (its data gradient is a GEMM against W); patch merging in the gather's channel order."""
        ver = self._version()
        hit = self._packed.get(("grad", dt))
        if hit is not None and hit[0] == ver:
            return hit[1]
        self.packed_weights(dt)
        dev = next(self.parameters()).device
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        linT = lambda w: pack_conv_weight(w.float().t().contiguous().reshape(w.shape[1], w.shape[0], 1, 1), dt)
        G = {"peT": linT(sd["patch_embed.proj.weight"].float().reshape(self.embed_dim, -1)),
             "head.wT": sd["head.fc.weight"].float().t().contiguous()}
        for i, nb in enumerate(self.depths):
            c = self.embed_dim << i
            if i > 0:
                p = f"layers.{i}.downsample"
                cp = c // 2
                perm = torch.cat([torch.arange(cp) + blk * cp for blk in (0, 2, 1, 3)]).to(dev)
                G[p + ".wT"] = linT(sd[p + ".reduction.weight"].float()[:, perm])
            for j in range(nb):
                p = f"layers.{i}.blocks.{j}"
                for src, dst in ((".attn.qkv", ".qkvT"), (".attn.proj", ".projT"), (".mlp.fc1", ".fc1T"), (".mlp.fc2", ".fc2T")):
                    G[p + dst] = linT(sd[p + src + ".weight"])
        self._packed[("grad", dt)] = (ver, G)
        return G

    def grad_engine(self, batch, size=None, dtype=None):
        """Static plan of forward + backward-to-the-image (d cross_entropy / d input) for [batch,3,S,S] inputs."""
        if size is not None and size != self.image_size:
            raise ValueError(f"SwinVictim was built for {self.image_size}x{self.image_size} inputs, not {size}")
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W, G = self.packed_weights(dt), self.packed_grad_weights(dt)
        eng = self._engines.get(("grad", batch, dt))
        if eng is None:
            eng = _SwinGradEngine(self, W, G, batch, dt)
            self._engines[("grad", batch, dt)] = eng
        return eng

    def forward(self, x):
        if x.shape[2] != self.image_size or x.shape[3] != self.image_size:
            raise ValueError(f"SwinVictim was built for {self.image_size}x{self.image_size} inputs, got {tuple(x.shape[2:])}")
        eng = self.engine(x.shape[0])
        cur = torch.cuda.current_stream(x.device)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x.to(torch.float32), non_blocking=True)
            eng.run()
            out = eng.logits.clone()
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        return out


class _SwinGradEngine:
    """Swin forward with what the reverse sweep needs retained (each block's input, qkv, attention output, the stream after the
    attention branch and the MLP pre-activation; the gathered input of every patch-merging LayerNorm), then the blocks backwards:
        d f = d out W_fc2';  d pre = d f * gelu'(pre);  d a = d out + LN2'(d pre W_fc1'; a)
        d pr = partition(shift(d a));  d qkv = attention'(d pr W_proj'; qkv, att, bias);  d in = d a + LN1'(unshift(merge(d qkv W_qkv')); in)
    -- the gradient of the inverse window gather is the forward gather and vice versa (both are permutations)."""

    def __init__(self, model, W, G, batch, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        S, ps = model.image_size, model.patch_size
        geo = model._stage_geometry()
        for res, win, _ in geo:
            if res % win:
                raise ValueError(f"SwinVictim: a {res}x{res} map is not a whole number of {win}-pixel windows")
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib, plan = bld.lib, bld.plan
            self.x = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            self.labels = torch.zeros((batch,), dtype=torch.int64, device=dev)
            self.grad = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            g = S // ps
            kcols = 3 * ps * ps
            # ---- forward
            patches = bld.buf((batch, g, g, kcols))
            plan.add(lib.advs_patchify, ptr(self.x), ptr(patches), batch, 3, S, S, ps, dt, keep=(self.x, patches))
            e = bld.conv(patches, W["pe.w"], model.embed_dim, bias=W["pe.b"], ksize=1, pad=0)
            bld.free(patches)
            h = bld.layernorm(e, W["pe.g"], W["pe.beta"], 1e-5)
            tape = [("pe", e)]
            max_scratch = 0
            for i, nb in enumerate(model.depths):
                c = model.embed_dim << i
                res, win, shift = geo[i]
                heads = model.num_heads[i]
                d = c // heads
                if i > 0:
                    p = f"layers.{i}.downsample"
                    s2d = bld.buf((batch, res, res, 2 * c))
                    plan.add(lib.advs_space_to_depth2, ptr(h), ptr(s2d), batch, 2 * res, 2 * res, c // 2, dt, keep=(h, s2d))
                    bld.free(h)
                    ln = bld.layernorm(s2d, W[p + ".g"], W[p + ".beta"], 1e-5)
                    h = bld.conv(ln, W[p + ".w"], c, ksize=1, pad=0)
                    bld.free(ln)
                    tape.append(("down", p, s2d, res, c))
                nW = (res // win) ** 2
                max_scratch = max(max_scratch, lib.advs_attention_bwd_scratch_bytes(batch * nW, win * win, heads))
                for j in range(nb):
                    p = f"layers.{i}.blocks.{j}"
                    sh = shift if j % 2 else 0
                    ln = bld.layernorm(h, W[p + ".norm1.g"], W[p + ".norm1.beta"], 1e-5)
                    wins = bld.window_shift(ln, win, sh)
                    bld.free(ln)
                    qkv = bld.conv(wins, W[p + ".qkv.w"], 3 * c, bias=W[p + ".qkv.b"], ksize=1, pad=0)
                    bld.free(wins)
                    att = bld.attention_bias(qkv, heads, d, 0, c, 2 * c, d, W[p + ".bias"], nW if sh else 1)
                    pr = bld.conv(att, W[p + ".proj.w"], c, bias=W[p + ".proj.b"], ksize=1, pad=0)
                    a = bld.window_shift(pr, win, sh, inverse=True, residual=h, image_hw=(res, res))
                    bld.free(pr)
                    ln = bld.layernorm(a, W[p + ".norm2.g"], W[p + ".norm2.beta"], 1e-5)
                    pre = bld.conv(ln, W[p + ".fc1.w"], W[p + ".fc1.b"].numel(), bias=W[p + ".fc1.b"], ksize=1, pad=0)
                    bld.free(ln)
                    f = bld.buf(tuple(pre.shape))
                    plan.add(lib.advs_gelu, ptr(pre), ptr(f), pre.numel(), dt, keep=(pre, f))
                    out = bld.conv(f, W[p + ".fc2.w"], c, bias=W[p + ".fc2.b"], residual=a, ksize=1, pad=0)
                    bld.free(f)
                    tape.append(("block", p, h, qkv, att, a, pre, res, win, sh, c, heads, d, nW))
                    h = out
            cl = model.embed_dim << (len(model.depths) - 1)
            rl = geo[-1][0]
            ln = bld.layernorm(h, W["norm.g"], W["norm.beta"], 1e-5)
            pooled = bld.buf((batch, cl), torch.float32)
            plan.add(lib.advs_global_avgpool, ptr(ln), ptr(pooled), batch, rl * rl, cl, dt, keep=(ln, pooled))
            bld.free(ln)
            self.logits = bld.linear(pooled, W["head.w"], W["head.b"])
            # ---- backward
            K = self.logits.shape[1]
            gl = bld.buf((batch, K), torch.float32)
            plan.add(lib.advs_softmax_ce_grad, ptr(self.logits), ptr(self.labels), ptr(gl), batch, K, 1.0, keep=(self.logits, self.labels, gl))
            gp = bld.linear(gl, G["head.wT"], None)                                  # [B, cl] f32
            dln = bld.buf((batch, rl, rl, cl))
            plan.add(lib.advs_avgpool_bwd, ptr(gp), ptr(dln), batch, rl * rl, cl, dt, keep=(gp, dln))
            scratch = torch.empty(max_scratch, dtype=torch.uint8, device=dev)

            def ln_bwd(dy, x, gamma, add):
                dx = bld.buf(tuple(x.shape))
                rows = x.numel() // x.shape[-1]
                plan.add(lib.advs_layernorm_bwd, ptr(dy), ptr(x), ptr(gamma), ptr(add), ptr(dx), rows, x.shape[-1], 1e-5, dt,
                         keep=(dy, x, gamma, add, dx))
                return dx

            dh = ln_bwd(dln, h, W["norm.g"], None)
            bld.free(dln); bld.free(h)
            for rec in reversed(tape):
                if rec[0] == "block":
                    _, p, hin, qkv, att, a, pre, res, win, sh, c, heads, d, nW = rec
                    df = bld.conv(dh, G[p + ".fc2T"], pre.shape[-1], ksize=1, pad=0)
                    dpre = bld.buf(tuple(pre.shape))
                    plan.add(lib.advs_gelu_bwd, ptr(pre), ptr(df), ptr(dpre), pre.numel(), dt, keep=(pre, df, dpre))
                    bld.free(df); bld.free(pre)
                    dl2 = bld.conv(dpre, G[p + ".fc1T"], c, ksize=1, pad=0)
                    bld.free(dpre)
                    da = ln_bwd(dl2, a, W[p + ".norm2.g"], dh)
                    bld.free(dl2); bld.free(dh); bld.free(a)
                    dpr = bld.window_shift(da, win, sh)                               # gradient of the inverse gather = the gather
                    datt = bld.conv(dpr, G[p + ".projT"], c, ksize=1, pad=0)
                    bld.free(dpr)
                    dqkv = bld.buf(tuple(qkv.shape))
                    plan.add(lib.advs_attention_bias_bwd, ptr(qkv), ptr(att), ptr(datt), ptr(dqkv), ptr(scratch), ptr(W[p + ".bias"]),
                             nW if sh else 1, batch * nW, win * win, heads, d, 3 * c, 0, c, 2 * c, d, dt,
                             keep=(qkv, att, datt, dqkv, scratch, W[p + ".bias"]))
                    bld.free(datt); bld.free(qkv); bld.free(att)
                    dwins = bld.conv(dqkv, G[p + ".qkvT"], c, ksize=1, pad=0)
                    bld.free(dqkv)
                    dl1 = bld.window_shift(dwins, win, sh, inverse=True, residual=None, image_hw=(res, res))
                    bld.free(dwins)
                    dh = ln_bwd(dl1, hin, W[p + ".norm1.g"], da)
                    bld.free(dl1); bld.free(da); bld.free(hin)
                elif rec[0] == "down":
                    _, p, s2d, res, c = rec
                    dlo = bld.conv(dh, G[p + ".wT"], 2 * c, ksize=1, pad=0)
                    bld.free(dh)
                    ds2d = ln_bwd(dlo, s2d, W[p + ".g"], None)
                    bld.free(dlo); bld.free(s2d)
                    dh = bld.buf((batch, 2 * res, 2 * res, c // 2))
                    plan.add(lib.advs_depth_to_space2, ptr(ds2d), ptr(dh), batch, 2 * res, 2 * res, c // 2, dt, keep=(ds2d, dh))
                    bld.free(ds2d)
                else:
                    _, e = rec
                    de = ln_bwd(dh, e, W["pe.g"], None)
                    bld.free(dh); bld.free(e)
                    dcols = bld.conv(de, G["peT"], kcols, ksize=1, pad=0)
                    bld.free(de)
                    plan.add(lib.advs_unpatchify_padded, ptr(dcols), ptr(self.grad), batch, 3, S, S, ps, kcols, g * g, 0, dt, keep=(dcols, self.grad))
            self.plan, self.captured = plan, False
            torch.cuda.synchronize(dev)

    def run(self):
        if self.model.use_graph and not self.captured:
            self.plan.run_eager()
            self.stream.synchronize()
            self.plan.capture()
            self.captured = True
        self.plan.run()


class _SwinEngine:
    def __init__(self, model, W, batch, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        S, ps = model.image_size, model.patch_size
        geo = model._stage_geometry()
        for res, win, _ in geo:
            if res % win:
                raise ValueError(f"SwinVictim: a {res}x{res} map is not a whole number of {win}-pixel windows")
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib = bld.lib
            self.x = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            g = S // ps
            patches = bld.buf((batch, g, g, 3 * ps * ps))
            bld.plan.add(lib.advs_patchify, ptr(self.x), ptr(patches), batch, 3, S, S, ps, dt, keep=(self.x, patches))
            e = bld.conv(patches, W["pe.w"], model.embed_dim, bias=W["pe.b"], ksize=1, pad=0)
            bld.free(patches)
            h = bld.layernorm(e, W["pe.g"], W["pe.beta"], 1e-5)
            bld.free(e)
            for i, nb in enumerate(model.depths):
                c = model.embed_dim << i
                res, win, shift = geo[i]
                heads = model.num_heads[i]
                d = c // heads
                if i > 0:
                    p = f"layers.{i}.downsample"
                    s2d = bld.buf((batch, res, res, 2 * c))
                    bld.plan.add(lib.advs_space_to_depth2, ptr(h), ptr(s2d), batch, 2 * res, 2 * res, c // 2, dt, keep=(h, s2d))
                    bld.free(h)
                    ln = bld.layernorm(s2d, W[p + ".g"], W[p + ".beta"], 1e-5)
                    bld.free(s2d)
                    h = bld.conv(ln, W[p + ".w"], c, ksize=1, pad=0)
                    bld.free(ln)
                nW = (res // win) ** 2
                for j in range(nb):
                    p = f"layers.{i}.blocks.{j}"
                    sh = shift if j % 2 else 0
                    ln = bld.layernorm(h, W[p + ".norm1.g"], W[p + ".norm1.beta"], 1e-5)
                    wins = bld.window_shift(ln, win, sh)
                    bld.free(ln)
                    qkv = bld.conv(wins, W[p + ".qkv.w"], 3 * c, bias=W[p + ".qkv.b"], ksize=1, pad=0)
                    bld.free(wins)
                    att = bld.attention_bias(qkv, heads, d, 0, c, 2 * c, d, W[p + ".bias"], nW if sh else 1)
                    bld.free(qkv)
                    pr = bld.conv(att, W[p + ".proj.w"], c, bias=W[p + ".proj.b"], ksize=1, pad=0)
                    bld.free(att)
                    a = bld.window_shift(pr, win, sh, inverse=True, residual=h, image_hw=(res, res))
                    bld.free(pr)
                    bld.free(h)
                    ln = bld.layernorm(a, W[p + ".norm2.g"], W[p + ".norm2.beta"], 1e-5)
                    f = bld.conv(ln, W[p + ".fc1.w"], W[p + ".fc1.b"].numel(), bias=W[p + ".fc1.b"], act="gelu", ksize=1, pad=0)
                    bld.free(ln)
                    h = bld.conv(f, W[p + ".fc2.w"], c, bias=W[p + ".fc2.b"], residual=a, ksize=1, pad=0)
                    bld.free(f)
                    bld.free(a)
            ln = bld.layernorm(h, W["norm.g"], W["norm.beta"], 1e-5)
            bld.free(h)
            cl = model.embed_dim << (len(model.depths) - 1)
            rl = geo[-1][0]
            pooled = bld.buf((batch, cl), torch.float32)
            bld.plan.add(lib.advs_global_avgpool, ptr(ln), ptr(pooled), batch, rl * rl, cl, dt, keep=(ln, pooled))
            self.logits = bld.linear(pooled, W["head.w"], W["head.b"])
            self.plan, self.captured = bld.plan, False
            torch.cuda.synchronize(dev)

    def run(self):
        if self.model.use_graph and not self.captured:
            self.plan.run_eager()
            self.stream.synchronize()
            self.plan.capture()
            self.captured = True
        self.plan.run()



# ============================================================================ DINOv2 (HF names)
class Dinov2Victim(_InputGradient, nn.Module):
    """HF ``Dinov2ForImageClassification`` (the other ``AutoModelForImageClassification`` checkpoint family of
    ASR_fast.py:47-58) on the ViT engine: patch 14, LayerScale after attention and MLP (folded into the projection /
    fc2 weights), position embeddings interpolated bicubically on the host when the checkpoint's grid differs from the
    input's (Dinov2Embeddings.interpolate_pos_encoding), classifier on [cls | mean of the patch tokens].
    The SwiGLU feed-forward of the giant variant is not built."""

    def __init__(self, num_labels=37, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4,
                 patch_size=14, image_size=224, pos_grid=None, layer_norm_eps=1e-6, compute_dtype="fp32", use_graph=True):
        super().__init__()
        if image_size % patch_size:
            raise ValueError("Dinov2Victim: image_size must be a multiple of patch_size")
        grid = pos_grid or image_size // patch_size          # pos_grid: the grid the checkpoint was trained with (e.g. 37)
        self.cfg = dict(num_labels=num_labels, hidden=hidden_size, layers=num_hidden_layers, heads=num_attention_heads,
                        mlp=hidden_size * mlp_ratio, patch=patch_size, image=image_size, eps=layer_norm_eps, head="cls_mean")
        self.pos_grid, self.compute_dtype, self.use_graph = grid, compute_dtype, use_graph
        C = hidden_size
        emb = nn.Module()
        emb.cls_token = nn.Parameter(torch.randn(1, 1, C) * 0.02)
        emb.mask_token = nn.Parameter(torch.zeros(1, C))
        emb.position_embeddings = nn.Parameter(torch.randn(1, grid * grid + 1, C) * 0.02)
        _attach(self, "dinov2.embeddings", emb)
        _attach(self, "dinov2.embeddings.patch_embeddings.projection", nn.Conv2d(3, C, patch_size, stride=patch_size))
        for i in range(num_hidden_layers):
            p = f"dinov2.encoder.layer.{i}"
            _attach(self, p + ".norm1", nn.LayerNorm(C, eps=layer_norm_eps))
            for n in ("query", "key", "value"):
                _attach(self, f"{p}.attention.attention.{n}", nn.Linear(C, C))
            _attach(self, p + ".attention.output.dense", nn.Linear(C, C))
            for k in (1, 2):
                ls = nn.Module()
                ls.lambda1 = nn.Parameter(torch.ones(C))
                _attach(self, f"{p}.layer_scale{k}", ls)
            _attach(self, p + ".norm2", nn.LayerNorm(C, eps=layer_norm_eps))
            _attach(self, p + ".mlp.fc1", nn.Linear(C, C * mlp_ratio))
            _attach(self, p + ".mlp.fc2", nn.Linear(C * mlp_ratio, C))
        _attach(self, "dinov2.layernorm", nn.LayerNorm(C, eps=layer_norm_eps))
        _attach(self, "classifier", nn.Linear(2 * C, num_labels))
        self._packed, self._engines = {}, {}

    def _version(self):
        return (str(next(self.parameters()).device), sum(p._version for p in self.parameters()))

    def _pos(self, pe):
        """Dinov2Embeddings.interpolate_pos_encoding on the host (bicubic, align_corners=False), once per weight set."""
        g_in, g_out = self.pos_grid, self.cfg["image"] // self.cfg["patch"]
        if g_in == g_out:
            return pe[0].float().contiguous()
        C = pe.shape[-1]
        patch = pe[:, 1:].float().reshape(1, g_in, g_in, C).permute(0, 3, 1, 2)
        patch = torch.nn.functional.interpolate(patch, size=(g_out, g_out), mode="bicubic", align_corners=False)
        return torch.cat([pe[0, :1].float(), patch.permute(0, 2, 3, 1).reshape(-1, C)], 0).contiguous()

    def packed_weights(self, dt):
        ver = self._version()
        hit = self._packed.get(dt)
        if hit is not None and hit[0] == ver:
            return hit[1]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AdvsError(f"Dinov2Victim parameters are on {dev}: move the model to the GPU; there is no CPU fallback")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        f32 = lambda k: sd[k].float().contiguous()
        lin = lambda w: pack_conv_weight(w.float().reshape(w.shape[0], -1, 1, 1), dt)
        W = {"cls": f32("dinov2.embeddings.cls_token").reshape(-1), "pos": self._pos(sd["dinov2.embeddings.position_embeddings"]),
             "proj.w": _patch_proj(sd["dinov2.embeddings.patch_embeddings.projection.weight"], dt),
             "proj.b": f32("dinov2.embeddings.patch_embeddings.projection.bias")}
        for i in range(self.cfg["layers"]):
            s, p = f"dinov2.encoder.layer.{i}", f"vit.encoder.layer.{i}"            # the ViT engine's key scheme
            a = s + ".attention.attention."
            W[p + ".qkv.w"] = lin(torch.cat([sd[a + n + ".weight"] for n in ("query", "key", "value")], 0))
            W[p + ".qkv.b"] = torch.cat([sd[a + n + ".bias"].float() for n in ("query", "key", "value")], 0).contiguous()
            l1, l2 = sd[s + ".layer_scale1.lambda1"].float(), sd[s + ".layer_scale2.lambda1"].float()
            W[p + ".o.w"] = lin(sd[s + ".attention.output.dense.weight"].float() * l1[:, None])
            W[p + ".o.b"] = (sd[s + ".attention.output.dense.bias"].float() * l1).contiguous()
            W[p + ".fc1.w"], W[p + ".fc1.b"] = lin(sd[s + ".mlp.fc1.weight"]), f32(s + ".mlp.fc1.bias")
            W[p + ".fc2.w"] = lin(sd[s + ".mlp.fc2.weight"].float() * l2[:, None])
            W[p + ".fc2.b"] = (sd[s + ".mlp.fc2.bias"].float() * l2).contiguous()
            for src, dst in ((".norm1", ".layernorm_before"), (".norm2", ".layernorm_after")):
                W[p + dst + ".g"], W[p + dst + ".b"] = f32(s + src + ".weight"), f32(s + src + ".bias")
        W["ln.g"], W["ln.b"] = f32("dinov2.layernorm.weight"), f32("dinov2.layernorm.bias")
        W["cls.w"], W["cls.b"] = f32("classifier.weight"), f32("classifier.bias")
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[-1] == dt]:      # (batch, dt) and ("grad", batch, dt)
            del self._engines[key]
        return W

    def engine(self, batch, dtype=None):
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        eng = self._engines.get((batch, dt))
        if eng is None:
            eng = _ViTEngine(self, W, batch, dt)
            self._engines[(batch, dt)] = eng
        return eng

    forward = ViTVictim.forward

    # ---- backward to the image: the ViT plan with LayerScale folded into the transposed projection / fc2 weights and the
    # [cls | mean] head run backwards (advs_scatter_cls_mean)
    def packed_grad_weights(self, dt):
        ver = self._version()
        hit = self._packed.get(("grad", dt))
        if hit is not None and hit[0] == ver:
            return hit[1]
        self.packed_weights(dt)
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        C = self.cfg["hidden"]
        linT = lambda w: pack_conv_weight(w.float().t().contiguous().reshape(w.shape[1], w.shape[0], 1, 1), dt)
        G = {}
        for i in range(self.cfg["layers"]):
            s, p = f"dinov2.encoder.layer.{i}", f"vit.encoder.layer.{i}"
            a = s + ".attention.attention."
            G[p + ".qkvT"] = linT(torch.cat([sd[a + n + ".weight"].float() for n in ("query", "key", "value")], 0))
            l1, l2 = sd[s + ".layer_scale1.lambda1"].float(), sd[s + ".layer_scale2.lambda1"].float()
            G[p + ".oT"] = linT(sd[s + ".attention.output.dense.weight"].float() * l1[:, None])
            G[p + ".fc1T"] = linT(sd[s + ".mlp.fc1.weight"])
            G[p + ".fc2T"] = linT(sd[s + ".mlp.fc2.weight"].float() * l2[:, None])
        wp = sd["dinov2.embeddings.patch_embeddings.projection.weight"].float().reshape(C, -1)        # [C, 3*ps*ps]
        k = wp.shape[1]
        kpad = -(-k // 64) * 64
        wt = torch.zeros((kpad, C), dtype=torch.float32, device=wp.device)
        wt[:k] = wp.t()
        G["projT"] = pack_conv_weight(wt.reshape(kpad, C, 1, 1), dt)
        G["cls.wT"] = sd["classifier.weight"].float().t().contiguous()
        self._packed[("grad", dt)] = (ver, G)
        return G

    def grad_engine(self, batch, size=None, dtype=None):
        """Static plan of forward + backward-to-the-image (d cross_entropy / d pixel_values) for [batch,3,S,S] inputs."""
        if size is not None and size != self.cfg["image"]:
            raise ValueError(f"Dinov2Victim was built for {self.cfg['image']}x{self.cfg['image']} inputs, not {size}")
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W, G = self.packed_weights(dt), self.packed_grad_weights(dt)
        eng = self._engines.get(("grad", batch, dt))
        if eng is None:
            eng = _ViTGradEngine(self, W, G, batch, dt)
            self._engines[("grad", batch, dt)] = eng
        return eng


# ============================================================================ EfficientNetV2-S (torchvision names)
_EFFNETV2_S = [("fused", 1, 3, 1, 24, 24, 2), ("fused", 4, 3, 2, 24, 48, 4), ("fused", 4, 3, 2, 48, 64, 4),
               ("mb", 4, 3, 2, 64, 128, 6), ("mb", 6, 3, 1, 128, 160, 9), ("mb", 6, 3, 2, 160, 256, 15)]


class EfficientNetV2S(_InputGradient, nn.Module):
    """torchvision ``efficientnet_v2_s`` with ``classifier[1] = Linear(1280, 37)`` (ASR_fast.py:59-65) on the HIP
    kernels: BatchNorm (eps 1e-3) folded into every conv, FusedMBConv = 3x3 GEMM (+1x1 projection), MBConv = 1x1
    expansion -> depthwise 3x3 (+SiLU) -> squeeze-excitation (pool, two tiny Linears, per-channel scale) -> 1x1
    projection (+residual in the epilogue).  Parameter names follow torchvision (``features.s.b.block.k.{0,1}``,
    ``block.2.fc{1,2}``, ``classifier.1``).  torchvision is not installed here: PARITY UNPINNED (the architecture is
    restated in oracle/victims.py and its parameter count equals the published 21,458,488 for 1000 classes)."""

    def __init__(self, num_classes=37, setting=None, last_channel=1280, image_size=224, compute_dtype="fp32", use_graph=True):
        super().__init__()
        self.num_classes, self.setting, self.last = num_classes, [tuple(s) for s in (setting or _EFFNETV2_S)], last_channel
        self.image_size, self.compute_dtype, self.use_graph = image_size, compute_dtype, use_graph
        self.layout = [("cna", "features.0", 3, self.setting[0][4], 3, 2)]
        for si, (kind, e, k, s, cin, cout, n) in enumerate(self.setting, start=1):
            for b in range(n):
                ci, st = (cin, s) if b == 0 else (cout, 1)
                self.layout.append((kind, f"features.{si}.{b}", e, k, st, ci, cout))
        self.layout.append(("cna", f"features.{len(self.setting) + 1}", self.setting[-1][5], last_channel, 1, 1))
        for item in self.layout:
            if item[0] == "cna":
                self._cna(item[1], item[2], item[3], item[4], item[5])
                continue
            kind, p, e, k, st, ci, co = item
            ce = ci * e
            if kind == "fused":
                if e != 1:
                    self._cna(p + ".block.0", ci, ce, k, st); self._cna(p + ".block.1", ce, co, 1, 1)
                else:
                    self._cna(p + ".block.0", ci, co, k, st)
            else:
                self._cna(p + ".block.0", ci, ce, 1, 1); self._cna(p + ".block.1", ce, ce, k, st, groups=ce)
                sq = max(1, ci // 4)
                _attach(self, p + ".block.2.fc1", nn.Conv2d(ce, sq, 1)); _attach(self, p + ".block.2.fc2", nn.Conv2d(sq, ce, 1))
                self._cna(p + ".block.3", ce, co, 1, 1)
        _attach(self, "classifier.1", nn.Linear(last_channel, num_classes))
        self._packed, self._engines = {}, {}

    def _cna(self, p, cin, cout, k, stride, groups=1):
        _attach(self, p + ".0", nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, groups=groups, bias=False))
        _attach(self, p + ".1", nn.BatchNorm2d(cout, eps=1e-3))

    def _version(self):
        dev = next(self.parameters()).device
        return (str(dev), sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers()))

    @staticmethod
    def _pad8(c):
        return -(-c // 32) * 32 if c < 32 else c              # the 24-channel stage is carried in 32 channels (zeros)

    def packed_weights(self, dt):
        ver = self._version()
        hit = self._packed.get(dt)
        if hit is not None and hit[0] == ver:
            return hit[1]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AdvsError(f"EfficientNetV2S parameters are on {dev}: move the model to the GPU; there is no CPU fallback")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        W = {}
        pad = self._pad8

        def fold(p):                                           # conv weight * bn scale, bn shift as bias (eps 1e-3)
            scale = sd[p + ".1.weight"].float() / torch.sqrt(sd[p + ".1.running_var"].float() + 1e-3)
            return sd[p + ".0.weight"].float() * scale[:, None, None, None], sd[p + ".1.bias"].float() - sd[p + ".1.running_mean"].float() * scale

        def gemm(p, name):                                     # zero-pad narrow channel counts on both sides, then pack
            w, b = fold(p)
            co, ci = pad(w.shape[0]), pad(w.shape[1])
            wp = torch.zeros((co, ci) + tuple(w.shape[2:]), device=dev)
            wp[:w.shape[0], :w.shape[1]] = w
            bp = torch.zeros(co, device=dev)
            bp[:b.numel()] = b
            W[name + ".w"], W[name + ".b"] = pack_conv_weight(wp, dt), bp.contiguous()

        w0, b0 = fold("features.0")
        c0 = pad(w0.shape[0])
        ws = torch.zeros((c0, 3, 3, 3), device=dev); ws[:w0.shape[0]] = w0
        bs = torch.zeros(c0, device=dev); bs[:b0.numel()] = b0
        W["stem.w"], W["stem.b"] = ws.contiguous(), bs.contiguous()
        for item in self.layout[1:]:
            if item[0] == "cna":
                gemm(item[1], "head")
                continue
            kind, p, e, k, st, ci, co = item
            if kind == "fused":
                gemm(p + ".block.0", p + ".c0")
                if e != 1:
                    gemm(p + ".block.1", p + ".c1")
            else:
                gemm(p + ".block.0", p + ".c0")
                wd, bd = fold(p + ".block.1")
                W[p + ".dw.w"] = wd.reshape(wd.shape[0], k * k).t().contiguous()                     # [k*k][C]
                W[p + ".dw.b"] = bd.contiguous()
                for n in ("fc1", "fc2"):
                    W[f"{p}.{n}.w"] = sd[f"{p}.block.2.{n}.weight"].float().reshape(sd[f"{p}.block.2.{n}.weight"].shape[0], -1).contiguous()
                    W[f"{p}.{n}.b"] = sd[f"{p}.block.2.{n}.bias"].float().contiguous()
                gemm(p + ".block.3", p + ".c3")
        W["cls.w"], W["cls.b"] = sd["classifier.1.weight"].float().contiguous(), sd["classifier.1.bias"].float().contiguous()
        self._packed[dt] = (ver, W)
        for key in [k for k in self._engines if k[-1] == dt]:      # (batch, dt) and ("grad", batch, dt)
            del self._engines[key]
        return W

    def engine(self, batch, dtype=None):
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W = self.packed_weights(dt)
        eng = self._engines.get((batch, dt))
        if eng is None:
            eng = _EffNetEngine(self, W, batch, dt)
            self._engines[(batch, dt)] = eng
        return eng

    # ---- backward to the image (the gradient attack of tools/train_shadow.py:177-221 with this victim) ------------------
    def packed_grad_weights(self, dt):
        """Weights of the data-gradient convs: BatchNorm folded and channels zero-padded as forwards, then 1x1 transposed,
        3x3 transposed and flipped; the classifier weight transposed."""
        ver = self._version()
        hit = self._packed.get(("grad", dt))
        if hit is not None and hit[0] == ver:
            return hit[1]
        self.packed_weights(dt)
        dev = next(self.parameters()).device
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        pad = self._pad8
        G = {}

        def fold(p):
            scale = sd[p + ".1.weight"].float() / torch.sqrt(sd[p + ".1.running_var"].float() + 1e-3)
            return sd[p + ".0.weight"].float() * scale[:, None, None, None]

        def gemmT(p, name):
            w = fold(p)
            co, ci = pad(w.shape[0]), pad(w.shape[1])
            wp = torch.zeros((co, ci) + tuple(w.shape[2:]), device=dev)
            wp[:w.shape[0], :w.shape[1]] = w
            G[name] = pack_conv_weight(wp.permute(1, 0, 2, 3).flip(2, 3).contiguous(), dt)

        w0 = fold("features.0")
        c0 = pad(w0.shape[0])
        ws = torch.zeros((c0, 3, 3, 3), device=dev); ws[:w0.shape[0]] = w0
        G["stem.w"] = ws.contiguous()                         # advs_conv_stem_bwd reads the forward OIHW weight
        for item in self.layout[1:]:
            if item[0] == "cna":
                gemmT(item[1], "headT")
                continue
            kind, p, e, k, st, ci, co = item
            gemmT(p + ".block.0", p + ".c0T")
            if kind == "fused":
                if e != 1:
                    gemmT(p + ".block.1", p + ".c1T")
            else:
                gemmT(p + ".block.3", p + ".c3T")
                for n in ("fc1", "fc2"):                     # the squeeze-excitation Linears backwards: advs_linear_f32 on W'
                    wse = sd[f"{p}.block.2.{n}.weight"].float()
                    G[f"{p}.{n}T"] = wse.reshape(wse.shape[0], -1).t().contiguous()
        G["cls.wT"] = sd["classifier.1.weight"].float().t().contiguous()
        self._packed[("grad", dt)] = (ver, G)
        return G

    def grad_engine(self, batch, size=None, dtype=None):
        """Static plan of forward + backward-to-the-image (d cross_entropy / d input) for [batch,3,S,S] inputs."""
        if size is not None and size != self.image_size:
            raise ValueError(f"EfficientNetV2S was built for {self.image_size}x{self.image_size} inputs, not {size}")
        dt = dtype_code(dtype if dtype is not None else self.compute_dtype)
        W, G = self.packed_weights(dt), self.packed_grad_weights(dt)
        eng = self._engines.get(("grad", batch, dt))
        if eng is None:
            eng = _EffNetGradEngine(self, W, G, batch, dt)
            self._engines[("grad", batch, dt)] = eng
        return eng

    def forward(self, x):
        if x.shape[2] != self.image_size or x.shape[3] != self.image_size:
            raise ValueError(f"EfficientNetV2S was built for {self.image_size}x{self.image_size} inputs, got {tuple(x.shape[2:])}")
        eng = self.engine(x.shape[0])
        cur = torch.cuda.current_stream(x.device)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x.to(torch.float32), non_blocking=True)
            eng.run()
            out = eng.logits.clone()
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        return out


class _EffNetGradEngine:
    """EfficientNetV2-S forward with every SiLU as its own pass (the pre-activations are what the reverse sweep needs), then backwards:
      FusedMBConv: d pre = d m * silu'(pre), d in = conv3x3'(d pre) (zero insertion in front of it for stride 2) (+ d out for the residual);
      MBConv: d sc = d out W3';  gs = sum_p d sc * d;  SE backwards (advs_se_mlp_bwd);  d pre_dw = (d sc * s + d pooled / HW) * silu'(pre_dw);
              d m = depthwise'(d pre_dw);  d in = (d m * silu'(pre_m)) W0' (+ d out);
      head conv, average pool and classifier as in the other victims; BatchNorm (eval) is folded into the weights both ways."""

    def __init__(self, model, W, G, batch, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        S, pad = model.image_size, model._pad8
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib, plan = bld.lib, bld.plan
            self.x = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            self.labels = torch.zeros((batch,), dtype=torch.int64, device=dev)
            self.grad = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)

            def silu(pre, add=None):
                y = bld.buf(tuple(pre.shape))
                plan.add(lib.advs_silu, ptr(pre), ptr(add), ptr(y), pre.numel(), dt, keep=(pre, add, y))
                return y

            def silu_bwd(pre, dy):
                dx = bld.buf(tuple(pre.shape))
                plan.add(lib.advs_silu_bwd, ptr(pre), ptr(dy), ptr(dx), pre.numel(), dt, keep=(pre, dy, dx))
                return dx

            def zero_insert(t, hh, ww):
                z = bld.buf((batch, hh, ww, t.shape[3]))
                plan.add(lib.advs_zero_insert2x, ptr(t), ptr(z), batch, t.shape[1], t.shape[2], t.shape[3], hh, ww, dt, keep=(t, z))
                bld.free(t)
                return z

            # ---- forward
            c0 = W["stem.b"].numel()
            side = (S + 2 - 3) // 2 + 1
            pre0 = bld.buf((batch, side, side, c0))
            plan.add(lib.advs_conv_stem, ptr(self.x), ptr(W["stem.w"]), ptr(W["stem.b"]), ptr(pre0), batch, 3, S, S, c0, 3, 2, 1,
                     _lib.ACT["none"], dt, keep=(self.x, pre0))
            h = silu(pre0)
            tape = [("stem", pre0, c0)]
            for item in model.layout[1:]:
                if item[0] == "cna":
                    preh = bld.conv(h, W["head.w"], model.last, bias=W["head.b"], ksize=1, pad=0)
                    bld.free(h)
                    h = silu(preh)
                    tape.append(("head", preh))
                    continue
                kind, p, e, k, st, ci, co = item
                has_res = st == 1 and ci == co
                cop, cep = pad(co), pad(ci * e)
                hin_shape = tuple(h.shape)
                if kind == "fused" and e == 1:
                    pre = bld.conv(h, W[p + ".c0.w"], cop, bias=W[p + ".c0.b"], stride=st)
                    new = silu(pre, h if has_res else None)
                    tape.append(("fused1", p, pre, st, has_res, hin_shape))
                elif kind == "fused":
                    pre = bld.conv(h, W[p + ".c0.w"], cep, bias=W[p + ".c0.b"], stride=st)
                    m = silu(pre)
                    new = bld.conv(m, W[p + ".c1.w"], cop, bias=W[p + ".c1.b"], residual=h if has_res else None, ksize=1, pad=0)
                    bld.free(m)
                    tape.append(("fused", p, pre, st, has_res, hin_shape, cep))
                else:
                    prem = bld.conv(h, W[p + ".c0.w"], cep, bias=W[p + ".c0.b"], ksize=1, pad=0)
                    m = silu(prem)
                    si = m.shape[1]
                    so = (si + 2 - 3) // st + 1
                    pred = bld.buf((batch, so, so, cep))
                    plan.add(lib.advs_dwconv2d, ptr(m), ptr(W[p + ".dw.w"]), ptr(W[p + ".dw.b"]), ptr(pred), batch, si, si, cep, k, st, dt,
                             keep=(m, pred))
                    bld.free(m)
                    d = silu(pred)
                    pooled = bld.buf((batch, cep), torch.float32)
                    plan.add(lib.advs_global_avgpool, ptr(d), ptr(pooled), batch, so * so, cep, dt, keep=(d, pooled))
                    z1 = bld.linear(pooled, W[p + ".fc1.w"], W[p + ".fc1.b"])
                    s2 = bld.linear(z1, W[p + ".fc2.w"], W[p + ".fc2.b"], act_in="silu", act_out="sigmoid")
                    sc = bld.buf((batch, so, so, cep))
                    plan.add(lib.advs_scale_channels, ptr(d), ptr(s2), ptr(sc), batch, so * so, cep, dt, keep=(d, s2, sc))
                    new = bld.conv(sc, W[p + ".c3.w"], cop, bias=W[p + ".c3.b"], residual=h if has_res else None, ksize=1, pad=0)
                    bld.free(sc)
                    tape.append(("mb", p, prem, pred, d, z1, s2, k, st, has_res, hin_shape, cep, si, so))
                bld.free(h)
                h = new
            hw_last = h.shape[1] * h.shape[2]
            pooled = bld.buf((batch, model.last), torch.float32)
            plan.add(lib.advs_global_avgpool, ptr(h), ptr(pooled), batch, hw_last, model.last, dt, keep=(h, pooled))
            last_shape = tuple(h.shape)
            bld.free(h)
            self.logits = bld.linear(pooled, W["cls.w"], W["cls.b"])
            # ---- backward
            K = self.logits.shape[1]
            gl = bld.buf((batch, K), torch.float32)
            plan.add(lib.advs_softmax_ce_grad, ptr(self.logits), ptr(self.labels), ptr(gl), batch, K, 1.0, keep=(self.logits, self.labels, gl))
            gp = bld.linear(gl, G["cls.wT"], None)
            g = bld.buf(last_shape)
            plan.add(lib.advs_avgpool_bwd, ptr(gp), ptr(g), batch, hw_last, model.last, dt, keep=(gp, g))
            for rec in reversed(tape):
                kind = rec[0]
                if kind == "head":
                    dpre = silu_bwd(rec[1], g)
                    bld.free(g); bld.free(rec[1])
                    g = bld.conv(dpre, G["headT"], pad(model.setting[-1][5]), ksize=1, pad=0)
                    bld.free(dpre)
                elif kind in ("fused1", "fused"):
                    if kind == "fused1":
                        _, p, pre, st, has_res, hin_shape = rec
                        dm = g
                    else:
                        _, p, pre, st, has_res, hin_shape, cep = rec
                        dm = bld.conv(g, G[p + ".c1T"], cep, ksize=1, pad=0)
                    dpre = silu_bwd(pre, dm)
                    if dm is not g:
                        bld.free(dm)
                    bld.free(pre)
                    if st == 2:
                        dpre = zero_insert(dpre, hin_shape[1], hin_shape[2])
                    din = bld.conv(dpre, G[p + ".c0T"], hin_shape[3], residual=g if has_res else None, ksize=3, stride=1, pad=1)
                    bld.free(dpre); bld.free(g)
                    g = din
                elif kind == "mb":
                    _, p, prem, pred, d, z1, s2, k, st, has_res, hin_shape, cep, si, so = rec
                    dsc = bld.conv(g, G[p + ".c3T"], cep, ksize=1, pad=0)
                    gs = bld.buf((batch, cep), torch.float32)
                    plan.add(lib.advs_channel_dot, ptr(dsc), ptr(d), ptr(gs), batch, so * so, cep, dt, keep=(dsc, d, gs))
                    dz2 = bld.buf((batch, cep), torch.float32)
                    plan.add(lib.advs_sigmoid_gate_bwd, ptr(gs), ptr(s2), ptr(dz2), batch * cep, keep=(gs, s2, dz2))
                    da1 = bld.linear(dz2, G[p + ".fc2T"], None)                            # [B, sq]
                    dz1 = bld.buf(tuple(da1.shape), torch.float32)
                    plan.add(lib.advs_silu_bwd, ptr(z1), ptr(da1), ptr(dz1), da1.numel(), _lib.F32, keep=(z1, da1, dz1))
                    dpool = bld.linear(dz1, G[p + ".fc1T"], None)                          # [B, cep]
                    dpd = bld.buf(tuple(pred.shape))
                    plan.add(lib.advs_se_scale_bwd, ptr(dsc), ptr(s2), ptr(dpool), ptr(pred), ptr(dpd), batch, so * so, cep, dt,
                             keep=(dsc, s2, dpool, pred, dpd))
                    bld.free(dsc); bld.free(pred); bld.free(d)
                    dm = bld.buf((batch, si, si, cep))
                    plan.add(lib.advs_dwconv2d_bwd_strided, ptr(dpd), ptr(W[p + ".dw.w"]), ptr(dm), batch, si, si, cep, k, st, dt, keep=(dpd, dm))
                    bld.free(dpd)
                    dprem = silu_bwd(prem, dm)
                    bld.free(dm); bld.free(prem)
                    din = bld.conv(dprem, G[p + ".c0T"], hin_shape[3], residual=g if has_res else None, ksize=1, pad=0)
                    bld.free(dprem); bld.free(g)
                    g = din
                else:
                    _, pre0, c0 = rec
                    dpre = silu_bwd(pre0, g)
                    bld.free(g)
                    plan.add(lib.advs_conv_stem_bwd, ptr(dpre), ptr(G["stem.w"]), ptr(self.grad), batch, 3, S, S, c0, 3, 2, 1, dt,
                             keep=(dpre, self.grad))
            self.plan, self.captured = plan, False
            torch.cuda.synchronize(dev)

    def run(self):
        if self.model.use_graph and not self.captured:
            self.plan.run_eager()
            self.stream.synchronize()
            self.plan.capture()
            self.captured = True
        self.plan.run()


class _EffNetEngine:
    def __init__(self, model, W, batch, dt):
        dev = next(model.parameters()).device
        self.model, self.stream = model, torch.cuda.Stream(device=dev)
        S, pad = model.image_size, model._pad8
        with torch.cuda.device(dev):
            bld = Builder(dev, dt, self.stream, batch)
            lib = bld.lib
            self.x = torch.zeros((batch, 3, S, S), dtype=torch.float32, device=dev)
            c0 = W["stem.b"].numel()
            side = (S + 2 - 3) // 2 + 1
            h = bld.buf((batch, side, side, c0))
            bld.plan.add(lib.advs_conv_stem, ptr(self.x), ptr(W["stem.w"]), ptr(W["stem.b"]), ptr(h), batch, 3, S, S, c0, 3, 2, 1,
                         _lib.ACT["silu"], dt, keep=(self.x, h))
            for item in model.layout[1:]:
                if item[0] == "cna":
                    new = bld.conv(h, W["head.w"], model.last, bias=W["head.b"], act="silu", ksize=1, pad=0)
                    bld.free(h)
                    h = new
                    continue
                kind, p, e, k, st, ci, co = item
                res = h if (st == 1 and ci == co) else None
                cop, cep = pad(co), pad(ci * e)
                if kind == "fused" and e == 1:
                    new = bld.conv(h, W[p + ".c0.w"], cop, bias=W[p + ".c0.b"], act="silu", stride=st, residual=res,
                                   residual_after_act=res is not None)
                elif kind == "fused":
                    m = bld.conv(h, W[p + ".c0.w"], cep, bias=W[p + ".c0.b"], act="silu", stride=st)
                    new = bld.conv(m, W[p + ".c1.w"], cop, bias=W[p + ".c1.b"], residual=res, ksize=1, pad=0)
                    bld.free(m)
                else:
                    m = bld.conv(h, W[p + ".c0.w"], cep, bias=W[p + ".c0.b"], act="silu", ksize=1, pad=0)
                    so = (m.shape[1] + 2 - 3) // st + 1
                    d = bld.buf((batch, so, so, cep))
                    bld.plan.add(lib.advs_dwconv2d_act, ptr(m), ptr(W[p + ".dw.w"]), ptr(W[p + ".dw.b"]), ptr(d), batch, m.shape[1],
                                 m.shape[2], cep, k, st, _lib.ACT["silu"], dt, keep=(m, d))
                    bld.free(m)
                    pooled = bld.buf((batch, cep), torch.float32)
                    bld.plan.add(lib.advs_global_avgpool, ptr(d), ptr(pooled), batch, so * so, cep, dt, keep=(d, pooled))
                    s1 = bld.linear(pooled, W[p + ".fc1.w"], W[p + ".fc1.b"], act_out="silu")
                    s2 = bld.linear(s1, W[p + ".fc2.w"], W[p + ".fc2.b"], act_out="sigmoid")
                    sc = bld.buf((batch, so, so, cep))
                    bld.plan.add(lib.advs_scale_channels, ptr(d), ptr(s2), ptr(sc), batch, so * so, cep, dt, keep=(d, s2, sc))
                    bld.free(d)
                    new = bld.conv(sc, W[p + ".c3.w"], cop, bias=W[p + ".c3.b"], residual=res, ksize=1, pad=0)
                    bld.free(sc)
                bld.free(h)
                h = new
            pooled = bld.buf((batch, model.last), torch.float32)
            bld.plan.add(lib.advs_global_avgpool, ptr(h), ptr(pooled), batch, h.shape[1] * h.shape[2], model.last, dt, keep=(h, pooled))
            bld.free(h)
            self.logits = bld.linear(pooled, W["cls.w"], W["cls.b"])
            self.plan, self.captured = bld.plan, False
            torch.cuda.synchronize(dev)

    def run(self):
        if self.model.use_graph and not self.captured:
            self.plan.run_eager()
            self.stream.synchronize()
            self.plan.capture()
            self.captured = True
        self.plan.run()
