"""Oracle: ResNet-50 victim forward on CPU (torchvision architecture, eval-mode BatchNorm).

Test infrastructure only.  timm and torchvision are absent from the image and the reference ships
no victim weights or outputs, so this restatement of the public ResNet-50 v1.5 layer specification
(stride on the 3x3 conv; state_dict names shared by torchvision and timm) is PARITY UNPINNED: it is
the CPU truth the HIP path is compared with, not a checked copy of timm's numerics.
"""
import torch
import torch.nn.functional as F

LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=1e-5)


def resnet50_logits(sd, x):
    """The forward with autograd left on (oracle/adversarial.py differentiates it with respect to x)."""
    h = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)))
    h = F.max_pool2d(h, 3, 2, 1)
    for li, (width, n, stride) in enumerate(LAYERS, start=1):
        for bi in range(n):
            p = f"layer{li}.{bi}"
            s = stride if bi == 0 else 1
            o = F.relu(_bn(sd, p + ".bn1", F.conv2d(h, sd[p + ".conv1.weight"])))
            o = F.relu(_bn(sd, p + ".bn2", F.conv2d(o, sd[p + ".conv2.weight"], stride=s, padding=1)))
            o = _bn(sd, p + ".bn3", F.conv2d(o, sd[p + ".conv3.weight"]))
            if p + ".downsample.0.weight" in sd:
                h = _bn(sd, p + ".downsample.1", F.conv2d(h, sd[p + ".downsample.0.weight"], stride=s))
            h = F.relu(o + h)
    h = F.adaptive_avg_pool2d(h, 1).flatten(1)
    return F.linear(h, sd["fc.weight"], sd["fc.bias"])


@torch.no_grad()
def resnet50_forward(sd, x):
    return resnet50_logits(sd, x)


def randomize_bn(sd, seed):
    """Give every BatchNorm non-trivial affine parameters and running statistics (in place)."""
    g = torch.Generator().manual_seed(seed)
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.5 + 0.75
        elif ".bn" in k or k.startswith("bn") or "downsample.1" in k:
            if k.endswith("weight"):
                sd[k] = torch.rand(sd[k].shape, generator=g) * 0.5 + 0.75
            elif k.endswith("bias"):
                sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    return sd


# ---------------------------------------------------------------------------- VGG / ViT
VGG_CFG = {16: [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
           19: [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]}


def vgg_logits(sd, x, depth=16):
    """torchvision VGG (features / avgpool / classifier), eval mode, autograd left on.  PARITY UNPINNED (torchvision absent)."""
    idx = 0
    for v in VGG_CFG[depth]:
        if v == "M":
            x = F.max_pool2d(x, 2)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"], padding=1))
            idx += 2
    x = F.adaptive_avg_pool2d(x, 7).flatten(1)
    x = F.relu(F.linear(x, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    x = F.relu(F.linear(x, sd["classifier.3.weight"], sd["classifier.3.bias"]))
    return F.linear(x, sd["classifier.6.weight"], sd["classifier.6.bias"])


@torch.no_grad()
def vgg_forward(sd, x, depth=16):
    return vgg_logits(sd, x, depth)


def hf_vit(num_labels=37, seed=2, **cfg):
    """The installed transformers ViTForImageClassification, random init: the one victim whose REAL
    implementation is importable here, used as its own oracle."""
    from transformers import ViTConfig, ViTForImageClassification
    torch.manual_seed(seed)
    return ViTForImageClassification(ViTConfig(num_labels=num_labels, **cfg)).eval()


def hf_convnext(num_labels=37, seed=5, **cfg):
    """The installed transformers ConvNextForImageClassification, random init (layer scale re-drawn so that it is
    not the near-zero 1e-6 initial value, which would hide errors in the block bodies): the ConvNeXt victim's oracle.
    The reference loads ``timm.create_model('convnext_base.fb_in1k')`` (ASR_fast.py:21-26); timm is absent, the
    architecture is the same, the parameter names differ (the product maps them)."""
    from transformers import ConvNextConfig, ConvNextForImageClassification
    torch.manual_seed(seed)
    m = ConvNextForImageClassification(ConvNextConfig(num_labels=num_labels, **cfg)).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("layer_scale_parameter"):
                p.copy_(torch.rand_like(p) * 0.5 + 0.25)
            elif "layernorm" in n or "downsampling_layer.0" in n:
                p.copy_(torch.randn_like(p) * 0.1 + (1.0 if n.endswith("weight") else 0.0))
            elif n.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.05)
    return m


def hf_swin(num_labels=37, seed=7, **cfg):
    """The installed transformers SwinForImageClassification, random init with non-trivial relative position bias
    tables: the Swin victim's oracle.  The reference loads ``timm.create_model('swin_base_patch4_window7_224')``
    (ASR_fast.py:27-32); timm is absent, the architecture is the same, parameter names differ (the product maps)."""
    from transformers import SwinConfig, SwinForImageClassification
    torch.manual_seed(seed)
    m = SwinForImageClassification(SwinConfig(num_labels=num_labels, **cfg)).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("relative_position_bias_table"):
                p.copy_(torch.randn_like(p) * 0.5)
            elif "layernorm" in n or n.endswith("norm.weight") or n.endswith("norm.bias"):
                p.copy_(torch.randn_like(p) * 0.1 + (1.0 if n.endswith("weight") else 0.0))
            elif n.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.05)
    return m


def hf_dinov2(num_labels=37, seed=9, **cfg):
    """The installed transformers Dinov2ForImageClassification, random init with non-trivial LayerScale: the DINOv2
    victim's oracle (the reference loads such checkpoints with AutoModelForImageClassification, ASR_fast.py:47-58)."""
    from transformers import Dinov2Config, Dinov2ForImageClassification
    torch.manual_seed(seed)
    m = Dinov2ForImageClassification(Dinov2Config(num_labels=num_labels, **cfg)).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("lambda1"):
                p.copy_(torch.rand_like(p) * 0.8 + 0.2)
            elif n.endswith("position_embeddings") or n.endswith("cls_token"):
                p.copy_(torch.randn_like(p) * 0.3)
            elif n.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.05)
    return m


# --------------------------------------------------------------------------- EfficientNetV2-S (torchvision)
# torchvision.models.efficientnet_v2_s as published (torchvision/models/efficientnet.py): (block, expand, kernel,
# stride, in, out, layers); BatchNorm eps 1e-3; SE squeeze = max(1, in // 4) with SiLU / Sigmoid; stem 3->24 s2;
# head 256->1280; classifier[1] = Linear(1280, classes) (ASR_fast.py:59-65).  torchvision is absent from this image:
# PARITY UNPINNED (architecture and parameter names restated from the published source, not checked against it).
EFFNETV2_S = [("fused", 1, 3, 1, 24, 24, 2), ("fused", 4, 3, 2, 24, 48, 4), ("fused", 4, 3, 2, 48, 64, 4),
              ("mb", 4, 3, 2, 64, 128, 6), ("mb", 6, 3, 1, 128, 160, 9), ("mb", 6, 3, 2, 160, 256, 15)]


def effnetv2_layout(setting=EFFNETV2_S, last=1280):
    """[(kind, prefix, args...)] in torchvision's module order with its state_dict prefixes."""
    out = [("cna", "features.0", 3, setting[0][4], 3, 2, True)]
    for si, (kind, e, k, s, cin, cout, n) in enumerate(setting, start=1):
        for b in range(n):
            ci, st = (cin, s) if b == 0 else (cout, 1)
            out.append((kind, f"features.{si}.{b}", e, k, st, ci, cout))
    out.append(("cna", f"features.{len(setting) + 1}", setting[-1][5], last, 1, 1, True))
    return out


def effnetv2_init(seed, num_classes=37, setting=EFFNETV2_S, last=1280):
    """Random parameters with torchvision's names (BatchNorm statistics randomised so that folding is exercised)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def cna(p, cin, cout, k, groups=1):
        fan = cin // groups * k * k
        sd[p + ".0.weight"] = torch.randn(cout, cin // groups, k, k, generator=g) * (1.6 / fan) ** 0.5
        sd[p + ".1.weight"] = torch.rand(cout, generator=g) * 0.5 + 0.75
        sd[p + ".1.bias"] = torch.randn(cout, generator=g) * 0.1
        sd[p + ".1.running_mean"] = torch.randn(cout, generator=g) * 0.1
        sd[p + ".1.running_var"] = torch.rand(cout, generator=g) * 0.5 + 0.75
        sd[p + ".1.num_batches_tracked"] = torch.tensor(0)

    for item in effnetv2_layout(setting, last):
        if item[0] == "cna":
            cna(item[1], item[2], item[3], item[4])
            continue
        kind, p, e, k, st, ci, co = item
        ce = ci * e
        if kind == "fused":
            if e != 1:
                cna(p + ".block.0", ci, ce, k); cna(p + ".block.1", ce, co, 1)
            else:
                cna(p + ".block.0", ci, co, k)
        else:
            cna(p + ".block.0", ci, ce, 1); cna(p + ".block.1", ce, ce, k, groups=ce)
            sq = max(1, ci // 4)
            sd[p + ".block.2.fc1.weight"] = torch.randn(sq, ce, 1, 1, generator=g) * (1.0 / ce) ** 0.5
            sd[p + ".block.2.fc1.bias"] = torch.randn(sq, generator=g) * 0.1
            sd[p + ".block.2.fc2.weight"] = torch.randn(ce, sq, 1, 1, generator=g) * (1.0 / sq) ** 0.5
            sd[p + ".block.2.fc2.bias"] = torch.randn(ce, generator=g) * 0.1 + 1.0
            cna(p + ".block.3", ce, co, 1)
    sd["classifier.1.weight"] = torch.randn(num_classes, last, generator=g) * (1.0 / last) ** 0.5
    sd["classifier.1.bias"] = torch.randn(num_classes, generator=g) * 0.1
    return sd


@torch.no_grad()
def effnetv2_forward(sd, x, setting=EFFNETV2_S, last=1280):
    """EfficientNet.forward (eval mode: BatchNorm running statistics, StochasticDepth = identity)."""
    def cna(p, h, stride=1, groups=1, act=True):
        w = sd[p + ".0.weight"]
        h = F.conv2d(h, w, stride=stride, padding=(w.shape[-1] - 1) // 2, groups=groups)
        h = F.batch_norm(h, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"], False, 0.0, 1e-3)
        return F.silu(h) if act else h

    h = x
    for item in effnetv2_layout(setting, last):
        if item[0] == "cna":
            h = cna(item[1], h, stride=item[5])
            continue
        kind, p, e, k, st, ci, co = item
        inp = h
        if kind == "fused":
            if e != 1:
                h = cna(p + ".block.1", cna(p + ".block.0", h, stride=st), act=False)
            else:
                h = cna(p + ".block.0", h, stride=st)
        else:
            h = cna(p + ".block.0", h)
            h = cna(p + ".block.1", h, stride=st, groups=h.shape[1])
            s = F.adaptive_avg_pool2d(h, 1)
            s = F.silu(F.conv2d(s, sd[p + ".block.2.fc1.weight"], sd[p + ".block.2.fc1.bias"]))
            s = torch.sigmoid(F.conv2d(s, sd[p + ".block.2.fc2.weight"], sd[p + ".block.2.fc2.bias"]))
            h = cna(p + ".block.3", h * s, act=False)
        if st == 1 and ci == co:
            h = h + inp
    h = F.adaptive_avg_pool2d(h, 1).flatten(1)
    return F.linear(h, sd["classifier.1.weight"], sd["classifier.1.bias"])
