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


@torch.no_grad()
def resnet50_forward(sd, x):
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


@torch.no_grad()
def vgg_forward(sd, x, depth=16):
    """torchvision VGG (features / avgpool / classifier), eval mode.  PARITY UNPINNED (torchvision absent)."""
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
