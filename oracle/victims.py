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
