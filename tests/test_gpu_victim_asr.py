"""Victim forward, ASR evaluation and conv-stem / pooling kernels on the MI355X vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

pytestmark = pytest.mark.gpu

from gpu_helpers import OneOp, bf16_round, dev, nchw, nhwc  # noqa: E402
from advshadow_amd import _lib, asr  # noqa: E402
from advshadow_amd.engine import ptr  # noqa: E402
from advshadow_amd.victims import ResNet50  # noqa: E402
from oracle import victims as ov  # noqa: E402


def make_victim(seed=1, **kw):
    torch.manual_seed(seed)
    net = ResNet50(num_classes=37, **kw)
    sd = ov.randomize_bn({k: v.clone() for k, v in net.state_dict().items()}, seed + 100)
    net.load_state_dict(sd)
    return net.to("cuda").eval(), {k: v.cpu() for k, v in sd.items()}


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_stem_pool_avg_kernels(dt):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 37, 41, generator=g)
    w, b = torch.randn(64, 3, 7, 7, generator=g) * 0.05, torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, b, stride=2, padding=3))
    op = OneOp(dt, 2)
    lib = op.b.lib
    xd, wd, bd = x.to(dev()), w.to(dev()), b.to(dev())
    y = op.b.buf((2, ref.shape[2], ref.shape[3], 64))
    op.b.plan.add(lib.advs_conv_stem, ptr(xd), ptr(wd), ptr(bd), ptr(y), 2, 3, 37, 41, 64, 7, 2, 3, _lib.ACT["relu"], op.b.dt)
    refp = F.max_pool2d(bf16_round(ref) if dt == "bf16" else ref, 3, 2, 1)
    yp = op.b.buf((2, refp.shape[2], refp.shape[3], 64))
    op.b.plan.add(lib.advs_maxpool3x3s2, ptr(y), ptr(yp), 2, ref.shape[2], ref.shape[3], 64, op.b.dt)
    ya = torch.empty(2, 64, device=dev())
    op.b.plan.add(lib.advs_global_avgpool, ptr(yp), ptr(ya), 2, refp.shape[2] * refp.shape[3], 64, op.b.dt)
    op.go()
    tol = 1e-5 if dt == "fp32" else 3e-2
    assert (nchw(y) - ref).abs().max().item() < tol
    assert (nchw(yp) - refp).abs().max().item() < tol
    assert (ya.cpu() - refp.mean((2, 3))).abs().max().item() < tol


def test_resnet50_fp32_matches_oracle_and_top1():
    net, sd = make_victim()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 3, 224, 224, generator=g)
    ref = ov.resnet50_forward(sd, x)
    for _ in range(2):
        got = net(x.cuda()).cpu()
        assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
        assert torch.equal(got.argmax(1), ref.argmax(1))


def test_resnet50_bf16_top1_agreement():
    net, sd = make_victim(compute_dtype="bf16")
    g = torch.Generator().manual_seed(6)
    x = torch.rand(8, 3, 224, 224, generator=g)
    ref = ov.resnet50_forward(sd, x)
    got = net(x.cuda()).cpu()
    rel = (got - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 0.05, rel
    # decisions agree wherever the fp32 margin between the top two classes exceeds the bf16 noise
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.1 * ref.abs().max()
    assert torch.equal(got.argmax(1)[clear], ref.argmax(1)[clear])


def test_compute_asr_on_a_folder(tmp_path):
    net, sd = make_victim()
    rng = np.random.default_rng(0)
    names = ["Abyssinian_1.jpg", "Bengal_7.png", "american_bulldog_22.png", "notes.txt"]
    arrs = {}
    for n in names[:3]:
        a = rng.integers(0, 256, (96, 120, 3), dtype=np.uint8)
        Image.fromarray(a).save(tmp_path / n)
        arrs[n] = np.asarray(Image.open(tmp_path / n).convert("RGB"))
    (tmp_path / names[3]).write_text("x")
    # oracle: PIL resize + CPU forward
    preds = {}
    for n, a in arrs.items():
        pil = Image.fromarray(a).resize((224, 224), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(pil).transpose(2, 0, 1).astype(np.float32) / 255.0)[None]
        preds[n] = int(ov.resnet50_forward(sd, x).argmax(1))
    int_to_label = {i: f"class{i}" for i in range(37)}
    int_to_label[preds["Bengal_7.png"]] = "Bengal"            # make exactly one file a "failed attack"
    expect = sum(int_to_label[preds[n]] != n.rsplit("_", 1)[0] for n in arrs) / 3
    got = asr.compute_asr(str(tmp_path), net, int_to_label)
    assert got == expect
    x = asr.preprocess_image(str(tmp_path / names[0]))
    assert x.shape == (1, 3, 224, 224) and x.is_cuda and float(x.max()) <= 1.0


def test_label_maps_from_reference_style_config(tmp_path):
    import json
    p = tmp_path / "config2.json"
    p.write_text(json.dumps({"id2label": {"0": "Abyssinian", "1": "Bengal"}}))
    l2i, i2l = asr.load_label_maps(str(p))
    assert l2i == {"Abyssinian": 0, "Bengal": 1} and i2l == {0: "Abyssinian", 1: "Bengal"}


def test_vit_victim_matches_hf_transformers():
    """ViT-B/16 (config C4's victim) against the installed transformers implementation itself."""
    from advshadow_amd.victims import ViTVictim
    hf = ov.hf_vit(37, seed=2)
    net = ViTVictim(37)
    net.load_state_dict(hf.state_dict())                  # transformers-5 key names are remapped
    net = net.to("cuda").eval()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(3, 3, 224, 224, generator=g)
    with torch.no_grad():
        ref = hf(pixel_values=x).logits
    for _ in range(2):
        got = net(x.cuda()).logits.cpu()
        assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
        assert torch.equal(got.argmax(1), ref.argmax(1))
    for dt, bound in (("bf16", 0.05), ("fp16", 0.01)):     # fp16: the dtype of BASELINE.json's ViT config
        lp = ViTVictim(37, compute_dtype=dt)
        lp.load_state_dict(hf.state_dict())
        got = lp.to("cuda").eval()(x.cuda()).logits.cpu()
        assert (got - ref).abs().max().item() < bound * max(1.0, ref.abs().max().item()), dt


def test_vit_small_config_and_masking():
    """A small ViT (2 layers, 4 heads, 64 px, 8 px patches -> 65 tokens padded to 128) vs transformers."""
    from advshadow_amd.victims import ViTVictim
    cfg = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256, patch_size=8, image_size=64)
    hf = ov.hf_vit(5, seed=3, **cfg)
    net = ViTVictim(5, **cfg)
    net.load_state_dict(hf.state_dict())
    net = net.to("cuda").eval()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = hf(pixel_values=x).logits
    got = net(x.cuda()).logits.cpu()
    assert (got - ref).abs().max().item() < 2e-5


def test_vgg16_matches_oracle():
    from advshadow_amd.victims import VGG
    torch.manual_seed(6)
    net = VGG(16, 37)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to("cuda").eval()
    x = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(7))
    ref = ov.vgg_forward(sd, x, 16)
    got = net(x.cuda()).cpu()
    assert (got - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())
    assert torch.equal(got.argmax(1), ref.argmax(1))


# ------------------------------------------------------------------------------ ConvNeXt (ASR_fast.py:21-26)
def test_convnext_small_config_matches_hf_transformers():
    """A small ConvNeXt (depths 1-1-2-1, dims 64..512, 64 px) against the installed transformers implementation;
    the HF parameter names are remapped to the timm names the reference's loader produces."""
    from advshadow_amd.victims import ConvNeXtVictim
    cfg = dict(depths=[1, 1, 2, 1], hidden_sizes=[64, 128, 256, 512], image_size=64)
    hf = ov.hf_convnext(7, seed=5, **cfg)
    x = torch.rand(3, 3, 64, 64, generator=torch.Generator().manual_seed(10))
    with torch.no_grad():
        ref = hf(pixel_values=x).logits
    scale = max(1.0, ref.abs().max().item())
    for dt, bound in (("fp32", 3e-4), ("fp16", 0.02), ("bf16", 0.08)):
        net = ConvNeXtVictim(7, depths=cfg["depths"], dims=cfg["hidden_sizes"], image_size=64, head_norm_eps=1e-12, compute_dtype=dt)
        net.load_state_dict(hf.state_dict())
        net = net.to("cuda").eval()
        for _ in range(2):
            got = net(x.cuda()).cpu()
            assert (got - ref).abs().max().item() < bound * scale, (dt, (got - ref).abs().max().item())
        if dt == "fp32":
            assert torch.equal(got.argmax(1), ref.argmax(1))
    keys = set(net.state_dict().keys())
    assert {"stem.0.weight", "stem.1.bias", "stages.1.downsample.1.weight", "stages.2.blocks.1.conv_dw.weight",
            "stages.2.blocks.1.mlp.fc2.bias", "stages.0.blocks.0.gamma", "head.norm.weight", "head.fc.bias"} <= keys


def test_convnext_base_top1_vs_hf():
    """Full convnext_base geometry (depths 3-3-27-3, dims 128..1024, 224 px, 37 classes)."""
    from advshadow_amd.victims import ConvNeXtVictim
    hf = ov.hf_convnext(37, seed=6, depths=[3, 3, 27, 3], hidden_sizes=[128, 256, 512, 1024])
    x = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        ref = hf(pixel_values=x).logits
    net = ConvNeXtVictim(37, head_norm_eps=1e-12)
    net.load_state_dict(hf.state_dict())
    got = net.to("cuda").eval()(x.cuda()).cpu()
    assert (got - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())
    assert torch.equal(got.argmax(1), ref.argmax(1))
    pred = asr.evaluate_batch((x * 255).to(torch.uint8).cuda(), net)           # plugs into the ASR path like any victim
    assert pred.shape == (2,)


# ------------------------------------------------------------------------------ Swin (ASR_fast.py:27-32)
def test_swin_small_config_matches_hf_transformers():
    """A small Swin (embed 32, depths 2-2, heads 2-4, window 4, 64 px: shifted windows with masks in stage 1,
    whole-map windows in stage 2, one patch merging) against the installed transformers implementation."""
    from advshadow_amd.victims import SwinVictim
    cfg = dict(image_size=64, patch_size=4, embed_dim=32, depths=[2, 2], num_heads=[2, 4], window_size=4)
    hf = ov.hf_swin(7, seed=7, **cfg)
    x = torch.rand(3, 3, 64, 64, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        ref = hf(pixel_values=x).logits
    scale = max(1.0, ref.abs().max().item())
    for dt, bound in (("fp32", 3e-4), ("fp16", 0.02), ("bf16", 0.08)):
        net = SwinVictim(7, embed_dim=32, depths=[2, 2], num_heads=[2, 4], window_size=4, image_size=64, compute_dtype=dt)
        net.load_state_dict(hf.state_dict())
        net = net.to("cuda").eval()
        for _ in range(2):
            got = net(x.cuda()).cpu()
            assert (got - ref).abs().max().item() < bound * scale, (dt, (got - ref).abs().max().item())
        if dt == "fp32":
            assert torch.equal(got.argmax(1), ref.argmax(1))


def test_swin_base_top1_vs_hf():
    """swin_base_patch4_window7_224 geometry: embed 128, depths 2-2-18-2, heads 4-8-16-32, window 7, 49-token windows."""
    from advshadow_amd.victims import SwinVictim
    hf = ov.hf_swin(37, seed=8, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7)
    x = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(13))
    with torch.no_grad():
        ref = hf(pixel_values=x).logits
    net = SwinVictim(37)
    net.load_state_dict(hf.state_dict())
    got = net.to("cuda").eval()(x.cuda()).cpu()
    assert (got - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())
    assert torch.equal(got.argmax(1), ref.argmax(1))


def test_window_shift_roundtrip_and_bias_attention():
    """advs_window_shift forward == roll(-s) + window_partition; inverse restores the image (+ residual);
    advs_attention_bias == softmax(qk^T/sqrt(d) + bias) v with per-window bias blocks."""
    import math
    B, H, W, C, win, sh = 2, 8, 12, 32, 4, 2
    g = torch.Generator().manual_seed(14)
    x = torch.randn(B, H, W, C, generator=g)
    op = OneOp("fp32", B)
    wins = op.b.window_shift(x.to(dev()), win, sh)
    back = op.b.window_shift(wins, win, sh, inverse=True, residual=x.to(dev()), image_hw=(H, W))
    op.go()
    rolled = torch.roll(x, shifts=(-sh, -sh), dims=(1, 2))
    ref = rolled.view(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, win * win, C)
    assert torch.equal(wins.cpu().view(-1, win * win, C), ref)
    assert torch.equal(back.cpu(), 2 * x)
    nW, heads, d, N = 6, 2, 16, 16
    qkv = torch.randn(B * nW, N, 3 * heads * d, generator=g)
    bias = torch.randn(nW, heads, N, N, generator=g)
    t = qkv.view(B * nW, N, 3, heads, d)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    sc = q @ k.transpose(-1, -2) / math.sqrt(d) + bias.repeat(B, 1, 1, 1)
    refa = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * nW, N, heads * d)
    op = OneOp("fp32", B * nW)
    y = op.b.attention_bias(qkv.view(B * nW, 1, N, -1).to(dev()), heads, d, 0, heads * d, 2 * heads * d, d,
                            (bias * 1.4426950408889634).contiguous().to(dev()), nW)
    op.go()
    assert (y.cpu().view(B * nW, N, heads * d) - refa).abs().max().item() < 2e-5


# ------------------------------------------------------------------------------ DINOv2 (ASR_fast.py:47-58)
def test_dinov2_matches_hf_transformers():
    """Small DINOv2 (2 layers, 4 heads, patch 14) vs transformers: same-grid position embeddings at 56 px, and a
    checkpoint grid (4x4) interpolated bicubically to the 8x8 grid of a 112 px input."""
    from advshadow_amd.victims import Dinov2Victim
    cfg = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, patch_size=14, mlp_ratio=4)
    hf = ov.hf_dinov2(5, seed=9, image_size=56, **cfg)
    for size, dts in ((56, (("fp32", 3e-4), ("fp16", 0.02), ("bf16", 0.08))), (112, (("fp32", 3e-4),))):
        x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(15))
        with torch.no_grad():
            ref = hf(pixel_values=x).logits
        scale = max(1.0, ref.abs().max().item())
        for dt, bound in dts:
            net = Dinov2Victim(5, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, image_size=size, pos_grid=4,
                               compute_dtype=dt)
            net.load_state_dict(hf.state_dict())
            got = net.to("cuda").eval()(x.cuda()).logits.cpu()
            assert (got - ref).abs().max().item() < bound * scale, (size, dt, (got - ref).abs().max().item())
            if dt == "fp32":
                assert torch.equal(got.argmax(1), ref.argmax(1))


# ------------------------------------------------------------------------------ EfficientNetV2-S (ASR_fast.py:59-65)
def test_efficientnet_v2_s_vs_restatement():
    """PARITY UNPINNED (torchvision absent): the product against oracle/victims.py's restatement of the published
    architecture, whose parameter count is checked against torchvision's published 21,458,488 (1000 classes)."""
    from advshadow_amd.victims import EfficientNetV2S
    sd1000 = ov.effnetv2_init(0, 1000)
    assert sum(v.numel() for k, v in sd1000.items() if "running" not in k and "num_batches" not in k) == 21458488
    sd = ov.effnetv2_init(3, 37)
    net = EfficientNetV2S(37, image_size=64)
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())
    net.load_state_dict(sd)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(16))
    ref = ov.effnetv2_forward(sd, x)
    scale = max(1.0, ref.abs().max().item())
    got = net.to("cuda").eval()(x.cuda()).cpu()
    assert (got - ref).abs().max().item() < 5e-4 * scale, (got - ref).abs().max().item()
    assert torch.equal(got.argmax(1), ref.argmax(1))
    for dt, bound in (("fp16", 0.03), ("bf16", 0.15)):
        lp = EfficientNetV2S(37, image_size=64, compute_dtype=dt)
        lp.load_state_dict(sd)
        got = lp.to("cuda").eval()(x.cuda()).cpu()
        assert (got - ref).abs().max().item() < bound * scale, (dt, (got - ref).abs().max().item())
