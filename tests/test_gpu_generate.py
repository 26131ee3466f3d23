"""generate(args) end to end on the MI355X: checkpoint dict -> UNet -> DDIM+CFG -> image files."""
import argparse
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from advshadow_amd.model.networks.unet import UNet  # noqa: E402
from advshadow_amd.tools.generate import generate  # noqa: E402
from advshadow_amd.utils.utils import make_grid  # noqa: E402
from oracle import lineage_a as oa  # noqa: E402


def test_generate_writes_reference_file_set(tmp_path):
    sd = oa.init_state_dict(1, num_classes=37)
    ckpt = {"start_epoch": 3, "model": {"module." + k: v for k, v in sd.items()}, "ema_model": None, "optimizer": None,
            "num_classes": 37, "classes_name": None, "conditional": True, "image_size": 64, "sample": "ddim",
            "network": "unet", "act": "silu"}
    wpath = str(tmp_path / "ckpt_last.pt")
    torch.save(ckpt, wpath)
    g = torch.Generator().manual_seed(8)
    xT = torch.randn(2, 3, 64, 64, generator=g)
    args = argparse.Namespace(weight_path=wpath, conditional=True, network="unet", image_size=128, num_classes=10,
                              act="gelu", generate_name="df", sample="ddim", num_images=2, use_ema=True,
                              image_format="png", result_path=str(tmp_path / "vis"), class_name=5, cfg_scale=3,
                              sample_steps=4, x_T=xT)
    out_dir = generate(args)
    files = sorted(os.listdir(out_dir))
    # grid + per-image + resized copies (args.image_size 128 != checkpoint image_size 64)
    assert files == ["df.png", "df_0.png", "df_1.png", "df_128_0.png", "df_128_1.png"]
    ref = oa.ddim_sample(lambda x, t, y: oa.unet_forward(sd, x, t, y), xT, labels=torch.tensor([5, 5]), cfg_scale=3,
                         sample_steps=4).numpy()
    for k in range(2):
        got = np.asarray(Image.open(os.path.join(out_dir, f"df_{k}.png"))).transpose(2, 0, 1)
        d = (got.astype(np.int16) - ref[k].astype(np.int16)) % 256
        d = np.minimum(d, 256 - d)
        assert d.max() <= 1 and (d > 0).mean() < 0.01
    assert Image.open(os.path.join(out_dir, "df_128_0.png")).size == (128, 128)
    assert Image.open(os.path.join(out_dir, "df.png")).size == (2 * 66 + 2, 66 + 2)


def test_generate_with_cspdarkunet_and_plms(tmp_path):
    """--network cspdarkunet (utils/initializer.py:90-91) with --sample plms, unconditional, jpg output."""
    from oracle import cspdark as oc
    sd = oc.init_state_dict(3, num_classes=None)
    wpath = str(tmp_path / "csp.pt")
    torch.save({"model": sd, "network": "cspdarkunet", "image_size": 64, "conditional": False, "act": "silu",
                "num_classes": None, "sample": "plms"}, wpath)
    g = torch.Generator().manual_seed(9)
    xT = torch.randn(1, 3, 64, 64, generator=g)
    args = argparse.Namespace(weight_path=wpath, conditional=False, network="unet", image_size=64, num_classes=10,
                              act="silu", generate_name="csp", sample="plms", num_images=1, use_ema=False,
                              image_format="png", result_path=str(tmp_path / "vis"), class_name=-1, cfg_scale=3,
                              sample_steps=4, x_T=xT)
    out_dir = generate(args)
    assert sorted(os.listdir(out_dir)) == ["csp.png", "csp_0.png"]
    ref = oa.plms_sample(lambda x, t, y: oc.cspdarkunet_forward(sd, x, t, y), xT, sample_steps=4).numpy()
    got = np.asarray(Image.open(os.path.join(out_dir, "csp_0.png"))).transpose(2, 0, 1)
    d = (got.astype(np.int16) - ref[0].astype(np.int16)) % 256
    d = np.minimum(d, 256 - d)
    assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_bare_state_dict_and_shape_filter(tmp_path):
    from advshadow_amd.utils.checkpoint import load_ckpt
    sd = oa.init_state_dict(2, num_classes=37)
    sd["label_emb.weight"] = torch.zeros(5, 256)             # wrong shape: must be dropped, not fatal
    wpath = str(tmp_path / "w.pt")
    torch.save(sd, wpath)
    torch.manual_seed(0)
    net = UNet(num_classes=37, image_size=64)
    before = net.state_dict()["label_emb.weight"].clone()
    load_ckpt(wpath, net, "cpu")
    after = net.state_dict()
    assert torch.equal(after["label_emb.weight"], before)
    assert torch.equal(after["inc.double_conv.0.weight"], sd["inc.double_conv.0.weight"])


def test_make_grid_layout():
    imgs = torch.arange(3 * 3 * 4 * 4, dtype=torch.uint8).reshape(3, 3, 4, 4)
    g = make_grid(imgs, nrow=2)
    assert g.shape == (3, 2 * 6 + 2, 2 * 6 + 2)
    assert torch.equal(g[:, 2:6, 2:6], imgs[0]) and torch.equal(g[:, 2:6, 8:12], imgs[1]) and torch.equal(g[:, 8:12, 2:6], imgs[2])
    assert int(g[:, 8:12, 8:12].sum()) == 0
