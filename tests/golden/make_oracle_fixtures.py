#!/usr/bin/env python
"""Fixtures computed with the CPU ORACLE (not the reference) where evaluating it inside the GPU suite would take minutes.

    python tests/golden/make_oracle_fixtures.py

``oracle_lineage_a_256_loop.npz``: DDIMDiffusion.sample (model/samples/ddim.py:48-100) with classifier-free guidance on
UNet(num_classes=37, image_size=256) -- B = 1, sample_steps = 4, cfg_scale = 3, label 23, x_T from Generator(2560), weights from
torch.manual_seed(1) -- evaluated by oracle/lineage_a.py (itself pinned at 64 x 64 by the reference's goldens lineage_a_*.npz):
eight forwards with 65 536-token attention, ~3.5 minutes on 16 cores.  Stored: the float end point and a digest of the state_dict
so the test can prove it rebuilt the same weights.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import lineage_a as oa  # noqa: E402


def main():
    torch.set_num_threads(min(16, torch.get_num_threads()))
    sd = oa.init_state_dict(1, num_classes=37, act="silu")
    keys = sorted(sd.keys())
    digest = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])
    g = torch.Generator().manual_seed(2560)
    xT = torch.randn(1, 3, 256, 256, generator=g)
    labels = torch.tensor([23])
    ref = oa.ddim_sample(lambda x, t, y: oa.unet_forward(sd, x, t, y), xT, labels=labels, cfg_scale=3, sample_steps=4, to_uint8=False)
    path = os.path.join(HERE, "oracle_lineage_a_256_loop.npz")
    np.savez_compressed(path, sd_keys=np.array(keys), sd_digest=digest, out=ref.numpy(),
                        pairs=np.array(oa.time_pairs(1000, 4)), label=labels.numpy(), xT_seed=np.array(2560))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
