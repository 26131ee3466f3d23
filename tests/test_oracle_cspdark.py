"""The CPU oracle vs golden vectors of the imported reference (model/networks/cspdarkunet.py)."""
import numpy as np
import pytest
import torch

from oracle import cspdark as oc
from oracle import lineage_a as oa

CASES = {"silu": (3, "silu"), "lrelu": (4, "lrelu")}


def wrap_diff(a, b):
    d = (a.astype(np.int16) - b.astype(np.int16)) % 256
    return np.minimum(d, 256 - d)


@pytest.mark.parametrize("tag", list(CASES))
def test_seeded_init_matches_reference(golden, tag):
    seed, _ = CASES[tag]
    g = golden(f"cspdark_{tag}.npz")
    sd = oc.init_state_dict(seed, num_classes=37)
    keys = list(g["sd_keys"])
    assert sorted(sd.keys()) == keys
    mine = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])
    assert np.array_equal(mine, g["sd_digest"])


@pytest.mark.parametrize("tag", list(CASES))
def test_forward_matches_reference(golden, tag):
    seed, act = CASES[tag]
    g = golden(f"cspdark_{tag}.npz")
    sd = oc.init_state_dict(seed, num_classes=37)
    x, t, y = torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["y"])
    assert np.abs(oc.cspdarkunet_forward(sd, x, t, y, act=act).numpy() - g["eps_cond"]).max() < 2e-5
    assert np.abs(oc.cspdarkunet_forward(sd, x, t, None, act=act).numpy() - g["eps_uncond"]).max() < 2e-5


def test_ddim_sample_matches_reference(golden):
    g = golden("cspdark_silu.npz")
    sd = oc.init_state_dict(3, num_classes=37)
    fn = lambda x, t, y: oc.cspdarkunet_forward(sd, x, t, y)
    out = oa.ddim_sample(fn, torch.from_numpy(g["sample_xT"]), labels=torch.from_numpy(g["sample_labels"]),
                         cfg_scale=3, sample_steps=10).numpy()
    d = wrap_diff(out, g["sample_cfg3"])
    assert out.dtype == np.uint8 and d.max() <= 1 and (d > 0).mean() < 0.01
