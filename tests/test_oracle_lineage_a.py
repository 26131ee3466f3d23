"""The CPU oracle vs golden vectors of the imported reference (model/networks/unet.py, model/samples/ddim.py)."""
import numpy as np
import pytest
import torch

from oracle import lineage_a as oa

CASES = {"silu": (1, "silu"), "gelu": (2, "gelu")}


def wrap_diff(a, b):
    """uint8 distance modulo 256 (the reference's cast wraps, ddim.py:97-99)."""
    d = (a.astype(np.int16) - b.astype(np.int16)) % 256
    return np.minimum(d, 256 - d)


def test_schedule_and_pairs(golden):
    g = golden("lineage_a_schedule.npz")
    assert np.array_equal(oa.alpha_hat().numpy(), g["alpha_hat"])            # float32, bit-exact
    assert [list(p) for p in oa.time_pairs(1000, 500)] == g["time_pairs_500"].tolist()
    s = golden("lineage_a_silu.npz")
    assert [list(p) for p in oa.time_pairs(1000, 10)] == s["time_pairs"].tolist()
    assert oa.time_pairs(1000, 10)[0] == (901, 801) and oa.time_pairs(1000, 10)[-1] == (1, 0)


@pytest.mark.parametrize("tag", list(CASES))
def test_seeded_init_matches_reference(golden, tag):
    seed, act = CASES[tag]
    g = golden(f"lineage_a_{tag}.npz")
    sd = oa.init_state_dict(seed, num_classes=37, act=act)
    keys = list(g["sd_keys"])
    assert sorted(sd.keys()) == keys
    mine = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])
    assert np.array_equal(mine, g["sd_digest"])


@pytest.mark.parametrize("tag", list(CASES))
def test_forward_matches_reference(golden, tag):
    seed, act = CASES[tag]
    g = golden(f"lineage_a_{tag}.npz")
    sd = oa.init_state_dict(seed, num_classes=37, act=act)
    x, t, y = torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["y"])
    assert np.abs(oa.unet_forward(sd, x, t, y, act=act).numpy() - g["eps_cond"]).max() < 2e-5
    assert np.abs(oa.unet_forward(sd, x, t, None, act=act).numpy() - g["eps_uncond"]).max() < 2e-5


def test_ddim_sample_matches_reference(golden):
    g = golden("lineage_a_silu.npz")
    sd = oa.init_state_dict(1, num_classes=37, act="silu")
    fn = lambda x, t, y: oa.unet_forward(sd, x, t, y)
    xT = torch.from_numpy(g["sample_xT"])
    out = oa.ddim_sample(fn, xT, labels=torch.from_numpy(g["sample_labels"]), cfg_scale=3, sample_steps=10).numpy()
    d = wrap_diff(out, g["sample_cfg3"])
    assert out.dtype == np.uint8 and d.max() <= 1 and (d > 0).mean() < 0.01
    out = oa.ddim_sample(fn, xT, sample_steps=10).numpy()
    d = wrap_diff(out, g["sample_uncond"])
    assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_ddpm_and_plms_match_reference(golden):
    g = golden("lineage_a_samplers.npz")
    sd = oa.init_state_dict(1, num_classes=37, act="silu")
    fn = lambda x, t, y: oa.unet_forward(sd, x, t, y)
    labels = torch.from_numpy(g["labels"])
    noises = {i: torch.from_numpy(g["ddpm_noise"][11 - i]) for i in range(2, 12)}
    out = oa.ddpm_sample(fn, torch.from_numpy(g["ddpm_xT"]), noises, labels=labels, cfg_scale=3, noise_steps=12).numpy()
    d = wrap_diff(out, g["ddpm_cfg3"])
    assert d.max() <= 1 and (d > 0).mean() < 0.01
    out = oa.plms_sample(fn, torch.from_numpy(g["plms_xT"]), labels=labels, cfg_scale=3, sample_steps=8).numpy()
    d = wrap_diff(out, g["plms_cfg3"])
    assert d.max() <= 1 and (d > 0).mean() < 0.01
    out = oa.plms_sample(fn, torch.from_numpy(g["plms_xT"]), sample_steps=8).numpy()
    d = wrap_diff(out, g["plms_uncond"])
    assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_oracle_fixture_256_loop_is_this_oracle(golden):
    """tests/golden/oracle_lineage_a_256_loop.npz (made by make_oracle_fixtures.py, 3.5 min of this oracle) belongs to the weights
    and step pairs this oracle builds today; the loop itself is re-run only by that script."""
    g = golden("oracle_lineage_a_256_loop.npz")
    sd = oa.init_state_dict(1, num_classes=37, act="silu")
    keys = list(g["sd_keys"])
    assert sorted(sd.keys()) == keys
    assert np.array_equal(np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys]), g["sd_digest"])
    assert [tuple(p) for p in g["pairs"]] == [tuple(p) for p in oa.time_pairs(1000, 4)]
    assert g["out"].shape == (1, 3, 256, 256) and np.isfinite(g["out"]).all() and np.abs(g["out"]).max() < 1.5
