"""The CPU oracle vs golden vectors produced by the imported reference (diff_model.py)."""
import numpy as np
import pytest
import torch

from oracle import lineage_b as ob

CASES = {
    "small": (3, dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4)),
    "mid": (5, dict(model_channels=64, channel_mult=(1, 2, 3), num_res_blocks=2, attention_resolutions=(1, 4), num_heads=2)),
    "default": (0, {}),
}
TOL = 2e-5   # fp32 run-to-run noise of the reference itself is 7.6e-6


def test_schedules(golden):
    g = golden("lineage_b_schedules.npz")
    for sched in ("cosine", "linear"):
        ac = ob.alphas_cumprod(1000, sched).numpy()
        assert np.array_equal(ac, g[f"ac_{sched}"])        # float64, bit-exact


def test_step_sequences():
    # SURVEY §8 a10 [probed]: 10 -> [1,101,..,901]; 50 -> [1,21,..,981]; 100 -> [1,11,..,991]
    for steps, c in ((10, 100), (50, 20), (100, 10)):
        seq, prev = ob.ddim_sequences(1000, steps)
        assert list(seq) == [1 + c * i for i in range(steps)]
        assert list(prev) == [0] + list(seq[:-1])


@pytest.mark.parametrize("tag", list(CASES))
def test_seeded_init_matches_reference(golden, tag):
    seed, over = CASES[tag]
    g = golden(f"lineage_b_{tag}.npz")
    sd = ob.init_state_dict(seed, ob.hparams(**over))
    keys = list(g["sd_keys"])
    assert sorted(sd.keys()) == keys
    dg = ob.state_dict_digest(sd)
    mine = np.array([dg[k] for k in keys])
    assert np.array_equal(mine, g["sd_digest"])


@pytest.mark.parametrize("tag", list(CASES))
def test_forward_matches_reference(golden, tag):
    seed, over = CASES[tag]
    g = golden(f"lineage_b_{tag}.npz")
    hp = ob.hparams(**over)
    sd = ob.init_state_dict(seed, hp)
    x = torch.from_numpy(g["x"])
    for t in g["ts"]:
        eps = ob.unet_forward(sd, hp, x, torch.full((x.shape[0],), int(t), dtype=torch.long)).numpy()
        assert np.abs(eps - g[f"eps_t{t}"]).max() < TOL


@pytest.mark.parametrize("tag", list(CASES))
@pytest.mark.parametrize("sched", ["cosine", "linear"])
def test_ddim_loop_matches_reference(golden, tag, sched):
    seed, over = CASES[tag]
    g = golden(f"lineage_b_{tag}.npz")
    hp = ob.hparams(**over)
    sd = ob.init_state_dict(seed, hp)
    xT = torch.from_numpy(g[f"ddim_xT_{sched}"])
    out = ob.ddim_sample(lambda x, t: ob.unet_forward(sd, hp, x, t), xT, schedule=sched, steps=int(g["ddim_steps"]))
    ref = g[f"ddim_out_{sched}"]
    assert out.dtype == np.float32 and out.shape == ref.shape
    assert np.abs(out - ref).max() < 1e-4


@pytest.mark.parametrize("sched", ["cosine", "linear"])
def test_ancestral_loop_matches_reference(golden, sched):
    """GaussianDiffusion.sample (p_sample_loop, diff_model.py:398-413) with timesteps=24: every recorded step."""
    g = golden("lineage_b_ancestral.npz")
    hp = ob.hparams(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4)
    sd = ob.init_state_dict(3, hp)
    noises = {23 - k: torch.from_numpy(g["noise"][k]) for k in range(24)}
    imgs = ob.p_sample_loop(lambda x, t: ob.unet_forward(sd, hp, x, t), torch.from_numpy(g["xT"]), noises, T=24, schedule=sched)
    assert len(imgs) == int(g[f"len_{sched}"]) == 24
    for j, k in enumerate((0, 11, 22, 23)):
        assert np.abs(imgs[k].numpy() - g[f"imgs_{sched}"][j]).max() < 2e-5, k


WIDE = {   # the widths the reference's drivers instantiate: ddim2/main2.py:118-127 and gen.py:522-528 ("cs2")
    "ddim2": (11, dict(num_res_blocks=2, attention_resolutions=(4, 8, 16, 32), channel_mult=(1, 1, 2, 2, 4, 4))),
    "cs2": (12, dict(num_res_blocks=2, channel_mult=(1, 2, 3, 4), attention_resolutions=(2,))),
}


@pytest.mark.parametrize("tag", list(WIDE))
def test_real_widths_match_reference(golden, tag):
    """Six levels / 512-channel levels / attention at four resolutions (121.5 M parameters) and the cs2 widths (81.3 M): seeded
    re-initialisation identical to the reference's state_dict, forward at 64x64 within fp32 noise."""
    seed, over = WIDE[tag]
    g = golden("lineage_b_wide.npz")
    hp = ob.hparams(**over)
    sd = ob.init_state_dict(seed, hp)
    keys = list(g[f"{tag}_sd_keys"])
    assert sorted(sd.keys()) == keys
    dg = ob.state_dict_digest(sd)
    assert np.array_equal(np.array([dg[k] for k in keys]), g[f"{tag}_sd_digest"])
    assert sum(v.numel() for v in sd.values()) == int(g[f"{tag}_nparams"]) == {"ddim2": 121_543_683, "cs2": 81_311_363}[tag]
    x = torch.from_numpy(g[f"{tag}_x"])
    for t in (21, 801):
        eps = ob.unet_forward(sd, hp, x, torch.full((1,), t, dtype=torch.long)).numpy()
        assert np.abs(eps - g[f"{tag}_eps_t{t}"]).max() < TOL


def test_quad_discretisation_matches_reference(golden):
    """ddim_discr_method='quad' (diff_model.py:431-434): the sequence and a 7-step loop on the small net."""
    g = golden("lineage_b_wide.npz")
    seq, prev = ob.ddim_sequences(1000, 7, "quad")
    assert list(seq) == list(g["quad_seq"]) and list(prev) == [0] + list(seq[:-1])
    hp = ob.hparams(**CASES["small"][1])
    sd = ob.init_state_dict(3, hp)
    out = ob.ddim_sample(lambda x, t: ob.unet_forward(sd, hp, x, t), torch.from_numpy(g["quad_xT"]), steps=7, method="quad")
    assert np.abs(out - g["quad_out"]).max() < 1e-4
