#!/usr/bin/env python
"""Generate golden vectors by importing the reference in the BUILD container.

Run from the repo root:  ``python tests/golden/make_golden.py [group ...]``.
It needs ``/root/reference`` (absent on the GPU box, where only the committed
``.npz`` files are used).  Modules the reference imports but never uses on this
path (torchvision, coloredlogs) are replaced by inert stubs, as SURVEY.md
Appendix B documents.  Nothing is written outside ``tests/golden/``.

Weights are never stored: every network is built under ``torch.manual_seed``
and the fixture keeps a per-tensor digest of the reference's ``state_dict`` so
the tests can prove the oracle's seeded re-initialisation is identical.
"""
import os
import sys
import types

os.environ["MPLBACKEND"] = "Agg"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _stub(name, subs=()):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    for s in subs:
        sm = types.ModuleType(f"{name}.{s}")
        setattr(m, s, sm)
        sys.modules[f"{name}.{s}"] = sm
    return m


tv = _stub("torchvision", ("models", "datasets", "transforms", "utils"))
tv.utils.save_image = lambda *a, **k: None
_stub("coloredlogs").install = lambda *a, **k: None
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)


def digest(sd):
    keys = sorted(sd.keys())
    return (np.array(keys), np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys]))


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------- lineage B
def gen_lineage_b():
    import diff_model as dm
    dm.tqdm = lambda it, **k: it

    # schedules and step sequences (diff_model.py:269-331, 428-440)
    out = {}
    for sched in ("cosine", "linear"):
        gd = dm.GaussianDiffusion(beta_schedule=sched)
        out[f"ac_{sched}"] = gd.alphas_cumprod.numpy()
    save("lineage_b_schedules.npz", **out)

    def run(tag, seed, hp, S, B, ts, steps):
        torch.manual_seed(seed)
        net = dm.UNetModel(**hp).eval()
        keys, dg = digest(net.state_dict())
        g = torch.Generator().manual_seed(1000 + seed)
        x = torch.randn(B, 3, S, S, generator=g)
        arrs = dict(sd_keys=keys, sd_digest=dg, x=x.numpy(), ts=np.array(ts))
        with torch.no_grad():
            for t in ts:
                arrs[f"eps_t{t}"] = net(x, torch.full((B,), t, dtype=torch.long)).numpy()
        for sched in ("cosine", "linear"):
            gd = dm.GaussianDiffusion(beta_schedule=sched)
            torch.manual_seed(1234)
            xT = torch.randn((B, 3, S, S))          # what ddim_sample draws first (diff_model.py:444)
            torch.manual_seed(1234)
            res = gd.ddim_sample(net, S, batch_size=B, channels=3, ddim_timesteps=steps)
            arrs[f"ddim_xT_{sched}"] = xT.numpy()
            arrs[f"ddim_out_{sched}"] = res
        arrs["ddim_steps"] = np.array(steps)
        save(f"lineage_b_{tag}.npz", **arrs)

    small = dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4)
    run("small", 3, small, 32, 2, [1, 501, 981], 5)
    # GaussianDiffusion.sample = the full-length ancestral loop (what main.py:124 / gen.py:562 call), here with
    # timesteps=24 so that the CPU suite can replay it: RNG stream = x_T, then one randn_like per step
    torch.manual_seed(3)
    net = dm.UNetModel(**small).eval()
    arrs = {}
    for sched in ("cosine", "linear"):
        gd = dm.GaussianDiffusion(timesteps=24, beta_schedule=sched)
        torch.manual_seed(4242)
        xT = torch.randn((2, 3, 32, 32))
        noise = np.stack([torch.randn((2, 3, 32, 32)).numpy() for _ in range(24)])           # steps 23, 22, ..., 0
        torch.manual_seed(4242)
        imgs = gd.sample(net, 32, batch_size=2, channels=3)
        arrs["xT"], arrs["noise"] = xT.numpy(), noise                  # same seed: identical stream for both schedules
        arrs[f"imgs_{sched}"] = np.stack([imgs[k] for k in (0, 11, 22, 23)])
        arrs[f"len_{sched}"] = np.array(len(imgs))
    save("lineage_b_ancestral.npz", **arrs)
    # attention at two levels + three res blocks + no-attention level, odd head count
    mid = dict(model_channels=64, channel_mult=(1, 2, 3), num_res_blocks=2, attention_resolutions=(1, 4), num_heads=2)
    run("mid", 5, mid, 32, 1, [21, 741], 4)
    # BASELINE config 0: UNetModel() defaults, 1x3x64x64, 10-step DDIM
    run("default", 0, {}, 64, 1, [1, 501, 981], 10)


# --------------------------------------------------------------------------- lineage A
def gen_lineage_a():
    from model.networks.unet import UNet
    from model.samples.ddim import DDIMDiffusion
    import model.samples.ddim as ddim_mod
    ddim_mod.tqdm = lambda it, **k: it

    def run(tag, seed, act, with_sample):
        torch.manual_seed(seed)
        net = UNet(num_classes=37, device="cpu", image_size=64, act=act).eval()
        keys, dg = digest(net.state_dict())
        g = torch.Generator().manual_seed(2000 + seed)
        x = torch.randn(2, 3, 64, 64, generator=g)
        t = torch.tensor([31, 901])
        y = torch.tensor([17, 3])
        arrs = dict(sd_keys=keys, sd_digest=dg, x=x.numpy(), t=t.numpy(), y=y.numpy())
        with torch.no_grad():
            arrs["eps_cond"] = net(x, t, y).numpy()
            arrs["eps_uncond"] = net(x, t).numpy()
        if with_sample:
            diff = DDIMDiffusion(sample_steps=10, img_size=64, device="cpu")
            arrs["alpha_hat"] = diff.alpha_hat.numpy()
            arrs["time_pairs"] = np.array([[int(a), int(b)] for a, b in diff.time_step])
            labels = torch.tensor([5, 30])
            torch.manual_seed(4321)
            xT = torch.randn((2, 3, 64, 64))                       # first draw of sample() (ddim.py:62)
            torch.manual_seed(4321)
            arrs["sample_xT"] = xT.numpy()
            arrs["sample_labels"] = labels.numpy()
            arrs["sample_cfg3"] = diff.sample(net, 2, labels=labels, cfg_scale=3).numpy()
            torch.manual_seed(4321)
            arrs["sample_uncond"] = diff.sample(net, 2).numpy()
            net.eval()
        save(f"lineage_a_{tag}.npz", **arrs)

    run("silu", 1, "silu", True)
    run("gelu", 2, "gelu", False)
    d500 = DDIMDiffusion(img_size=64, device="cpu")                # defaults: 500 sample steps
    save("lineage_a_schedule.npz", alpha_hat=d500.alpha_hat.numpy(),
         time_pairs_500=np.array([[int(a), int(b)] for a, b in d500.time_step]),
         **{f"beta_{n}": d500.prepare_noise_schedule(n).numpy() for n in ("linear", "cosine", "sqrt_linear", "sqrt")})


# --------------------------------------------------------------------------- DDPM / PLMS samplers
def gen_samplers():
    from model.networks.unet import UNet
    import model.samples.ddpm as ddpm_mod
    import model.samples.plms as plms_mod
    ddpm_mod.tqdm = lambda it, **k: it
    plms_mod.tqdm = lambda it, **k: it
    torch.manual_seed(1)
    net = UNet(num_classes=37, device="cpu", image_size=64, act="silu").eval()
    keys, dg = digest(net.state_dict())
    labels = torch.tensor([5, 30])
    arrs = dict(sd_keys=keys, sd_digest=dg, labels=labels.numpy())
    # DDPM: 12 noise steps -> 11 ancestral steps; the RNG stream is x_T then one randn per step with i > 1
    d = ddpm_mod.DDPMDiffusion(noise_steps=12, img_size=64, device="cpu")
    torch.manual_seed(777)
    arrs["ddpm_xT"] = torch.randn((2, 3, 64, 64)).numpy()
    arrs["ddpm_noise"] = np.stack([torch.randn((2, 3, 64, 64)).numpy() for _ in range(10)])     # i = 11 .. 2
    torch.manual_seed(777)
    arrs["ddpm_cfg3"] = d.sample(net, 2, labels=labels, cfg_scale=3, save_path=None).numpy()
    net.eval()
    p = plms_mod.PLMSDiffusion(sample_steps=8, img_size=64, device="cpu")
    torch.manual_seed(888)
    arrs["plms_xT"] = torch.randn((2, 3, 64, 64)).numpy()
    torch.manual_seed(888)
    arrs["plms_cfg3"] = p.sample(net, 2, labels=labels, cfg_scale=3, save_path=None).numpy()
    torch.manual_seed(888)
    arrs["plms_uncond"] = p.sample(net, 2, save_path=None).numpy()
    save("lineage_a_samplers.npz", **arrs)


# --------------------------------------------------------------------------- CSPDarkUnet (second --network)
def gen_cspdark():
    from model.networks.cspdarkunet import CSPDarkUnet
    from model.samples.ddim import DDIMDiffusion
    import model.samples.ddim as ddim_mod
    ddim_mod.tqdm = lambda it, **k: it

    def run(tag, seed, act, with_sample):
        torch.manual_seed(seed)
        net = CSPDarkUnet(num_classes=37, device="cpu", image_size=64, act=act).eval()
        keys, dg = digest(net.state_dict())
        g = torch.Generator().manual_seed(3000 + seed)
        x = torch.randn(2, 3, 64, 64, generator=g)
        t = torch.tensor([47, 873])
        y = torch.tensor([9, 36])
        arrs = dict(sd_keys=keys, sd_digest=dg, x=x.numpy(), t=t.numpy(), y=y.numpy())
        with torch.no_grad():
            arrs["eps_cond"] = net(x, t, y).numpy()
            arrs["eps_uncond"] = net(x, t).numpy()
        if with_sample:
            diff = DDIMDiffusion(sample_steps=10, img_size=64, device="cpu")
            labels = torch.tensor([12, 2])
            torch.manual_seed(5151)
            arrs["sample_xT"] = torch.randn((2, 3, 64, 64)).numpy()
            arrs["sample_labels"] = labels.numpy()
            torch.manual_seed(5151)
            arrs["sample_cfg3"] = diff.sample(net, 2, labels=labels, cfg_scale=3).numpy()
            net.eval()
        save(f"cspdark_{tag}.npz", **arrs)

    run("silu", 3, "silu", True)
    run("lrelu", 4, "lrelu", False)


# --------------------------------------------------------------------------- the reference's REAL widths + the 'quad' sequence (round 3)
def gen_wide():
    """The hyper-parameters the reference's drivers actually instantiate: ddim2/main2.py:118-127
    (six levels (1,1,2,2,4,4), attention at ds 4/8/16/32, 121.5 M parameters, 128-channel heads at 512 channels) and gen.py:522-528
    "cs2" ((1,2,3,4), attention at ds 2, 81.3 M) -- one forward each at 64x64, outputs only -- and ddim_sample's
    ddim_discr_method='quad' branch (diff_model.py:431-434) on the small net."""
    import diff_model as dm
    dm.tqdm = lambda it, **k: it
    arrs = {}
    for tag, seed, hp in (("ddim2", 11, dict(num_res_blocks=2, attention_resolutions=(4, 8, 16, 32), channel_mult=(1, 1, 2, 2, 4, 4), dropout=0.1)),
                          ("cs2", 12, dict(num_res_blocks=2, channel_mult=(1, 2, 3, 4), attention_resolutions=(2,), dropout=0.1))):
        torch.manual_seed(seed)
        net = dm.UNetModel(**hp).eval()
        keys, dg = digest(net.state_dict())
        arrs[f"{tag}_sd_keys"], arrs[f"{tag}_sd_digest"] = keys, dg
        arrs[f"{tag}_nparams"] = np.array(sum(p.numel() for p in net.parameters()))
        g = torch.Generator().manual_seed(1000 + seed)
        x = torch.randn(1, 3, 64, 64, generator=g)
        arrs[f"{tag}_x"] = x.numpy()
        with torch.no_grad():
            for t in (21, 801):
                arrs[f"{tag}_eps_t{t}"] = net(x, torch.full((1,), t, dtype=torch.long)).numpy()
    small = dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4)
    torch.manual_seed(3)
    net = dm.UNetModel(**small).eval()
    gd = dm.GaussianDiffusion()
    torch.manual_seed(1234)
    arrs["quad_xT"] = torch.randn((2, 3, 32, 32)).numpy()
    torch.manual_seed(1234)
    arrs["quad_out"] = gd.ddim_sample(net, 32, batch_size=2, channels=3, ddim_timesteps=7, ddim_discr_method="quad")
    arrs["quad_seq"] = ((np.linspace(0, np.sqrt(1000 * .8), 7)) ** 2).astype(int) + 1
    save("lineage_b_wide.npz", **arrs)


GROUPS = {"wide": gen_wide, "lineage_b": gen_lineage_b, "lineage_a": gen_lineage_a, "samplers": gen_samplers, "cspdark": gen_cspdark}

if __name__ == "__main__":
    for g in (sys.argv[1:] or list(GROUPS)):
        GROUPS[g]()
