"""The fused attack loop (sampler -> victim -> ASR; composite -> PSNR/SSIM) on the MI355X against the
same stages evaluated with the CPU oracles, incl. identical top-1 decisions."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from advshadow_amd import attack, parallel  # noqa: E402
from advshadow_amd.diff_model import GaussianDiffusion, UNetModel  # noqa: E402
from advshadow_amd.victims import ResNet50  # noqa: E402
from oracle import lineage_b as ob, metrics as om, shadow as osh, victims as ov  # noqa: E402


def test_attack_loop_matches_oracle_pipeline():
    n, S = 4, 32
    over = dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4)
    torch.manual_seed(3)
    net = UNetModel(**over).to("cuda").eval()
    hp = ob.hparams(**over)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    torch.manual_seed(1)
    victim = ResNet50(37)
    vsd = ov.randomize_bn({k: v.clone() for k, v in victim.state_dict().items()}, 9)
    victim.load_state_dict(vsd)
    victim = victim.to("cuda").eval()
    xT = parallel.image_noise(range(n), (3, S, S))
    g = torch.Generator().manual_seed(7)
    clean = torch.rand(n, 3, S, S, generator=g)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    fmask = (((xx - 16.0) ** 2 + (yy - 16.0) ** 2) <= 10 ** 2).float()[None, None].expand(n, 1, S, S).contiguous()
    centers = torch.tensor([[16.0, 16.0]] * n)
    radii = torch.tensor([5.0] * n)
    gd = GaussianDiffusion()

    def sample_fn():
        x = gd.ddim_sample(net, S, batch_size=n, ddim_timesteps=4, x_T=xT, return_tensor=True)
        out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        from advshadow_amd import _lib
        _lib.check(_lib.load().advs_to_uint8(x.data_ptr(), out.data_ptr(), x.numel(), 1, torch.cuda.current_stream().cuda_stream))
        return out

    gen, pred, psnr, ssim = attack.attack_shard(sample_fn, victim, clean.cuda(), fmask.cuda(), centers, radii)
    # ---- oracle pipeline on the CPU
    xf = torch.from_numpy(ob.ddim_sample(lambda x, t: ob.unet_forward(sd, hp, x, t), xT, steps=4))
    ref_u8 = (((xf + 1) * 0.5) * 255).clamp(0, 255).type(torch.uint8)
    d = (gen.cpu().to(torch.int16) - ref_u8.to(torch.int16)).abs()
    assert d.max().item() <= 1 and (d > 0).float().mean().item() < 0.01
    for i in range(n):
        pil = Image.fromarray(gen[i].cpu().numpy().transpose(1, 2, 0)).resize((224, 224), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(pil).transpose(2, 0, 1).astype(np.float32) / 255.0)[None]
        assert int(ov.resnet50_forward(vsd, x).argmax(1)) == int(pred[i])          # identical top-1 decision
        sh = osh.apply_shadow(clean[i], (16.0, 16.0), 5.0, fmask[i], 0.43, 5)
        to64 = lambda t: np.asarray(Image.fromarray((t * 255).clamp(0, 255).byte().numpy().transpose(1, 2, 0))
                                    .resize((64, 64), Image.BILINEAR)).transpose(2, 0, 1).astype(np.float32) / 255.0
        s, p = om.calculate_ssim_psnr(to64(clean[i]), to64(sh), 7)
        assert abs(float(ssim[i]) - s) < 1e-5 and abs(float(psnr[i]) - p) < 1e-3
    m, _ = attack.run_attack(n, lambda lo, hi: (gen, pred, psnr, ssim), labels=torch.arange(n) % 37)
    assert m["n"] == n and 0.0 <= m["asr"] <= 1.0 and m["psnr"] > 0


def test_attack_shard_with_gradient_attack_composite():
    """gradient_attack= switches the composite to the classifier branch of apply_shadow (train_shadow.py:242-266):
    equals the direct call, stays within epsilon * mask of the closed-form shadow, and is shard-invariant."""
    from advshadow_amd import adversarial, shadow
    n, S = 4, 64
    torch.manual_seed(1)
    victim = ResNet50(37)
    victim.load_state_dict(ov.randomize_bn({k: v.clone() for k, v in victim.state_dict().items()}, 9))
    victim = victim.to("cuda").eval()
    g = torch.Generator().manual_seed(8)
    clean = torch.rand(n, 3, S, S, generator=g).cuda()
    fmask = (torch.rand(n, 1, S, S, generator=g) > 0.2).float().cuda()
    centers, radii = torch.tensor([[30.0, 34.0]] * n), torch.tensor([14.0] * n)
    labels = torch.arange(n) % 37
    gen0 = torch.zeros(n, 3, S, S, dtype=torch.uint8, device="cuda")
    ga = dict(labels=labels, epsilon=0.01, alpha=0.005, iterations=5)
    _, pred, psnr, ssim = attack.attack_shard(lambda: gen0, victim, clean, fmask, centers, radii, gradient_attack=ga)
    direct = adversarial.apply_shadow_adversarial_batch(victim, clean, centers, radii, fmask, labels, iterations=5)
    plain = shadow.apply_shadow_batch(clean, centers, radii, fmask)
    assert (direct - plain).abs().max().item() <= 0.01 + 1e-6 and not torch.equal(direct, plain)
    from advshadow_amd.metrics import ssim_psnr_batch
    sp = ssim_psnr_batch(attack.to_64(clean), attack.to_64(direct), 7)
    assert torch.equal(psnr, sp[:, 1].float()) and torch.equal(ssim, sp[:, 0].float())
    # two shards of two images give the same per-image numbers
    for lo in (0, 2):
        sl = slice(lo, lo + 2)
        _, _, p2, s2 = attack.attack_shard(lambda: gen0[sl], victim, clean[sl], fmask[sl], centers[sl], radii[sl],
                                           gradient_attack=dict(ga, labels=labels[sl]))
        assert torch.equal(p2, psnr[sl]) and torch.equal(s2, ssim[sl])
    with pytest.raises(TypeError):
        attack.attack_shard(lambda: gen0, victim, clean, fmask, centers, radii, gradient_attack=dict(ga, step=1))


def test_config4_reduced_fp16_ddim100_vit_victim():
    """BASELINE config 4 at reduced size: fp16 eps-predictor, ``ddim_sample(ddim_timesteps=100)`` -- the sequence
    [1, 11, ..., 991] of diff_model.py:428-440, one captured step graph replayed 100 times -- then ``attack_shard`` with
    the ViT victim (configvit.json's architecture: HF ViTForImageClassification, 37 labels, ASR_fast.py:47-58) in fp16.
    Checked: the step sequence; replays are bit-identical (images, decisions, metrics); the fp32 ViT plan takes the same
    top-1 decision as the installed transformers ViT on the SAME uint8 images for every image, the fp16 plan wherever
    transformers' own top-1 margin exceeds twice the fp16 logit error bound of test_vit_victim_matches_hf_transformers
    (random-init logits sit 0.01-0.08 apart; a tie inside the rounding error has no right answer); the fp16 sample
    eps stays within 0.02 of the fp32 eps at the first, middle and last of the 100 steps (teacher-forced)."""
    from advshadow_amd import _lib
    from advshadow_amd.victims import ViTVictim
    n, S = 4, 32
    over = dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4)
    gd = GaussianDiffusion()
    seq, prev = gd.ddim_sequences(1000, 100)
    assert list(seq) == list(range(1, 992, 10)) and list(prev) == [0] + list(range(1, 982, 10))
    xT = parallel.image_noise(range(n), (3, S, S))
    g = torch.Generator().manual_seed(7)
    clean = torch.rand(n, 3, S, S, generator=g).cuda()
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    fmask = (((xx - 16.0) ** 2 + (yy - 16.0) ** 2) <= 10 ** 2).float()[None, None].expand(n, 1, S, S).contiguous().cuda()
    centers, radii = torch.tensor([[16.0, 16.0]] * n), torch.tensor([5.0] * n)
    hf = ov.hf_vit(37, seed=2)

    def sampler(dt):
        torch.manual_seed(3)
        net = UNetModel(compute_dtype=dt, **over).to("cuda").eval()

        def sample_fn():
            x = gd.ddim_sample(net, S, batch_size=n, ddim_timesteps=100, x_T=xT, return_tensor=True)
            out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
            _lib.check(_lib.load().advs_to_uint8(x.data_ptr(), out.data_ptr(), x.numel(), 1, torch.cuda.current_stream().cuda_stream))
            return out
        return net, sample_fn

    def victim(dt):
        v = ViTVictim(37, compute_dtype=dt)
        v.load_state_dict(hf.state_dict())
        return v.to("cuda").eval()

    net16, fn16 = sampler("fp16")
    v16 = victim("fp16")
    gen, pred, psnr, ssim = attack.attack_shard(fn16, v16, clean, fmask, centers, radii)
    gen2, pred2, psnr2, ssim2 = attack.attack_shard(fn16, v16, clean, fmask, centers, radii)     # graph replays
    assert torch.equal(gen, gen2) and torch.equal(pred, pred2) and torch.equal(psnr, psnr2) and torch.equal(ssim, ssim2)
    assert gen.dtype == torch.uint8 and gen.shape == (n, 3, S, S)
    # the victim's decisions against transformers' ViT on the same uint8 images (ASR_fast.py:90-97 preprocessing)
    xs = []
    for i in range(n):
        pil = Image.fromarray(gen[i].cpu().numpy().transpose(1, 2, 0)).resize((224, 224), Image.BILINEAR)
        xs.append(torch.from_numpy(np.asarray(pil).transpose(2, 0, 1).astype(np.float32) / 255.0))
    with torch.no_grad():
        ref = hf(pixel_values=torch.stack(xs)).logits
    top2 = ref.topk(2, 1).values
    margin = top2[:, 0] - top2[:, 1]
    from advshadow_amd.asr import evaluate_batch
    pred32 = evaluate_batch(gen, victim("fp32"))
    assert torch.equal(pred32.cpu().long(), ref.argmax(1))
    bound16 = 0.01 * max(1.0, ref.abs().max().item())
    sure = margin > 2 * bound16
    assert torch.equal(pred.cpu().long()[sure], ref.argmax(1)[sure]), (pred, ref.argmax(1), margin)
    # fp16 against fp32 along the fp32 trajectory of the same 100-step sequence (teacher forcing: a random-init net
    # amplifies 16-bit rounding over a free-running loop -- BASELINE.md sec. 2 -- so the bound is per forward)
    net32, _ = sampler("fp32")
    trace = []
    ob.ddim_sample(lambda x, t: net32(x.cuda(), t.cuda()).cpu(), xT, steps=100, trace=trace)
    assert [tr[0] for tr in trace[:3]] == [991, 981, 971] and trace[-1][0] == 1 and len(trace) == 100
    worst = 0.0
    for k in (0, 49, 99):
        t, eps32, _ = trace[k]
        x_in = xT if k == 0 else trace[k - 1][2]
        e16 = net16(x_in.cuda(), torch.full((n,), t, dtype=torch.long, device="cuda")).cpu()
        worst = max(worst, (e16 - eps32).abs().max().item())
    print("config-4 reduced: fp16 vs fp32 eps along the 100-step trajectory, max", worst)
    assert worst < 0.02


def test_config2_full_size_attack_loop():
    """BASELINE config 2 at FULL size on one GPU: batch 64, 3x256x256, bf16 default UNetModel (4 DDIM steps keep it short) ->
    uint8 -> Pillow-exact resize 224 -> ResNet-50 victim -> argmax; apply_shadow closed form -> 64x64 -> PSNR/SSIM.  For ALL 64
    images: the fp32 victim plan takes the CPU oracle's top-1 decision (oracle/victims.resnet50_forward on the SAME uint8
    images, ASR_fast.py:90-97 preprocessing), the bf16 victim plan wherever the oracle's top-1 margin exceeds twice the bf16
    logit error; PSNR / SSIM equal oracle/metrics on oracle/shadow's composite; and the two shards of 32 that a 2-GPU run would
    hold (x_T drawn per GLOBAL image id) reproduce images, decisions and metrics bit for bit (ASR_fast.py:101-126,
    PSNR_SSIM_fast.py:21-56)."""
    from advshadow_amd import _lib
    from advshadow_amd.asr import evaluate_batch
    n, S, steps = 64, 256, 4
    torch.manual_seed(0)
    net = UNetModel(compute_dtype="bf16").to("cuda").eval()
    gd = GaussianDiffusion()
    torch.manual_seed(1)
    v32 = ResNet50(37)
    vsd = ov.randomize_bn({k: v.clone() for k, v in v32.state_dict().items()}, 9)
    v32.load_state_dict(vsd)
    v32 = v32.to("cuda").eval()
    v16 = ResNet50(37, compute_dtype="bf16")
    v16.load_state_dict(vsd)
    v16 = v16.to("cuda").eval()
    g = torch.Generator().manual_seed(7)
    clean = torch.rand(n, 3, S, S, generator=g)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    fm1 = (((xx - 128.0) ** 2 + (yy - 128.0) ** 2) <= 80.0 ** 2).float()
    fmask = fm1[None, None].expand(n, 1, S, S).contiguous()
    centers, radii = torch.tensor([[128.0, 128.0]] * n), torch.tensor([40.0] * n)

    def shard(lo, hi, victim):
        xT = parallel.image_noise(range(lo, hi), (3, S, S)).cuda()

        def sample_fn():
            x = gd.ddim_sample(net, S, batch_size=hi - lo, ddim_timesteps=steps, x_T=xT, return_tensor=True)
            out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
            _lib.check(_lib.load().advs_to_uint8(x.data_ptr(), out.data_ptr(), x.numel(), 1, torch.cuda.current_stream().cuda_stream))
            return out
        return attack.attack_shard(sample_fn, victim, clean[lo:hi].cuda(), fmask[lo:hi].cuda(), centers[lo:hi], radii[lo:hi])

    gen, pred16, psnr, ssim = shard(0, n, v16)
    assert gen.shape == (n, 3, S, S) and gen.dtype == torch.uint8 and pred16.shape == (n,)
    # ---- shard invariance: 64 = 2 x 32
    for lo in (0, 32):
        g2, p2, q2, s2 = shard(lo, lo + 32, v16)
        sl = slice(lo, lo + 32)
        assert torch.equal(g2, gen[sl]) and torch.equal(p2, pred16[sl]) and torch.equal(q2, psnr[sl]) and torch.equal(s2, ssim[sl])
    # ---- decisions against the CPU oracle on the same uint8 images
    torch.set_num_threads(min(16, torch.get_num_threads()))
    xs = []
    for i in range(n):
        pil = Image.fromarray(gen[i].cpu().numpy().transpose(1, 2, 0)).resize((224, 224), Image.BILINEAR)
        xs.append(torch.from_numpy(np.asarray(pil).transpose(2, 0, 1).astype(np.float32) / 255.0))
    with torch.no_grad():
        ref = torch.cat([ov.resnet50_forward(vsd, torch.stack(xs[i:i + 16])) for i in range(0, n, 16)])
    pred32 = evaluate_batch(gen, v32)
    assert torch.equal(pred32.cpu().long(), ref.argmax(1))                                   # all 64, fp32 plan
    top2 = ref.topk(2, 1).values
    margin = top2[:, 0] - top2[:, 1]
    sure = margin > 2 * 0.01 * max(1.0, ref.abs().max().item())
    assert int(sure.sum()) >= n // 2, margin
    assert torch.equal(pred16.cpu().long()[sure], ref.argmax(1)[sure]), (pred16, ref.argmax(1), margin)
    # ---- PSNR / SSIM of all 64 against the CPU restatements
    to64 = lambda t: np.asarray(Image.fromarray((t * 255).clamp(0, 255).byte().numpy().transpose(1, 2, 0))
                                .resize((64, 64), Image.BILINEAR)).transpose(2, 0, 1).astype(np.float32) / 255.0
    for i in range(n):
        sh = osh.apply_shadow(clean[i], (128.0, 128.0), 40.0, fmask[i], 0.43, 5)
        s, p = om.calculate_ssim_psnr(to64(clean[i]), to64(sh), 7)
        assert abs(float(ssim[i]) - s) < 1e-5 and abs(float(psnr[i]) - p) < 1e-3, i
    labels = torch.arange(n) % 37
    m, _ = attack.run_attack(n, lambda lo, hi: (gen, pred16, psnr, ssim), labels=labels)
    assert m["n"] == n and abs(m["asr"] - float((pred16.cpu().long() != labels).float().mean())) < 1e-9
    print("config 2 full size:", m, "decisions beyond the bf16 margin:", int(sure.sum()), "of", n)
