"""Handle-level C entry points (advs_unet_*, advs_ddim_*; csrc/unet_handle.hip) against the Python plan of the same network:
the two hosts issue the same kernels in the same order, so eps and the DDIM sample must agree BIT FOR BIT (diff_model.py:245-267,
442-474); the table helper against GaussianDiffusion's torch.float64 tables."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from advshadow_amd.diff_model import GaussianDiffusion, UNetModel  # noqa: E402
from advshadow_amd.handle import CUNet, ddim_tables  # noqa: E402

CASES = {
    # name: (kwargs, batch, size, dtype)
    "small-fp32": (dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4), 2, 32, "fp32"),
    "mid-bf16": (dict(model_channels=64, channel_mult=(1, 2, 3), num_res_blocks=2, attention_resolutions=(1, 4), num_heads=2), 2, 64, "bf16"),
    "default-bf16-128": ({}, 1, 128, "bf16"),      # 128 x 128 at level 0: GroupNorm on load, sub-pixel upsample, fused shortcuts, tiles 17 / 19
    "default-fp16-64": ({}, 2, 64, "fp16"),
}


def _pair(kw, dtype):
    torch.manual_seed(11)
    net = UNetModel(compute_dtype=dtype, **kw).to("cuda").eval()
    with torch.no_grad():                    # non-trivial norms and biases
        for n, p in net.named_parameters():
            if n.endswith("bias") or ".0.weight" in n and p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    c = CUNet(compute_dtype=dtype, **{k: v for k, v in kw.items()})
    sd = net.state_dict()
    assert [n for n, _ in c.param_names()] == list(sd.keys())                 # the reference's construction order
    assert all(sd[n].numel() == k for n, k in c.param_names())
    c.load_state_dict(sd)
    half = net.model_channels // 2
    c.set_param("freqs", torch.exp(-np.log(10000) * torch.arange(0, half, dtype=torch.float32) / half))     # the table the Python plan uploads
    return net, c


@pytest.mark.parametrize("name", list(CASES))
def test_forward_is_bit_identical_to_the_python_plan(name):
    kw, B, S, dtype = CASES[name]
    net, c = _pair(kw, dtype)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, S, S, generator=g).cuda()
    t = torch.tensor([501, 17][:B]).cuda()
    want = net(x, t)
    c.plan(B, S)
    got = c.forward(x, t)
    assert torch.equal(got, want), (got - want).abs().max().item()
    got2 = c.forward(x, t)                  # the captured graph replays
    assert torch.equal(got2, want)
    c.close()


@pytest.mark.parametrize("name", ["small-fp32", "default-bf16-128"])
def test_ddim_loop_is_bit_identical_to_the_python_sampler(name):
    kw, B, S, dtype = CASES[name]
    net, c = _pair(kw, dtype)
    diff = GaussianDiffusion(timesteps=1000, beta_schedule="cosine")
    coef, tseq = diff._tables(6, "uniform", 0.0, torch.device("cpu"))
    g = torch.Generator().manual_seed(6)
    xT = torch.randn(B, 3, S, S, generator=g).cuda()
    want = diff.ddim_sample(net, S, batch_size=B, ddim_timesteps=6, x_T=xT, return_tensor=True)
    c.plan(B, S, uniform_t=True)
    got = c.ddim_run(xT, coef.numpy(), tseq.numpy())
    assert torch.equal(got, want), (got - want).abs().max().item()
    got = c.ddim_run(xT, coef.numpy(), tseq.numpy())          # second run: captured step, counters reset
    assert torch.equal(got, want)
    c.close()


@pytest.mark.parametrize("sched,method,n", [("cosine", "uniform", 50), ("linear", "uniform", 50), ("cosine", "quad", 20), ("linear", "uniform", 30)])
def test_table_helper_vs_the_float64_tables(sched, method, n):
    diff = GaussianDiffusion(timesteps=1000, beta_schedule=sched)
    coef, tseq = diff._tables(n, method, 0.0, torch.device("cpu"))
    c2, t2 = ddim_tables(1000, n, sched, method, 0.0)
    assert np.array_equal(t2, tseq.numpy())
    # libm cos / torch's vectorised cos may differ in the last place of a float64; after .float() that is at most one f32 ulp
    assert np.allclose(c2, coef.numpy(), rtol=2e-7, atol=0)


def test_handle_errors_are_loud():
    c = CUNet(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4)
    from advshadow_amd._lib import AdvsError
    with pytest.raises(AdvsError, match="never set"):
        c.plan(1, 32)
    with pytest.raises(AdvsError, match="not a parameter"):
        c.set_param("nope.weight", torch.zeros(3))
    with pytest.raises(AdvsError, match="elements"):
        c.set_param("out.2.bias", torch.zeros(7))
    x = torch.zeros(1, 3, 32, 32, device="cuda")
    c.batch, c.size, c.stream = 1, 32, torch.cuda.Stream()
    with pytest.raises(AdvsError, match="no plan"):
        c.forward(x, torch.zeros(1, dtype=torch.int64, device="cuda"))
    c.close()


def _lcg_units(n, state=12345):
    """examples/c_host_ddim.c: next_unit(), n values, vectorised (x_k = a^k x_0 + c (1 + a + ... + a^(k-1)) mod 2^32)."""
    a, c = np.uint32(1664525), np.uint32(1013904223)
    with np.errstate(over="ignore"):
        pw = np.cumprod(np.full(n, a, dtype=np.uint32), dtype=np.uint32)                  # a^1 .. a^n
        geo = np.cumsum(np.concatenate([np.ones(1, np.uint32), pw[:-1]]), dtype=np.uint32)   # 1 + a + ... + a^(k-1)
        st = pw * np.uint32(state) + c * geo
    return (st >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0) - np.float32(0.5)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_c_host_program(tmp_path, dtype):
    """examples/c_host_ddim.c -- plain C, gcc, no Python in the process -- samples with the handle-level entry points; the same
    weights and x_T through this package's Python plan (same tables: the C helper's) must give the same bits."""
    import os
    import subprocess
    from advshadow_amd import _lib
    from advshadow_amd.diff_model import _DDIMLoop
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe, dump = str(tmp_path / "c_host_ddim"), str(tmp_path / "x0.bin")
    subprocess.run(["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "examples", "c_host_ddim.c"), "-I" + os.path.join(root, "include"),
                    "-I/opt/rocm/include", "-L" + libdir, "-ladvshadow_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True)
    steps, B, S = 5, 2, 32
    r = subprocess.run([exe, dtype, str(steps), dump], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = torch.from_numpy(np.fromfile(dump, dtype=np.float32).reshape(B, 3, S, S))

    net = UNetModel(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4, compute_dtype=dtype).eval()
    sd = net.state_dict()
    total = sum(v.numel() for v in sd.values()) + B * 3 * S * S
    u = _lcg_units(total)
    o = 0
    for name, v in sd.items():
        n = v.numel()
        x = torch.from_numpy(u[o:o + n].copy())
        o += n
        bias = name.endswith(".bias")
        norm_w = not bias and any(k in name for k in (".conv1.0.", ".conv2.0.", ".norm.")) or name.startswith("out.0.") and not bias
        sd[name] = (1.0 + 0.25 * x if norm_w else (0.125 * x if bias else 0.25 * x)).reshape(v.shape)
    xT = torch.from_numpy(4.0 * u[o:o + B * 3 * S * S]).reshape(B, 3, S, S)
    net.load_state_dict(sd)
    net = net.to("cuda")
    coef, tseq = ddim_tables(1000, steps, "cosine", "uniform", 0.0)
    # the C host has no torch: its frequency table of the sinusoidal embedding comes from libm's expf (csrc/unet_handle.hip), one f32 ulp
    # from torch's in a few entries.  Give the Python plan libm's table, so that the comparison is of the two HOSTS, bit for bit.
    import ctypes
    import math
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype, libm.expf.argtypes = ctypes.c_float, [ctypes.c_float]
    half = net.model_channels // 2
    fr = [libm.expf(float(np.float32(-math.log(10000.0)) * np.float32(i) / np.float32(half))) for i in range(half)]
    from advshadow_amd.engine import dtype_code
    net.packed_weights(dtype_code(dtype))["freqs"].copy_(torch.tensor(fr, dtype=torch.float32))
    eng = net.engine(B, S, uniform_t=True)
    with torch.cuda.stream(eng.stream):
        loop = _DDIMLoop(eng, torch.from_numpy(coef).cuda(), torch.from_numpy(tseq).cuda(), True, 0.0)
        eng.x.copy_(xT.cuda())
        loop.run()
        want = eng.x.clone()
    eng.stream.synchronize()
    assert torch.equal(got, want.cpu()), (got - want.cpu()).abs().max().item()
    assert np.isfinite(got.numpy()).all()


# ------------------------------------------------------------------------------ the victim and the evaluation chain behind the C handle
@pytest.mark.parametrize("dtype,size,batch", [("fp32", 64, 2), ("bf16", 224, 4), ("fp16", 96, 3)])
def test_resnet50_handle_vs_python_plan(dtype, size, batch):
    from advshadow_amd.handle import CResNet50
    from advshadow_amd.victims import ResNet50
    torch.manual_seed(3)
    net = ResNet50(num_classes=37, compute_dtype=dtype).to("cuda").eval()
    with torch.no_grad():                    # running statistics and affine parameters away from their defaults
        for n, b in net.named_buffers():
            if n.endswith("running_mean"):
                b.normal_(0, 0.2)
            elif n.endswith("running_var"):
                b.uniform_(0.5, 1.5)
        for n, p in net.named_parameters():
            if ".bn" in n or n.startswith("bn1") or "downsample.1" in n:
                p.add_(0.1 * torch.randn_like(p))
    c = CResNet50(37, dtype)
    sd = net.state_dict()
    assert [n for n, _ in c.param_names()] == [k for k in sd if not k.endswith("num_batches_tracked")]
    c.load_state_dict(sd)
    c.plan(batch, size)
    x = torch.rand(batch, 3, size, size, generator=torch.Generator().manual_seed(4)).cuda()
    want = net(x)
    got = c.forward(x)
    # the fold (gamma / sqrt(var + eps), beta - mean * scale) runs in host f32 here and in torch's GPU f32 there: the same IEEE operations
    assert torch.equal(got, want), (got - want).abs().max().item()
    c.close()


@pytest.mark.parametrize("src", [256, 224, 64])
def test_resnet50_eval_chain_vs_evaluate_batch(src):
    """advs_resnet50_eval_u8 against asr.evaluate_batch (ASR_fast.py:90-126) on the same uint8 images: identical predictions, and the
    resize tables against imageops.bilinear_coeffs (Pillow's precompute_coeffs)."""
    from advshadow_amd.asr import evaluate_batch
    from advshadow_amd.handle import CResNet50, resize_tables
    from advshadow_amd.imageops import bilinear_coeffs
    from advshadow_amd.victims import ResNet50
    torch.manual_seed(8)
    net = ResNet50(num_classes=37, compute_dtype="bf16").to("cuda").eval()
    c = CResNet50(37, "bf16")
    c.load_state_dict(net.state_dict())
    B = 5
    c.plan(B, 224, src_size=src)
    img = torch.randint(0, 256, (B, 3, src, src), dtype=torch.uint8, generator=torch.Generator().manual_seed(9)).cuda()
    want = evaluate_batch(img, net)
    got = c.eval_u8(img)
    assert torch.equal(got.cpu(), want.cpu().to(torch.int32))
    for a, b in ((256, 224), (256, 64), (64, 224), (37, 224)):
        b1, k1, ks1 = bilinear_coeffs(a, b)
        b2, k2, ks2 = resize_tables(a, b)
        assert ks1 == ks2 and np.array_equal(b1, b2) and np.array_equal(k1, k2)
    c.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_c_host_attack_program(tmp_path, dtype):
    """examples/c_host_attack.c: BASELINE configs 2/3's loop -- DDIM sample, uint8 cast, Resize 224, ResNet-50, argmax; apply_shadow, 64 x 64
    PSNR / SSIM -- in plain C.  The same generator-made networks and inputs through this package's Python pipeline (attack.attack_shard's
    stages) must give the same predictions and the same metrics to the last bit of the printed float64."""
    import ctypes
    import math
    import os
    import re
    import subprocess
    from advshadow_amd import _lib
    from advshadow_amd.asr import evaluate_batch
    from advshadow_amd.attack import to_64
    from advshadow_amd.diff_model import _DDIMLoop
    from advshadow_amd.engine import dtype_code
    from advshadow_amd.metrics import ssim_psnr_batch
    from advshadow_amd.shadow import apply_shadow_batch
    from advshadow_amd.victims import ResNet50
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "c_host_attack")
    subprocess.run(["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "examples", "c_host_attack.c"), "-I" + os.path.join(root, "include"),
                    "-I/opt/rocm/include", "-L" + libdir, "-ladvshadow_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir,
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe, dtype], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = re.findall(r"image (\d+) pred (\d+) ssim (\S+) psnr (\S+)", r.stdout)
    B, S, steps = 4, 64, 4
    assert len(rows) == B, r.stdout

    net = UNetModel(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,), num_heads=4, compute_dtype=dtype).eval()
    vic = ResNet50(num_classes=37, compute_dtype=dtype).eval()
    sd_u, sd_v = net.state_dict(), vic.state_dict()
    vkeys = [k for k in sd_v if not k.endswith("num_batches_tracked")]
    cnt = B * 3 * S * S
    u = _lcg_units(sum(v.numel() for v in sd_u.values()) + sum(sd_v[k].numel() for k in vkeys) + 2 * cnt)
    o = 0

    def take(n):
        nonlocal o
        x = torch.from_numpy(u[o:o + n].copy())
        o += n
        return x
    for name, v in sd_u.items():
        x = take(v.numel())
        bias = name.endswith(".bias")
        norm_w = not bias and (any(k in name for k in (".conv1.0.", ".conv2.0.", ".norm.")) or name.startswith("out.0."))
        sd_u[name] = (1.0 + 0.25 * x if norm_w else (0.125 * x if bias else 0.25 * x)).reshape(v.shape)
    for name in vkeys:
        v = sd_v[name]
        x = take(v.numel())
        is_bn = "bn" in name or "downsample.1." in name
        if name.endswith("running_var"):
            y = 1.0 + 0.5 * x
        elif name.endswith("running_mean"):
            y = 0.25 * x
        elif is_bn and name.endswith(".weight"):
            y = 1.0 + 0.25 * x
        elif name.endswith(".bias"):
            y = 0.125 * x
        elif "downsample.0." in name or name.startswith("fc."):
            y = 0.25 * x
        else:
            y = 0.125 * x
        sd_v[name] = y.reshape(v.shape)
    xT = (4.0 * take(cnt)).reshape(B, 3, S, S)
    clean = (take(cnt) + 0.5).reshape(B, 3, S, S).cuda()
    net.load_state_dict(sd_u)
    vic.load_state_dict(sd_v)
    net, vic = net.to("cuda"), vic.to("cuda")
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype, libm.expf.argtypes = ctypes.c_float, [ctypes.c_float]
    half = net.model_channels // 2
    fr = [libm.expf(float(np.float32(-math.log(10000.0)) * np.float32(i) / np.float32(half))) for i in range(half)]
    net.packed_weights(dtype_code(dtype))["freqs"].copy_(torch.tensor(fr, dtype=torch.float32))      # the C host's libm table (see test_c_host_program)
    coef, tseq = ddim_tables(1000, steps, "cosine", "uniform", 0.0)
    eng = net.engine(B, S, uniform_t=True)
    with torch.cuda.stream(eng.stream):
        loop = _DDIMLoop(eng, torch.from_numpy(coef).cuda(), torch.from_numpy(tseq).cuda(), True, 0.0)
        eng.x.copy_(xT.cuda())
        loop.run()
        x0 = eng.x.clone()
    eng.stream.synchronize()
    gen = torch.empty(x0.shape, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.load().advs_to_uint8(x0.data_ptr(), gen.data_ptr(), x0.numel(), 1, torch.cuda.current_stream().cuda_stream))
    pred = evaluate_batch(gen, vic).cpu().tolist()
    centers = torch.tensor([[20.0 + 6.0 * b, 40.0 - 5.0 * b] for b in range(B)])
    radii = torch.tensor([10.0 + 2.0 * b for b in range(B)])
    shadowed = apply_shadow_batch(clean, centers, radii, torch.ones(B, 1, S, S, device="cuda"), 0.43, 5)
    sp = ssim_psnr_batch(to_64(clean), to_64(shadowed), 7).cpu().numpy()
    for i, p, ssim, psnr in rows:
        i = int(i)
        assert int(p) == pred[i], (i, p, pred[i])
        assert float(ssim) == sp[i, 0] and float(psnr) == sp[i, 1], (i, ssim, sp[i, 0], psnr, sp[i, 1])
    wrong = sum(int(p) != (int(i) * 7) % 37 for i, p, _, _ in rows)
    assert f"asr {wrong / B:.6f}" in r.stdout
    assert 5.0 < sp[:, 1].min() and sp[:, 0].max() < 1.0                      # a visible shadow: finite PSNR, SSIM below 1
