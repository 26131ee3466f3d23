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
