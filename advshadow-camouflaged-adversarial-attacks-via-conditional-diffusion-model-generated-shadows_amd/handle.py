"""ctypes view of the handle-level C entry points (include/advshadow.h: advs_unet_*, advs_ddim_*; csrc/unet_handle.hip).

What a host WITHOUT this package's Python plan builder would write against ``libadvshadow_hip.so``: the network, its launch plan
and the DDIM loop live behind an opaque ``advs_unet*``; the caller hands over state_dict tensors and device pointers.  The Python
product path (``diff_model.UNetModel``) does not go through here -- tests/test_gpu_handle.py holds the two paths bit-identical."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import UNetConfig, check
from .engine import dtype_code


class CUNet:
    """``advs_unet`` for one architecture / dtype.  ``load_state_dict`` takes any mapping name -> array-like (torch layout)."""

    def __init__(self, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(8, 16),
                 channel_mult=(1, 2, 2, 2), num_heads=4, compute_dtype="fp32"):
        _lib.init_device()
        self.lib = _lib.load()
        cfg = UNetConfig()
        cfg.in_channels, cfg.model_channels, cfg.out_channels, cfg.num_res_blocks = in_channels, model_channels, out_channels, num_res_blocks
        cfg.n_attention_resolutions, cfg.n_channel_mult = len(attention_resolutions), len(channel_mult)
        for i, v in enumerate(attention_resolutions):
            cfg.attention_resolutions[i] = v
        for i, v in enumerate(channel_mult):
            cfg.channel_mult[i] = v
        cfg.num_heads, cfg.dtype = num_heads, dtype_code(compute_dtype)
        self.h = C.c_void_p()
        check(self.lib.advs_unet_create(C.byref(cfg), C.byref(self.h)), "advs_unet_create")
        self.stream = None
        self.batch = self.size = 0
        self.channels = (in_channels, out_channels)

    def param_names(self):
        out = []
        buf = C.create_string_buffer(256)
        n = C.c_longlong()
        for i in range(self.lib.advs_unet_param_count(self.h)):
            check(self.lib.advs_unet_param_name(self.h, i, buf, 256, C.byref(n)), "advs_unet_param_name")
            out.append((buf.value.decode(), n.value))
        return out

    def set_param(self, name, value):
        a = np.ascontiguousarray(torch.as_tensor(value).detach().float().cpu().numpy().reshape(-1))
        check(self.lib.advs_unet_set_param(self.h, name.encode(), a.ctypes.data, a.size), f"advs_unet_set_param({name})")

    def load_state_dict(self, sd):
        for name, _ in self.param_names():
            self.set_param(name, sd[name])

    def plan(self, batch, size, uniform_t=False, stream=None):
        self.stream = stream if stream is not None else torch.cuda.Stream()
        check(self.lib.advs_unet_plan(self.h, batch, size, 1 if uniform_t else 0, self.stream.cuda_stream), "advs_unet_plan")
        self.batch, self.size = batch, size

    def forward(self, x, t):
        """x [B, C, S, S] f32 and t [B] int64 on the GPU -> eps, on the plan's stream."""
        x = x.contiguous().float()
        t = t.contiguous().to(torch.int64)
        eps = torch.empty((self.batch, self.channels[1], self.size, self.size), dtype=torch.float32, device=x.device)
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        check(self.lib.advs_unet_forward(self.h, x.data_ptr(), t.data_ptr(), eps.data_ptr()), "advs_unet_forward")
        self.stream.synchronize()
        return eps

    def ddim_run(self, x_T, coef, tseq, clip_denoised=True):
        """coef [n, 3] f32 and tseq [n] int64 (host, loop order; ``ddim_tables`` or GaussianDiffusion._tables) -> the sample."""
        x = x_T.contiguous().float().clone()
        coef = np.ascontiguousarray(np.asarray(coef, dtype=np.float32))
        tseq = np.ascontiguousarray(np.asarray(tseq, dtype=np.int64))
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        check(self.lib.advs_ddim_run(self.h, x.data_ptr(), coef.ctypes.data, tseq.ctypes.data, len(tseq), 1 if clip_denoised else 0), "advs_ddim_run")
        self.stream.synchronize()
        return x

    def close(self):
        if self.h:
            self.lib.advs_unet_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ddim_tables(timesteps=1000, ddim_timesteps=50, beta_schedule="cosine", method="uniform", eta=0.0):
    """advs_ddim_tables: (coef [n, 3] f32, tseq [n] int64) in loop order."""
    lib = _lib.load()
    n = C.c_int()
    args = (1 if beta_schedule == "cosine" else 0, timesteps, ddim_timesteps, 1 if method == "quad" else 0, float(eta))
    check(lib.advs_ddim_tables(*args, None, None, C.byref(n)), "advs_ddim_tables")
    coef = np.zeros((n.value, 3), np.float32)
    tseq = np.zeros((n.value,), np.int64)
    check(lib.advs_ddim_tables(*args, coef.ctypes.data, tseq.ctypes.data, C.byref(n)), "advs_ddim_tables")
    return coef, tseq


class CResNet50:
    """``advs_resnet50`` (the victim of ASR_fast.py:16-20) behind the C handle: ``forward`` and the uint8 evaluation chain."""

    def __init__(self, num_classes=37, compute_dtype="fp32"):
        _lib.init_device()
        self.lib = _lib.load()
        self.h = C.c_void_p()
        self.num_classes = num_classes
        check(self.lib.advs_resnet50_create(num_classes, dtype_code(compute_dtype), C.byref(self.h)), "advs_resnet50_create")
        self.stream, self.batch, self.size = None, 0, 0

    def param_names(self):
        out, buf, n = [], C.create_string_buffer(256), C.c_longlong()
        for i in range(self.lib.advs_resnet50_param_count(self.h)):
            check(self.lib.advs_resnet50_param_name(self.h, i, buf, 256, C.byref(n)), "advs_resnet50_param_name")
            out.append((buf.value.decode(), n.value))
        return out

    def load_state_dict(self, sd):
        for name, _ in self.param_names():
            a = np.ascontiguousarray(torch.as_tensor(sd[name]).detach().float().cpu().numpy().reshape(-1))
            check(self.lib.advs_resnet50_set_param(self.h, name.encode(), a.ctypes.data, a.size), f"advs_resnet50_set_param({name})")

    def plan(self, batch, size=224, src_size=0, stream=None):
        self.stream = stream if stream is not None else torch.cuda.Stream()
        check(self.lib.advs_resnet50_plan(self.h, batch, size, src_size, self.stream.cuda_stream), "advs_resnet50_plan")
        self.batch, self.size = batch, size

    def forward(self, x):
        x = x.contiguous().float()
        out = torch.empty((self.batch, self.num_classes), dtype=torch.float32, device=x.device)
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        check(self.lib.advs_resnet50_forward(self.h, x.data_ptr(), out.data_ptr()), "advs_resnet50_forward")
        self.stream.synchronize()
        return out

    def eval_u8(self, images_u8_nchw):
        x = images_u8_nchw.contiguous()
        pred = torch.empty((self.batch,), dtype=torch.int32, device=x.device)
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        check(self.lib.advs_resnet50_eval_u8(self.h, x.data_ptr(), pred.data_ptr()), "advs_resnet50_eval_u8")
        self.stream.synchronize()
        return pred

    def close(self):
        if self.h:
            self.lib.advs_resnet50_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resize_tables(in_size, out_size):
    """advs_resize_tables: (bounds int32 [out, 2], coefs int32 [out, ksize], ksize)."""
    lib = _lib.load()
    ks = C.c_int()
    check(lib.advs_resize_tables(in_size, out_size, None, None, C.byref(ks)), "advs_resize_tables")
    b, k = np.zeros((out_size, 2), np.int32), np.zeros((out_size, ks.value), np.int32)
    check(lib.advs_resize_tables(in_size, out_size, b.ctypes.data, k.ctypes.data, C.byref(ks)), "advs_resize_tables")
    return b, k, ks.value
