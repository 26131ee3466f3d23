"""Helpers for the -m gpu tests: run single C-ABI ops on torch tensors."""
import torch

from advshadow_amd import _lib
from advshadow_amd.engine import Builder, pack_conv_weight, dtype_code, TORCH_DT


def dev():
    return torch.device("cuda", 0)


def nhwc(x, dt):
    """NCHW f32 cpu -> NHWC device tensor in compute dtype."""
    return x.permute(0, 2, 3, 1).contiguous().to(dev(), TORCH_DT[dtype_code(dt)])


def nchw(y):
    return y.float().permute(0, 3, 1, 2).contiguous().cpu()


class OneOp:
    """A throwaway Builder on its own stream; call .go() after emitting ops."""

    def __init__(self, dt, batch):
        self.stream = torch.cuda.Stream(device=dev())
        torch.cuda.synchronize()
        self.b = Builder(dev(), dt, self.stream, batch)

    def go(self):
        torch.cuda.synchronize()
        self.b.plan.run_eager()
        self.stream.synchronize()


def bf16_round(x):
    return x.to(torch.bfloat16).float()


def tdt(dt):
    return {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dt]


def lp_round(dt, x):
    """x rounded to the storage type of compute mode dt (identity for fp32)."""
    return x if dt == "fp32" else x.to(tdt(dt)).float()
