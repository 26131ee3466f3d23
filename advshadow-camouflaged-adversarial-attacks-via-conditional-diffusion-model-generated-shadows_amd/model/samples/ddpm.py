"""Drop-in for ``model/samples/ddpm.py::DDPMDiffusion`` (the default ``--sample`` of generate()).

``sample(model, n, labels=None, cfg_scale=None)`` runs the 999 ancestral steps on the GPU: UNet
forward(s) + one fused update kernel per step.  The per-step Gaussian noise comes from torch's
generator on the model's device, as in the reference (ddpm.py:80-83); ``noise_fn(i, shape)`` injects it
instead (tests).  The reference's per-step PNG dump (``save_path``, a hard-coded Windows path) is not
reproduced.  Returns the clamped uint8 image (ddpm.py:95-97).
"""
import torch

from ... import _lib
from ...engine import Plan, ptr
from .ddim import BaseDiffusion, guidance_mode


class DDPMDiffusion(BaseDiffusion):
    def __init__(self, noise_steps=1000, beta_start=1e-4, beta_end=2e-2, img_size=256, device="cpu"):
        super().__init__(noise_steps, beta_start, beta_end, img_size, device)
        self._loops = {}

    @torch.no_grad()
    def sample(self, model, n, labels=None, cfg_scale=None, save_path=None, x_T=None, noise_fn=None, steps=None,
               return_float=False):
        dev = next(model.parameters()).device
        model.eval()
        mode = guidance_mode(labels, cfg_scale, "ddpm.py:68")
        eng = model.engine(n)
        ts = list(reversed(range(1, self.noise_steps)))
        if steps is not None:
            ts = ts[:steps]                                   # truncated chains (tests)
        idx = torch.as_tensor(ts)
        coef = torch.stack([self.alpha[idx], self.alpha_hat[idx], self.beta[idx]], 1).contiguous().to(dev)
        tseq = idx.to(torch.int64).to(dev)
        counter = torch.zeros((1,), dtype=torch.int32, device=dev)
        noise = torch.zeros_like(eng.x)
        fwd = eng.plan(mode)
        plan = Plan(eng.stream)
        plan.ops = list(fwd.ops)
        plan.keep = [fwd.keep, coef, tseq, counter, noise]
        eps = eng.eps_u if mode == "uncond" else eng.eps_c
        plan.add(_lib.load().advs_ddpm_step, ptr(eng.x), ptr(eps), ptr(eng.eps_u) if mode == "cfg" else 0,
                 float(cfg_scale or 0.0), ptr(noise), ptr(coef), ptr(tseq), len(ts), ptr(counter), ptr(eng.t), eng.B,
                 eng.x[0].numel())
        if x_T is None:
            x_T = torch.randn((n, 3, self.img_size, self.img_size))
        cur = torch.cuda.current_stream(dev)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x_T.to(dev, torch.float32), non_blocking=True)
            if labels is not None:
                eng.labels.copy_(labels.to(dev, torch.int64), non_blocking=True)
            eng.t.fill_(ts[0])
            captured = False
            for k, i in enumerate(ts):
                if i > 1:
                    if noise_fn is not None:
                        noise.copy_(noise_fn(i, tuple(noise.shape)).to(dev, torch.float32))
                    else:
                        noise.normal_()
                else:
                    noise.zero_()
                if model.use_graph and not captured and k == 1:
                    # step 0 ran eagerly (warm-up); capture from the second step on
                    eng.stream.synchronize()
                    x_keep, c_keep, t_keep = eng.x.clone(), counter.clone(), eng.t.clone()
                    plan.capture()                               # capture does not execute
                    eng.x.copy_(x_keep); counter.copy_(c_keep); eng.t.copy_(t_keep)
                    captured = True
                plan.run() if captured else plan.run_eager()
            xo = eng.x.clamp(-1, 1) if not return_float else eng.x.clone()
            if return_float:
                out = xo
            else:
                out = torch.empty(eng.x.shape, dtype=torch.uint8, device=dev)
                _lib.check(_lib.load().advs_to_uint8(xo.data_ptr(), out.data_ptr(), xo.numel(), 1, eng.stream.cuda_stream))
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        model.train()
        return out
