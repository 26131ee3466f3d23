"""Drop-in for ``model/samples/ddim.py::DDIMDiffusion`` (and the ``BaseDiffusion`` tables it needs).

``sample(model, n, labels=None, cfg_scale=None)`` keeps the reference's signature and return value
(uint8 ``[n,3,S,S]`` on the device, cast WITHOUT clamping so out-of-range pixels wrap mod 256,
ddim.py:97-99).  The whole loop runs on the GPU: per step one (or, with classifier-free guidance,
two) UNet forwards and one fused update kernel, captured once as a hipGraph and replayed.
Extra keywords: ``x_T`` (inject the start noise; otherwise drawn with torch's CPU generator and
moved to the device exactly as ddim.py:62 does), ``clamp`` (saturate instead of wrapping),
``return_float`` (skip the uint8 cast).
"""
import math

import torch

from ... import _lib
from ...engine import Plan, ptr


def guidance_mode(labels, cfg_scale, where):
    """Which forwards one sampler step needs (``where`` = the reference line for the error text).  With ``labels=None`` the
    reference calls ``model(x, t, None)`` for both branches and ``lerp(u, u, w)`` is ``u`` exactly, so that case is
    the unconditional plan whatever ``cfg_scale`` is -- never a conditional forward on stale labels."""
    if labels is None:
        return "uncond"
    if cfg_scale is None:
        raise TypeError(f"cfg_scale must be a number when labels are given ({where} compares it with 0)")
    return "cfg" if cfg_scale > 0 else "cond"


class BaseDiffusion:
    """model/samples/base.py:18-45: linear beta schedule in float32, alpha_hat = cumprod."""

    def __init__(self, noise_steps=1000, beta_start=1e-4, beta_end=2e-2, img_size=256, device="cpu"):
        self.noise_steps, self.beta_start, self.beta_end = noise_steps, beta_start, beta_end
        self.img_size, self.device = img_size, device
        # computed on the host so the fp32 cumprod is identical on every device
        self.beta = self.prepare_noise_schedule()
        self.alpha = 1.0 - self.beta
        self.alpha_hat = torch.cumprod(self.alpha, dim=0)

    def prepare_noise_schedule(self, schedule_name="linear"):
        """model/samples/base.py:40-85 (the constructor always takes "linear", base.py:34)."""
        if schedule_name == "linear":
            return torch.linspace(self.beta_start, self.beta_end, self.noise_steps)
        if schedule_name == "cosine":
            f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
            n = self.noise_steps
            return torch.tensor([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)])
        if schedule_name == "sqrt_linear":
            return torch.linspace(self.beta_start ** 0.5, self.beta_end ** 0.5, self.noise_steps) ** 2
        if schedule_name == "sqrt":
            return torch.linspace(self.beta_start, self.beta_end, self.noise_steps) ** 0.5
        raise NotImplementedError(f"Unknown beta schedule: {schedule_name}")

    def noise_images(self, x, time):
        """model/samples/base.py:86-98: (sqrt(a_hat) x + sqrt(1 - a_hat) eps, eps) on x's device."""
        ah = self.alpha_hat.to(x.device)[time]
        eps = torch.randn_like(x)
        return torch.sqrt(ah)[:, None, None, None] * x + torch.sqrt(1 - ah)[:, None, None, None] * eps, eps

    def sample_time_steps(self, n):
        """model/samples/base.py:100-108."""
        return torch.randint(low=1, high=self.noise_steps, size=(n,))


class DDIMDiffusion(BaseDiffusion):
    def __init__(self, noise_steps=1000, sample_steps=500, beta_start=1e-4, beta_end=2e-2, img_size=64, device="cpu"):
        super().__init__(noise_steps, beta_start, beta_end, img_size, device)
        self.sample_steps, self.eta = sample_steps, 0
        ts = torch.arange(0, noise_steps, noise_steps // sample_steps).long() + 1      # ddim.py:44
        ts = reversed(torch.cat((torch.tensor([0], dtype=torch.long), ts)))
        self.time_step = list(zip(ts[:-1], ts[1:]))
        self._loops = {}

    def _tables(self, dev):
        cur = torch.stack([a for a, _ in self.time_step])
        prev = torch.stack([b for _, b in self.time_step])
        if int(cur.max()) >= self.noise_steps:
            raise IndexError(f"time step {int(cur.max())} indexes alpha_hat[{self.noise_steps}] "
                             f"(the reference fails the same way for sample_steps == noise_steps)")
        a_t, a_p = self.alpha_hat[cur], self.alpha_hat[prev]
        c1 = self.eta * torch.sqrt((1 - a_t / a_p) * (1 - a_p) / (1 - a_t))
        coef = torch.stack([a_t, a_p, c1.float()], dim=1).contiguous().to(dev)
        return coef, cur.to(torch.int64).to(dev)

    @torch.no_grad()
    def sample(self, model, n, labels=None, cfg_scale=None, x_T=None, clamp=False, return_float=False):
        dev = next(model.parameters()).device
        model.eval()
        mode = guidance_mode(labels, cfg_scale, "ddim.py:83")
        eng = model.engine(n)
        cfg = float(cfg_scale or 0.0) if mode == "cfg" else 0.0
        key = (id(eng), mode, cfg)
        loop = self._loops.get(key)
        if loop is None:
            coef, tseq = self._tables(dev)
            loop = _Loop(eng, mode, cfg, coef, tseq)
            self._loops = {key: loop}
        if x_T is None:
            x_T = torch.randn((n, 3, self.img_size, self.img_size))
        cur = torch.cuda.current_stream(dev)
        eng.stream.wait_stream(cur)
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x_T.to(dev, torch.float32), non_blocking=True)
            if labels is not None:
                eng.labels.copy_(labels.to(dev, torch.int64), non_blocking=True)
            loop.run()
            if return_float:
                out = eng.x.clone()
            else:
                out = torch.empty(eng.x.shape, dtype=torch.uint8, device=dev)
                _lib.check(_lib.load().advs_to_uint8(eng.x.data_ptr(), out.data_ptr(), eng.x.numel(), 1 if clamp else 0,
                                                     eng.stream.cuda_stream), "to_uint8")
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        model.train()                     # the reference leaves the model in train mode (ddim.py:95)
        return out


class _Loop:
    def __init__(self, eng, mode, cfg, coef, tseq):
        self.eng, self.coef, self.tseq, self.nsteps = eng, coef, tseq, tseq.numel()
        self.counter = torch.zeros((1,), dtype=torch.int32, device=eng.dev)
        fwd = eng.plan(mode)
        self.plan = Plan(eng.stream)
        self.plan.ops = list(fwd.ops)
        self.plan.keep = [fwd.keep, coef, tseq, self.counter]
        eps = eng.eps_u if mode == "uncond" else eng.eps_c
        eps_u = eng.eps_u if mode == "cfg" else None
        self.plan.add(_lib.load().advs_ddim_step, ptr(eng.x), ptr(eps), ptr(eps_u), cfg, 0, ptr(coef), ptr(tseq),
                      self.nsteps, ptr(self.counter), ptr(eng.t), eng.B, eng.x[0].numel(), 1)
        self.captured = False

    def run(self):
        eng = self.eng
        first_t = int(self.tseq[0].item())
        self.counter.zero_()
        eng.t.fill_(first_t)
        if eng.model.use_graph and not self.captured:
            x0 = eng.x.clone()
            self.plan.run_eager()
            eng.stream.synchronize()
            self.plan.capture()
            self.captured = True
            eng.x.copy_(x0)
            self.counter.zero_()
            eng.t.fill_(first_t)
        for _ in range(self.nsteps):
            self.plan.run()
