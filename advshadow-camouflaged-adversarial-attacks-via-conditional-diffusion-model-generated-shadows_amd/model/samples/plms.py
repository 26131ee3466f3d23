"""Drop-in for ``model/samples/plms.py::PLMSDiffusion``: DDIM-style steps whose eps is a linear-multistep
combination of the last four predictions (one extra forward on the first step).  Per step: UNet
forward(s) (replayed plan), ``advs_plms_combine``, then the DDIM update kernel on the combined eps.
"""
import torch

from ... import _lib
from ...engine import ptr
from .ddim import BaseDiffusion, guidance_mode


class PLMSDiffusion(BaseDiffusion):
    def __init__(self, noise_steps=1000, sample_steps=500, beta_start=1e-4, beta_end=2e-2, img_size=256, device="cpu"):
        super().__init__(noise_steps, beta_start, beta_end, img_size, device)
        self.sample_steps, self.eta = sample_steps, 0
        ts = torch.arange(0, noise_steps, noise_steps // sample_steps).long() + 1
        ts = reversed(torch.cat((torch.tensor([0], dtype=torch.long), ts)))
        self.time_step = list(zip(ts[:-1], ts[1:]))

    @torch.no_grad()
    def sample(self, model, n, labels=None, cfg_scale=None, save_path=None, x_T=None, return_float=False):
        dev = next(model.parameters()).device
        lib = _lib.load()
        model.eval()
        mode = guidance_mode(labels, cfg_scale, "plms.py:88")
        cfg = float(cfg_scale or 0.0)
        eng = model.engine(n)
        cur_t = torch.stack([a for a, _ in self.time_step])
        prev_t = torch.stack([b for _, b in self.time_step])
        a_t, a_p = self.alpha_hat[cur_t], self.alpha_hat[prev_t]
        coef = torch.stack([a_t, a_p, torch.zeros_like(a_t)], 1).contiguous().to(dev)     # eta = 0 -> c1 = 0
        tseq = cur_t.to(torch.int64).to(dev)
        nsteps = len(self.time_step)
        if x_T is None:
            x_T = torch.randn((n, 3, self.img_size, self.img_size))
        cur = torch.cuda.current_stream(dev)
        eng.stream.wait_stream(cur)
        s = eng.stream.cuda_stream
        per = eng.x[0].numel()
        with torch.cuda.stream(eng.stream):
            eng.x.copy_(x_T.to(dev, torch.float32), non_blocking=True)
            if labels is not None:
                eng.labels.copy_(labels.to(dev, torch.int64), non_blocking=True)
            counter = torch.zeros((1,), dtype=torch.int32, device=dev)
            scratch_counter = torch.zeros((1,), dtype=torch.int32, device=dev)
            t_scratch = torch.zeros_like(eng.t)
            prime = torch.empty_like(eng.x)
            olds = []                                       # most recent first
            eps = eng.eps_u if mode == "uncond" else eng.eps_c
            eps_u = eng.eps_u if mode == "cfg" else None
            for k in range(nsteps):
                eng.t.fill_(int(cur_t[k]))
                eng.run(mode)
                guided = torch.empty_like(eng.x)
                if not olds:
                    # first step: provisional DDIM step with the plain eps, then a second forward at p_t
                    x_save = eng.x.clone()
                    _lib.check(lib.advs_plms_combine(ptr(eps), ptr(eps_u), cfg, 0, 0, 0, 0, 0, ptr(guided), ptr(prime), n * per, s))
                    scratch_counter.fill_(k)
                    _lib.check(lib.advs_ddim_step(ptr(eng.x), ptr(prime), 0, 0.0, 0, ptr(coef), ptr(tseq), nsteps,
                                                  ptr(scratch_counter), ptr(t_scratch), n, per, 1, s))
                    eng.t.fill_(int(prev_t[k]))
                    # the reference's second forward is conditional WITHOUT guidance (plms.py:90-93)
                    eng.run("uncond" if mode == "uncond" else "cond")
                    nxt = (eng.eps_u if mode == "uncond" else eng.eps_c).clone()
                    _lib.check(lib.advs_plms_combine(ptr(guided), 0, 0.0, ptr(nxt), 0, 0, 0, 0, 0, ptr(prime), n * per, s))
                    eng.x.copy_(x_save)
                else:
                    o = olds + [None] * 3
                    _lib.check(lib.advs_plms_combine(ptr(eps), ptr(eps_u), cfg, 0, ptr(o[0]), ptr(o[1]), ptr(o[2]),
                                                     min(len(olds), 3), ptr(guided), ptr(prime), n * per, s))
                counter.fill_(k)
                _lib.check(lib.advs_ddim_step(ptr(eng.x), ptr(prime), 0, 0.0, 0, ptr(coef), ptr(tseq), nsteps, ptr(counter),
                                              ptr(t_scratch), n, per, 1, s))
                olds = [guided] + olds[:2]
            if return_float:
                out = eng.x.clone()
            else:
                out = torch.empty(eng.x.shape, dtype=torch.uint8, device=dev)
                _lib.check(lib.advs_to_uint8(eng.x.data_ptr(), out.data_ptr(), eng.x.numel(), 0, s))
        cur.wait_stream(eng.stream)
        out.record_stream(cur)
        model.train()
        return out
