"""Batch sharding of the attack loop across the GPUs of a node and the final metric exchange.

Every image's trajectory depends only on its own x_T, label and the shared weights, so the global
batch splits contiguously by image id with no data-path collective; the only exchange is one
all-gather of the per-image results (12 B per image) over RCCL/xGMI (``backend="nccl"`` on ROCm),
or gloo on CPU in the tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous split of ``total`` image ids: rank r owns [lo, hi)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def image_noise(global_ids, shape, seed=1234, device="cpu"):
    """x_T per GLOBAL image id (so 1-GPU and N-GPU runs sample identical images)."""
    out = torch.empty((len(global_ids),) + tuple(shape), dtype=torch.float32)
    for i, gid in enumerate(global_ids):
        g = torch.Generator().manual_seed(seed * 1000003 + int(gid))
        out[i] = torch.randn(shape, generator=g)
    return out.to(device)


def gather_results(pred, psnr, ssim, total):
    """All-gather per-image records {pred: int32, psnr: f32, ssim: f32} (12 bytes) from every rank into global
    order: one collective on an int32 [cap, 3] buffer whose last two columns carry the floats' bit patterns.
    Shards may be ragged: each rank pads to the largest shard and the pad is dropped.  The buffer lives where the
    backend can reach it: on the GPU for nccl (RCCL over xGMI), on the host for gloo (CPU tests, rehearsals)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return pred, psnr, ssim
    world = dist.get_world_size()
    cap = max(shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world))
    out_dev = pred.device
    dev = out_dev if dist.get_backend() == "nccl" else torch.device("cpu")
    rec = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    n = pred.numel()
    rec[:n, 0] = pred.to(dev, torch.int32)
    rec[:n, 1] = psnr.to(dev, torch.float32).view(torch.int32)
    rec[:n, 2] = ssim.to(dev, torch.float32).view(torch.int32)
    bufs = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(bufs, rec)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        parts.append(bufs[r][:hi - lo])
    allr = torch.cat(parts, 0).to(out_dev)
    return allr[:, 0].contiguous(), allr[:, 1].contiguous().view(torch.float32), allr[:, 2].contiguous().view(torch.float32)


def reduce_metrics(pred, labels, psnr, ssim):
    """ASR = fraction mispredicted (ASR_fast.py:118-123); mean PSNR / SSIM (PSNR_SSIM_fast.py:53-54)."""
    pred, labels = pred.to(torch.int64).cpu(), torch.as_tensor(labels).to(torch.int64).cpu()
    return {"asr": float((pred != labels).float().mean()), "psnr": float(psnr.double().mean()),
            "ssim": float(ssim.double().mean()), "n": int(pred.numel())}
