"""Multi-rank path on CPU: world_size-2 gloo run of the shard + all-gather + reduction logic."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from advshadow_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_every_image_once():
    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            ids = []
            for r in range(world):
                lo, hi = parallel.shard_bounds(total, r, world)
                ids += list(range(lo, hi))
            assert ids == list(range(total))


def test_image_noise_is_indexed_globally():
    full = parallel.image_noise(range(6), (3, 4, 4))
    part = parallel.image_noise(range(4, 6), (3, 4, 4))
    assert torch.equal(full[4:], part)


def _worker(rank, world, port, total, out):
    sys.path.insert(0, ROOT)
    import advshadow_amd  # noqa: F401
    from advshadow_amd import parallel as par
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = par.shard_bounds(total, rank, world)
    ids = torch.arange(lo, hi)
    pred = (ids * 7 % 37).to(torch.int32)                    # stand-ins for per-image results
    psnr, ssim = 20.0 + ids.float() * 0.5, 1.0 / (1.0 + ids.float())
    p, q, s = par.gather_results(pred, psnr, ssim, total)
    if rank == 0:
        torch.save({"pred": p, "psnr": q, "ssim": s}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_gather_results_two_ranks_gloo(tmp_path, total):
    out = str(tmp_path / "r.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, total, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=True)
    ids = torch.arange(total)
    assert torch.equal(r["pred"], (ids * 7 % 37).to(torch.int32))
    assert torch.allclose(r["psnr"], 20.0 + ids.float() * 0.5) and torch.allclose(r["ssim"], 1.0 / (1.0 + ids.float()))
    m = parallel.reduce_metrics(r["pred"], ids % 37, r["psnr"], r["ssim"])
    assert m["n"] == total and 0.0 <= m["asr"] <= 1.0
