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


_STUB = '''
import json, os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["LOCAL_RANK"] == str(rank)
if "--fail" in sys.argv and rank == 1:
    sys.exit(3)
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if rank != 0:
    print("noise from rank", rank)          # must not reach the parent's stdout
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": t.item(), "argv": sys.argv[1:]}))
dist.barrier()
dist.destroy_process_group()
'''


def test_bench_self_launch_two_ranks(tmp_path, capsys):
    """bench.py --gpus N without a launcher starts its own N ranks: rendezvous on 127.0.0.1, RANK /
    LOCAL_RANK / WORLD_SIZE exported, rank 0's line relayed alone on stdout, exit codes propagated.  The children
    here are a gloo stand-in for bench.py's GPU body."""
    import json
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB)
    assert bench.self_launch(2, ["--gpus", "2"], script=str(stub)) == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1
    line = json.loads(out[0])
    assert line == {"n_gpus": 2, "sum": 3.0, "argv": ["--gpus", "2"]}
    env = bench.child_env(1, 4, 1234, base={})
    assert env["RANK"] == "1" and env["WORLD_SIZE"] == "4" and env["MASTER_PORT"] == "1234" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_self_launch_propagates_failure(tmp_path):
    """Rank 1 dies before the rendezvous; rank 0 then sits in init_process_group with the DEFAULT (30 min) timeout.  The parent
    polls every child, so it must end the run with rank 1's code within seconds and take rank 0 down itself."""
    import time
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB)
    t0 = time.time()
    assert bench.self_launch(2, ["--fail"], script=str(stub)) == 3
    assert time.time() - t0 < 60


def _run_bench_stub(extra, launcher=False, ranks=2):
    """bench.py's REAL control path at N = 2 on the CPU: ADVS_BENCH_BACKEND=gloo, ADVS_BENCH_STUB=1 (host stand-ins for the
    sampler and the attack shard; rendezvous, barriers, max-over-ranks timing, gather and JSON are the shipped code)."""
    import json
    import subprocess
    env = dict(os.environ, ADVS_BENCH_BACKEND="gloo", ADVS_BENCH_STUB="1", ADVS_BENCH_PG_TIMEOUT_S="60")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    bench_py = os.path.join(ROOT, "bench.py")
    args = ["--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--batch", "5"] + extra
    if launcher:                              # the driver's N > 1 command form
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + os.getpid() % 200), bench_py] + args
    else:                                     # ... and the launcher-less form (bench.py starts its own ranks)
        cmd = [sys.executable, bench_py] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("launcher", [False, True])
def test_bench_two_rank_control_path_gloo(launcher):
    line = _run_bench_stub([], launcher=launcher)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["rccl_ranks"] == 2 and line["collective"]["ranks_seen"] == 2 and line["collective"]["backend"] == "gloo"
    per = line["collective"]["per_rank_elapsed_s"]
    assert len(per) == 2 and abs(max(per) * 1e3 / 2 - line["ms_per_step"]) < 1e-6          # MAX over ranks is what is reported
    assert line["ms_per_step"] >= 20.0                                                        # rank 1 sleeps 20 ms per pass
    assert abs(line["value"] - 2 * 5 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6              # whole-job images / time
    pa = line["pipeline_attack"]                                                              # config 3's loop rides along at N > 1
    assert pa["images"] == 10 and pa["gathered_records"] == 10 and 0.0 <= pa["asr"] <= 1.0 and len(pa["per_rank_s"]) == 2
    assert "roofline" not in line and "cpu_baseline" not in line                              # rank-0, N = 1 objects only


def test_bench_two_rank_attack_pipeline_gloo():
    line = _run_bench_stub(["--pipeline", "attack"])
    cfg = line["config"]
    assert cfg["pipeline"] == "attack" and line["rccl_ranks"] == 2 and "pipeline_attack" not in line
    assert cfg["asr"] == 0.0 and abs(cfg["psnr"] - (10.0 + 4.5)) < 1e-6                       # the stub's records, gathered in global order


def test_bench_rejects_mismatched_world(monkeypatch):
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(os.environ, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True)
    assert r.returncode != 0 and "must agree" in r.stderr
