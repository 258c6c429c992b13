"""N > 1 path on CPU: two gloo ranks exercise the rank sharding, the gather in global order, the max-over-ranks timing
reduction of bench.py and the bucketed gradient all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests._common import join_all
from worddiffusion_amd.dist import GradAllReducer, gather_rows, max_over_ranks, shard_range, sharded_sampling


def test_shard_range_covers_disjointly():
    for n in (0, 1, 7, 64, 339, 512):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                s, c = shard_range(n, r, world)
                got.extend(range(s, s + c))
            assert got == list(range(n))
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


class FakeDiffusion:
    """Stands in for Diffusion.sampling (which needs the GPU): row g of the result depends on (seed, global index,
    word, label) only - exactly the contract the device Philox stream gives."""

    def sampling(self, model, vae, n, x_text, labels, args, seed=0, sample_offset=0, **kw):
        rows = []
        for i in range(n):
            g = sample_offset + i
            gen = torch.Generator().manual_seed(seed * 1000003 + g)
            rows.append(torch.randn(4, 2, 3, generator=gen) + len(x_text[i]) + float(labels[i]))
        return torch.stack(rows)


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        words = ["w" * (1 + i % 5) for i in range(n_total)]
        labels = torch.arange(n_total) % 7
        out, (start, count) = sharded_sampling(FakeDiffusion(), None, None, words, labels, None, seed=5)
        ref = FakeDiffusion().sampling(None, None, n_total, words, labels, None, seed=5, sample_offset=0)
        ok_gather = torch.equal(out, ref)
        s2, c2 = shard_range(n_total, rank, world)
        ok_shard = (start, count) == (s2, c2)
        # uneven gather helper
        loc = torch.full((c2, 2), float(rank))
        allr = gather_rows(loc, n_total)
        ok_rows = allr.shape[0] == n_total and float(allr[s2:s2 + c2].mean() if c2 else rank) == float(rank)
        # bench timing reduction
        mx = max_over_ranks(1.0 + rank)
        # gradient all-reduce (mean) in buckets
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
        for i, p in enumerate(m.parameters()):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        red = GradAllReducer(list(m.parameters()), bucket_mb=16 * 4 / (1024 * 1024))  # tiny buckets -> several
        nb = red.num_buckets
        red.allreduce()
        exp = (1 + world) / 2.0
        ok_grad = all(torch.allclose(p.grad, torch.full_like(p, exp * (i + 1))) for i, p in enumerate(m.parameters()))
        q.put((rank, ok_gather, ok_shard, ok_rows, mx, nb, ok_grad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 8])
def test_two_rank_sharded_sampling_and_grad_allreduce(n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=120) for _ in procs]
    finally:
        join_all(procs, 60)
    for rank, ok_gather, ok_shard, ok_rows, mx, nb, ok_grad in res:
        assert ok_gather and ok_shard and ok_rows and ok_grad, (rank, ok_gather, ok_shard, ok_rows, ok_grad)
        assert mx == 2.0 and nb >= 2


def test_bench_gpus_flag_launches_that_many_ranks():
    """``python bench.py --gpus 2`` with no launcher around it starts two ranks by itself (fresh child processes, torchrun
    environment) and rank 0 prints one JSON line with ``n_gpus: 2``.  ``--launch-check`` keeps it to the control flow
    (rendezvous, barrier, max over ranks) so that it runs without a GPU; the GPU form is tests/test_gpu_dist.py."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["launch_check"] is True and line["max_over_ranks"] == 2.0
    # the multi-GPU training leg (BASELINE configs[3]) runs in fresh per-rank children with a wall-clock limit: the rehearsal
    # children rendezvous on their own port and rank 0's object is merged into the line ...
    assert line["train_step"] == {"launch_check": True, "n_gpus": 2, "ms_per_step": None}
    # ... and children that hang cost `train_step: {"error": "timeout"}`, not the line
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--train-timeout", "3"],
                       env=dict(env, WDIFF_BENCH_TEST_HANG="1"), capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["train_step"]["error"] == "timeout"
    # under a launcher whose world size disagrees with --gpus the script refuses instead of printing a wrong n_gpus
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
