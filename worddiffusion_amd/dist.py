"""One-process-per-GPU data parallelism for the denoising path.

Sampling is embarrassingly parallel: rows of (writer, word) are cut into contiguous shards, every rank runs
``Diffusion.sampling`` on its shard with ``sample_offset`` = global index of its first row (the on-device Philox noise is
keyed by that global index, so results do not depend on the number of ranks) and there is NO collective inside the loop;
an optional final all-gather returns the latents in global order.  This replaces the reference's single-process
``nn.DataParallel`` (``regenerateFromtrain2.py:1118``), which re-broadcast all 161 MB of weights on every forward.

Training needs one exchange per step: a mean all-reduce of the gradients (RCCL over xGMI through
``torch.distributed``'s ``nccl`` backend; ``gloo`` in the CPU tests).  ``GradAllReducer`` packs them into a few large
contiguous fp32 buckets - on a fully connected xGMI node a handful of big collectives beats hundreds of small ones.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(start, count) of the contiguous shard of ``rank``; the first ``n_total % world`` ranks get one extra row."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: Optional[str] = None):
    """torchrun-style initialisation (RANK / WORLD_SIZE / MASTER_* from the environment)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def max_over_ranks(value: float, device="cpu") -> float:
    """MAX all-reduce of a scalar (bench.py: the step time of the slowest rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather row shards produced by ``shard_range`` back into global order (every rank gets the result)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    counts = [shard_range(n_total, r, world)[1] for r in range(world)]
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:c] for o, c in zip(outs, counts)], dim=0)


def sharded_sampling(diffusion, model, vae, words: Sequence[str], labels: torch.Tensor, args, seed: int = 0,
                     gather: bool = True, **kw):
    """Rank-sharded ``Diffusion.sampling`` over rows (words[i], labels[i]).  Returns (rows of this rank or, with
    ``gather``, all rows in global order; (start, count) of this rank)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    n_total = len(words)
    start, count = shard_range(n_total, rank, world)
    if count:
        out = diffusion.sampling(model, vae, count, list(words[start:start + count]), labels[start:start + count], args,
                                 seed=seed, sample_offset=start, **kw)
    else:
        out = None
    if gather and world > 1:
        # ranks with an empty shard still join the collective: they learn the row shape from the others
        shapes = [None] * world
        dist.all_gather_object(shapes, tuple(out.shape[1:]) if out is not None else None)
        trailing = next(sh for sh in shapes if sh is not None)
        if out is None:
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            out = torch.zeros((0,) + tuple(trailing), dtype=torch.float32, device=dev)
        out = gather_rows(out, n_total)
    return out, (start, count)


class GradAllReducer:
    """Mean all-reduce of ``.grad`` over the process group in a few large contiguous buckets (config 4: 40.4 M fp32
    gradients = 161 MB per step).  ``bucket_mb`` defaults to one 64 MiB bucket per call: with 7 xGMI links per GPU a
    direct reduce-scatter + all-gather moves 7 x (S/8) concurrently, so large messages are what fills the links."""

    def __init__(self, params: Sequence[torch.nn.Parameter], bucket_mb: float = 64.0):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_elems = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self._buckets: List[List[torch.nn.Parameter]] = []
        cur, n = [], 0
        for p in self.params:
            if cur and n + p.numel() > self.bucket_elems:
                self._buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self._buckets.append(cur)
        self._flat = [None] * len(self._buckets)

    @property
    def num_buckets(self) -> int:
        return len(self._buckets)

    def allreduce(self) -> None:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        works = []
        for i, bucket in enumerate(self._buckets):
            n = sum(p.numel() for p in bucket)
            dev = bucket[0].device
            if self._flat[i] is None or self._flat[i].device != dev:
                self._flat[i] = torch.empty(n, dtype=torch.float32, device=dev)
            flat = self._flat[i]
            off = 0
            for p in bucket:
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                flat[off:off + p.numel()].copy_(g.reshape(-1))
                off += p.numel()
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for i, bucket in enumerate(self._buckets):
            works[i].wait()
            flat = self._flat[i]
            flat.div_(world)
            off = 0
            for p in bucket:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
