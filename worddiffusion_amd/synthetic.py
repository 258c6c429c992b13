"""Deterministic synthetic weights and inputs.

There are no trained checkpoints offline, and a freshly
constructed reference UNet outputs exactly zero because of its zero-initialised
convolutions (reference ``unet.py:152-158,620-622,375-379,1457``).  Every
parity test, the smoke run and the benchmark therefore fill *all* entries of a
``state_dict`` from a generator that depends only on ``(key, shape, seed)``:
the golden-vector script fills the reference model with it, the tests fill the
oracle and the HIP model with it, and nothing heavier than a seed has to be
committed.

numpy's legacy ``RandomState`` stream is frozen by numpy's compatibility
policy, so the values are reproducible across machines.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

def synthetic_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """Value of state_dict entry ``key`` for ``seed`` (fp32 numpy array)."""
    rs = np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    n = int(np.prod(shape)) if len(shape) else 1
    z = rs.standard_normal(n).astype(np.float32).reshape(shape)
    if key.endswith("weight") and len(shape) == 1:
        # the only 1-D weights are GroupNorm / LayerNorm scales: ~ 1 + 0.1 z
        return (1.0 + 0.1 * z).astype(np.float32)
    if key.endswith("bias"):
        return (0.05 * z).astype(np.float32)
    if "embedding.weight" in key or key.startswith("label_emb"):
        return (0.5 * z).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (z / np.sqrt(fan_in)).astype(np.float32)
    return (0.05 * z).astype(np.float32)


def synthetic_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(synthetic_tensor(k, tuple(s), seed)) for k, s in shapes}


def fill_module_(module: torch.nn.Module, seed: int = 0) -> torch.nn.Module:
    """Overwrite every parameter/buffer of ``module`` in place (zero-init ones too)."""
    sd = module.state_dict()
    new = {k: torch.from_numpy(synthetic_tensor(k, tuple(v.shape), seed)).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


def synthetic_inputs(batch: int, seed: int = 2, hw=(8, 32), in_ch: int = 4, num_classes: int = 339,
                     max_len: int = 10, phosc_len: int = 0, t_max: int = 1000):
    """Inputs of SURVEY.md §8(d) config 2: x~N(0,1), word ids for a random
    length 1..max_len in [1,52] padded with 52, writer ids uniform."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((batch, in_ch, hw[0], hw[1])).astype(np.float32)
    ctx = np.full((batch, max_len), 52, dtype=np.int64)
    for b in range(batch):
        n = int(rs.randint(1, max_len + 1))
        ctx[b, :n] = rs.randint(1, 53, size=n)
    y = rs.randint(0, num_classes, size=batch).astype(np.int64)
    t = rs.randint(1, t_max, size=batch).astype(np.int64)
    out = dict(x=torch.from_numpy(x), context=torch.from_numpy(ctx), y=torch.from_numpy(y),
               t=torch.from_numpy(t))
    if phosc_len:
        out["phosc"] = torch.from_numpy(rs.randint(0, 3, size=(batch, phosc_len)).astype(np.int64))
    return out
