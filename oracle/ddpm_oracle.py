"""CPU oracle for the DDPM process around the UNet (a-T).

TEST INFRASTRUCTURE - not part of the product (see ``oracle/unet_oracle.py`` for who may import).

Restates ``Diffusion`` / ``EMA`` / ``label_padding`` of the reference's ``train.py:42-52,140-251`` with
plain fp32 torch ops in the same order, so tables and updates are bit-comparable.
Pinned by ``tests/golden/primitives.npz`` (schedules, label_padding), ``ddpm_traj.npz`` (an 8-step
reverse trajectory of the reference with recorded noise, ``noise_images``) and ``train_step.npz``.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

C_CLASSES = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"  # train.py:30
MAX_CHARS = 10  # train.py:28
PAD_TOKEN = 52  # train.py:73-77 (tok == False)
NUM_TOKENS = 1


def label_padding(word: str, num_tokens: int = NUM_TOKENS, max_len: int = MAX_CHARS) -> List[int]:
    """train.py:42-52: letter index + num_tokens, right-padded with PAD_TOKEN to max_len."""
    ll = [C_CLASSES.index(c) + num_tokens for c in word]
    ll = ll + [PAD_TOKEN] * (max_len - len(ll))
    return ll


def schedule(noise_steps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02):
    """train.py:180-188: beta = linspace, alpha = 1 - beta, alpha_hat = cumprod(alpha) (fp32)."""
    beta = torch.linspace(beta_start, beta_end, noise_steps)
    alpha = 1.0 - beta
    alpha_hat = torch.cumprod(alpha, dim=0)
    return beta, alpha, alpha_hat


def noise_images(alpha_hat: torch.Tensor, x: torch.Tensor, t: torch.Tensor, eps: torch.Tensor):
    """train.py:190-194 with the noise passed in."""
    a = torch.sqrt(alpha_hat[t])[:, None, None, None]
    b = torch.sqrt(1 - alpha_hat[t])[:, None, None, None]
    return a * x + b * eps


def reverse_step(beta, alpha, alpha_hat, x, eps_hat, i: int, z: Optional[torch.Tensor]):
    """train.py:229-236 for integer step i (same i for the whole batch)."""
    n = x.shape[0]
    t = (torch.ones(n) * i).long()
    a = alpha[t][:, None, None, None]
    ah = alpha_hat[t][:, None, None, None]
    b = beta[t][:, None, None, None]
    noise = z if i > 1 else torch.zeros_like(x)
    return 1 / torch.sqrt(a) * (x - ((1 - a) / (torch.sqrt(1 - ah))) * eps_hat) + torch.sqrt(b) * noise


def sampling(model: Callable, x_T: torch.Tensor, noises: Sequence[torch.Tensor], noise_steps: int,
             record: Optional[list] = None):
    """train.py:221-236: i = T-1 .. 1; ``model(x, t)`` returns predicted noise; ``noises[k]`` is the
    k-th ``randn_like`` draw (one per step with i > 1).  Returns x_0 (before the 1/0.18215 scale)."""
    beta, alpha, alpha_hat = schedule(noise_steps)
    x = x_T
    k = 0
    for i in reversed(range(1, noise_steps)):
        if record is not None:
            record.append(x.clone())
        t = (torch.ones(x.shape[0]) * i).long()
        eps_hat = model(x, t)
        z = None
        if i > 1:
            z = noises[k]
            k += 1
        x = reverse_step(beta, alpha, alpha_hat, x, eps_hat, i, z)
    return x


def ema_update(ema_sd: Dict[str, torch.Tensor], sd: Dict[str, torch.Tensor], beta: float, keys) -> None:
    """train.py:151-159: old * beta + (1 - beta) * new, parameters only."""
    for k in keys:
        ema_sd[k] = ema_sd[k] * beta + (1 - beta) * sd[k]


def adamw_step(p, g, m, v, step: int, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor update (defaults of train.py:405)."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v
