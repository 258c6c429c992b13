"""CPU oracle for the DDPM process around the UNet (a-T).

TEST INFRASTRUCTURE - not part of the product (see ``oracle/unet_oracle.py`` for who may import).

Restates ``Diffusion`` / ``EMA`` / ``label_padding`` of the reference's ``train.py:42-52,140-251`` with
plain fp32 torch ops in the same order, so tables and updates are bit-comparable.
Pinned by ``tests/golden/primitives.npz`` (schedules, label_padding), ``ddpm_traj.npz`` (an 8-step
reverse trajectory of the reference with recorded noise, ``noise_images``), ``train_step.npz``, and the variant
scripts' loops: ``ddpm_traj_phosc_{small,full}.npz`` (PHOSC-conditioned), ``ddpm_traj_modcond.npz`` +
``primitives_modcond.npz`` (trainModifyCondition.py: s_id = ones, '_' alphabet, T = 600).
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


C_CLASSES_UNDERSCORE = C_CLASSES + "_"  # trainModifyCondition.py:68


def label_padding_underscore(word: str, num_tokens: int = NUM_TOKENS, max_len: int = MAX_CHARS) -> List[int]:
    """trainModifyCondition.py:166-180: ``labels.replace(" ", "_")``, index in the 53-class alphabet + num_tokens, right-padded
    with PAD_TOKEN (52) - '_' is id 53, so the models of that script have vocab_size 54.  Pinned by
    ``tests/golden/primitives_modcond.npz`` (the reference function's own outputs)."""
    ll = [C_CLASSES_UNDERSCORE.index(c) + num_tokens for c in word.replace(" ", "_")]
    return ll + [PAD_TOKEN] * (max_len - len(ll))


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


def sampling3_calls_model(i: int, noise_steps: int, epoch: int = 0) -> bool:
    """The predicate of regenerateFromtrain2.py:536 (non-fullSampling): the UNet is evaluated when
    ``i%100==0 or i%5==0 or i==T or i==T-1 or (epoch>3 and i%25==0) or (epoch>5 and i%15==0) or (epoch>10 and i%10==0) or
    epoch>50==0`` - the last clause is the chained comparison ``epoch > 50 and 50 == 0`` (always False), the epoch clauses
    are implied by ``i%5==0``.  Restated literally."""
    return bool(i % 100 == 0 or i % 5 == 0 or i == noise_steps or i == noise_steps - 1 or (epoch > 3 and i % 25 == 0) or
                (epoch > 5 and i % 15 == 0) or (epoch > 10 and i % 10 == 0) or (epoch > 50 == 0))


def sampling3(model: Callable, x_T: torch.Tensor, noise_steps: int, epoch: int = 0, full_sampling: bool = False,
              noises: Optional[Sequence[torch.Tensor]] = None, record: Optional[list] = None):
    """regenerateFromtrain2.py:521-618 (``sampling3``): the predicted noise is refreshed only on the steps of
    ``sampling3_calls_model`` and reused in between; without ``fullSampling`` the update drops the noise term
    (``x = 1/sqrt(alpha) * (x - (1-alpha)/sqrt(1-alpha_hat) * eps)``, :618).  PINNED by ``tests/golden/ddpm_traj_sampling3.npz``:
    trajectories recorded from that script's own ``Diffusion.sampling3`` (``oracle/make_golden_sampling3.py`` imports it with
    empty stand-ins for the modules its import lines name and the sampler never touches).  ``record`` receives the x handed
    to the model on every model call."""
    beta, alpha, alpha_hat = schedule(noise_steps)
    x = x_T
    eps_hat = None
    k = 0
    calls = 0
    for i in reversed(range(1, noise_steps)):
        t = (torch.ones(x.shape[0]) * i).long()
        if full_sampling or sampling3_calls_model(i, noise_steps, epoch):
            if record is not None:
                record.append(x.clone())
            eps_hat = model(x, t)
            calls += 1
        z = None
        if full_sampling and i > 1:
            z = noises[k]
            k += 1
        if full_sampling:
            x = reverse_step(beta, alpha, alpha_hat, x, eps_hat, i, z)
        else:
            a, ah = alpha[t][:, None, None, None], alpha_hat[t][:, None, None, None]
            x = 1 / torch.sqrt(a) * (x - ((1 - a) / (torch.sqrt(1 - ah))) * eps_hat)
    return x, calls


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
