"""Host-side helpers of the backward pass: gather tables and weight layouts that turn the data / weight gradients of the
convolutions and linears into calls of the forward GEMM kernel (``wd_gemm``).

Forward (``engine.conv_gather_table``): out[m] = sum_t W_t . in[g(m, t)].  Hence
  * d in[p]  = sum_t W_t^T . d out[g'(p, t)]   - the same tap-gather GEMM over the planes of d out, with the *inverse*
    table g' (``conv_bwd_table``) and the weights repacked as [C_in][tap][C_out] (``pack_dx_weight``);
  * d W_t[n][c] = sum_m d out[m][n] . in[g(m, t)][c] - a GEMM whose reduction runs over the tokens: rows = output
    channels (planes of d out^T), columns = (tap, c) (planes of the gathered input, transposed by ``wd_transpose_planes``).
Reference: autograd of ``nn.Conv2d`` / ``nn.Linear`` in ``loss.backward()`` (train.py:291).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from .engine import conv_gather_table


def conv_bwd_table(h: int, w: int, mode: str) -> Tuple[np.ndarray, int, int]:
    """int32 [9][h_in*w_in]: for input position p and tap t the output position m with g(m, t) == p, or -1.
    (h, w) is the forward INPUT size; modes 'same' and 'down' (for 'up' use 'same' at (2h, 2w) + 2x2 sum pooling)."""
    fwd, ho, wo = conv_gather_table(h, w, mode)
    if mode == "up":
        raise ValueError("upsample backward = 'same' backward at the upsampled size followed by wd_pool2x2_sum")
    tab = np.full((9, h * w), -1, dtype=np.int32)
    for t in range(9):
        m = np.nonzero(fwd[t] >= 0)[0]
        tab[t, fwd[t, m]] = m  # g(., t) is injective for stride-1 and stride-2 3x3 convolutions
    return tab, ho, wo


def pack_dx_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW conv weight (or [out, in] linear weight) -> [C_in][tap * C_out + n]: the wd_gemm weight of the data gradient."""
    if w.dim() == 2:
        return w.t().contiguous()
    n, c, kh, kw = w.shape
    return w.permute(1, 2, 3, 0).reshape(c, kh * kw * n).contiguous()


def unpack_dw(dw_packed: torch.Tensor, shape) -> torch.Tensor:
    """[N][tap * C + c] (the forward packed layout) -> parameter layout (OIHW or [out, in])."""
    if len(shape) == 2:
        return dw_packed.reshape(shape)
    n, c, kh, kw = shape
    return dw_packed.reshape(n, kh, kw, c).permute(0, 3, 1, 2).contiguous()
