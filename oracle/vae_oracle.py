"""CPU restatement (plain torch, fp32/fp64) of the decoder of the Stable-Diffusion-v1.5 ``AutoencoderKL``.

TEST INFRASTRUCTURE - not part of the product.  Only ``tests/`` may import it.

**Parity unpinned.**  The algorithm lives in a third-party dependency that is absent here: ``diffusers`` (the reference's
``requirment.txt`` pins ``diffusers==0.30.x``; call sites ``train.py:415`` ``AutoencoderKL.from_pretrained(...,
subfolder="vae")`` and ``train.py:239-247`` ``vae.decode(latents).sample``), and no VAE weights exist offline
(SURVEY.md section 8c).  This file restates the published forward of ``AutoencoderKL.decode`` for the SD-v1.5 config:

  post_quant_conv (1x1) -> Decoder.conv_in (3x3) -> UNetMidBlock2D [ResnetBlock2D, Attention, ResnetBlock2D]
  -> 4 x UpDecoderBlock2D [3 x ResnetBlock2D (+ Upsample2D: nearest x2, 3x3 conv)] -> GroupNorm -> SiLU -> conv_out (3x3)

  ResnetBlock2D (temb None): x + conv2(silu(norm2(conv1(silu(norm1(x))))))   (1x1 conv_shortcut on x when channels change)
  Attention: x + to_out(softmax(q k^T / sqrt(C)) v) over the h*w positions of group_norm(x), one head of C channels
  every GroupNorm: 32 groups, eps 1e-6

State-dict keys are the ``diffusers`` ones (``decoder.up_blocks.2.resnets.0.conv_shortcut.weight`` ...).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def _gn(sd, k, x):
    return F.group_norm(x, 32, sd[k + ".weight"], sd[k + ".bias"], eps=1e-6)


def _conv(sd, k, x, pad):
    return F.conv2d(x, sd[k + ".weight"], sd[k + ".bias"], padding=pad)


def _resnet(sd, k, x):
    h = _conv(sd, k + ".conv1", F.silu(_gn(sd, k + ".norm1", x)), 1)
    h = _conv(sd, k + ".conv2", F.silu(_gn(sd, k + ".norm2", h)), 1)
    if k + ".conv_shortcut.weight" in sd:
        x = _conv(sd, k + ".conv_shortcut", x, 0)
    return x + h


def _attention(sd, k, x):
    b, c, h, w = x.shape
    t = _gn(sd, k + ".group_norm", x).reshape(b, c, h * w).transpose(1, 2)
    q = F.linear(t, sd[k + ".to_q.weight"], sd[k + ".to_q.bias"])
    kk = F.linear(t, sd[k + ".to_k.weight"], sd[k + ".to_k.bias"])
    v = F.linear(t, sd[k + ".to_v.weight"], sd[k + ".to_v.bias"])
    p = torch.softmax(q @ kk.transpose(1, 2) * (c ** -0.5), dim=-1)
    o = F.linear(p @ v, sd[k + ".to_out.0.weight"], sd[k + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(b, c, h, w)


def vae_decode(sd: Dict[str, torch.Tensor], z: torch.Tensor, block_out_channels: Sequence[int] = (128, 256, 512, 512),
               layers_per_block: int = 2) -> torch.Tensor:
    """``AutoencoderKL.decode(z).sample`` for a state dict in ``diffusers`` naming (dtype of ``z`` / ``sd``)."""
    x = _conv(sd, "post_quant_conv", z, 0)
    x = _conv(sd, "decoder.conv_in", x, 1)
    x = _resnet(sd, "decoder.mid_block.resnets.0", x)
    x = _attention(sd, "decoder.mid_block.attentions.0", x)
    x = _resnet(sd, "decoder.mid_block.resnets.1", x)
    n = len(block_out_channels)
    for i in range(n):
        for j in range(layers_per_block + 1):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i != n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x, 1)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x))
    return _conv(sd, "decoder.conv_out", x, 1)
