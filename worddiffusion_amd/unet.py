"""``UNetModel`` - drop-in for the reference's base denoiser (``unet.py:1096-1836``).

Same constructor kwargs, same ``state_dict`` keys/shapes (including the never-called ``res.*`` and
``wrd_proj.*``), same forward signature; the arithmetic runs in HIP kernels (``engine.py`` / ``csrc/``).

Semantics worth knowing: this variant has NO spatial self-attention - each
transformer block is cross-attn(attn1, norm2) + cross-attn(attn2, norm2) + GEGLU-FF(norm3) over the text tokens
(``unet.py:337-345``); ``forward``'s second positional argument is ``wrdChrWrStyl`` (ignored unless
``args.wrdChrWrStyl``, which is out of scope), so ``timesteps/context/y`` are passed by keyword as in
``trainModifyCondition.py:584``.
"""
from __future__ import annotations

import torch.nn as nn

from .layers import ResBlockConditionalParams
from .model import UNetBase, _arg


class UNetModel(UNetBase):
    variant = "base"

    def _extra_heads_before_label(self):
        self.wrd_proj = nn.Linear(4096, 320)  # unet.py:1243 - in every checkpoint, used only when wrdChrWrStyl==1

    def _extra_heads_after_out(self):
        self.res = ResBlockConditionalParams()  # unet.py:1472 - dead parameters kept for checkpoint parity

    def forward(self, x, wrdChrWrStyl=None, original_images=None, timesteps=None, context=None, y=None,
                charContextImages=None, original_context=None, or_images=None, mix_rate=None, **kwargs):
        """Predicted noise [B, out_channels, H, W] (``unet.py:1499``; returns ``h`` as at ``:1821/:1836``)."""
        self._check_common(x, timesteps, mix_rate)
        self._check_context(context)
        if self.num_classes is not None:
            if _arg(self.args, "imgConditioned", 0) == 1:
                raise NotImplementedError("args.imgConditioned=1 drops the writer embedding (unet.py:1578-1579)")
            assert y is not None and tuple(y.shape) == (x.shape[0],), "y must be [B] (unet.py:1555)"
        return self._run(x, timesteps, context, y)
