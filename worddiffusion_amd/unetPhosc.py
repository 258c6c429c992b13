"""``UNetModelPhosc`` - drop-in for the reference's PHOSC-conditioned denoiser
(``unetPhosc.py:751-1159``; ``unetPhosc2.py`` is the same arithmetic plus file logging).

Standard transformer blocks: self-attn(norm1) + cross-attn(norm2) + GEGLU-FF(norm3) (``unetPhosc.py:241-246``).
With ``args.phosc == 1`` or ``args.phos == 1`` the integer PHOSC vector is embedded through the *character*
table and concatenated to the text context (``unetPhosc.py:1120-1130``); the positional encoding is skipped for
sequences longer than ``max_seq_len`` (``unetPhosc.py:726-729``).
"""
from __future__ import annotations

from .model import UNetBase, _arg


class UNetModelPhosc(UNetBase):
    variant = "phosc"

    def forward(self, x, phoscLabels=None, timesteps=None, context=None, y=None, mix_rate=None, **kwargs):
        self._check_common(x, timesteps, mix_rate)
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"  # unetPhosc.py:1079-1081
        if y is not None and y.shape[0] != x.shape[0]:
            y = y[: x.shape[0]]  # unetPhosc.py:1089-1090
        phosc = None
        if context is not None and (_arg(self.args, "phosc", 0) == 1 or _arg(self.args, "phos", 0) == 1):
            if phoscLabels is None:
                raise ValueError("args.phosc/phos is set: phoscLabels [B, n] is required (unetPhosc.py:1120-1123)")
            phosc = phoscLabels.int()
        return self._run(x, timesteps, context, y, phosc)
