"""Parameter containers with the reference's ``state_dict`` layout.

These modules only *hold* parameters under the names and shapes the reference checkpoints use
(``ckpt.pt`` / ``ema_ckpt.pt``, reference ``train.py:314-316``; key layout: ``unet.py:570-632`` ResBlock,
``:164-183`` CrossAttention, ``:305-317`` BasicTransformerBlock, ``:347-380`` SpatialTransformer,
``:472-551`` Up/Downsample, ``:839-849`` CharacterEncoder).  They have no ``forward``: the arithmetic runs in
the HIP engine (``engine.py``), which reads the tensors and repacks them for the kernels.
Never-used reference entries (``attnc``, ``to_kv``, ``norm1`` in the base model, ``res.*``, ``wrd_proj``)
are kept so that ``load_state_dict(strict=True)`` works in both directions.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - by construction
        raise RuntimeError(f"{type(self).__name__} holds parameters only; the forward runs in the HIP engine")


def _zero(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


def _seq(*mods) -> nn.Sequential:
    return nn.Sequential(*mods)


class ResBlockParams(Holder):
    kind = "res"

    def __init__(self, cin: int, emb_ch: int, cout: int, dropout: float = 0.0):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.in_layers = _seq(nn.GroupNorm(32, cin), nn.SiLU(), nn.Conv2d(cin, cout, 3, padding=1))
        self.emb_layers = _seq(nn.SiLU(), nn.Linear(emb_ch, cout))
        self.out_layers = _seq(nn.GroupNorm(32, cout), nn.SiLU(), nn.Dropout(p=dropout),
                               _zero(nn.Conv2d(cout, cout, 3, padding=1)))
        self.skip_connection = nn.Identity() if cin == cout else nn.Conv2d(cin, cout, 1)


class DownsampleParams(Holder):
    kind = "down"

    def __init__(self, ch: int, cout: int):
        super().__init__()
        self.cin, self.cout = ch, cout
        self.op = nn.Conv2d(ch, cout, 3, stride=2, padding=1)


class UpsampleParams(Holder):
    kind = "up"

    def __init__(self, ch: int, cout: int):
        super().__init__()
        self.cin, self.cout = ch, cout
        self.conv = nn.Conv2d(ch, cout, 3, padding=1)


class CrossAttentionParams(Holder):
    def __init__(self, query_dim: int, context_dim, heads: int, dim_head: int, dropout: float = 0.0):
        super().__init__()
        inner = heads * dim_head
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_kv = nn.Linear(context_dim, inner * 2, bias=False)  # never used by the reference forward
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = _seq(nn.Linear(inner, query_dim), nn.Dropout(dropout))


class GEGLUParams(Holder):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForwardParams(Holder):
    def __init__(self, dim: int, mult: int = 4, dropout: float = 0.0):
        super().__init__()
        inner = int(dim * mult)
        self.net = _seq(GEGLUParams(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim))


class TransformerBlockParams(Holder):
    def __init__(self, dim: int, heads: int, d_head: int, context_dim, dropout: float = 0.0):
        super().__init__()
        self.attn1 = CrossAttentionParams(dim, None, heads, d_head, dropout)
        self.attnc = CrossAttentionParams(dim, None, heads, d_head, dropout)  # dead in the reference too
        self.ff = FeedForwardParams(dim, dropout=dropout)
        self.attn2 = CrossAttentionParams(dim, context_dim, heads, d_head, dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)


class SpatialTransformerParams(Holder):
    kind = "st"

    def __init__(self, ch: int, heads: int, d_head: int, depth: int, context_dim, dropout: float = 0.0):
        super().__init__()
        inner = heads * d_head
        self.ch, self.heads, self.d_head = ch, heads, d_head
        self.norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [TransformerBlockParams(inner, heads, d_head, context_dim, dropout) for _ in range(depth)])
        self.proj_out = _zero(nn.Conv2d(inner, ch, 1))


class WordAttentionParams(Holder):
    def __init__(self, dim: int):
        super().__init__()
        self.linear_query = nn.Linear(dim, dim)
        self.linear_key = nn.Linear(dim, dim)
        self.linear_value = nn.Linear(dim, dim)


def positional_encoding_table(max_seq_len: int, dim: int) -> torch.Tensor:
    """The reference's table (``unet.py:876-882``).  Entry (pos, j) uses the angle pos / 1e4^(j/dim) for EVERY
    column j (sin for even j, cos for odd j) - unlike the textbook encoding, the odd columns do not share the
    exponent of their even neighbour.  Evaluated in Python doubles and rounded once to fp32, as the reference."""
    rows = [[(math.sin if j % 2 == 0 else math.cos)(pos / (10000 ** (j / dim))) for j in range(dim)]
            for pos in range(max_seq_len)]
    return torch.tensor(rows, dtype=torch.float64).float()


class CharacterEncoderParams(Holder):
    def __init__(self, vocab: int, dim: int, max_seq_len: int):
        super().__init__()
        self.embedding = nn.Embedding(vocab, dim)
        self.attention = WordAttentionParams(dim)
        self.embedding_dim, self.max_seq_len = dim, max_seq_len
        # plain attribute like the reference (not a buffer, not in the state_dict: unet.py:849)
        self.positional_encoding = positional_encoding_table(max_seq_len, dim)


class ResBlockConditionalParams(Holder):
    """``self.res = ResBlockConditional(32, 1280, 0.2, 320, use_conv=True, down=True)`` (unet.py:1472):
    built by the reference constructor, present in its checkpoints, never called (``if 0:`` at unet.py:1593)."""

    def __init__(self):
        super().__init__()
        self.in_layers = _seq(nn.GroupNorm(32, 32), nn.SiLU(), nn.Conv2d(32, 320, 3, padding=1))
        self.h_upd = DownsampleParams(32, 32)
        self.x_upd = DownsampleParams(32, 32)
        self.emb_layers = _seq(nn.SiLU(), nn.Linear(1280, 320))
        self.out_layers = _seq(nn.GroupNorm(32, 320), nn.SiLU(), nn.Dropout(p=0.2),
                               _zero(nn.Conv2d(320, 320, 3, padding=1)))
        self.skip_connection = nn.Conv2d(32, 320, 3, padding=1)
