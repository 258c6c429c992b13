"""CPU oracle for the UNet forward.

TEST INFRASTRUCTURE - not part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``worddiffusion_amd``) never does.

This is a restatement, in plain functional fp32 PyTorch on the CPU, of what the
reference's ``unet.UNetModel.forward`` (``unet.py:1499-1836``) and
``unetPhosc.UNetModelPhosc.forward`` (``unetPhosc.py:1068-1159``) compute.  It
works directly on a reference-layout ``state_dict`` (reference key names, OIHW /
[out,in] shapes) and carries no module tree of its own.  Each function cites
the reference lines it follows.

Pinned by: ``tests/golden/fwd_*.npz`` - outputs (and per-block hook outputs) of
the reference's own modules, produced by ``oracle/make_golden.py``;
``tests/test_oracle_golden.py`` checks this file against every one of them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------- a1
def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    """unet.py:96-116: [cos(t f_k), sin(t f_k)], f_k = exp(-ln(max_period) k / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# ------------------------------------------------------------------------------------------- a3
def positional_encoding(max_seq_len: int, dim: int) -> Tensor:
    """unet.py:876-882 (non-standard: the odd entries use exponent (i+1)/d)."""
    pe = torch.zeros(max_seq_len, dim)
    for pos in range(max_seq_len):
        for i in range(0, dim, 2):
            pe[pos, i] = math.sin(pos / (10000 ** (i / dim)))
            pe[pos, i + 1] = math.cos(pos / (10000 ** ((i + 1) / dim)))
    return pe


def character_encoder(sd: Dict[str, Tensor], ids: Tensor, max_seq_len: int, always_pe: bool, pe: Tensor,
                      prefix: str = "word_emb.") -> Tensor:
    """CharacterEncoder + Word_Attention.

    unet.py:851-874 adds PE[:L] unconditionally (``always_pe``); unetPhosc.py:724-733 skips it
    when L > max_seq_len.  Word_Attention (unet.py:823-836): softmax(q k^T) v with NO 1/sqrt(d).
    """
    x = F.embedding(ids.long(), sd[prefix + "embedding.weight"])
    L = x.shape[1]
    if always_pe or L <= max_seq_len:
        x = x + pe[:L]
    q = F.linear(x, sd[prefix + "attention.linear_query.weight"], sd[prefix + "attention.linear_query.bias"])
    k = F.linear(x, sd[prefix + "attention.linear_key.weight"], sd[prefix + "attention.linear_key.bias"])
    v = F.linear(x, sd[prefix + "attention.linear_value.weight"], sd[prefix + "attention.linear_value.bias"])
    scores = torch.softmax(q @ k.transpose(-2, -1), dim=-1)
    return scores @ v


# ------------------------------------------------------------------------------------------- a4
def group_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.group_norm(x.float(), 32, w, b, eps)


def resblock(sd: Dict[str, Tensor], p: str, x: Tensor, emb: Tensor) -> Tensor:
    """ResBlock._forward, unet.py:646-671 with use_scale_shift_norm=False, no up/down.
    GroupNorm32 = 32 groups, eps 1e-5 (unet.py:427-431)."""
    h = F.silu(group_norm(x, sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"], 1e-5))
    h = F.conv2d(h, sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = F.silu(group_norm(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"], 1e-5))
    h = F.conv2d(h, sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    if p + "skip_connection.weight" in sd:
        w = sd[p + "skip_connection.weight"]
        x = F.conv2d(x, w, sd[p + "skip_connection.bias"], padding=w.shape[-1] // 2)
    return x + h


# ------------------------------------------------------------------------------------------- a5
def downsample(sd, p, x):
    """Downsample with conv: unet.py:540-551 (3x3, stride 2, pad 1)."""
    return F.conv2d(x, sd[p + "op.weight"], sd[p + "op.bias"], stride=2, padding=1)


def upsample(sd, p, x):
    """Upsample: nearest x2 then 3x3 conv, unet.py:490-500."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)


# ------------------------------------------------------------------------------------------- a8
def cross_attention(sd, p, x: Tensor, context: Optional[Tensor], heads: int):
    """CrossAttention.forward, unetPhosc.py:176-198 / unet.py:185-279: q,k,v without bias,
    softmax(q k^T * d_head^-0.5) v, Linear+bias.  Returns (out, probs[B,h,N,M])."""
    ctx = x if context is None else context
    q = F.linear(x, sd[p + "to_q.weight"])
    k = F.linear(ctx, sd[p + "to_k.weight"])
    v = F.linear(ctx, sd[p + "to_v.weight"])
    B, N, inner = q.shape
    d = inner // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v).permute(0, 2, 1, 3).reshape(B, N, inner)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"]), attn


# ------------------------------------------------------------------------------------------- a9
def feed_forward(sd, p, x):
    """FeedForward with GEGLU, unet.py:122-149: Linear(d,8d) -> a*gelu(b) (erf) -> Linear(4d,d)."""
    h = F.linear(x, sd[p + "net.0.proj.weight"], sd[p + "net.0.proj.bias"])
    a, g = h.chunk(2, dim=-1)
    return F.linear(a * F.gelu(g), sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


# ------------------------------------------------------------------------------------------- a7
def transformer_block(sd, p, x, context, heads, variant):
    """BasicTransformerBlock._forward.

    variant 'phosc' (unetPhosc.py:241-246): self-attn(norm1) + cross-attn(norm2) + ff(norm3).
    variant 'base'  (unet.py:337-345): attn1(norm2(x), context) + attn2(norm2(x), context) + ff(norm3);
    norm1 / attnc are never used."""
    if variant == "phosc":
        x = cross_attention(sd, p + "attn1.", layer_norm(sd, p + "norm1.", x), None, heads)[0] + x
        x = cross_attention(sd, p + "attn2.", layer_norm(sd, p + "norm2.", x), context, heads)[0] + x
    else:
        x = cross_attention(sd, p + "attn1.", layer_norm(sd, p + "norm2.", x), context, heads)[0] + x
        x = cross_attention(sd, p + "attn2.", layer_norm(sd, p + "norm2.", x), context, heads)[0] + x
    x = feed_forward(sd, p + "ff.", layer_norm(sd, p + "norm3.", x)) + x
    return x


# ------------------------------------------------------------------------------------------- a6
def spatial_transformer(sd, p, x, context, heads, variant, depth=1, taps=None):
    """SpatialTransformer.forward, unet.py:381-412 / unetPhosc.py:282-300.
    GroupNorm(32, eps 1e-6) (unet.py:161-162) -> 1x1 conv -> tokens -> blocks -> 1x1 conv -> + x_in."""
    B, C, H, W = x.shape
    x_in = x
    h = group_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    h = F.conv2d(h, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, -1)
    for d in range(depth):
        h = transformer_block(sd, f"{p}transformer_blocks.{d}.", h, context, heads, variant)
        if taps is not None:
            taps[f"{p}transformer_blocks.{d}"] = h
    h = h.reshape(B, H, W, -1).permute(0, 3, 1, 2)
    h = F.conv2d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return h + x_in


# ------------------------------------------------------------------------------------------- a10
def build_layout(cfg) -> dict:
    """The block list the reference constructor produces (unet.py:1248-1458): which layer
    types sit in each input/middle/output block.  Re-derived here from the kwargs only."""
    mc = cfg["model_channels"]
    mult = tuple(cfg.get("channel_mult", (1, 2, 4, 8)))
    nres = cfg["num_res_blocks"]
    attn_res = tuple(cfg["attention_resolutions"])
    num_heads = cfg.get("num_heads", -1)
    nhc = cfg.get("num_head_channels", -1)

    def heads_for(ch):
        return num_heads if nhc == -1 else ch // nhc

    inputs: List[list] = [[("conv", cfg["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in attn_res:
                layers.append(("st", ch, heads_for(ch)))
            inputs.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inputs.append([("down", ch, ch)])
            chans.append(ch)
            ds *= 2
    middle = [("res", ch, ch), ("st", ch, heads_for(ch)), ("res", ch, ch)]
    outputs: List[list] = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * m)]
            ch = mc * m
            if ds in attn_res:
                layers.append(("st", ch, heads_for(ch)))
            if level and i == nres:
                layers.append(("up", ch, ch))
                ds //= 2
            outputs.append(layers)
    return dict(inputs=inputs, middle=middle, outputs=outputs, out_ch=ch)


class UNetOracle:
    """Functional forward over a reference-layout state_dict.

    variant: 'base' = unet.UNetModel, 'phosc' = unetPhosc.UNetModelPhosc.
    phosc_on: args.phosc==1 or args.phos==1 (unetPhosc.py:1120)."""

    def __init__(self, cfg: dict, sd: Dict[str, Tensor], variant: str = "base", phosc_on: bool = False):
        assert variant in ("base", "phosc")
        self.cfg = dict(cfg)
        self.sd = {k: v.detach().float() if not v.requires_grad else v for k, v in sd.items()}
        self.variant = variant
        self.phosc_on = phosc_on
        self.layout = build_layout(cfg)
        self.depth = cfg.get("transformer_depth", 1)
        self.pe = positional_encoding(cfg.get("max_seq_len", 20), cfg["context_dim"])

    def _run(self, layers, prefix, h, emb, context, taps):
        sd = self.sd
        for j, (kind, cin, cout) in enumerate(layers):
            p = f"{prefix}{j}."
            if kind == "conv":
                h = F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1)
            elif kind == "res":
                h = resblock(sd, p, h, emb)
            elif kind == "st":
                h = spatial_transformer(sd, p, h, context, cout, self.variant, self.depth, taps)
            elif kind == "down":
                h = downsample(sd, p, h)
            elif kind == "up":
                h = upsample(sd, p, h)
            if taps is not None:
                taps[p[:-1]] = h
        return h

    def embed(self, t: Tensor, y: Optional[Tensor]) -> Tensor:
        """unet.py:1550-1581: time MLP + label embedding."""
        sd = self.sd
        e = timestep_embedding(t, self.cfg["model_channels"])
        e = F.linear(e, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
        e = F.linear(F.silu(e), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
        if self.cfg.get("num_classes") is not None:
            e = e + F.embedding(y.long(), sd["label_emb.weight"])
        return e

    def context(self, ids: Optional[Tensor], phosc: Optional[Tensor] = None) -> Optional[Tensor]:
        """unet.py:1605-1609; unetPhosc.py:1115-1130 (PHOSC ints through the same table, concatenated)."""
        if ids is None:
            return None
        msl = self.cfg.get("max_seq_len", 20)
        always = self.variant == "base"
        ctx = character_encoder(self.sd, ids, msl, always, self.pe)
        if self.variant == "phosc" and self.phosc_on:
            cp = character_encoder(self.sd, phosc.int(), msl, always, self.pe)
            ctx = torch.cat([ctx, cp], dim=1)
        return ctx

    def forward(self, x: Tensor, t: Tensor, context: Optional[Tensor], y: Optional[Tensor],
                phosc: Optional[Tensor] = None, taps: Optional[dict] = None) -> Tensor:
        sd = self.sd
        if self.variant == "phosc" and y is not None and y.shape[0] != x.shape[0]:
            y = y[: x.shape[0]]  # unetPhosc.py:1089-1090
        emb = self.embed(t, y)
        if taps is not None:
            taps["emb"] = emb
        ctx = self.context(context, phosc)
        if taps is not None and ctx is not None:
            taps["context"] = ctx
        h = x.float()
        hs = []
        for i, layers in enumerate(self.layout["inputs"]):
            h = self._run(layers, f"input_blocks.{i}.", h, emb, ctx, taps)
            hs.append(h)
            if taps is not None:
                taps[f"input_blocks.{i}"] = h
        h = self._run(self.layout["middle"], "middle_block.", h, emb, ctx, taps)
        if taps is not None:
            taps["middle_block"] = h
        for i, layers in enumerate(self.layout["outputs"]):
            h = torch.cat([h, hs.pop()], dim=1)
            h = self._run(layers, f"output_blocks.{i}.", h, emb, ctx, taps)
            if taps is not None:
                taps[f"output_blocks.{i}"] = h
        h = F.silu(group_norm(h, sd["out.0.weight"], sd["out.0.bias"], 1e-5))
        return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)

    __call__ = forward


def state_dict_shapes(cfg: dict, variant: str):
    """(key, shape) list of the reference state_dict for ``cfg`` - including entries the forward
    never touches (``attnc``, ``to_kv``, ``norm1``; base only: ``res.*``, ``wrd_proj.*``,
    unet.py:1243,1472).  Order follows module registration order in the reference constructor."""
    mc = cfg["model_channels"]
    ted = 4 * mc
    cd = cfg["context_dim"]
    out: list = []

    def lin(p, i, o, bias=True):
        out.append((p + "weight", (o, i)))
        if bias:
            out.append((p + "bias", (o,)))

    def conv(p, i, o, k=3):
        out.append((p + "weight", (o, i, k, k)))
        out.append((p + "bias", (o,)))

    def norm(p, c):
        out.append((p + "weight", (c,)))
        out.append((p + "bias", (c,)))

    def attn(p, qd, cdim, inner):
        lin(p + "to_q.", qd, inner, False)
        lin(p + "to_kv.", cdim, 2 * inner, False)
        lin(p + "to_k.", cdim, inner, False)
        lin(p + "to_v.", cdim, inner, False)
        lin(p + "to_out.0.", inner, qd)

    def res(p, cin, cout):
        norm(p + "in_layers.0.", cin)
        conv(p + "in_layers.2.", cin, cout)
        lin(p + "emb_layers.1.", ted, cout)
        norm(p + "out_layers.0.", cout)
        conv(p + "out_layers.3.", cout, cout)
        if cin != cout:
            conv(p + "skip_connection.", cin, cout, 1)

    def st(p, ch, heads):
        norm(p + "norm.", ch)
        conv(p + "proj_in.", ch, ch, 1)
        for d in range(cfg.get("transformer_depth", 1)):
            q = f"{p}transformer_blocks.{d}."
            attn(q + "attn1.", ch, ch, ch)
            attn(q + "attnc.", ch, ch, ch)
            lin(q + "ff.net.0.proj.", ch, 8 * ch)
            lin(q + "ff.net.2.", 4 * ch, ch)
            attn(q + "attn2.", ch, cd, ch)
            norm(q + "norm1.", ch)
            norm(q + "norm2.", ch)
            norm(q + "norm3.", ch)
        conv(p + "proj_out.", ch, ch, 1)

    lin("time_embed.0.", mc, ted)
    lin("time_embed.2.", ted, ted)
    out.append(("word_emb.embedding.weight", (cfg["vocab_size"], cd)))
    lin("word_emb.attention.linear_query.", cd, cd)
    lin("word_emb.attention.linear_key.", cd, cd)
    lin("word_emb.attention.linear_value.", cd, cd)
    if variant == "base":
        lin("wrd_proj.", 4096, 320)
    if cfg.get("num_classes") is not None:
        out.append(("label_emb.weight", (cfg["num_classes"], ted)))
    layout = build_layout(cfg)

    def emit(layers, prefix):
        for j, (kind, cin, cout) in enumerate(layers):
            p = f"{prefix}{j}."
            if kind == "conv":
                conv(p, cin, cout)
            elif kind == "res":
                res(p, cin, cout)
            elif kind == "st":
                st(p, cin, cout)
            elif kind == "down":
                conv(p + "op.", cin, cout)
            elif kind == "up":
                conv(p + "conv.", cin, cout)

    for i, layers in enumerate(layout["inputs"]):
        emit(layers, f"input_blocks.{i}.")
    emit(layout["middle"], "middle_block.")
    for i, layers in enumerate(layout["outputs"]):
        emit(layers, f"output_blocks.{i}.")
    norm("out.0.", layout["out_ch"])
    conv("out.2.", mc, cfg["out_channels"])
    if variant == "base":
        # self.res = ResBlockConditional(32, 1280, 0.2, 320, use_conv=True, down=True)  (unet.py:1472)
        norm("res.in_layers.0.", 32)
        conv("res.in_layers.2.", 32, 320)
        conv("res.h_upd.op.", 32, 32)
        conv("res.x_upd.op.", 32, 32)
        lin("res.emb_layers.1.", 1280, 320)
        norm("res.out_layers.0.", 320)
        conv("res.out_layers.3.", 320, 320)
        conv("res.skip_connection.", 32, 320)
    return out
