"""Common body of ``UNetModel`` / ``UNetModelPhosc``: the reference constructor surface and parameter tree,
with the forward handed to the HIP engine.

Constructor kwargs: reference ``unet.py:1126-1156`` / ``unetPhosc.py:781-811`` (identical lists).
Block construction order: ``unet.py:1248-1458``.
"""
from __future__ import annotations

import copy
import os
from typing import Optional

import torch
import torch.nn as nn

from .layers import (CharacterEncoderParams, DownsampleParams, ResBlockParams, SpatialTransformerParams,
                     UpsampleParams, _zero)


def _arg(args, name, default=0):
    return getattr(args, name, default) if args is not None else default


class UNetBase(nn.Module):
    variant = "base"

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=True, transformer_depth=1,
                 context_dim=768, vocab_size=256, n_embed=None, legacy=False, args=None, max_seq_len=20):
        super().__init__()
        # ---- combinations the reference itself cannot run, or that lie outside the hot path
        if not use_spatial_transformer:
            raise NotImplementedError("use_spatial_transformer=False (AttentionBlock) is never used by the reference")
        if context_dim is None:
            raise AssertionError("context_dim is required with the spatial transformer (unet.py:1163-1164)")
        if dims != 2:
            raise NotImplementedError("dims != 2")
        if resblock_updown:
            raise NotImplementedError("resblock_updown=True raises TypeError in the reference (unet.py:547)")
        if not conv_resample:
            raise NotImplementedError("conv_resample=False builds nn.AvgPool2d(dims, ...) which the reference cannot run")
        if use_scale_shift_norm:
            raise NotImplementedError("use_scale_shift_norm=True is not on the accelerated path")
        if n_embed is not None:
            raise NotImplementedError("n_embed (codebook id prediction) is not on the accelerated path")
        if isinstance(context_dim, (list, tuple)):
            context_dim = list(context_dim)[0]
        for flag in ("charImages", "attentionMaps", "ocrTraining", "wrdChrWrStyl"):
            if _arg(args, flag, 0):
                raise NotImplementedError(f"args.{flag}=1 selects an auxiliary head / debugging output that is "
                                          "outside the accelerated denoising path (SURVEY.md 8b)")
        # args.charLevelEmb=1 (the default of unet.py:1871; checkpoints ``ckpt_ema_charLevelEmb.pt``, config.py:61): the base
        # CharacterEncoder flattens the ids, embeds them and views the result back as (B, 10, 320) (unet.py:855-864) - the same
        # tensor as without the flag (tests/golden/fwd_base_full_charlevel.npz: bit-identical output).  The view hard-codes
        # 10 tokens x 320 channels, so the reference itself fails for any other context width.  unetPhosc.py never reads it.
        self.char_level_emb = bool(_arg(args, "charLevelEmb", 0)) and self.variant == "base"
        if self.char_level_emb and context_dim != 320:
            raise NotImplementedError("args.charLevelEmb=1 needs context_dim=320: unet.py:864 views the embedding as "
                                      "(B, 10, 320) and raises for any other width")
        if num_heads == -1 and num_head_channels == -1:
            raise AssertionError("Either num_heads or num_head_channels has to be set")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads

        self.args = args
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float32  # use_fp16 only retypes the input in the reference (convert_* are no-ops)
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.predict_codebook_ids = False
        self.context_dim = context_dim
        self.max_seq_len = max_seq_len
        self.transformer_depth = transformer_depth
        self.interpolation = bool(_arg(args, "interpolation", False))

        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.word_emb = CharacterEncoderParams(vocab_size, context_dim, max_seq_len)
        self._extra_heads_before_label()
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)

        def heads_for(ch):
            if num_head_channels == -1:
                return num_heads, ch // num_heads
            return ch // num_head_channels, num_head_channels

        def st(ch):
            h, d = heads_for(ch)
            return SpatialTransformerParams(ch, h, d, transformer_depth, context_dim)

        self.input_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlockParams(ch, ted, mult * model_channels, dropout)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(st(ch))
                self.input_blocks.append(nn.Sequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(nn.Sequential(DownsampleParams(ch, ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = nn.Sequential(ResBlockParams(ch, ted, ch, dropout), st(ch),
                                          ResBlockParams(ch, ted, ch, dropout))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlockParams(ch + ich, ted, model_channels * mult, dropout)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(st(ch))
                if level and i == num_res_blocks:
                    layers.append(UpsampleParams(ch, ch))
                    ds //= 2
                self.output_blocks.append(nn.Sequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(),
                                 _zero(nn.Conv2d(model_channels, out_channels, 3, padding=1)))
        self._extra_heads_after_out()
        self._engine = None
        self._train_engine = None
        dev = _arg(args, "device", None)
        if dev is not None and str(dev) != "cpu":
            # the reference moves word_emb to args.device in the constructor (unet.py:1210-1213); callers then
            # .to(device) the whole model, which we leave to them.
            pass

    # hooks for the two variants --------------------------------------------------------------------------
    def _extra_heads_before_label(self):
        pass

    def _extra_heads_after_out(self):
        pass

    # ---------------------------------------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            from .engine import UNetEngine
            self._engine = UNetEngine(self, self.variant)
            mode = os.environ.get("WDIFF_PRECISION")
            if mode:
                self._engine.set_precision(mode)
        return self._engine

    def __deepcopy__(self, memo):
        # ``copy.deepcopy(model)`` is how the reference makes its EMA model (train.py:409); the engine (ctypes
        # handles, device plans) is per-instance and rebuilt lazily by the copy.
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_engine", "_train_engine", "_anchor") else copy.deepcopy(v, memo)
        return new

    def set_precision(self, mode: str):
        """'bf16x3' (default; split-bf16 MFMA, fp32-class results) or 'bf16' (single pass)."""
        self.__dict__["_precision"] = mode
        self.engine.set_precision(mode)
        if self._train_engine is not None:
            self._train_engine.set_precision(mode)
        return self

    def convert_to_fp16(self):  # no-ops in the reference as well (unet.py:415-419)
        pass

    def convert_to_fp32(self):
        pass

    def _check_common(self, x, timesteps, mix_rate):
        if self.interpolation and mix_rate is not None:
            raise NotImplementedError("writer-style interpolation (mix_rate) draws two random writers on the host "
                                      "(unet.py:1558-1573); not on the accelerated path")
        if timesteps is None:
            raise ValueError("timesteps is required")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"x must be [B,{self.in_channels},H,W], got {tuple(x.shape)}")

    def _check_context(self, context):
        if self.char_level_emb and context is not None and context.shape[1] != 10:
            raise ValueError("args.charLevelEmb=1: context must be [B, 10] (unet.py:864 views the embedding as (B, 10, 320))")

    @property
    def train_engine(self):
        """Forward-with-saved-intermediates + backward launch lists (``train_engine.py``); base variant only."""
        if self._train_engine is None:
            from .train_engine import TrainEngine
            self._train_engine = TrainEngine(self, self.variant)
            mode = self.__dict__.get("_precision") or os.environ.get("WDIFF_PRECISION")
            if mode:
                self._train_engine.set_precision(mode)
        return self._train_engine

    def _run(self, x, timesteps, context, y, phosc=None):
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            # ``predicted_noise = model(...)`` inside the training loop (train.py:287): the result carries a grad_fn whose
            # backward runs the HIP backward list and fills ``param.grad`` (reference layouts), so ``loss.backward()``,
            # ``optimizer.step()`` and ``ema.step_ema`` of the reference loop work unchanged.
            return _HipUNetStep.apply(self, x, timesteps, context, y, phosc, self._grad_anchor(x.device)).type(x.dtype)
        out = self.engine.forward(x.float(), timesteps, context, y, phosc)
        return out.type(x.dtype)

    def _grad_anchor(self, device):
        a = self.__dict__.get("_anchor")
        if a is None or a.device != device:
            a = torch.zeros(1, device=device, requires_grad=True)
            self.__dict__["_anchor"] = a
        return a


class _HipUNetStep(torch.autograd.Function):
    """Autograd node of the HIP training forward.  The parameters are not inputs of the node: their gradients are written
    by the backward kernels into persistent buffers that become ``param.grad`` (added to an existing ``.grad``, as autograd
    would).  ``anchor`` is a dummy leaf that makes autograd schedule the node; x / context get no gradient (train.py never
    asks for one)."""

    @staticmethod
    def forward(ctx, model, x, timesteps, context, y, phosc, anchor):
        eng = model.train_engine
        ctx.eng = eng
        return eng.forward_train(x.float(), timesteps, context, y, phosc).clone()

    @staticmethod
    def backward(ctx, gout):
        ctx.eng.backward(gout.contiguous().float())
        return None, None, None, None, None, None, torch.zeros_like(ctx.eng.model._anchor)
