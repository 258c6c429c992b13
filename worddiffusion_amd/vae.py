"""The decoder half of the Stable-Diffusion-v1.5 ``AutoencoderKL`` on the HIP kernels of this package.

Reference call sites: ``AutoencoderKL.from_pretrained(args.stable_dif_path, subfolder="vae")`` (``train.py:415``,
``sampling.py:108``) and, at the end of every sampler, ``latents = 1 / 0.18215 * x; image = vae.decode(latents).sample;
image = (image / 2 + 0.5).clamp(0, 1)`` (``train.py:239-247``).  ``diffusers`` is not part of the reference tree: the
architecture below restates the published ``diffusers`` model (``models/autoencoder_kl.py``, ``models/vae.py::Decoder``,
``UNetMidBlock2D``, ``UpDecoderBlock2D``, ``ResnetBlock2D``, ``Attention``, ``Upsample2D``) for the SD-v1.5 VAE config
(block_out_channels (128, 256, 512, 512), layers_per_block 2, latent_channels 4, norm_num_groups 32, eps 1e-6, one
attention head of 512 channels in the mid block).

**Parity unpinned**: neither ``diffusers`` nor any VAE weights exist offline, so the HIP path is
checked against ``oracle/vae_oracle.py`` - a plain-torch restatement of the same published architecture - on synthetic
weights; a real checkpoint in ``diffusers`` layout loads through ``load_state_dict`` / ``from_pretrained`` (state-dict
keys and shapes are the ``diffusers`` ones, both attention namings).

Only the decoder is built (the training loop of the reference reads cached latents, ``vaeFromDict``; the encoder is not on
any path of section 8).  There is no CPU fallback: ``decode`` needs the model on an MI355X.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _native as N
from .engine import Act, Plan, RecipeBook, UNetEngine

SD15_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)


class _Resnet(nn.Module):
    """``ResnetBlock2D`` without time embedding (``temb_channels=None``), eps 1e-6, SiLU."""

    def __init__(self, cin: int, cout: int, groups: int):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)


class _Attention(nn.Module):
    """``Attention`` of the VAE mid block: GroupNorm, one head over all channels, biased projections, residual."""

    def __init__(self, ch: int, groups: int):
        super().__init__()
        self.ch = ch
        self.group_norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])


class _Mid(nn.Module):
    def __init__(self, ch: int, groups: int):
        super().__init__()
        self.attentions = nn.ModuleList([_Attention(ch, groups)])
        self.resnets = nn.ModuleList([_Resnet(ch, ch, groups), _Resnet(ch, ch, groups)])


class _Upsampler(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.cin = self.cout = ch
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)


class _UpBlock(nn.Module):
    def __init__(self, cin: int, cout: int, nlayers: int, groups: int, upsample: bool):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if j == 0 else cout, cout, groups) for j in range(nlayers)])
        if upsample:
            self.upsamplers = nn.ModuleList([_Upsampler(cout)])


class _Decoder(nn.Module):
    def __init__(self, latent: int, out_ch: int, boc: Sequence[int], layers_per_block: int, groups: int):
        super().__init__()
        self.conv_in = nn.Conv2d(latent, boc[-1], 3, padding=1)
        self.mid_block = _Mid(boc[-1], groups)
        rev = list(reversed(boc))
        ups, prev = [], rev[0]
        for i, ch in enumerate(rev):
            ups.append(_UpBlock(prev, ch, layers_per_block + 1, groups, upsample=i != len(rev) - 1))
            prev = ch
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], out_ch, 3, padding=1)


class DecoderOutput:
    """What ``vae.decode`` returns in ``diffusers`` (the samplers read ``.sample``)."""

    def __init__(self, sample: torch.Tensor):
        self.sample = sample


class AutoencoderKL(nn.Module):
    """Decoder-only ``AutoencoderKL``: ``decode(z).sample`` = image in [-1, 1] nominal range, [B, 3, 8h, 8w]."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, latent_channels: int = 4,
                 block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block: int = 2,
                 norm_num_groups: int = 32, scaling_factor: float = 0.18215, **_ignored):
        super().__init__()
        if norm_num_groups != 32:
            raise NotImplementedError("the GroupNorm kernels are built for 32 groups (every SD VAE uses 32)")
        if any(c % 64 for c in block_out_channels):
            raise NotImplementedError("block_out_channels must be multiples of 64 (the GEMM streams 64-channel chunks)")
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      norm_num_groups=norm_num_groups, scaling_factor=scaling_factor)
        self.decoder = _Decoder(latent_channels, out_channels, tuple(block_out_channels), layers_per_block, norm_num_groups)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self._engine: Optional[VAEDecoderEngine] = None

    # ---- weights -----------------------------------------------------------------------------------------------------
    @staticmethod
    def _remap(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Decoder entries of a ``diffusers`` AutoencoderKL state dict; the pre-0.14 attention names (query / key / value /
        proj_attn) are mapped to to_q / to_k / to_v / to_out.0, 1x1-conv shaped projections are flattened."""
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        out = {}
        for k, v in sd.items():
            if not (k.startswith("decoder.") or k.startswith("post_quant_conv.")):
                continue
            parts = k.split(".")
            if "attentions" in parts and parts[-2] in ren:
                k = ".".join(parts[:-2] + [ren[parts[-2]], parts[-1]])
            if "attentions" in k and v.dim() == 4 and ("to_" in k):
                v = v.reshape(v.shape[0], v.shape[1])
            out[k] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        return super().load_state_dict(self._remap(dict(state_dict)), strict=strict, **kw)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, **kw) -> "AutoencoderKL":
        """LOCAL directory in ``diffusers`` layout (``config.json`` + ``diffusion_pytorch_model.safetensors`` or ``.bin``);
        nothing is ever downloaded.  The ``.bin`` is read with ``weights_only=True``."""
        root = os.path.join(path, subfolder) if subfolder else path
        cfg = dict(SD15_VAE_CONFIG)
        cfg_file = os.path.join(root, "config.json")
        if os.path.isfile(cfg_file):
            with open(cfg_file) as f:
                raw = json.load(f)
            cfg.update({k: raw[k] for k in cfg if k in raw})
        model = cls(**cfg)
        st = os.path.join(root, "diffusion_pytorch_model.safetensors")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(root, "diffusion_pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(sd)
        return model.eval().requires_grad_(False)

    # ---- the one operation the samplers use -------------------------------------------------------------------------------
    @property
    def engine(self) -> "VAEDecoderEngine":
        if self._engine is None:
            self._engine = VAEDecoderEngine(self)
        return self._engine

    def set_precision(self, mode: str):
        self.engine.set_precision(mode)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        if z.dim() != 4 or z.shape[1] != self.config.latent_channels:
            raise ValueError(f"latents must be [B, {self.config.latent_channels}, h, w], got {tuple(z.shape)}")
        if not z.is_cuda:
            raise N.NativeError("worddiffusion_amd runs on an MI355X only: move the latents and the VAE to cuda "
                                "(there is no CPU / eager fallback)")
        sample = self.engine.decode(z.float().contiguous())
        return DecoderOutput(sample) if return_dict else (sample,)

    def forward(self, z):
        return self.decode(z).sample


class VAEDecoderEngine(UNetEngine):
    """Launch plan of the decoder out of the UNet engine's building blocks: im2col + GEMM for the two 4-channel
    convolutions, tap-gather GEMMs with fused GroupNorm statistics for every 3x3, nearest-x2 folded into the gather table of
    the upsampler's convolution, the shortcut 1x1 as a second K segment of a block's last 3x3, one attention launch."""

    TILE = 128128  # channel counts are multiples of 128, not of 160

    def __init__(self, model: AutoencoderKL):
        super().__init__(model, "vae")

    # ---- operands -----------------------------------------------------------------------------------------------------
    def _resnets(self):
        d = self.model.decoder
        yield "mid.r0", d.mid_block.resnets[0]
        yield "mid.r1", d.mid_block.resnets[1]
        for i, ub in enumerate(d.up_blocks):
            for j, r in enumerate(ub.resnets):
                yield f"up{i}.r{j}", r

    def _recipes(self) -> RecipeBook:
        m = self.model
        d = m.decoder
        R = RecipeBook()
        lat = m.config.latent_channels
        self.kpad_in = ((9 * lat + 31) // 32) * 32
        # post_quant_conv (1x1) rides the same im2col operand as a 3x3 whose only non-zero tap is the centre one
        R.matrix("pq.w", lat, self.kpad_in).fwd(m.post_quant_conv.weight, col_off=4 * lat)
        R.vector("pq.b", m.post_quant_conv.bias)
        R.matrix("in.w", d.conv_in.out_channels, self.kpad_in).fwd(d.conv_in.weight)
        R.vector("in.b", d.conv_in.bias)
        for name, r in self._resnets():
            R.vector(name + ".gn1.g", r.norm1.weight)
            R.vector(name + ".gn1.b", r.norm1.bias)
            R.matrix(name + ".c1.w", r.cout, 9 * r.cin).fwd(r.conv1.weight)
            R.vector(name + ".c1.b", r.conv1.bias)
            R.vector(name + ".gn2.g", r.norm2.weight)
            R.vector(name + ".gn2.b", r.norm2.bias)
            if r.cin != r.cout:
                R.matrix(name + ".c2.w", r.cout, 9 * r.cout + r.cin).fwd(r.conv2.weight) \
                    .fwd(r.conv_shortcut.weight, col_off=9 * r.cout)
                R.vector(name + ".c2.b", r.conv2.bias, r.conv_shortcut.bias)
            else:
                R.matrix(name + ".c2.w", r.cout, 9 * r.cout).fwd(r.conv2.weight)
                R.vector(name + ".c2.b", r.conv2.bias)
        at = d.mid_block.attentions[0]
        ch = at.ch
        R.vector("mid.at.gn.g", at.group_norm.weight)
        R.vector("mid.at.gn.b", at.group_norm.bias)
        R.matrix("mid.at.qkv.w", 3 * ch, ch)
        for i, l in enumerate((at.to_q, at.to_k, at.to_v)):
            R["mid.at.qkv.w"].fwd(l.weight, row_off=i * ch)
        R.vector_cat("mid.at.qkv.b", [at.to_q.bias, at.to_k.bias, at.to_v.bias])
        R.linear("mid.at.o", at.to_out[0])
        for i, ub in enumerate(d.up_blocks):
            if hasattr(ub, "upsamplers"):
                up = ub.upsamplers[0]
                R.matrix(f"up{i}.us.w", up.cout, 9 * up.cin).fwd(up.conv.weight)
                R.vector(f"up{i}.us.b", up.conv.bias)
        R.vector("out.gn.g", d.conv_norm_out.weight)
        R.vector("out.gn.b", d.conv_norm_out.bias)
        R.matrix("out.w", d.conv_out.out_channels, 9 * d.conv_out.in_channels).fwd(d.conv_out.weight)
        R.vector("out.b", d.conv_out.bias)
        return R

    # ---- plan ---------------------------------------------------------------------------------------------------------
    def _vae_resnet(self, P: Plan, name: str, r: _Resnet, x: Act) -> Act:
        ops = P.step
        B, h, w = self._B, x.h, x.w
        hw, M = h * w, B * h * w
        tab, _, _ = self._table(h, w, "same")
        need_raw = r.cin != r.cout
        a1, raw = self._gn(P, ops, name + ".gn1", [x], name + ".gn1", 1e-6, True, want_raw=need_raw)
        h1 = self._f32(P, M, r.cout)
        g1 = self._gemm(ops, name + ".conv1", [self._src(a1, r.cin, 9, tab, hw)], name + ".c1.w", M, hw,
                        bias=self._w[name + ".c1.b"], out_f32=h1, out_ld=r.cout, want_stats=True, tile=self.TILE)
        a2, _ = self._gn(P, ops, name + ".gn2", [Act(h1, r.cout, h, w, g1._stats)], name + ".gn2", 1e-6, True)
        out = self._f32(P, M, r.cout)
        if need_raw:
            g2 = self._gemm(ops, name + ".conv2+shortcut", [self._src(a2, r.cout, 9, tab, hw), self._src(raw, r.cin)],
                            name + ".c2.w", M, hw, bias=self._w[name + ".c2.b"], out_f32=out, out_ld=r.cout,
                            want_stats=True, tile=self.TILE)
        else:
            g2 = self._gemm(ops, name + ".conv2", [self._src(a2, r.cout, 9, tab, hw)], name + ".c2.w", M, hw,
                            bias=self._w[name + ".c2.b"], resid=x.t.data_ptr(), resid_ld=r.cout, out_f32=out,
                            out_ld=r.cout, want_stats=True, tile=self.TILE)
        return Act(out, r.cout, h, w, g2._stats)

    def _vae_attention(self, P: Plan, x: Act) -> Act:
        ops = P.step
        B, h, w, c = self._B, x.h, x.w, x.c
        hw, M = h * w, B * h * w
        g, _ = self._gn(P, ops, "mid.at.gn", [x], "mid.at.gn", 1e-6, False)
        qkv = self._f32(P, M, 3 * c)
        self._gemm(ops, "mid.attn.qkv", [self._src(g, c)], "mid.at.qkv.w", M, hw, bias=self._w["mid.at.qkv.b"], out_f32=qkv,
                   out_ld=3 * c, tile=self.TILE)
        o = self._planes(P, M, c)
        self._attention(ops, "mid.attn", qkv.data_ptr(), 3 * c, qkv.data_ptr() + 4 * c, 3 * c, qkv.data_ptr() + 8 * c, 3 * c,
                        1, hw, hw, c, float(c) ** -0.5, o)
        out = self._f32(P, M, c)
        gg = self._gemm(ops, "mid.attn.to_out+residual", [self._src(o, c)], "mid.at.o.w", M, hw, bias=self._w["mid.at.o.b"],
                        resid=x.t.data_ptr(), resid_ld=c, out_f32=out, out_ld=c, want_stats=True, tile=self.TILE)
        return Act(out, c, h, w, gg._stats)

    def plan_decode(self, B: int, H: int, W: int) -> Plan:
        key = ("vae", B, H, W, self.npass)
        if key in self._plans:
            return self._plans[key]
        m, lib, dev = self.model, self.lib, self.device
        d = m.decoder
        lat = m.config.latent_channels
        lo_ok = self.npass == 3
        P = Plan()
        self._cur_plan = P
        self._B = B
        step = P.step
        M = B * H * W
        P.z_in = torch.zeros((B, lat, H, W), dtype=torch.float32, device=dev)
        zc = self._planes(P, M, self.kpad_in)
        step.append((lib.wd_im2col3x3, (P.z_in.data_ptr(), B, lat, H, W, zc[0].data_ptr(), zc[1].data_ptr() if lo_ok else None,
                                        self.kpad_in), "im2col(z)"))
        zq_tok = self._f32(P, M, lat)
        self._gemm(step, "post_quant_conv", [self._src(zc, self.kpad_in)], "pq.w", M, H * W, bias=self._w["pq.b"],
                   out_f32=zq_tok, out_ld=lat)
        zq = self._f32(P, B, lat, H, W)
        step.append((lib.wd_tokens_to_nchw, (zq_tok.data_ptr(), lat, B, lat, H * W, zq.data_ptr()), "tokens_to_nchw(z)"))
        zi = self._planes(P, M, self.kpad_in)
        step.append((lib.wd_im2col3x3, (zq.data_ptr(), B, lat, H, W, zi[0].data_ptr(), zi[1].data_ptr() if lo_ok else None,
                                        self.kpad_in), "im2col(post_quant z)"))
        c0 = d.conv_in.out_channels
        h0 = self._f32(P, M, c0)
        g0 = self._gemm(step, "decoder.conv_in", [self._src(zi, self.kpad_in)], "in.w", M, H * W, bias=self._w["in.b"],
                        out_f32=h0, out_ld=c0, want_stats=True, tile=self.TILE)
        cur = Act(h0, c0, H, W, g0._stats)
        cur = self._vae_resnet(P, "mid.r0", d.mid_block.resnets[0], cur)
        cur = self._vae_attention(P, cur)
        cur = self._vae_resnet(P, "mid.r1", d.mid_block.resnets[1], cur)
        for i, ub in enumerate(d.up_blocks):
            for j, r in enumerate(ub.resnets):
                cur = self._vae_resnet(P, f"up{i}.r{j}", r, cur)
            if hasattr(ub, "upsamplers"):
                cur = self._resample(P, f"up{i}.us", ub.upsamplers[0], cur, "up", tile=self.TILE)
        g, _ = self._gn(P, step, "out.gn", [cur], "out.gn", 1e-6, True)
        tab, _, _ = self._table(cur.h, cur.w, "same")
        oc = d.conv_out.out_channels
        otok = self._f32(P, B * cur.h * cur.w, oc)
        self._gemm(step, "decoder.conv_out", [self._src(g, cur.c, 9, tab, cur.h * cur.w)], "out.w", B * cur.h * cur.w,
                   cur.h * cur.w, bias=self._w["out.b"], out_f32=otok, out_ld=oc)
        P.out = torch.empty((B, oc, cur.h, cur.w), dtype=torch.float32, device=dev)
        step.append((lib.wd_tokens_to_nchw, (otok.data_ptr(), oc, B, oc, cur.h * cur.w, P.out.data_ptr()), "tokens_to_nchw"))
        self._plans[key] = P
        return P

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        self.refresh_weights()
        B, _, H, W = z.shape
        P = self.plan_decode(B, H, W)
        P.z_in.copy_(z, non_blocking=True)
        P.run_step(torch.cuda.current_stream(self.device).cuda_stream)
        return P.out.clone()
