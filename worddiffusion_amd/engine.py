"""Host side of the HIP UNet forward: weight repacking, buffer planning and the launch sequence.

The reference runs ``UNetModel.forward`` (``unet.py:1499-1836``) / ``UNetModelPhosc.forward``
(``unetPhosc.py:1068-1159``) as ~150 ATen kernels over NCHW fp32 tensors.  Here the same arithmetic is a static
list of launches of the kernels in ``csrc/`` over token-major (NHWC) buffers:

  * every convolution / linear is ``wd_gemm`` (tap-gather GEMM on MFMA, split-bf16 operands);
  * GroupNorm(+SiLU) is ``wd_gn_stats`` + ``wd_gn_apply`` which writes the GEMM operand planes directly - the
    channel concat of the decoder (``torch.cat([h, hs.pop()], 1)``, unet.py:1750) is never materialised, the two
    halves are normalised into one plane buffer; the 1x1 skip projection of a ResBlock is a second K-segment of
    its last 3x3 GEMM; nearest-x2 upsampling / stride-2 are gather tables of the same GEMM;
  * step-invariant work (CharacterEncoder, K/V projections of every cross-attention) lives in ``cond`` ops that
    ``Diffusion.sampling`` runs once per call, the per-step ops are captured into one hipGraph.

Torch is used for device memory, the weight repack at load time and stream handles only.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _native as N
from .layers import (DownsampleParams, ResBlockParams, SpatialTransformerParams, UpsampleParams)


# ----------------------------------------------------------------------------------------------------------
def conv_gather_table(h: int, w: int, mode: str) -> Tuple[np.ndarray, int, int]:
    """int32 [9][h_out*w_out] source position (or -1 = zero padding) per 3x3 tap.

    mode 'same': 3x3 pad 1 (unet.py:595); 'down': 3x3 stride 2 pad 1 (unet.py:540-542);
    'up': nearest x2 then 3x3 pad 1 (unet.py:497-499) - (h, w) is the INPUT size in all modes."""
    if mode == "same":
        ho, wo = h, w
    elif mode == "down":
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    elif mode == "up":
        ho, wo = 2 * h, 2 * w
    else:
        raise ValueError(mode)
    yy, xx = np.meshgrid(np.arange(ho), np.arange(wo), indexing="ij")
    tab = np.full((9, ho * wo), -1, dtype=np.int32)
    for tap in range(9):
        dy, dx = tap // 3 - 1, tap % 3 - 1
        if mode == "same":
            sy, sx = yy + dy, xx + dx
            ok = (sy >= 0) & (sy < h) & (sx >= 0) & (sx < w)
        elif mode == "down":
            sy, sx = 2 * yy + dy, 2 * xx + dx
            ok = (sy >= 0) & (sy < h) & (sx >= 0) & (sx < w)
        else:
            uy, ux = yy + dy, xx + dx  # coordinates in the upsampled map
            ok = (uy >= 0) & (uy < ho) & (ux >= 0) & (ux < wo)
            sy, sx = uy // 2, ux // 2
        idx = sy * w + sx
        tab[tap] = np.where(ok, idx, -1).reshape(-1)
    return tab, ho, wo


def upsample_phase_tables(h: int, w: int) -> Tuple[np.ndarray, np.ndarray]:
    """Nearest x2 upsample + 3x3 / pad 1 convolution (Upsample.forward, unet.py:488-499) as four 2x2 convolutions of the SOURCE map,
    one per output phase (py, px) = (row parity, column parity): the taps of the 3x3 kernel that fall on the same source pixel are
    summed (``upsample_phase_weights``), 4 taps instead of 9.  Output rows of a sample are phase-major: q = (2 py + px) * h*w + y*w + x
    is output pixel (2y + py, 2x + px).

    Returns (int32 [4 taps][4*h*w] source position or -1, tap = 2 dy + dx reads source (y + dy - 1 + py, x + dx - 1 + px);
             int32 [4*h*w]: the row q that holds raster position (Y, X) of the 2h x 2w map)."""
    hw = h * w
    tab = np.full((4, 4 * hw), -1, dtype=np.int32)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    for py in range(2):
        for px in range(2):
            ph = 2 * py + px
            for dy in range(2):
                for dx in range(2):
                    sy, sx = yy + dy - 1 + py, xx + dx - 1 + px
                    ok = (sy >= 0) & (sy < h) & (sx >= 0) & (sx < w)
                    tab[2 * dy + dx, ph * hw:(ph + 1) * hw] = np.where(ok, sy * w + sx, -1).reshape(-1)
    Y, X = np.meshgrid(np.arange(2 * h), np.arange(2 * w), indexing="ij")
    perm = ((2 * (Y % 2) + (X % 2)) * hw + (Y // 2) * w + (X // 2)).reshape(-1).astype(np.int32)
    return tab, perm


def upsample_phase_weights(w: torch.Tensor) -> torch.Tensor:
    """[N, C, 3, 3] -> [4 phases][4 taps][N][C] fp32: kernel rows {0} / {1, 2} feed source rows y - 1 / y of an even output row,
    {0, 1} / {2} feed y / y + 1 of an odd one (the same for columns); phase = 2 py + px, tap = 2 dy + dx."""
    sel = torch.tensor([[[1., 0., 0.], [0., 1., 1.]], [[1., 1., 0.], [0., 0., 1.]]], dtype=torch.float32, device=w.device)  # [parity][d][k]
    out = torch.einsum("pdk,qel,nckl->pqdenc", sel, sel, w.detach().float())
    n, c = w.shape[0], w.shape[1]
    return out.reshape(4, 4, n, c).contiguous()


def geglu_interleave(w: torch.Tensor, g: int = 32) -> torch.Tensor:
    """Rows [x | gate] (unet.py:128 ``chunk(2)``) -> blocks of ``g`` x-rows followed by their ``g`` gate rows, so
    that one BN = 2g column tile of ``wd_gemm`` holds both halves of ``g`` output columns."""
    inner = w.shape[0] // 2
    assert inner % g == 0
    x = w[:inner].reshape(inner // g, g, *w.shape[1:])
    gt = w[inner:].reshape(inner // g, g, *w.shape[1:])
    return torch.stack([x, gt], dim=1).reshape(w.shape)


def ff_fused_supported(c: int, inner: int) -> bool:
    """The shapes csrc/wd_ff.hip covers (== wd_ff_supported of the library, which wd_ff_fused itself enforces)."""
    return c == 320 and inner > 0 and inner % 128 == 0


def geglu_tile(inner: int) -> int:
    """GEMM tile for a GEGLU projection with ``inner`` output columns: 128x160 when 80 divides it."""
    return 128160 if inner % 80 == 0 else 128064


def slab_order(wp: torch.Tensor, ntaps0: int, c0: int, c1: int = 0) -> torch.Tensor:
    """[..., N, ktot] (k = tap-major/channel-minor of src0, then src1) -> "slab order" [..., stages, N, 32]:
    stage = (32-channel chunk of src0, tap) chunk-major, then the 32-channel chunks of src1 (wd_gemm w_layout 1)."""
    lead = wp.shape[:-2]
    n = wp.shape[-2]
    k0 = ntaps0 * c0
    w0 = wp[..., :k0].reshape(*lead, n, ntaps0, c0 // 32, 32)
    nl = len(lead)
    w0 = w0.permute(*range(nl), nl + 2, nl + 1, nl, nl + 3).reshape(*lead, (c0 // 32) * ntaps0, n, 32)
    if c1:
        w1 = wp[..., k0:].reshape(*lead, n, c1 // 32, 32).permute(*range(nl), nl + 1, nl, nl + 2)
        w0 = torch.cat([w0, w1], dim=nl)
    return w0.contiguous()


def slab_span(tab: Optional[np.ndarray], hw_out: int, hw_src: int, m: int, bm: int = 128) -> int:
    """Largest (max - min + 1) of the gathered source rows over the bm-row output panels (wd_gemm slab_rows)."""
    if tab is None:
        return bm
    period = int(np.lcm(bm, hw_out))
    mt = min(m, 2 * period)
    rows = np.arange(mt)
    b, p = rows // hw_out, rows % hw_out
    src = np.where(tab[:, p] >= 0, b[None, :] * hw_src + tab[:, p], -1)  # [ntaps, mt]
    span = 0
    for r0 in range(0, mt, bm):
        blk = src[:, r0:r0 + bm]
        valid = blk[blk >= 0]
        if valid.size:
            span = max(span, int(valid.max() - valid.min() + 1))
    return span


_PARAM_GEN = [0]  # bumped whenever any nn.Module registers a Parameter (see UNetEngine._signature)
_NATIVE_WRITES = [0]  # bumped whenever a kernel of this package rewrites parameters in place (optimiser step, EMA)


def note_native_write():
    """``wd_adamw_multi`` / ``wd_ema_update`` change parameter storage behind autograd's back (no ``Tensor._version`` bump): every
    engine re-derives its packed operands before its next use.  (A public counter of our own instead of the private
    ``torch._C._autograd._unsafe_set_version_counter``.)"""
    _NATIVE_WRITES[0] += 1

PLAN_CACHE_SIZE = int(os.environ.get("WDIFF_PLAN_CACHE", "4"))  # launch plans kept per engine (least recently used evicted)


def _bump_param_gen(*_a):
    _PARAM_GEN[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump_param_gen)

FILM_CHUNK_ROWS = int(os.environ.get("WDIFF_FILM_CHUNK_ROWS", "8192"))  # rows (timesteps x batch) of the resident FiLM chunk


class _Recipe:
    __slots__ = ("name", "planes", "rows", "cols", "shape", "pieces")

    def __init__(self, name, planes, rows=0, cols=0, shape=None):
        self.name, self.planes, self.rows, self.cols, self.shape, self.pieces = name, planes, rows, cols, shape, []

    @staticmethod
    def _nct(w):
        return w.shape[0], w.shape[1], (w.shape[2] * w.shape[3] if w.dim() == 4 else 1)

    def fwd(self, w, row_off=0, col_off=0, g=0):
        """[N, C(, kh, kw)] parameter -> rows row_off.., columns col_off + tap*C + c of this matrix."""
        n, c, t = self._nct(w)
        assert row_off + n <= self.rows and col_off + t * c <= self.cols, self.name
        self.pieces.append((0, w, None, n, c, t, row_off, col_off, 0, g))
        return self

    def bwd(self, w, col_off=0, npad=None):
        """[N, C(, kh, kw)] parameter -> rows c, columns col_off + tap*npad + n (the data-gradient operand)."""
        n, c, t = self._nct(w)
        npad = n if npad is None else npad
        assert c <= self.rows and col_off + t * npad <= self.cols, self.name
        self.pieces.append((1, w, None, n, c, t, 0, col_off, npad, 0))
        return self

    def vec(self, p, p2=None, off=0, g=0):
        self.pieces.append((2, p, p2, p.numel(), 1, 1, 0, off, 0, g))
        return self

    def host(self) -> torch.Tensor:
        """The packed operand in fp32, computed with torch on the parameters' device: the specification of
        ``wd_repack_multi``'s index mapping (used by the tests; the product path never calls it)."""
        first = self.pieces[0][1]
        out = torch.zeros((self.rows, self.cols) if self.planes else (int(torch.tensor(self.shape).prod()),),
                          dtype=torch.float32, device=first.device)
        for (mode, p, p2, n, c, t, row_off, col_off, npad, g) in self.pieces:
            src = p.detach().float()
            if mode == 2:
                v = src.reshape(-1) + (p2.detach().float().reshape(-1) if p2 is not None else 0)
                out[col_off:col_off + n] = geglu_interleave(v, g) if g else v
                continue
            w3 = src.reshape(n, c, t)
            if mode == 0:
                blk = w3.permute(0, 2, 1).reshape(n, t * c)
                out[row_off:row_off + n, col_off:col_off + t * c] = geglu_interleave(blk, g) if g else blk
            else:
                blk = torch.zeros(c, t, npad, dtype=torch.float32, device=first.device)
                blk[:, :, :n] = w3.permute(1, 2, 0)
                out[:c, col_off:col_off + t * npad] = blk.reshape(c, t * npad)
        return out if self.planes else out.reshape(self.shape)


class RecipeBook(dict):
    """name -> _Recipe.  Matrices become split-bf16 planes [2, rows, cols]; vectors stay fp32."""

    def matrix(self, name, rows, cols) -> _Recipe:
        self[name] = _Recipe(name, True, rows, cols)
        return self[name]

    def vector(self, name, p, p2=None, g=0) -> _Recipe:
        self[name] = _Recipe(name, False, shape=tuple(p.shape)).vec(p, p2, 0, g)
        return self[name]

    def vector_cat(self, name, ps) -> _Recipe:
        r = _Recipe(name, False, shape=(sum(p.numel() for p in ps),))
        off = 0
        for p in ps:
            r.vec(p, None, off)
            off += p.numel()
        self[name] = r
        return r

    def linear(self, name, lin):
        self.matrix(name + ".w", lin.weight.shape[0], lin.weight.shape[1]).fwd(lin.weight)
        if lin.bias is not None:
            self.vector(name + ".b", lin.bias)

    @staticmethod
    def frag_ok(r) -> bool:
        """Can ``wd_repack_multi`` write this matrix fragment-major itself (mode 3)?"""
        return r.planes and r.rows % 16 == 0 and r.cols % 32 == 0 and all(
            mode == 0 and c % 32 == 0 and t in (1, 9) and col_off % 2 == 0 and row_off >= 0
            for (mode, p, p2, n, c, t, row_off, col_off, npad, g) in r.pieces)

    def device_table(self, lib, dst: Dict[str, torch.Tensor], dev, frag: Optional[Dict[str, torch.Tensor]] = None):
        """Serialises the pieces into the entry table of ``wd_repack_multi`` (include/wdiff_hip.h).  ``frag``: name -> the
        fragment-major image to write as well (mode 3) for the matrices ``frag_ok`` accepts."""
        import struct
        assert lib.wd_repack_entry_bytes() == 80
        tile, vchunk = lib.wd_repack_tile(), lib.wd_repack_vchunk()
        recs, c0 = [], 0
        for name, r in self.items():
            d = dst[name]
            for (mode, p, p2, n, c, t, row_off, col_off, npad, g) in r.pieces:
                assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous(), name
                if mode == 2:
                    hi, lo, ld, ntc = d.data_ptr() + 4 * col_off, 0, 0, 1
                    nch = (n + vchunk - 1) // vchunk
                else:
                    ld = r.cols
                    o = 2 * (row_off * ld + col_off)
                    hi, lo = d[0].data_ptr() + o, d[1].data_ptr() + o
                    ntc = (c + tile - 1) // tile
                    nch = (((npad if mode == 1 else n) + tile - 1) // tile) * ntc
                recs.append(struct.pack("<QQQQiiiiiiiiqii", p.data_ptr(), p2.data_ptr() if p2 is not None else 0, hi, lo, n, c, t,
                                        mode, npad, ld, g, ntc, c0, 0, 0))
                c0 += nch
                if frag is not None and name in frag and self.frag_ok(r):
                    f = frag[name]
                    recs.append(struct.pack("<QQQQiiiiiiiiqii", p.data_ptr(), 0, f[0].data_ptr(), f[1].data_ptr(), n, c, t, 3, 0,
                                            r.rows, g, ntc, c0, row_off, col_off))
                    c0 += nch
        raw = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
        return raw, c0, len(recs)


class Act:
    """A token-major fp32 feature map [B*h*w, c] on the device."""
    __slots__ = ("t", "c", "h", "w", "stats", "prod", "perm")

    def __init__(self, t, c, h, w, stats=None, prod=None, perm=None):
        self.t, self.c, self.h, self.w = t, c, h, w
        self.perm = perm    # None, or int32 [h*w] on the device: the row (inside its sample) that holds raster position p - the
        #                     phase-major output of an Upsample (only the two-source GroupNorm of a decoder block reads such a map)
        self.stats = stats  # (part tensor [B, nchunk, c / part_cpg, 2] f64, nchunk, part_cpg) once known
        self.prod = prod    # the WdGemmArgs of the GEMM whose epilogue writes ``t``: a consumer that wants the operand planes of
        #                     this very tensor asks that epilogue for them instead of launching wd_split


class Plan:
    def __init__(self):
        self.cond: List[tuple] = []
        self.step: List[tuple] = []
        self.keep: List[object] = []
        self.out: Optional[torch.Tensor] = None

    @staticmethod
    def _run(ops, stream):
        for fn, args, what in ops:
            rc = fn(*args, stream)
            if rc != 0:
                N.check(rc, what)

    def run_cond(self, stream):
        self._run(self.cond, stream)

    def run_film(self, stream):
        """Per-call part of the tabulated FiLM path: the time MLP of every timestep.  The chunk of the table a step reads is
        brought in by ``film_prepare(t, stream)`` (a no-op closure when the plan has no table)."""
        self._run(getattr(self, "film", []), stream)
        self.film_loaded = -1  # writer ids / weights may have changed since the last call: no chunk is valid

    @staticmethod
    def film_prepare(t, stream):
        return False

    def run_step(self, stream):
        self._run(self.step, stream)


def _ptr(t: Optional[torch.Tensor], byte_off: int = 0):
    return None if t is None else t.data_ptr() + byte_off


class UNetEngine:
    """Builds and runs the launch plan for one model instance."""

    def __init__(self, model, variant: str):
        self.model = model
        self.variant = variant  # 'base' | 'phosc'
        self.lib = N.lib()
        self.npass = 3
        self._sig = None
        self._ps, self._ps_gen = None, -1
        self._pack = None
        self._w: Dict[str, torch.Tensor] = {}
        self._w3: Dict[str, torch.Tensor] = {}      # slab-order copies of the matrices the v3 kernel consumes
        self._w3_meta: Dict[str, tuple] = {}
        self._wf: Dict[str, torch.Tensor] = {}      # fragment-major copies of the matrices the weights-to-registers kernel reads
        # 64 x 320 tiles with the weights loaded straight into the MFMA operand registers (csrc/wd_gemmw.hip): the 320-column
        # layers whose grid fills the chip without a K cut
        self.use_wdirect = os.environ.get("WDIFF_GEMM_WDIRECT", "1") != "0"
        # Upsample as four 2x2 convolutions of the source map (one per output phase, 4 taps instead of 9; upsample_phase_tables)
        self.use_up_phases = os.environ.get("WDIFF_UPSAMPLE_PHASES", "1") != "0"
        self._derived: Dict[str, tuple] = {}   # name -> (persistent fp32 tensor, function that recomputes it from the parameters)
        self.use_smallmap = os.environ.get("WDIFF_GEMM_SMALLMAP", "1") != "0"  # 64 x 80 whole-K tiles for 3x3 layers over 64-position samples
        # GEGLU feed-forward + residual in one launch per 64-token panel, hidden activations on chip (csrc/wd_ff.hip)
        self.fuse_ff = os.environ.get("WDIFF_FUSE_FF", "1") != "0"
        self.fuse_proj = os.environ.get("WDIFF_FUSE_PROJ", "1") != "0"   # ... and proj_out + residual in the same launch
        # GroupNorm (+ SiLU) of a convolution's input applied while the weights-to-registers kernel stages its rows (no wd_gn_apply
        # launch, no operand planes of the normalised map)
        self.fuse_ln = os.environ.get("WDIFF_FUSE_LN", "1") != "0"  # LayerNorm of a 320-column GEMM result in its own epilogue
        self.fuse_gn_in = int(os.environ.get("WDIFF_FUSE_GN_IN", "1"))  # 0 off, 1 the 1x1 consumers, 2 the 3x3 consumers too
        self.use_slab = os.environ.get("WDIFF_SLAB", "0") != "0"
        self.fuse_stats = os.environ.get("WDIFF_FUSE_STATS", "1") != "0"
        self.fuse_xattn = os.environ.get("WDIFF_FUSE_XATTN", "1") != "0"
        # row-shared-taps convolution kernel (29 % fewer DMA pieces, but measured 15 % slower than the generic kernel so far)
        self.use_conv3 = os.environ.get("WDIFF_CONV3", "0") != "0"
        if (self.use_slab or self.use_conv3) and not self.lib.wd_gemm_experimental():
            raise N.NativeError("WDIFF_SLAB / WDIFF_CONV3 need a library built with WDIFF_EXPERIMENTAL=1 "
                                "(python -m worddiffusion_amd.build --force)")
        self.fuse_xattn_pair = os.environ.get("WDIFF_FUSE_XATTN_PAIR", "1") != "0"
        self.fuse_out = os.environ.get("WDIFF_FUSE_OUT", "1") != "0"
        self.pack_kv = os.environ.get("WDIFF_PACK_KV", "1") != "0"         # long-context cross-attention: K/V images built once per call
        self.fuse_gn2 = os.environ.get("WDIFF_FUSE_GN2", "1") != "0"       # GroupNorm over [h | skip]: one apply launch for both
        self.fuse_gn = os.environ.get("WDIFF_FUSE_GN", "1") != "0"         # GroupNorm in the producer's split-K combine launch
        self.fuse_split = os.environ.get("WDIFF_FUSE_SPLIT", "1") != "0"   # resample inputs: planes from the producer's epilogue
        self._plans: Dict[tuple, Plan] = {}
        self._tabs: Dict[tuple, torch.Tensor] = {}
        self._tab_np: Dict[int, np.ndarray] = {}
        self._same_w: Dict[int, int] = {}   # gather table pointer -> image width, for the 3x3 / pad 1 / stride 1 tables
        self._down_w: Dict[int, int] = {}   # ... -> OUTPUT image width, for the 3x3 / pad 1 / stride 2 tables
        self._ws = None
        self.device = None

    # ------------------------------------------------------------------------------------------ weights
    def set_precision(self, mode: str):
        npass = {"bf16x3": 3, "fp32": 3, "bf16": 1}[mode]
        if npass != self.npass:
            self.npass = npass
            self._plans.clear()

    def _recipes(self) -> "RecipeBook":
        """Declarative description of every operand the kernels consume: which parameter (reference layout) goes where in
        which packed matrix / vector.  ``refresh_weights`` turns it into the device table of ``wd_repack_multi``."""
        m = self.model
        R = RecipeBook()
        film_w = []
        self.film_off = {}
        off = 0
        cin_conv = m.input_blocks[0][0]
        self.kpad_in = ((9 * cin_conv.in_channels + 31) // 32) * 32
        R.matrix("in.w", cin_conv.out_channels, self.kpad_in).fwd(cin_conv.weight)
        R.vector("in.b", cin_conv.bias)
        R.linear("te0", m.time_embed[0])
        R.linear("te2", m.time_embed[2])
        if m.num_classes is not None:
            R.vector("label", m.label_emb.weight)
        we = m.word_emb
        cd = we.embedding.weight.shape[1]
        R.vector("we.table", we.embedding.weight)
        qkv = (we.attention.linear_query, we.attention.linear_key, we.attention.linear_value)
        R.matrix("we.qkv.w", 3 * cd, cd)
        R.vector_cat("we.qkv.b", [l.bias for l in qkv])
        for i, l in enumerate(qkv):
            R["we.qkv.w"].fwd(l.weight, row_off=i * cd)
        kv_w = []
        self.kv_off = {}
        kvo = 0
        for name, mod in self._walk():
            if isinstance(mod, ResBlockParams):
                R.vector(name + ".gn1.g", mod.in_layers[0].weight)
                R.vector(name + ".gn1.b", mod.in_layers[0].bias)
                R.matrix(name + ".c1.w", mod.cout, 9 * mod.cin).fwd(mod.in_layers[2].weight)
                R.vector(name + ".c1.b", mod.in_layers[2].bias)
                R.vector(name + ".gn2.g", mod.out_layers[0].weight)
                R.vector(name + ".gn2.b", mod.out_layers[0].bias)
                if mod.cin != mod.cout:
                    # the 1x1 skip projection is a second K segment of the last 3x3 GEMM; the biases add
                    R.matrix(name + ".c2.w", mod.cout, 9 * mod.cout + mod.cin).fwd(mod.out_layers[3].weight) \
                        .fwd(mod.skip_connection.weight, col_off=9 * mod.cout)
                    R.vector(name + ".c2.b", mod.out_layers[3].bias, mod.skip_connection.bias)
                else:
                    R.matrix(name + ".c2.w", mod.cout, 9 * mod.cout).fwd(mod.out_layers[3].weight)
                    R.vector(name + ".c2.b", mod.out_layers[3].bias)
                film_w.append(mod.emb_layers[1])
                self.film_off[name] = off
                off += mod.cout
            elif isinstance(mod, (DownsampleParams, UpsampleParams)):
                conv = mod.op if isinstance(mod, DownsampleParams) else mod.conv
                R.matrix(name + ".w", mod.cout, 9 * mod.cin).fwd(conv.weight)
                R.vector(name + ".b", conv.bias)
                if isinstance(mod, UpsampleParams) and self._up_phases_ok(mod):
                    # the four phase matrices [cout][4 taps x cin] from a derived tensor (the summed taps), refreshed before every repack
                    key = name + ".wph"
                    if key not in self._derived or self._derived[key][0].device != conv.weight.device:
                        self._derived[key] = (torch.empty((4, 4, mod.cout, mod.cin), dtype=torch.float32, device=conv.weight.device),
                                              (lambda cv=conv: upsample_phase_weights(cv.weight)))
                    wph = self._derived[key][0]
                    for ph in range(4):
                        r = R.matrix(f"{name}.wph{ph}", mod.cout, 4 * mod.cin)
                        for t in range(4):
                            r.fwd(wph[ph, t], col_off=t * mod.cin)
            elif isinstance(mod, SpatialTransformerParams):
                inner = mod.heads * mod.d_head
                R.vector(name + ".gn.g", mod.norm.weight)
                R.vector(name + ".gn.b", mod.norm.bias)
                R.matrix(name + ".pi.w", inner, mod.ch).fwd(mod.proj_in.weight)
                R.vector(name + ".pi.b", mod.proj_in.bias)
                R.matrix(name + ".po.w", mod.ch, inner).fwd(mod.proj_out.weight)
                R.vector(name + ".po.b", mod.proj_out.bias)
                for d, tb in enumerate(mod.transformer_blocks):
                    p = f"{name}.tb{d}"
                    for ln in ("norm1", "norm2", "norm3"):
                        R.vector(f"{p}.{ln}.g", getattr(tb, ln).weight)
                        R.vector(f"{p}.{ln}.b", getattr(tb, ln).bias)
                    if self.variant == "phosc":
                        R.matrix(p + ".a1.qkv.w", 3 * inner, inner)
                        for i, l in enumerate((tb.attn1.to_q, tb.attn1.to_k, tb.attn1.to_v)):
                            R[p + ".a1.qkv.w"].fwd(l.weight, row_off=i * inner)
                        cross = [("a2", tb.attn2)]
                    else:
                        R.matrix(p + ".a1.q.w", inner, inner).fwd(tb.attn1.to_q.weight)
                        cross = [("a1", tb.attn1), ("a2", tb.attn2)]
                    R.matrix(p + ".a2.q.w", inner, inner).fwd(tb.attn2.to_q.weight)
                    for tag, at in cross:
                        kv_w.append(at)
                        self.kv_off[f"{p}.{tag}"] = kvo
                        kvo += 2 * at.to_k.weight.shape[0]
                        # fp32 copies for the folded cross-attention (wd_xattn_fold)
                        R.vector(f"{p}.{tag}.q.f32", at.to_q.weight)
                        R.vector(f"{p}.{tag}.o.f32", at.to_out[0].weight)
                    for tag, at in (("a1", tb.attn1), ("a2", tb.attn2)):
                        R.linear(f"{p}.{tag}.o", at.to_out[0])
                    ffi = tb.ff.net[2].in_features
                    g = geglu_tile(ffi) % 1000 // 2
                    R.matrix(p + ".ff1.w", 2 * ffi, inner).fwd(tb.ff.net[0].proj.weight, g=g)
                    R.vector(p + ".ff1.b", tb.ff.net[0].proj.bias, g=g)
                    R.linear(p + ".ff2", tb.ff.net[2])
                    if ff_fused_supported(inner, ffi):
                        # the fused feed-forward (csrc/wd_ff.hip): x | gate rows in blocks of 16, one MFMA tile each
                        R.matrix(p + ".ff1f.w", 2 * ffi, inner).fwd(tb.ff.net[0].proj.weight, g=16)
                        R.vector(p + ".ff1f.b", tb.ff.net[0].proj.bias, g=16)
        self.film_total = off
        self.kv_total = kvo
        ted = m.time_embed[2].out_features
        R.matrix("film.w", off, ted)
        R.vector_cat("film.b", [l.bias for l in film_w])
        r0 = 0
        for l in film_w:
            R["film.w"].fwd(l.weight, row_off=r0)
            r0 += l.weight.shape[0]
        if kv_w:
            R.matrix("kv.w", kvo, kv_w[0].to_k.weight.shape[1])
            r0 = 0
            for at in kv_w:
                for l in (at.to_k, at.to_v):
                    R["kv.w"].fwd(l.weight, row_off=r0)
                    r0 += l.weight.shape[0]
        self._film_mods, self._kv_mods = film_w, kv_w
        R.vector("out.gn.g", m.out[0].weight)
        R.vector("out.gn.b", m.out[0].bias)
        R.matrix("out.w", m.out[2].out_channels, 9 * m.out[2].in_channels).fwd(m.out[2].weight)
        R.vector("out.b", m.out[2].bias)
        R.vector("out.w.f32", m.out[2].weight)  # the fused GroupNorm + SiLU + 3x3 kernel reads the parameter layout itself
        return R

    def _walk(self):
        """(name, module) of every block layer in execution order."""
        m = self.model
        for i, blk in enumerate(m.input_blocks):
            for j, mod in enumerate(blk):
                yield f"in{i}.{j}", mod
        for j, mod in enumerate(m.middle_block):
            yield f"mid.{j}", mod
        for i, blk in enumerate(m.output_blocks):
            for j, mod in enumerate(blk):
                yield f"out{i}.{j}", mod

    def _signature(self):
        # the module-tree walk of ``model.parameters()`` costs ~0.35 ms for the 264 tensors, 10x the rest of this function:
        # the list is kept until some module of the process registers a Parameter (object identity can only change then)
        if self._ps is None or self._ps_gen != _PARAM_GEN[0]:
            self._ps, self._ps_gen = list(self.model.parameters()), _PARAM_GEN[0]
        ps = self._ps
        return (sum(p._version for p in ps) + (_NATIVE_WRITES[0] << 32), hash(tuple(p.data_ptr() for p in ps)), str(ps[0].device))

    def refresh_weights(self, force: bool = False):
        """Re-derives the packed operands from the parameters when they changed (optimiser step, load_state_dict,
        .to(device)): one ``wd_repack_multi`` launch over a device table built once per parameter placement."""
        sig = self._signature()
        if not force and sig == self._sig:
            return
        dev = next(self.model.parameters()).device
        if dev.type != "cuda":
            raise N.NativeError("worddiffusion_amd runs on an MI355X only: move the model to cuda "
                                "(there is no CPU / eager fallback)")
        if self.device is not None and dev != self.device:
            self._w.clear()
            self._w3.clear()
            self._wf.clear()
            self._w3_meta.clear()
            self._plans.clear()
            self._tabs.clear()
            self._tab_np.clear()
            self._ws = None
            self._pack = None
        self.device = dev
        if self._pack is None or self._pack[0] != sig[1:]:
            book = self._recipes()
            for p in self.model.parameters():
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise N.NativeError("parameters must be contiguous fp32 (the reference never retypes them, unet.py:415)")
            for name, r in book.items():
                if name not in self._w:
                    self._w[name] = (torch.zeros((2, r.rows, r.cols), dtype=torch.bfloat16, device=dev) if r.planes
                                     else torch.zeros(r.shape, dtype=torch.float32, device=dev))
            table, chunks, n = book.device_table(self.lib, self._w, dev, self._wf)
            # (the fragment-major images the table does not write itself are re-made from the planes below)
            self._pack = (sig[1:], table, chunks, n, [k for k in self._wf if k not in book or not book.frag_ok(book[k])])
        _, table, chunks, n, wf_left = self._pack
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.no_grad():
            for buf, fn in self._derived.values():
                buf.copy_(fn())
        N.check(self.lib.wd_repack_multi(table.data_ptr(), n, chunks, stream), "wd_repack_multi")
        for name in wf_left:
            self._pack_wf(name, self._wf[name], stream)
        with torch.no_grad():
            for name, meta in self._w3_meta.items():
                self._w3[name].copy_(slab_order(self._w[name], *meta))
            if "freqs" not in self._w and hasattr(self.model, "word_emb"):  # (the VAE decoder engine has neither)
                half = self.model.model_channels // 2
                freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
                self._w["freqs"] = freqs.to(dev)
                self._w["pe"] = self.model.word_emb.positional_encoding.to(dev).contiguous()
        self._sig = sig

    def _pack_wf(self, name, wf, stream):
        wp = self._w[name]
        N.check(self.lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), wp.shape[1], wp.shape[2], wf[0].data_ptr(),
                                        wf[1].data_ptr(), stream), "wd_gemm_pack_w")

    # ------------------------------------------------------------------------------------------ helpers
    def _table(self, h, w, mode):
        key = (h, w, mode)
        if key not in self._tabs:
            tab, ho, wo = conv_gather_table(h, w, mode)
            dt = torch.from_numpy(tab).to(self.device)
            self._tabs[key] = (dt, ho, wo)
            self._tab_np[dt.data_ptr()] = tab
            if mode == "down":
                self._down_w[dt.data_ptr()] = wo
            if mode == "same":
                self._same_w[dt.data_ptr()] = w
        return self._tabs[key]

    def _f32(self, P: Plan, *shape):
        t = torch.empty(shape, dtype=torch.float32, device=self.device)
        P.keep.append(t)
        return t

    def _planes(self, P: Plan, rows, ld):
        t = torch.zeros((2, rows, ld), dtype=torch.bfloat16, device=self.device)
        P.keep.append(t)
        return t

    def _src(self, planes: torch.Tensor, c: int, ntaps: int = 1, gather: Optional[torch.Tensor] = None,
             hw_src: int = 0, col_off: int = 0) -> N.WdSrc:
        s = N.WdSrc()
        s._tab_np = self._tab_np.get(gather.data_ptr()) if gather is not None else None
        s._same_w = self._same_w.get(gather.data_ptr(), 0) if gather is not None else 0
        s._down_w = self._down_w.get(gather.data_ptr(), 0) if gather is not None else 0
        ld = planes.shape[2]
        s.hi = planes[0].data_ptr() + 2 * col_off
        s.lo = planes[1].data_ptr() + 2 * col_off
        s.gather = _ptr(gather)
        s.ld, s.c, s.ntaps, s.hw_src = ld, c, ntaps, hw_src
        return s

    def _gemm(self, ops, what, srcs, wname, m, hw_out, bias=None, rowvec=None, rowvec_ld=0, resid=None,
              resid_ld=0, resid_rows=None, act=N.ACT_NONE, out_f32=None, out_ld=0, out_pl=None, n=None, tile=0,
              w_row_off=0, want_stats=False, a32=None, ln=None):
        """a32 = (Act, norm name, eps, silu): src[0] is that fp32 map, normalised while it is staged (wd_gemm_args.a32*)."""
        a = N.WdGemmArgs()
        for i, s in enumerate(srcs):
            a.src[i] = s
        a.nsrc = len(srcs)
        a.npass = self.npass
        wp = self._w[wname]
        ktot = wp.shape[2]
        assert ktot == sum(s.ntaps * s.c for s in srcs), (what, ktot, [(s.ntaps, s.c) for s in srcs])
        nrows = wp.shape[1] if n is None else n
        span = 0
        if self.use_slab and w_row_off == 0 and (len(srcs) == 1 or (srcs[1].ntaps == 1 and not srcs[1].gather)):
            tabnp = getattr(srcs[0], "_tab_np", None)
            span = slab_span(tabnp, hw_out, srcs[0].hw_src, m)
            # the slab kernel runs 128-row panels: only worth it when they fill the chip
            if span > 192 or ((m + 127) // 128) * max(1, nrows // 160) < 96:
                span = 0
        wdirect = (self.use_wdirect and not span and not self.use_conv3 and w_row_off == 0 and n is None and act == N.ACT_NONE and
                   tile == 0 and nrows % 320 == 0 and ktot % 64 == 0 and all(s.c % 64 == 0 for s in srcs) and srcs[0].ntaps <= 9 and
                   (len(srcs) == 1 or (srcs[1].ntaps == 1 and not srcs[1].gather)) and
                   # (the K-cut layers of the 4 x 16 level stay on the LDS-staged kernel: this one is 5 % faster on their long loops
                   # in isolation and 2 % slower inside the step)
                   ((m + 63) // 64) * (nrows // 320) >= 256)
        # 3x3 layers over 64-position samples (the 4 x 16 level): 64 x 80 tiles with all of K inside the workgroup (wd_gemmq_kernel) instead
        # of a K cut over workgroups + combine launch
        s0 = srcs[0]
        sm_down = s0.ntaps == 9 and getattr(s0, "_down_w", 0) == 16 and s0.hw_src == 256   # the stride-2 convolution 8x32 -> 4x16
        sm_conv = (s0.ntaps == 9 and getattr(s0, "_same_w", 0) in (16, 32) and s0.hw_src == 64) or sm_down
        sm_ident = s0.ntaps == 1 and not s0.gather
        smallmap = (self.use_smallmap and not wdirect and not span and not self.use_conv3 and a32 is None and ln is None and
                    w_row_off == 0 and n is None and act == N.ACT_NONE and tile == 0 and self.npass == 3 and hw_out == 64 and
                    (sm_conv or sm_ident) and (len(srcs) == 1 or (srcs[1].ntaps == 1 and not srcs[1].gather)) and
                    m % 64 == 0 and nrows % 80 == 0 and all(q.c % 64 == 0 for q in srcs) and not resid_rows and
                    128 <= (m // 64) * (nrows // 80) <= 512)
        assert a32 is None or wdirect, what
        if a32 is not None:
            x32, gname, eps, silu = a32
            part, nchunk, pc = x32.stats
            a.a32, a.a32_ld, a.a32_part = x32.t.data_ptr(), x32.c, part.data_ptr()
            a.a32_nchunk, a.a32_pcpg, a.a32_cpg = nchunk, pc, x32.c // 32
            a.a32_gamma, a.a32_beta = self._w[gname + ".g"].data_ptr(), self._w[gname + ".b"].data_ptr()
            a.a32_eps, a.a32_silu = float(eps), int(silu)
        if smallmap:
            wf = self._wfrag(wname)
            a.w_hi, a.w_lo = wf[0].data_ptr(), wf[1].data_ptr()
            a.w_layout, a.slab_rows = 3, (16 if sm_down else srcs[0]._same_w if sm_conv else 0)
            tile = 64080
        elif wdirect:
            wf = self._wfrag(wname)
            a.w_hi, a.w_lo = wf[0].data_ptr(), wf[1].data_ptr()
            a.w_layout, a.slab_rows = 3, getattr(srcs[0], "_same_w", 0) if srcs[0].ntaps == 9 else 0
            tile = 64320
        elif span:
            meta = (srcs[0].ntaps, srcs[0].c, srcs[1].c if len(srcs) > 1 else 0)
            if wname not in self._w3:
                self._w3[wname] = slab_order(wp, *meta)
                self._w3_meta[wname] = meta
            assert self._w3_meta[wname] == meta, wname
            w3 = self._w3[wname]
            a.w_hi, a.w_lo = w3[0].data_ptr(), w3[1].data_ptr()
            a.w_layout, a.slab_rows = 1, span
            tile = 0 if tile == 0 else tile
        else:
            a.w_hi = wp[0].data_ptr() + 2 * w_row_off * ktot
            a.w_lo = wp[1].data_ptr() + 2 * w_row_off * ktot
            same_w = getattr(srcs[0], "_same_w", 0)  # (a hint: lets the kernel compute the source-row table; WDIFF_CONV3=1 also
            #                                           selects the row-shared-taps kernel for it)
            if (same_w and srcs[0].ntaps == 9 and srcs[0].c % 64 == 0 and act == N.ACT_NONE and tile == 0 and
                    (len(srcs) == 1 or (srcs[1].ntaps == 1 and not srcs[1].gather and srcs[1].c % 64 == 0))):
                a.w_layout, a.slab_rows = 2, same_w  # 3x3 same-convolution: the taps of a kernel row share their A tile
        a.m, a.n, a.ktot, a.hw_out = m, nrows, ktot, hw_out
        a.bias = _ptr(bias)
        a.rowvec, a.rowvec_ld = rowvec, rowvec_ld
        a.resid, a.resid_ld, a.resid_rows = resid, resid_ld, resid_rows
        a.act = act
        a.out_f32, a.out_ld = _ptr(out_f32), out_ld
        if out_pl is not None:
            a.out_hi, a.out_lo, a.out_pl_ld = out_pl[0].data_ptr(), out_pl[1].data_ptr(), out_pl.shape[2]
        a.tile = tile
        if ln is not None:  # out_pl receives LayerNorm(result) * gamma + beta (name of the norm's packed vectors)
            assert wdirect and nrows == 320 and out_pl is not None, what
            a.ln_gamma, a.ln_beta, a.ln_eps = self._w[ln + ".g"].data_ptr(), self._w[ln + ".b"].data_ptr(), 1e-5
        stats = None
        if want_stats and self.fuse_stats and nrows % 32 == 0 and (hw_out % 128 == 0 or hw_out == 64):
            cpg = nrows // 32
            bn = (tile % 1000) if tile else (160 if nrows % 160 == 0 else 64)
            if bn % cpg == 0 and (tile == 0 or tile // 1000 == 128 or wdirect or smallmap) and not span:
                nchunk = max(1, hw_out // (64 if (wdirect or smallmap) else 128))  # (the statistics are kept per row panel of the tile)
                part = torch.zeros((m // hw_out, nchunk, 32, 2), dtype=torch.float64, device=self.device)
                self._cur_plan.keep.append(part)
                a.stat_part, a.stat_cpg = part.data_ptr(), cpg
                stats = (part, nchunk, cpg)
        a._stats = stats
        if self._ws is None:
            self._ws = torch.empty(128 * 128 * 160 * 8, dtype=torch.float32, device=self.device)  # 84 MB split-K scratch
        a.ksplit, a.ws, a.ws_floats = 0, self._ws.data_ptr(), self._ws.numel()
        self._give_tickets(a, m, nrows)
        self._cur_plan.keep.append(a)
        ops.append((self.lib.wd_gemm, (C.byref(a),), what))
        return a

    def _give_tickets(self, a, m, n):
        """Arrival counters for the in-launch split-K combine of one wd_gemm launch: a range of its own inside a zeroed
        per-plan buffer (zero before and after every launch; only small grids ever split, so only they get one)."""
        ntick = ((m + 63) // 64) * ((n + 63) // 64)
        if ntick > 2048:
            return
        P = self._cur_plan
        tk = getattr(P, "_tickets", None)
        if tk is None or P._ticket_off + ntick > tk.numel():
            tk = torch.zeros(1 << 16, dtype=torch.int32, device=self.device)
            P.keep.append(tk)
            P._tickets, P._ticket_off = tk, 0
        a.tickets, a.ntickets = tk.data_ptr() + 4 * P._ticket_off, ntick
        P._ticket_off += ntick

    def _gn(self, P, ops, what, srcs: List[Act], gname, eps, silu, want_raw=False):
        """GroupNorm over the channel concat of ``srcs`` -> planes [M, sum c] (+ raw planes)."""
        B = self._B
        h, w = srcs[0].h, srcs[0].w
        hw = h * w
        ctot = sum(s.c for s in srcs)
        cpg = ctot // 32
        has_perm = any(s.perm is not None for s in srcs)
        if has_perm and not (len(srcs) == 2 and srcs[1].perm is None and self.fuse_gn2 and not any(s.c % cpg for s in srcs)):
            raise NotImplementedError(f"{what}: a phase-major map (Upsample) is read by the two-source GroupNorm of a decoder block only")
        if any(s.c % cpg for s in srcs):
            # groups straddle the concat boundary: materialise the concat (never the case at 320+320 channels)
            cat = self._f32(P, B * hw, ctot)
            coff = 0
            for s in srcs:
                ops.append((self.lib.wd_copy2d, (cat.data_ptr() + 4 * coff, 4 * ctot, s.t.data_ptr(), 4 * s.c, 4 * s.c,
                                                 B * hw), what + ":concat"))
                coff += s.c
            srcs = [Act(cat, ctot, h, w)]
        pl = self._planes(P, B * hw, ctot)
        raw = self._planes(P, B * hw, ctot) if want_raw else None
        for s in srcs:
            if s.stats is None:  # no producer-side statistics: one pass over the tensor (32 groups of c/32 channels)
                nchunk = self.lib.wd_gn_nchunk(hw)
                pc = s.c // 32
                part = torch.empty((B, nchunk, 32, 2), dtype=torch.float64, device=self.device)
                P.keep.append(part)
                ops.append((self.lib.wd_gn_stats, (s.t.data_ptr(), s.c, B, hw, s.c, pc, part.data_ptr()),
                            what + ":stats"))
                s.stats = (part, nchunk, pc)
        coff = 0
        gam, bet = self._w[gname + ".g"], self._w[gname + ".b"]
        todo = []  # (source, part, nchunk, pc, coff) of the sources that need an apply launch
        for s in srcs:
            part, nchunk, pc = s.stats
            assert cpg % pc == 0, (what, cpg, pc)
            if nchunk > 8:
                # many chunks per sample (large images): fold them once here, not in every workgroup of wd_gn_apply
                ngs = s.c // pc
                folded = torch.empty((B, 1, ngs, 2), dtype=torch.float64, device=self.device)
                P.keep.append(folded)
                ops.append((self.lib.wd_gn_fold_chunks, (part.data_ptr(), B, nchunk, ngs, folded.data_ptr()), what + ":fold chunks"))
                part, nchunk = folded, 1
                s.stats = (part, nchunk, pc)
            if not has_perm and self._gn_in_combine(s, raw, hw, cpg, pc, nchunk, coff):
                # the producer is a K-cut GEMM whose combine tiles (64 rows x 40 columns) hold whole (sample, group) blocks: its
                # combine launch normalises the rows it has just summed and writes these planes (wd_gemm_args.gn_*)
                pr = s.prod
                pr.gn_gamma, pr.gn_beta = gam.data_ptr() + 4 * coff, bet.data_ptr() + 4 * coff
                pr.gn_eps, pr.gn_silu, pr.gn_cpg = float(eps), int(silu), cpg
                pr.out_hi = pl[0].data_ptr() + 2 * coff
                pr.out_lo = (pl[1].data_ptr() + 2 * coff) if self.npass == 3 else None
                pr.out_pl_ld = ctot
            else:
                todo.append((s, part, nchunk, pc, coff))
            coff += s.c
        lo = pl[1].data_ptr() if self.npass == 3 else None
        rhi = raw[0].data_ptr() if raw is not None else None
        rlo = raw[1].data_ptr() if (raw is not None and self.npass == 3) else None
        if len(todo) == 2 and self.fuse_gn2:  # [h | skip] of a decoder block: one launch for both halves of the concat
            (sa, pa, na, pca, ca), (sb, pb, nb_, pcb, cb) = todo
            ops.append((self.lib.wd_gn_apply2,
                        (sa.t.data_ptr(), sa.c, sa.c, pa.data_ptr(), na, pca, ca, sb.t.data_ptr(), sb.c, sb.c, pb.data_ptr(), nb_, pcb, cb,
                         B, hw, cpg, gam.data_ptr(), bet.data_ptr(), eps, int(silu), pl[0].data_ptr(), lo, ctot, rhi, rlo,
                         sa.perm.data_ptr() if sa.perm is not None else None),
                        what + ":apply"))
        else:
            assert not has_perm, what
            for s, part, nchunk, pc, c0 in todo:
                ops.append((self.lib.wd_gn_apply,
                            (s.t.data_ptr(), s.c, B, hw, s.c, cpg, part.data_ptr(), nchunk, pc, gam.data_ptr(), bet.data_ptr(), eps,
                             int(silu), pl[0].data_ptr(), lo, ctot, c0, rhi, rlo), what + ":apply"))
        return pl, raw

    def _gn_in_consumer(self, P, ops, what, srcs: List[Act], want_raw, M, hw, ncols, taps=1) -> bool:
        """Can the convolution that consumes this GroupNorm apply it itself (wd_gemm_args.a32*)?  One fp32 source with known (or
        computable) statistics, a consumer that takes the 64 x 320 weights-to-registers kernel, tiles inside one sample.
        Measured at B = 64: a 1x1 consumer (proj_in: five stages, no SiLU) loses nothing and saves the wd_gn_apply launch; a 3x3
        consumer normalises and activates every element nine times (once per tap) and runs ~30 us longer per launch than the
        10 us launch it saves (step 2.095 -> 2.137 ms with all of them on) - WDIFF_FUSE_GN_IN=2 switches those on anyway."""
        if not (self.fuse_gn_in and self.use_wdirect and self.npass == 3 and len(srcs) == 1 and not want_raw and not self.use_conv3):
            return False
        if taps > 1 and self.fuse_gn_in < 2:
            return False
        s = srcs[0]
        if s.c % 64 or s.c > 1024 or hw % 64 or ncols % 320 or ((M + 63) // 64) * (ncols // 320) < 256:
            return False
        if s.stats is None:  # no producer-side statistics: one pass over the tensor
            nchunk = self.lib.wd_gn_nchunk(hw)
            pc = s.c // 32
            part = torch.empty((self._B, nchunk, 32, 2), dtype=torch.float64, device=self.device)
            P.keep.append(part)
            ops.append((self.lib.wd_gn_stats, (s.t.data_ptr(), s.c, self._B, hw, s.c, pc, part.data_ptr()), what + ":stats"))
            s.stats = (part, nchunk, pc)
        return (s.c // 32) % s.stats[2] == 0

    def _ln_in_producer(self, M, ncols) -> bool:
        """Does a 1x1 / linear GEMM with these output dimensions run as 64 x 320 weights-to-registers tiles holding whole rows
        (so that its epilogue can emit the next LayerNorm, wd_gemm_args.ln_*)?"""
        return (self.fuse_ln and self.use_wdirect and not self.use_slab and not self.use_conv3 and ncols == 320 and
                (M + 63) // 64 >= 256)

    def _src32(self, x: Act, ntaps=1, gather=None, hw_src=0) -> N.WdSrc:
        """src[0] of a GEMM that reads the fp32 map itself (a32): channel count, taps and gather table only."""
        s = N.WdSrc()
        s._tab_np = self._tab_np.get(gather.data_ptr()) if gather is not None else None
        s._same_w = self._same_w.get(gather.data_ptr(), 0) if gather is not None else 0
        s.hi, s.lo, s.gather = None, None, _ptr(gather)
        s.ld, s.c, s.ntaps, s.hw_src = x.c, x.c, ntaps, hw_src
        return s

    def _gn_in_combine(self, s: Act, raw, hw, cpg, pc, nchunk, coff) -> bool:
        """Can the GEMM that produced ``s`` apply this GroupNorm in its split-K combine launch (wd_gemm_args.gn_*)?  The
        conditions of include/wdiff_hip.h, checked here so that the plan never asks for what wd_gemm would refuse."""
        pr = s.prod
        if pr is None or not isinstance(pr, N.WdGemmArgs) or not self.fuse_gn or raw is not None or self.use_conv3:
            return False
        if pr.out_f32 != s.t.data_ptr() or pr.out_hi or pr.gn_gamma or pr.n != s.c or not pr.stat_part or not pr.ws:
            return False
        if hw != 64 or pr.hw_out != 64 or pr.m % 64 or s.c % 160 or 40 % cpg or cpg % pc or nchunk != 1 or coff % 4:
            return False
        if pr.act != N.ACT_NONE or pr.resid_rows or pr.ksplit != 0 or pr.w_layout == 1 or pr.dbg:
            return False
        if pr.tile == 64080:  # all of K in the producer's workgroups (wd_gemmq_kernel): the norm runs in its own epilogue
            return pr.w_layout == 3
        if pr.tile != (64320 if pr.w_layout == 3 else 0):
            return False
        return self.lib.wd_gemm_auto_ksplit(pr.m, pr.n, pr.ktot, pr.ws_floats) > 1

    def _ln(self, P, ops, what, x: torch.Tensor, rows, c, name):
        pl = self._planes(P, rows, c)
        ops.append((self.lib.wd_layernorm,
                    (x.data_ptr(), c, rows, c, self._w[name + ".g"].data_ptr(), self._w[name + ".b"].data_ptr(), 1e-5,
                     pl[0].data_ptr(), pl[1].data_ptr() if self.npass == 3 else None, c), what))
        return pl

    # ------------------------------------------------------------------------------------------ blocks
    def _resblock(self, P, name, mod: ResBlockParams, srcs: List[Act]) -> Act:
        ops = P.step
        B = self._B
        h, w = srcs[0].h, srcs[0].w
        hw, M = h * w, B * h * w
        cin, cout = mod.cin, mod.cout
        assert cin == sum(s.c for s in srcs)
        tab, _, _ = self._table(h, w, "same")
        need_raw = cin != cout
        h1 = self._f32(P, M, cout)
        if self._gn_in_consumer(P, ops, name + ".gn1", srcs, need_raw, M, hw, cout, taps=9):
            s1, in1 = self._src32(srcs[0], 9, tab, hw), (srcs[0], name + ".gn1", 1e-5, True)
            raw = None
        else:
            a1, raw = self._gn(P, ops, name + ".gn1", srcs, name + ".gn1", 1e-5, True, want_raw=need_raw)
            s1, in1 = self._src(a1, cin, 9, tab, hw), None
        g1 = self._gemm(ops, name + ".conv1", [s1], name + ".c1.w", M, hw,
                        bias=self._w[name + ".c1.b"], rowvec=self._film.data_ptr() + 4 * self.film_off[name],
                        rowvec_ld=self.film_total, out_f32=h1, out_ld=cout, want_stats=True, a32=in1)
        h1a = Act(h1, cout, h, w, g1._stats, prod=g1)
        if self._gn_in_consumer(P, ops, name + ".gn2", [h1a], False, M, hw, cout, taps=9):
            s2, in2 = self._src32(h1a, 9, tab, hw), (h1a, name + ".gn2", 1e-5, True)
        else:
            a2, _ = self._gn(P, ops, name + ".gn2", [h1a], name + ".gn2", 1e-5, True)
            s2, in2 = self._src(a2, cout, 9, tab, hw), None
        out = self._f32(P, M, cout)
        if need_raw:
            g2 = self._gemm(ops, name + ".conv2+skip", [s2, self._src(raw, cin)],
                            name + ".c2.w", M, hw, bias=self._w[name + ".c2.b"], out_f32=out, out_ld=cout,
                            want_stats=True, a32=in2)
        else:
            g2 = self._gemm(ops, name + ".conv2", [s2], name + ".c2.w", M, hw,
                            bias=self._w[name + ".c2.b"], resid=srcs[0].t.data_ptr(), resid_ld=cout, out_f32=out,
                            out_ld=cout, want_stats=True, a32=in2)
        return Act(out, cout, h, w, g2._stats, prod=g2)

    def _up_phases_ok(self, mod) -> bool:
        """Shapes the phase form of an Upsample covers (the 64 x 320 weights-to-registers kernel with weight groups)."""
        return (getattr(self, "use_up_phases", False) and self.use_wdirect and self.fuse_gn2 and self.npass == 3 and not self.use_slab and
                not self.use_conv3 and mod.cin % 64 == 0 and mod.cout % 320 == 0)

    def _resample(self, P, name, mod, x: Act, mode: str, tile: int = 0) -> Act:
        ops = P.step
        B = self._B
        assert x.perm is None, name
        tab, ho, wo = self._table(x.h, x.w, mode)
        pl = self._planes(P, B * x.h * x.w, x.c)
        pr = x.prod
        if (pr is not None and self.fuse_split and not pr.out_hi and pr.out_f32 == x.t.data_ptr() and
                (pr.n if isinstance(pr, N.WdGemmArgs) else pr.c) == x.c):
            # the producing GEMM writes the planes beside its fp32 output (one launch and one pass over the tensor less)
            pr.out_hi, pr.out_lo = pl[0].data_ptr(), (pl[1].data_ptr() if self.npass == 3 else None)
            pr.out_pl_ld = x.c
        else:
            ops.append((self.lib.wd_split, (x.t.data_ptr(), x.c, B * x.h * x.w, x.c, 0, pl[0].data_ptr(),
                                            pl[1].data_ptr() if self.npass == 3 else None, x.c), name + ":split"))
        out = self._f32(P, B * ho * wo, mod.cout)
        hw = x.h * x.w
        if (mode == "up" and tile == 0 and (name + ".wph0") in self._w and self._up_phases_ok(mod) and hw % 64 == 0 and
                (B * 4 * hw // 64) * (mod.cout // 320) >= 256):
            # four 2x2 convolutions of the source map, one per output phase: 4 taps instead of 9, the weight image picked per tile;
            # the rows of a sample come out phase-major (Act.perm: the decoder block's GroupNorm reads them in raster order)
            key = (x.h, x.w, "up4")
            if key not in self._tabs:
                t4, pm = upsample_phase_tables(x.h, x.w)
                dt, dp = torch.from_numpy(t4).to(self.device), torch.from_numpy(pm).to(self.device)
                self._tabs[key] = (dt, dp)
                self._tab_np[dt.data_ptr()] = t4
            tab4, perm = self._tabs[key]
            if (name + ".wph0") not in self._wf:   # the four fragment-major images side by side (one base + a stride for the kernel)
                grp = torch.empty((2, 4, mod.cout, 4 * x.c), dtype=torch.bfloat16, device=self.device)
                st = torch.cuda.current_stream(self.device).cuda_stream
                for ph in range(4):
                    self._wf[f"{name}.wph{ph}"] = grp[:, ph]
                    self._pack_wf(f"{name}.wph{ph}", grp[:, ph], st)
                self._pack = None
            gg = self._gemm(ops, name + ".conv (4 phases)", [self._src(pl, x.c, 4, tab4, hw)], name + ".wph0", B * 4 * hw, 4 * hw,
                            bias=self._w[name + ".b"], out_f32=out, out_ld=mod.cout, want_stats=True)
            assert gg.tile == 64320 and gg.w_layout == 3, name
            gg.w_ngroups, gg.w_group_stride = 4, mod.cout * 4 * x.c
            return Act(out, mod.cout, ho, wo, gg._stats, prod=None, perm=perm)
        gg = self._gemm(ops, name + ".conv", [self._src(pl, x.c, 9, tab, x.h * x.w)], name + ".w", B * ho * wo, ho * wo,
                        bias=self._w[name + ".b"], out_f32=out, out_ld=mod.cout, want_stats=True, tile=tile)
        return Act(out, mod.cout, ho, wo, gg._stats, prod=gg)

    def _attention(self, ops, what, q, ldq, k, ldk, v, ldv, heads, nq, nk, d, scale, out_pl, out_f32=None,
                   out_rows=None, out_row0=0):
        B = self._B
        ops.append((self.lib.wd_attention,
                    (q, ldq, k, ldk, v, ldv, B, heads, nq, nk, d, float(scale), _ptr(out_f32),
                     out_pl[0].data_ptr() if out_pl is not None else None,
                     out_pl[1].data_ptr() if (out_pl is not None and self.npass == 3) else None,
                     out_pl.shape[2] if out_pl is not None else out_f32.shape[-1],
                     nq if out_rows is None else out_rows, out_row0), what))

    def _transformer(self, P, name, mod: SpatialTransformerParams, x: Act) -> Act:
        ops = P.step
        B = self._B
        h, w, c = x.h, x.w, x.c
        hw, M = h * w, B * h * w
        heads, d = mod.heads, mod.d_head
        inner = heads * d
        L = self._ctx_len
        tok = self._f32(P, M, inner)
        # (PHOSC variant: the first block's norm1 planes come out of proj_in's epilogue where its tiles hold whole rows)
        n1_pre = None
        ln1 = None
        if self.variant == "phosc" and self._ln_in_producer(M, inner):
            n1_pre, ln1 = self._planes(P, M, inner), f"{name}.tb0.norm1"
        if self._gn_in_consumer(P, ops, name + ".gn", [x], False, M, hw, inner):
            self._gemm(ops, name + ".proj_in", [self._src32(x, hw_src=hw)], name + ".pi.w", M, hw, bias=self._w[name + ".pi.b"],
                       out_f32=tok, out_ld=inner, a32=(x, name + ".gn", 1e-6, False), out_pl=n1_pre, ln=ln1)
        else:
            g, _ = self._gn(P, ops, name + ".gn", [x], name + ".gn", 1e-6, False)
            self._gemm(ops, name + ".proj_in", [self._src(g, c)], name + ".pi.w", M, hw, bias=self._w[name + ".pi.b"],
                       out_f32=tok, out_ld=inner, out_pl=n1_pre, ln=ln1)
        xpl = None
        fuse = self.fuse_xattn and bool(self.lib.wd_xattn_supported(inner, heads, L))

        def fold(tag, p):
            """K/V/to_q/to_out of one cross-attention folded per sample in the conditioning phase (csrc/wd_xattn.hip)."""
            ko = self.kv_off[f"{p}.{tag}"]
            mq = self._f32(P, B, heads * L, inner)
            mo = self._f32(P, B, heads * L, inner)
            mq_pl = torch.zeros((B, 2, 64, inner), dtype=torch.bfloat16, device=self.device)   # MFMA operands (padded rows 0)
            mot_pl = torch.zeros((B, 2, inner, 64), dtype=torch.bfloat16, device=self.device)
            P.keep += [mq_pl, mot_pl]
            P.cond.append((self.lib.wd_xattn_fold,
                           (self._kv.data_ptr() + 4 * ko, self.kv_total, self._kv.data_ptr() + 4 * (ko + inner), self.kv_total, B,
                            heads, L, d, float(d ** -0.5), self._w[f"{p}.{tag}.q.f32"].data_ptr(),
                            self._w[f"{p}.{tag}.o.f32"].data_ptr(), inner, mq.data_ptr(), mo.data_ptr(), mq_pl.data_ptr(),
                            mot_pl.data_ptr()), f"{p}.{tag}:fold"))
            return mq, mo, mq_pl, mot_pl

        def folded(tag, p, x_in, x_out, ln_name, next_ln=None):
            """x_out = x_in + to_out(attention(to_q(LN(x_in)), K, V)) in one launch.  next_ln: also emit the following
            LayerNorm as operand planes."""
            mq, mo, mq_pl, mot_pl = fold(tag, p)
            npl = self._planes(P, M, inner) if next_ln else None
            ops.append((self.lib.wd_xattn_fused,
                        (x_in.data_ptr(), inner, B, hw, inner, self._w[f"{p}.{ln_name}.g"].data_ptr(),
                         self._w[f"{p}.{ln_name}.b"].data_ptr(), 1e-5, mq.data_ptr(), mo.data_ptr(), heads, L,
                         self._w[f"{p}.{tag}.o.b"].data_ptr(), x_out.data_ptr(), inner,
                         self._w[f"{p}.{next_ln}.g"].data_ptr() if next_ln else None,
                         self._w[f"{p}.{next_ln}.b"].data_ptr() if next_ln else None, 1e-5,
                         npl[0].data_ptr() if next_ln else None,
                         npl[1].data_ptr() if (next_ln and self.npass == 3) else None, inner, mq_pl.data_ptr(), mot_pl.data_ptr()),
                        f"{p}.{tag}:folded"))
            return npl

        def folded_pair(p, x_in, x_out):
            """Both cross-attentions of a base-model block (each behind norm2, unet.py:337-345) and norm3 in one launch."""
            _, _, qa, oa = fold("a1", p)
            _, _, qb, ob = fold("a2", p)
            npl = self._planes(P, M, inner)
            g2, b2 = self._w[f"{p}.norm2.g"].data_ptr(), self._w[f"{p}.norm2.b"].data_ptr()
            ops.append((self.lib.wd_xattn_pair,
                        (x_in.data_ptr(), inner, B, hw, inner, 1e-5, heads, L, g2, b2, qa.data_ptr(), oa.data_ptr(),
                         self._w[f"{p}.a1.o.b"].data_ptr(), g2, b2, qb.data_ptr(), ob.data_ptr(),
                         self._w[f"{p}.a2.o.b"].data_ptr(), x_out.data_ptr(), inner, self._w[f"{p}.norm3.g"].data_ptr(),
                         self._w[f"{p}.norm3.b"].data_ptr(), 1e-5, npl[0].data_ptr(),
                         npl[1].data_ptr() if self.npass == 3 else None, inner), f"{p}.a1+a2:folded"))
            return npl

        for di, tb in enumerate(mod.transformer_blocks):
            p = f"{name}.tb{di}"
            scale = d ** -0.5
            n3 = None
            # ---- attn1
            if fuse and self.variant != "phosc" and self.fuse_xattn_pair:
                tok2 = self._f32(P, M, inner)
                n3 = folded_pair(p, tok, tok2)
            tok1 = self._f32(P, M, inner) if n3 is None else None
            if n3 is not None:
                pass
            elif self.variant == "phosc":
                n1 = n1_pre if (di == 0 and n1_pre is not None) else self._ln(P, ops, p + ".norm1", tok, M, inner, p + ".norm1")
                qkv = self._f32(P, M, 3 * inner)
                self._gemm(ops, p + ".a1.qkv", [self._src(n1, inner)], p + ".a1.qkv.w", M, hw, out_f32=qkv,
                           out_ld=3 * inner)
                o1 = self._planes(P, M, inner)
                self._attention(ops, p + ".a1", qkv.data_ptr(), 3 * inner, qkv.data_ptr() + 4 * inner, 3 * inner,
                                qkv.data_ptr() + 8 * inner, 3 * inner, heads, hw, hw, d, scale, o1)
                n2_pre = self._planes(P, M, inner) if (not fuse and self._ln_in_producer(M, inner)) else None
                self._gemm(ops, p + ".a1.out", [self._src(o1, inner)], p + ".a1.o.w", M, hw, bias=self._w[p + ".a1.o.b"],
                           resid=tok.data_ptr(), resid_ld=inner, out_f32=tok1, out_ld=inner, out_pl=n2_pre,
                           ln=(p + ".norm2") if n2_pre is not None else None)
            elif fuse:
                folded("a1", p, tok, tok1, "norm2")  # the base model reads norm2 for both attentions (unet.py:337-345)
            else:
                n1 = self._ln(P, ops, p + ".norm2a", tok, M, inner, p + ".norm2")
                q1 = self._f32(P, M, inner)
                self._gemm(ops, p + ".a1.q", [self._src(n1, inner)], p + ".a1.q.w", M, hw, out_f32=q1, out_ld=inner)
                ko = self.kv_off[p + ".a1"]
                o1 = self._planes(P, M, inner)
                self._attention(ops, p + ".a1", q1.data_ptr(), inner, self._kv.data_ptr() + 4 * ko, self.kv_total,
                                self._kv.data_ptr() + 4 * (ko + inner), self.kv_total, heads, hw, L, d, scale, o1)
                self._gemm(ops, p + ".a1.out", [self._src(o1, inner)], p + ".a1.o.w", M, hw, bias=self._w[p + ".a1.o.b"],
                           resid=tok.data_ptr(), resid_ld=inner, out_f32=tok1, out_ld=inner)
            # ---- attn2 (cross)
            if n3 is not None:
                pass
            elif fuse:
                tok2 = self._f32(P, M, inner)
                n3 = folded("a2", p, tok1, tok2, "norm2", next_ln="norm3")
            else:
                tok2 = self._f32(P, M, inner)
                n2 = n2_pre if (self.variant == "phosc" and n2_pre is not None) else \
                    self._ln(P, ops, p + ".norm2", tok1, M, inner, p + ".norm2")
                q2 = self._f32(P, M, inner)
                self._gemm(ops, p + ".a2.q", [self._src(n2, inner)], p + ".a2.q.w", M, hw, out_f32=q2, out_ld=inner)
                ko = self.kv_off[p + ".a2"]
                o2 = self._planes(P, M, inner)
                kp, vp = self._kv.data_ptr() + 4 * ko, self._kv.data_ptr() + 4 * (ko + inner)
                nimg = self.lib.wd_attention_packed_elems(B, heads, L, d) if self.pack_kv else 0
                if nimg > 0:
                    # the context's keys / values do not change during a sampling call: their split-bf16 LDS images are built
                    # once per call (P.cond, after the K/V projection) and every step's attention copies them 16 bytes at a time
                    img = torch.empty(nimg, dtype=torch.bfloat16, device=self.device)
                    P.keep.append(img)
                    P.cond.append((self.lib.wd_attention_pack_kv,
                                   (kp, self.kv_total, vp, self.kv_total, B, heads, L, d, img.data_ptr()), p + ".a2:pack kv"))
                    ops.append((self.lib.wd_attention_packed,
                                (q2.data_ptr(), inner, img.data_ptr(), B, heads, hw, L, d, float(scale), None, o2[0].data_ptr(),
                                 o2[1].data_ptr() if self.npass == 3 else None, inner, hw, 0), p + ".a2"))
                else:
                    self._attention(ops, p + ".a2", q2.data_ptr(), inner, kp, self.kv_total, vp, self.kv_total, heads, hw, L, d,
                                    scale, o2)
                if self._ln_in_producer(M, inner):
                    n3 = self._planes(P, M, inner)
                self._gemm(ops, p + ".a2.out", [self._src(o2, inner)], p + ".a2.o.w", M, hw, bias=self._w[p + ".a2.o.b"],
                           resid=tok1.data_ptr(), resid_ld=inner, out_f32=tok2, out_ld=inner, out_pl=n3,
                           ln=(p + ".norm3") if n3 is not None else None)
            # ---- GEGLU feed-forward
            if n3 is None:
                n3 = self._ln(P, ops, p + ".norm3", tok2, M, inner, p + ".norm3")
            last = di == len(mod.transformer_blocks) - 1
            tok = self._f32(P, M, inner)
            xpl = self._planes(P, M, inner) if last else None
            ffi = tb.ff.net[2].in_features
            if self.fuse_ff and (p + ".ff1f.w") in self._w and (M + 63) // 64 >= 192:
                # one workgroup per 64 tokens: worth it once they fill the chip (the 8 x 32 level at batch >= 48)
                if last and self.fuse_proj and inner == c:
                    out = self._f32(P, M, c)
                    fa = self._ff_fused(ops, p, n3, tok2, M, inner, ffi, out, None,
                                        proj=(name + ".po.w", self._w[name + ".po.b"], x.t, hw))
                    return Act(out, c, h, w, fa._stats, prod=fa)
                self._ff_fused(ops, p, n3, tok2, M, inner, ffi, None if last else tok, xpl)
            else:
                ffh = self._planes(P, M, ffi)
                self._gemm(ops, p + ".ff1", [self._src(n3, inner)], p + ".ff1.w", M, hw, bias=self._w[p + ".ff1.b"],
                           act=N.ACT_GEGLU, out_pl=ffh, tile=geglu_tile(ffi))
                self._gemm(ops, p + ".ff2", [self._src(ffh, ffi)], p + ".ff2.w", M, hw, bias=self._w[p + ".ff2.b"],
                           resid=tok2.data_ptr(), resid_ld=inner, out_f32=None if last else tok, out_ld=inner,
                           out_pl=xpl)
        out = self._f32(P, M, c)
        gg = self._gemm(ops, name + ".proj_out", [self._src(xpl, inner)], name + ".po.w", M, hw,
                        bias=self._w[name + ".po.b"], resid=x.t.data_ptr(), resid_ld=c, out_f32=out, out_ld=c,
                        want_stats=True)
        return Act(out, c, h, w, gg._stats, prod=gg)

    def _wfrag(self, wname):
        """The fragment-major image (wd_gemm_pack_w) of a packed matrix, kept up to date by refresh_weights."""
        if wname not in self._wf:
            self._wf[wname] = torch.empty_like(self._w[wname])
            self._pack_wf(wname, self._wf[wname], torch.cuda.current_stream(self.device).cuda_stream)
            self._pack = None  # the next refresh_weights rebuilds the repack table with this image in it
        return self._wf[wname]

    def _ff_fused(self, ops, p, n3, resid, M, inner, ffi, out_f32, out_pl, proj=None):
        """x + FeedForward(LN3(x)) (unet.py:343-344, :122-149) as one wd_ff_fused launch.  proj = (weight name, bias, residual
        tensor, hw): the SpatialTransformer's proj_out + residual (unet.py:406-412) in the same launch, with the GroupNorm statistics
        of the result for the next ResBlock; returns the statistics tuple then."""
        a = N.WdFfArgs()
        lo_ok = self.npass == 3
        a.x_hi, a.x_lo, a.x_ld = n3[0].data_ptr(), (n3[1].data_ptr() if lo_ok else None), n3.shape[2]
        a.m, a.c, a.inner = M, inner, ffi
        w1, w2 = self._wfrag(p + ".ff1f.w"), self._wfrag(p + ".ff2.w")
        a.w1_hi, a.w1_lo, a.b1 = w1[0].data_ptr(), w1[1].data_ptr(), self._w[p + ".ff1f.b"].data_ptr()
        a.w2_hi, a.w2_lo, a.b2 = w2[0].data_ptr(), w2[1].data_ptr(), self._w[p + ".ff2.b"].data_ptr()
        a.resid, a.resid_ld = resid.data_ptr(), inner
        if out_f32 is not None:
            a.out_f32, a.out_ld = out_f32.data_ptr(), inner
        if out_pl is not None:
            a.out_hi, a.out_lo, a.out_pl_ld = out_pl[0].data_ptr(), (out_pl[1].data_ptr() if lo_ok else None), out_pl.shape[2]
        a.hw_out, a.npass = 1, self.npass
        stats = None
        if proj is not None:
            wname, b3, x_in, hw = proj
            w3 = self._wfrag(wname)
            a.w3_hi, a.w3_lo, a.b3 = w3[0].data_ptr(), w3[1].data_ptr(), b3.data_ptr()
            a.resid3, a.resid3_ld = x_in.data_ptr(), inner
            if self.fuse_stats and inner % 32 == 0 and hw % 64 == 0:
                nchunk = hw // 64
                part = torch.zeros((M // hw, nchunk, 32, 2), dtype=torch.float64, device=self.device)
                self._cur_plan.keep.append(part)
                a.stat_part, a.stat_cpg, a.hw_out = part.data_ptr(), inner // 32, hw
                stats = (part, nchunk, inner // 32)
        self._cur_plan.keep.append(a)
        ops.append((self.lib.wd_ff_fused, (C.byref(a),), p + (".ff + proj_out (fused)" if proj is not None else ".ff (fused)")))
        a._stats = stats
        return a

    # ------------------------------------------------------------------------------------------ plan
    def plan(self, B: int, H: int, W: int, ctx_len: int, phosc_len: int, film_steps: int = 0) -> Plan:
        """film_steps = T > 0 (the DDPM sampler): the whole time / writer embedding path (timestep_embedding, time_embed,
        label_emb, SiLU, every emb_layers) is tabulated for all T timesteps by ``P.film`` ops - one large GEMM per
        ``sampling()`` call instead of three 64-row GEMMs per step - and each step copies its rows (``wd_select_rows``)."""
        key = (B, H, W, ctx_len, phosc_len, self.npass, film_steps)
        if key in self._plans:
            self._plans[key] = self._plans.pop(key)  # most recently used last
            return self._plans[key]
        while len(self._plans) >= max(1, PLAN_CACHE_SIZE):  # a plan holds ~1 GB of buffers at B = 64: keep a few shapes only
            self._plans.pop(next(iter(self._plans)))
        m = self.model
        lib = self.lib
        if ctx_len + phosc_len == 0:
            raise NotImplementedError("context=None: every reference script conditions on the word (unet.py:1605)")
        P = Plan()
        self._cur_plan = P
        self._B = B
        L = ctx_len + phosc_len
        self._ctx_len = L
        dev = self.device
        mc = m.model_channels
        ted = 4 * mc
        cd = m.context_dim
        lo_ok = self.npass == 3

        # ---- persistent inputs (the caller copies into these; graph replays read them)
        P.x_in = torch.zeros((B, m.in_channels, H, W), dtype=torch.float32, device=dev)
        P.t_in = torch.zeros((B,), dtype=torch.int64, device=dev)
        P.y_in = torch.zeros((B,), dtype=torch.int64, device=dev)
        P.ctx_in = torch.zeros((B, max(ctx_len, 1)), dtype=torch.int64, device=dev)
        P.phosc_in = torch.zeros((B, max(phosc_len, 1)), dtype=torch.int32, device=dev)

        # ---- conditioning: CharacterEncoder (unet.py:851-874) + K/V of every cross-attention
        cond = P.cond
        ctx_pl = self._planes(P, B * L, cd)
        msl = m.max_seq_len
        for (ids, n_tok, row0, i64) in ((P.ctx_in, ctx_len, 0, 1), (P.phosc_in, phosc_len, ctx_len, 0)):
            if n_tok == 0:
                continue
            use_pe = (self.variant == "base") or (n_tok <= msl)
            if use_pe and n_tok > msl:
                raise ValueError(f"context length {n_tok} exceeds max_seq_len {msl} (the reference fails too)")
            e = self._planes(P, B * n_tok, cd)
            cond.append((lib.wd_embed_tokens,
                         (ids.data_ptr(), i64, B * n_tok, n_tok, self._w["we.table"].data_ptr(),
                          self._w["we.table"].shape[0], cd, self._w["pe"].data_ptr() if use_pe else None,
                          e[0].data_ptr(), e[1].data_ptr() if lo_ok else None, cd), "word_emb.embedding"))
            qkv = self._f32(P, B * n_tok, 3 * cd)
            self._gemm(cond, "word_emb.qkv", [self._src(e, cd)], "we.qkv.w", B * n_tok, n_tok,
                       bias=self._w["we.qkv.b"], out_f32=qkv, out_ld=3 * cd)
            # Word_Attention: softmax(q k^T) v without 1/sqrt(d) (unet.py:831-835)
            self._attention(cond, "word_emb.attention", qkv.data_ptr(), 3 * cd, qkv.data_ptr() + 4 * cd, 3 * cd,
                            qkv.data_ptr() + 8 * cd, 3 * cd, 1, n_tok, n_tok, cd, 1.0, ctx_pl, out_rows=L,
                            out_row0=row0)
        P.ctx_pl = ctx_pl
        if self.kv_total:
            self._kv = self._f32(P, B * L, self.kv_total)
            self._gemm(cond, "cross.kv", [self._src(ctx_pl, cd)], "kv.w", B * L, L, out_f32=self._kv,
                       out_ld=self.kv_total)

        # ---- per-step: time/label embedding (unet.py:1550-1581) + all emb_layers at once (unet.py:609-615,660)
        step = P.step
        has_lab = m.num_classes is not None
        self._film = self._f32(P, B, self.film_total)
        P.t_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        P.film = []
        if film_steps:
            # the table holds ``chunk`` consecutive timesteps (rows (t % chunk) * B + b), not all T: T*B*film_total fp32 was
            # 655 MB + 328 MB of operand planes at B = 64 and grew with B and T; a chunk of ~8192 rows is 84 + 42 MB at any
            # B.  The time MLP (T rows, cheap) is still evaluated for every t once per call (P.film); the SiLU(time + label)
            # planes and the emb_layers GEMM of a chunk run when the loop enters it (P.film_prepare, same stream, a fraction
            # of one step each).
            T = film_steps
            chunk = min(T, max(8, FILM_CHUNK_ROWS // B))
            nchunks = (T + chunk - 1) // chunk
            Tp = nchunks * chunk
            film = P.film
            tt = torch.arange(Tp, dtype=torch.int64, device=dev)
            P.keep.append(tt)
            te = self._planes(P, Tp, mc)
            film.append((lib.wd_timestep_embedding, (tt.data_ptr(), Tp, self._w["freqs"].data_ptr(), mc // 2, te[0].data_ptr(),
                                                     te[1].data_ptr() if lo_ok else None, mc), "timestep_embedding[all t]"))
            e1 = self._planes(P, Tp, ted)
            self._gemm(film, "time_embed.0[all t]", [self._src(te, mc)], "te0.w", Tp, 1, bias=self._w["te0.b"], act=N.ACT_SILU,
                       out_pl=e1)
            tm = self._f32(P, Tp, ted)
            self._gemm(film, "time_embed.2[all t]", [self._src(e1, ted)], "te2.w", Tp, 1, bias=self._w["te2.b"], out_f32=tm,
                       out_ld=ted)
            e2 = self._planes(P, chunk * B, ted)
            P.film_table = self._f32(P, chunk * B, self.film_total)
            P.film_chunk, P.film_nchunks, P.film_loaded = chunk, nchunks, -1
            chunk_gemm = []
            self._gemm(chunk_gemm, "emb_layers(all)[chunk of t]", [self._src(e2, ted)], "film.w", chunk * B, 1,
                       bias=self._w["film.b"], out_f32=P.film_table, out_ld=self.film_total)
            lab = self._w["label"].data_ptr() if has_lab else None
            yin = P.y_in.data_ptr() if has_lab else None
            ncls = m.num_classes if has_lab else 0

            def film_prepare(t, stream, P=P, tm=tm, e2=e2, chunk=chunk, B=B):
                """Makes the FiLM rows of timestep ``t`` resident: (re)computes the chunk ``t // chunk`` unless it is loaded."""
                c = int(t) // chunk
                if c == P.film_loaded:
                    return False
                N.check(lib.wd_emb_combine(tm.data_ptr() + 4 * c * chunk * ted, lab, yin, ncls, chunk, B, ted, e2[0].data_ptr(),
                                           e2[1].data_ptr() if lo_ok else None, ted, stream), "SiLU(time + label)[chunk of t]")
                Plan._run(chunk_gemm, stream)
                P.film_loaded = c
                return True

            P.film_prepare = film_prepare
            step.append((lib.wd_select_rows, (P.film_table.data_ptr(), P.t_dev.data_ptr(), B, self.film_total, chunk,
                                              self._film.data_ptr()), "film rows of step t"))
        else:
            te = self._planes(P, B, mc)
            step.append((lib.wd_timestep_embedding, (P.t_in.data_ptr(), B, self._w["freqs"].data_ptr(), mc // 2,
                                                     te[0].data_ptr(), te[1].data_ptr() if lo_ok else None, mc),
                         "timestep_embedding"))
            e1 = self._planes(P, B, ted)
            self._gemm(step, "time_embed.0", [self._src(te, mc)], "te0.w", B, 1, bias=self._w["te0.b"], act=N.ACT_SILU,
                       out_pl=e1)
            e2 = self._planes(P, B, ted)  # SiLU(emb): the only form any consumer reads (emb_layers start with SiLU)
            self._gemm(step, "time_embed.2+label", [self._src(e1, ted)], "te2.w", B, 1, bias=self._w["te2.b"],
                       resid=self._w["label"].data_ptr() if has_lab else None, resid_ld=ted if has_lab else 0,
                       resid_rows=P.y_in.data_ptr() if has_lab else None, act=N.ACT_SILU, out_pl=e2)
            self._gemm(step, "emb_layers(all)", [self._src(e2, ted)], "film.w", B, 1, bias=self._w["film.b"],
                       out_f32=self._film, out_ld=self.film_total)

        # ---- trunk
        xin = self._planes(P, B * H * W, self.kpad_in)
        step.append((lib.wd_im2col3x3, (P.x_in.data_ptr(), B, m.in_channels, H, W, xin[0].data_ptr(),
                                        xin[1].data_ptr() if lo_ok else None, self.kpad_in), "im2col"))
        h0 = self._f32(P, B * H * W, mc)
        g0 = self._gemm(step, "input_blocks.0", [self._src(xin, self.kpad_in)], "in.w", B * H * W, H * W,
                        bias=self._w["in.b"], out_f32=h0, out_ld=mc, want_stats=True)
        cur = Act(h0, mc, H, W, g0._stats)
        hs = [cur]

        def run_layers(prefix, blk, cur, extra=None):
            for j, mod in enumerate(blk):
                name = f"{prefix}.{j}"
                if isinstance(mod, ResBlockParams):
                    cur = self._resblock(P, name, mod, [cur] + ([extra] if (extra is not None and j == 0) else []))
                elif isinstance(mod, SpatialTransformerParams):
                    cur = self._transformer(P, name, mod, cur)
                elif isinstance(mod, DownsampleParams):
                    cur = self._resample(P, name, mod, cur, "down")
                elif isinstance(mod, UpsampleParams):
                    cur = self._resample(P, name, mod, cur, "up")
                else:
                    raise TypeError(type(mod))
            return cur

        for i, blk in enumerate(m.input_blocks):
            if i == 0:
                continue
            cur = run_layers(f"in{i}", blk, cur)
            hs.append(cur)
        cur = run_layers("mid", m.middle_block, cur)
        for i, blk in enumerate(m.output_blocks):
            cur = run_layers(f"out{i}", blk, cur, extra=hs.pop())
        oc = m.out_channels
        P.out = torch.empty((B, oc, cur.h, cur.w), dtype=torch.float32, device=dev)
        if self.fuse_out and lib.wd_gn_conv3x3_few_supported(cur.c, cur.w, oc) and cur.c % 32 == 0:
            # GroupNorm + SiLU + the 320 -> 4 convolution + NCHW in one fp32 launch (as a GEMM it fills 4 of 64 tile columns)
            if cur.stats is None:
                nchunk = lib.wd_gn_nchunk(cur.h * cur.w)
                part = torch.empty((B, nchunk, 32, 2), dtype=torch.float64, device=dev)
                P.keep.append(part)
                step.append((lib.wd_gn_stats, (cur.t.data_ptr(), cur.c, B, cur.h * cur.w, cur.c, cur.c // 32, part.data_ptr()),
                             "out.gn:stats"))
                cur.stats = (part, nchunk, cur.c // 32)
            part, nchunk, pc = cur.stats
            step.append((lib.wd_gn_conv3x3_few,
                         (cur.t.data_ptr(), cur.c, B, cur.h, cur.w, cur.c, cur.c // 32, part.data_ptr(), nchunk, pc,
                          self._w["out.gn.g"].data_ptr(), self._w["out.gn.b"].data_ptr(), 1e-5, 1,
                          self._w["out.w.f32"].data_ptr(), self._w["out.b"].data_ptr(), oc, P.out.data_ptr()),
                         "out: GroupNorm + SiLU + conv3x3 -> NCHW"))
        else:
            g, _ = self._gn(P, step, "out.gn", [cur], "out.gn", 1e-5, True)
            tab, _, _ = self._table(cur.h, cur.w, "same")
            otok = self._f32(P, B * cur.h * cur.w, oc)
            self._gemm(step, "out.conv", [self._src(g, cur.c, 9, tab, cur.h * cur.w)], "out.w", B * cur.h * cur.w,
                       cur.h * cur.w, bias=self._w["out.b"], out_f32=otok, out_ld=oc)
            step.append((lib.wd_tokens_to_nchw, (otok.data_ptr(), oc, B, oc, cur.h * cur.w, P.out.data_ptr()),
                         "tokens_to_nchw"))
        self._plans[key] = P
        return P

    # ------------------------------------------------------------------------------------------ run
    def check_ids(self, context=None, y=None, phosc=None, need_y=True):
        """Host-side range check of every integer the kernels use as a table row: writer ids against ``num_classes``
        (``label_emb``, unet.py:1581), word / PHOSC ids against the rows of the character table (unet.py:860).  The
        reference raises an index error for such an id; a raw device read past the table could fault the GPU instead."""
        m = self.model
        vocab = int(m.word_emb.embedding.weight.shape[0]) if hasattr(m, "word_emb") else None
        checks = []
        if getattr(m, "num_classes", None) is not None:
            if y is None:
                if need_y:
                    raise ValueError("y (writer ids) is required: the model is class-conditional (unet.py:1555)")
            else:
                checks.append(("writer id (y)", y, m.num_classes))
        for what, ids in (("word id (context)", context), ("PHOSC id (phoscLabels)", phosc)):
            if ids is not None and vocab is not None and ids.numel():
                checks.append((what, ids, vocab))
        if not checks:
            return
        vals = [v for _, ids, _ in checks for v in (ids.min().long(), ids.max().long())]
        ext = torch.stack(vals).tolist() if len({v.device for v in vals}) == 1 else [int(v) for v in vals]  # one sync
        for i, (what, _, bound) in enumerate(checks):
            lo, hi = int(ext[2 * i]), int(ext[2 * i + 1])
            if lo < 0 or hi >= bound:
                raise IndexError(f"{what} out of range: [{lo}, {hi}] does not fit a table of {bound} rows")

    def load_inputs(self, P: Plan, x=None, t=None, context=None, y=None, phosc=None, check=True):
        if check:
            self.check_ids(context, y, phosc, need_y=False)
        if x is not None:
            P.x_in.copy_(x, non_blocking=True)
        if t is not None:
            P.t_in.copy_(t, non_blocking=True)
        if y is not None:
            P.y_in.copy_(y, non_blocking=True)
        if context is not None:
            P.ctx_in.copy_(context, non_blocking=True)
        if phosc is not None:
            P.phosc_in.copy_(phosc.to(torch.int32) if phosc.dtype != torch.int32 else phosc, non_blocking=True)

    def forward(self, x, t, context, y, phosc=None):
        self.refresh_weights()
        B, _, H, W = x.shape
        ctx_len = 0 if context is None else context.shape[1]
        phosc_len = 0 if phosc is None else phosc.shape[1]
        P = self.plan(B, H, W, ctx_len, phosc_len)
        self.check_ids(context, y, phosc)
        self.load_inputs(P, x, t, context, y, phosc, check=False)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        P.run_cond(stream)
        P.run_step(stream)
        return P.out.clone()
