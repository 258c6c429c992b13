"""Training-step engine: the forward of ``engine.py`` with every intermediate kept, plus a hand-scheduled backward.

The reference trains with autograd over ~150 ATen modules (``train.py:287-293``: ``model(...)`` -> ``MSELoss`` ->
``loss.backward()``).  Here the backward is a static launch list, built once per input shape next to the forward list:

  * d(input) of every convolution / linear is ``wd_gemm`` again - over the planes of d(output), with the inverse gather
    table (``backward.conv_bwd_table``) and the weights repacked [C_in][tap][C_out] ("B:" entries of the weight cache);
  * d(weight) is ``wd_gemm`` with the token dimension as the reduction: both operands are transposed into split-bf16
    planes by ``wd_transpose_planes`` (rows ordered (c_in, tap), so the GEMM output IS the OIHW gradient) and the kernel's
    split-K spreads the long reduction over the chip;
  * GroupNorm(+SiLU), LayerNorm, attention over the <=16 context tokens, GEGLU, SiLU, nearest-x2 and the embeddings
    have their own kernels in ``csrc/wd_bwd.hip``; bias / FiLM / norm-parameter gradients are deterministic two-stage
    column sums (no float atomics anywhere: a replayed step is bit-identical).

Gradients land in persistent buffers that become ``param.grad`` (reference layout), so ``torch.optim.AdamW``, gradient
clipping, DDP-style all-reduce (``dist.GradAllReducer``) and ``optim.FusedAdamW`` all see what autograd would have given
them.  Forward + backward replay as one hipGraph.  Scope: the base ``unet.UNetModel`` (what ``train.py:403`` builds).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import os

import torch

from . import _native as N
from .backward import conv_bwd_table
from .engine import Act, Plan, UNetEngine, _ptr
from .layers import DownsampleParams, ResBlockParams, SpatialTransformerParams, UpsampleParams


class TAct(Act):
    """Feature map + its gradient buffer (``g``) and whether the backward list has written it yet (``gw``)."""
    __slots__ = ("g", "gw")

    def __init__(self, t, c, h, w, stats=None):
        super().__init__(t, c, h, w, stats)
        self.g, self.gw = None, False


class TrainPlan(Plan):
    def __init__(self):
        super().__init__()
        self.bwd: List[tuple] = []
        self.dout: Optional[torch.Tensor] = None

    def run_bwd(self, stream, begin: int = 0, end: Optional[int] = None):
        self._run(self.bwd[begin:end], stream)


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class TrainEngine(UNetEngine):
    def __init__(self, model, variant: str):
        super().__init__(model, variant)
        # the weights-to-registers GEMM is no faster on the training forward (11.03-11.14 ms per step without it, 11.18-11.25 with it at the
        # end of round 3 - its fragment-major weight images now come out of the repack launch itself, so that is the kernels alone).
        # The fused feed-forward does not keep the hidden activations the backward pass needs.
        self.use_wdirect = os.environ.get("WDIFF_TRAIN_WDIRECT", "0") != "0"
        self.use_smallmap = os.environ.get("WDIFF_TRAIN_SMALLMAP", "1") != "0"   # (forward only; -0.085 ms per step)
        self.fuse_ff = self.fuse_proj = False
        self.use_up_phases = False   # (the backward pass differentiates the 3x3 form)
        self.fuse_gn_in = 0
        self.use_dw = os.environ.get("WDIFF_TRAIN_DW", "1") != "0"  # weight gradients through wd_dw (csrc/wd_dw.hip) where it applies
        self.fuse_geglu_bwd = os.environ.get("WDIFF_FUSE_GEGLU_BWD", "1") != "0"  # GEGLU backward inside the d(out) preparation of ff1
        self.fuse_gn_bwd = os.environ.get("WDIFF_FUSE_GN_BWD", "1") != "0"  # GroupNorm backward in one pass (wd_gn_bwd_fused)
        self.dw_group_max = int(os.environ.get("WDIFF_DW_GROUP", "8"))  # single-tap layers of one shape per grouped launch (1: off)
        self._dw_pending: Dict[tuple, list] = {}
        self._tplans: Dict[tuple, TrainPlan] = {}
        self._grad: Dict[int, torch.Tensor] = {}     # id(param) -> gradient buffer (possibly a view into a group)
        self._params: Dict[int, torch.nn.Parameter] = {}
        self._btabs: Dict[tuple, torch.Tensor] = {}
        self._scr: Dict[str, torch.Tensor] = {}
        self._arena = None
        self._arena_used = 0
        self.defer_bias_sums = os.environ.get("WDIFF_DEFER_BIAS", "1") != "0"
        self._deferred, self._deferred_outs = [], {}
        # data-parallel overlap: the backward list is cut into NBUCKETS segments such that after segment j a PREFIX of the
        # gradient arena is final (the arena is carved in the order the backward list first writes the gradients, so a prefix
        # = the layers nearest the output) - TrainStep all-reduces that prefix while the next segment runs
        self.nbuckets = int(os.environ.get("WDIFF_GRAD_BUCKETS", "3"))
        self._carve_order: List[tuple] = []          # (data_ptr, arena offset, rounded numel) in carve order
        self._done_at: Dict[int, int] = {}           # gradient buffer data_ptr -> index of the last closure that writes it
        self._closure = 0
        self._cut_closures: Optional[List[int]] = None  # closures after which a bucket ends (fixed by the first plan)

    def set_precision(self, mode: str):
        old = self.npass
        super().set_precision(mode)
        if self.npass != old and self._tplans:
            self._tplans.clear()
            self._scr = {k: v for k, v in self._scr.items() if isinstance(k, tuple)}

    # ------------------------------------------------------------------------------------------ weights
    def _recipes(self):
        R = super()._recipes()
        m = self.model
        we = m.word_emb
        cd = we.embedding.weight.shape[1]
        R.matrix("B:we.qkv.w", cd, 3 * cd)
        for i, l in enumerate((we.attention.linear_query, we.attention.linear_key, we.attention.linear_value)):
            R["B:we.qkv.w"].bwd(l.weight, col_off=i * cd)
        for name, mod in self._walk():
            if isinstance(mod, ResBlockParams):
                R.matrix("B:" + name + ".c1.w", mod.cin, 9 * mod.cout).bwd(mod.in_layers[2].weight)
                R.matrix("B:" + name + ".c2.w", mod.cout, 9 * mod.cout).bwd(mod.out_layers[3].weight)
                if mod.cin != mod.cout:
                    R.matrix("B:" + name + ".skip.w", mod.cin, mod.cout).bwd(mod.skip_connection.weight)
            elif isinstance(mod, (DownsampleParams, UpsampleParams)):
                conv = mod.op if isinstance(mod, DownsampleParams) else mod.conv
                R.matrix("B:" + name + ".w", mod.cin, 9 * mod.cout).bwd(conv.weight)
            elif isinstance(mod, SpatialTransformerParams):
                inner = mod.heads * mod.d_head
                R.matrix("B:" + name + ".pi.w", mod.ch, inner).bwd(mod.proj_in.weight)
                R.matrix("B:" + name + ".po.w", inner, mod.ch).bwd(mod.proj_out.weight)
                for d, tb in enumerate(mod.transformer_blocks):
                    p = f"{name}.tb{d}"
                    for tag, at in (("a1", tb.attn1), ("a2", tb.attn2)):
                        if tag == "a1" and self.variant == "phosc":  # spatial self-attention: fused q|k|v projection
                            R.matrix(f"B:{p}.a1.qkv.w", inner, 3 * inner)
                            for i, l in enumerate((at.to_q, at.to_k, at.to_v)):
                                R[f"B:{p}.a1.qkv.w"].bwd(l.weight, col_off=i * inner)
                        else:
                            R.matrix(f"B:{p}.{tag}.q.w", inner, inner).bwd(at.to_q.weight)
                        R.matrix(f"B:{p}.{tag}.o.w", inner, inner).bwd(at.to_out[0].weight)
                    ffi = tb.ff.net[2].in_features
                    # the training forward keeps the GEGLU pre-activation: plain [x | gate] row order (unet.py:128)
                    R.linear(p + ".ff1u", tb.ff.net[0].proj)
                    R.matrix("B:" + p + ".ff1.w", inner, 2 * ffi).bwd(tb.ff.net[0].proj.weight)
                    R.matrix("B:" + p + ".ff2.w", ffi, inner).bwd(tb.ff.net[2].weight)
        ted = m.time_embed[2].out_features
        R.matrix("B:film.w", ted, self.film_total)
        c0 = 0
        for l in self._film_mods:
            R["B:film.w"].bwd(l.weight, col_off=c0)
            c0 += l.weight.shape[0]
        R.matrix("B:kv.w", cd, self.kv_total)
        c0 = 0
        for at in self._kv_mods:
            for l in (at.to_k, at.to_v):
                R["B:kv.w"].bwd(l.weight, col_off=c0)
                c0 += l.weight.shape[0]
        R.matrix("B:te2.w", ted, ted).bwd(m.time_embed[2].weight)
        # C_out padded to 32 columns per tap (the gradient planes of a 4-channel map are 32 wide)
        R.matrix("B:out.w", m.out[2].in_channels, 9 * 32).bwd(m.out[2].weight, npad=32)
        return R

    # ------------------------------------------------------------------------------------------ gradient buffers
    def _carve(self, shape) -> torch.Tensor:
        """A zeroed slice of the flat gradient arena (so that data-parallel training all-reduces ONE buffer)."""
        if self._arena is None:
            total = sum(_rup(p.numel(), 64) for p in self.model.parameters())
            self._arena = torch.zeros(total, dtype=torch.float32, device=self.device)
            self._arena_used = 0
        n = 1
        for d in shape:
            n *= int(d)
        out = self._arena[self._arena_used:self._arena_used + n].view(shape)
        self._carve_order.append((out.data_ptr(), self._arena_used, _rup(n, 64)))
        self._arena_used += _rup(n, 64)
        assert self._arena_used <= self._arena.numel()
        return out

    def _pgrad(self, p: torch.nn.Parameter) -> torch.Tensor:
        if id(p) not in self._grad:
            self._grad[id(p)] = self._carve(tuple(p.shape))
            self._params[id(p)] = p
        return self._grad[id(p)]

    def _pgroup(self, params: List[torch.nn.Parameter]) -> torch.Tensor:
        """One buffer whose row blocks are the gradients of ``params`` (weights the forward concatenates: all FiLM
        projections, every cross-attention's K/V, the word encoder's q/k/v)."""
        key = ("group",) + tuple(id(p) for p in params)
        if key not in self._scr:
            tail = tuple(params[0].shape[1:])
            rows = sum(p.shape[0] for p in params)
            buf = self._carve((rows,) + tail)
            r0 = 0
            for p in params:
                self._grad[id(p)] = buf[r0:r0 + p.shape[0]]
                self._params[id(p)] = p
                r0 += p.shape[0]
            self._scr[key] = buf
        return self._scr[key]

    def _ppair(self, p0: torch.nn.Parameter, p1: torch.nn.Parameter) -> torch.Tensor:
        """[2, c] buffer: row 0 is p0's gradient, row 1 p1's (norm scale / shift pairs come out of one column sum)."""
        key = ("pair", id(p0), id(p1))
        if key not in self._scr:
            buf = self._carve((2,) + tuple(p0.shape))
            for i, p in enumerate((p0, p1)):
                self._grad[id(p)] = buf[i]
                self._params[id(p)] = p
            self._scr[key] = buf
        return self._scr[key]

    def grad_arena(self) -> torch.Tensor:
        """All parameter gradients as one flat fp32 tensor (views of it are the ``param.grad``s)."""
        return self._arena[: self._arena_used]

    def _bias_finish(self, ops, what, colpart, nblk, n, b):
        """Bias gradient = sum over the 64-row blocks of ``colpart``."""
        self._param_colsum(ops, what, colpart.data_ptr(), n, nblk, n, b.data_ptr(), self._pacc(b), key=b.data_ptr())

    def _param_colsum(self, ops, what, src_ptr, ld, rows, c, out_ptr, acc, key=None):
        """out[0..c) (+)= column sums of src[rows, c] (row pitch ld) where ``out`` is a PARAMETER gradient, so nothing in the
        backward list reads it: deferred to the batched launches at the end of the list (``wd_colsum_finish_multi``), one
        launch per round.  A gradient with several writers (a norm shared by two attentions, the word encoder's projections
        run once per token group in the PHOSC variant) gets one entry per writer, in successive rounds.  ``src`` must stay
        intact until the end of the list (a plan-owned buffer, not a shared scratch)."""
        key = out_ptr if key is None else key
        rnd = self._deferred_outs.get(key, 0)
        if not self.defer_bias_sums or rows > 1024 or (acc and rnd == 0):
            # (acc and rnd == 0: an earlier, non-deferred op of this list already wrote the row - keep program order)
            self._colsum(ops, what, src_ptr, ld, rows, c, rows, out_ptr, c, acc)
            return
        self._deferred_outs[key] = rnd + 1
        while len(self._deferred) <= rnd:
            self._deferred.append([])
        self._deferred[rnd].append((src_ptr, out_ptr, rows, c, ld, 1 if (rnd or acc) else 0, 1.0, 0))

    def _flush_deferred(self, P):
        import struct
        for rnd, entries in enumerate(self._deferred):
            if not entries:
                continue
            rec = self.lib.wd_colsum_entry_bytes()
            assert rec == 40, rec
            blob = b"".join(struct.pack("<QQiiiifi", *e) for e in entries)
            table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.device)
            P.keep.append(table)
            P.bwd.append((self.lib.wd_colsum_finish_multi, (table.data_ptr(), len(entries), max(e[3] for e in entries)),
                          f"parameter-gradient column sums: deferred finishes (round {rnd})"))
        self._deferred, self._deferred_outs = [], {}

    def _pacc(self, t: torch.Tensor) -> int:
        """1 if the backward list has already written this parameter-gradient buffer (shared norm2), else 0."""
        k = t.data_ptr()
        acc = k in self._pw
        self._pw.add(k)
        self._done_at[k] = self._closure  # (a buffer with several writers is final after the last one)
        return int(acc)

    def _gacc(self, P, act: TAct):
        if act.g is None:
            act.g = self._f32(P, *act.t.shape)
        acc = act.gw
        act.gw = True
        return act.g, int(acc)

    def _scratch(self, key: str, numel: int, dtype) -> torch.Tensor:
        cur = self._scr.get(key)
        if cur is None or cur.numel() < numel:
            if cur is not None and self._tplans:
                raise RuntimeError("scratch grew after a plan was built")  # sized by _size_scratch before any use
            self._scr[key] = torch.zeros(numel, dtype=dtype, device=self.device)
        return self._scr[key]

    def _btable(self, h, w, mode):
        key = (h, w, mode)
        if key not in self._btabs:
            tab, _, _ = conv_bwd_table(h, w, mode)
            dt = torch.from_numpy(tab).to(self.device)
            self._btabs[key] = dt
            self._tab_np[dt.data_ptr()] = tab
        return self._btabs[key]

    # ------------------------------------------------------------------------------------------ backward emitters
    def _colsum(self, ops, what, x_ptr, ld, rows, c, seg, out_ptr, out_ld, acc, scale=1.0):
        scr = self._cs_scratch
        nseg, nblk = (rows + seg - 1) // seg, (seg + 63) // 64
        assert nseg * nblk * c <= scr.numel(), (what, nseg, nblk, c)
        ops.append((self.lib.wd_colsum, (x_ptr, ld, rows, c, seg, out_ptr, out_ld, int(acc), float(scale), scr.data_ptr(),
                                         scr.numel()), what))

    def _gemm_dw(self, ops, what, doutT, nrows, xT, ncols, mpad, out: torch.Tensor, out_ld, acc):
        a = N.WdGemmArgs()
        s = N.WdSrc()
        s.hi, s.lo = doutT[0].data_ptr(), doutT[1].data_ptr()
        s.ld, s.c, s.ntaps, s.hw_src = mpad, mpad, 1, 0
        a.src[0] = s
        a.nsrc, a.npass = 1, self.npass
        a.w_hi, a.w_lo = xT[0].data_ptr(), xT[1].data_ptr()
        a.m, a.n, a.ktot, a.hw_out = nrows, ncols, mpad, 1
        a.out_f32, a.out_ld = out.data_ptr(), out_ld
        if acc:
            a.resid, a.resid_ld = out.data_ptr(), out_ld
        a.ksplit, a.ws, a.ws_floats = 0, self._ws.data_ptr(), self._ws.numel()
        self._give_tickets(a, nrows, ncols)
        self._cur_plan.keep.append(a)
        ops.append((self.lib.wd_gemm, (C.byref(a),), what))

    def _dw(self, ops, what, dpl, d_ld, planes, col_off, c, ftab, ntaps, hw_out, hw_src, M, n, out: torch.Tensor, out_ld, acc):
        a = N.WdDwArgs()
        a.d_hi, a.d_lo = dpl[0].data_ptr(), (dpl[1].data_ptr() if self.npass == 3 else None)
        a.x_hi = planes[0].data_ptr() + 2 * col_off
        a.x_lo = planes[1].data_ptr() + 2 * col_off if self.npass == 3 else None
        a.gather = _ptr(ftab)
        a.grad, a.grad_ld, a.ws, a.ws_floats = out.data_ptr(), out_ld, self._ws.data_ptr(), self._ws.numel()
        a.d_ld, a.x_ld, a.ntaps, a.hw_out, a.hw_src = d_ld, planes.shape[2], ntaps, hw_out, hw_src
        a.m, a.n, a.c, a.npass, a.accumulate, a.nslice = M, n, c, self.npass, int(bool(acc)), 0
        self._cur_plan.keep.append(a)
        ops.append((self.lib.wd_dw, (C.byref(a),), what))

    def _flush_dw(self, P):
        """Emits the weight-gradient launches put off by _bwd_linear: layers of one shape as ONE grouped wd_dw_group launch."""
        pend, self._dw_pending = self._dw_pending, {}
        for (M, n, c, hw, ntaps, hw_src, _), items in pend.items():
            ftab = items[0]["ftab"]
            for lo in range(0, len(items), self.dw_group_max):
                grp = items[lo:lo + self.dw_group_max]
                if len(grp) == 1:
                    it = grp[0]
                    self._dw(P.bwd, it["what"], it["dpl"], it["d_ld"], it["planes"], it["col_off"], c, ftab, ntaps, hw, hw_src, M, n,
                             it["wg"], it["out_ld"], self._pacc(it["wg"]))
                    continue
                arr = (N.WdDwItem * len(grp))()
                for i, it in enumerate(grp):
                    arr[i].d_hi, arr[i].d_lo = it["dpl"][0].data_ptr(), (it["dpl"][1].data_ptr() if self.npass == 3 else None)
                    arr[i].x_hi = it["planes"][0].data_ptr() + 2 * it["col_off"]
                    arr[i].x_lo = it["planes"][1].data_ptr() + 2 * it["col_off"] if self.npass == 3 else None
                    arr[i].grad, arr[i].grad_ld = it["wg"].data_ptr(), it["out_ld"]
                    arr[i].d_ld, arr[i].x_ld, arr[i].accumulate = it["d_ld"], it["planes"].shape[2], self._pacc(it["wg"])
                dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
                a = N.WdDwArgs()
                a.ws, a.ws_floats = self._ws.data_ptr(), self._ws.numel()
                a.gather = _ptr(ftab)
                a.ntaps, a.hw_out, a.hw_src, a.m, a.n, a.c, a.npass, a.nslice = ntaps, hw, hw_src, M, n, c, self.npass, 0
                P.keep += [arr, dev, a]
                P.bwd.append((self.lib.wd_dw_group, (C.byref(a), C.cast(arr, C.c_void_p), dev.data_ptr(), len(grp)),
                              "dW group: " + ", ".join(it["what"] for it in grp)))

    def _bwd_linear(self, P, what, dout: torch.Tensor, M, n, hw_out, segs, bias=(), film_off=None, npad=None, geglu=None):
        """Backward of out[M, n] = sum_seg gather(planes_seg) . W_seg^T + bias (+ FiLM row vector).

        seg keys: planes [2, rows, ld], c, ntaps, ftab (forward table or None), hw_src (forward source positions per
        sample), wgrad (tensor [n, c*ntaps] in OIHW order, or None), wgrad_packed (tensor [n, ld_k] for the im2col input
        conv), wb (name of the data-gradient weight) and dx: list of (tensor, ld, acc, w_row_off, nrows) written by the
        data-gradient GEMM whose rows are dx_rows output positions (dx_hw per sample) through btab."""
        ops = P.bwd
        lib = self.lib
        npad = n if npad is None else npad
        # geglu = (u, dh, inner): d(output) is the GEGLU projection's and is NOT materialised - wd_dout_prep_geglu derives its planes
        # and column sums from the saved pre-activation and the gradient of the activation's output (dout is None then)
        ldd = dout.shape[1] if dout is not None else n
        mpad = _rup(M, 64)
        lo_ok = self.npass == 3
        need_dx = any(s.get("dx") for s in segs)
        # weight gradients straight from the row-major planes (wd_dw: transposed LDS reads) wherever the shape allows; the rest goes
        # through transposed copies of both operands and wd_gemm
        hw_dw = hw_out if (hw_out % 64 == 0 and M % hw_out == 0) else 64
        for s in segs:
            s["_dw"] = bool(self.use_dw and s.get("wgrad") is not None and npad % 8 == 0 and s["planes"].shape[2] % 8 == 0 and
                            s.get("col_off", 0) % 8 == 0 and (s.get("ftab") is not None or s["ntaps"] == 1) and
                            lib.wd_dw_supported(M, n, s["c"], s["ntaps"], hw_out if s.get("ftab") is not None else hw_dw) and
                            lib.wd_dw_slices(M, n, s["c"], s["ntaps"]) * n * s["c"] * s["ntaps"] <= self._ws.numel())
        need_dw = any((s.get("wgrad") is not None and not s["_dw"]) or s.get("wgrad_packed") is not None for s in segs)
        need_dpl = need_dx or any(s["_dw"] for s in segs)
        # single-tap layers wait for the others of their shape (one grouped launch at the end of the block's backward, _flush_dw):
        # their d(out) planes then need a buffer of their own
        # (3x3 layers join a group only where a launch of their own would run short token slices - the 4 x 16 level: < 16 units of 64
        # tokens per slice -; the two convolutions of a ResBlock there share a launch)
        for s in segs:
            taps = s["ntaps"]
            short = taps == 1 or (M // 64) // max(1, self.lib.wd_dw_slices(M, n, s["c"], taps)) < 16
            s["_defer"] = bool(s["_dw"] and self.dw_group_max > 1 and short and (taps == 1) == (s.get("ftab") is None) and
                               self.lib.wd_dw_group_slices(M, n, s["c"], taps, 2) * 2 * n * s["c"] * taps <= self._ws.numel())
        dpl = doutT = colpart = None
        if need_dpl and any(s["_defer"] for s in segs):
            dpl = torch.empty(2, M, npad, dtype=torch.bfloat16, device=self.device)
            P.keep.append(dpl)
        elif need_dpl:
            dpl = self._scratch("dpl", 2 * self._max_dpl, torch.bfloat16)[: 2 * M * npad].view(2, M, npad)
        if need_dw:
            doutT = self._scratch("doutT", 2 * self._max_doutT, torch.bfloat16)[: 2 * n * mpad].view(2, n, mpad)
        # bias / FiLM gradients come out of the same pass as per-64-row column sums whenever the segments line up
        fuse_cs = (bool(bias) or film_off is not None) and (film_off is None or hw_out % 64 == 0)
        if fuse_cs:
            if bias and self.defer_bias_sums:
                # the bias-gradient finish of this layer runs in the one batched launch at the end of the backward list, so
                # its partial sums need a buffer of their own
                colpart = torch.empty((mpad // 64) * n, dtype=torch.float32, device=self.device)
                P.keep.append(colpart)
            else:
                colpart = self._scratch("colpart", self._max_colpart, torch.float32)
            assert (mpad // 64) * n <= colpart.numel(), what
        if geglu is not None:
            u, dh, inner = geglu
            assert dout is None and n == 2 * inner and npad == n and fuse_cs and film_off is None, what
            ops.append((lib.wd_dout_prep_geglu, (u.data_ptr(), u.shape[1], dh.data_ptr(), dh.shape[1], M, inner, mpad,
                                                 _ptr(dpl[0]) if need_dpl else None,
                                                 (_ptr(dpl[1]) if lo_ok else None) if need_dpl else None,
                                                 _ptr(doutT[0]) if need_dw else None, _ptr(doutT[1]) if need_dw else None,
                                                 _ptr(colpart)), what + ":prep(geglu bwd)"))
        elif need_dpl or need_dw or fuse_cs:
            # one pass over d(output): row-major planes (data gradient, wd_dw), transposed planes (wd_gemm weight gradient), column sums
            ops.append((lib.wd_dout_prep, (dout.data_ptr(), ldd, M, n, npad, mpad, _ptr(dpl[0]) if need_dpl else None,
                                           (_ptr(dpl[1]) if lo_ok else None) if need_dpl else None,
                                           _ptr(doutT[0]) if need_dw else None, _ptr(doutT[1]) if need_dw else None,
                                           _ptr(colpart)), what + ":prep(dout)"))
        for si, s in enumerate(segs):
            c, ntaps = s["c"], s["ntaps"]
            planes = s["planes"]
            for (dx, dx_ld, acc, roff, nrows) in s.get("dx") or ():
                src = self._src(dpl, npad, ntaps, s.get("btab"), hw_out if s.get("btab") is not None else 0)
                self._gemm(ops, f"{what}:dX{si}", [src], s["wb"], s["dx_rows"], s["dx_hw"],
                           resid=dx.data_ptr() if acc else None, resid_ld=dx_ld if acc else 0, out_f32=dx, out_ld=dx_ld,
                           n=nrows, w_row_off=roff)
            wg, wgp = s.get("wgrad"), s.get("wgrad_packed")
            if wg is None and wgp is None:
                continue
            k = ntaps * c
            ftab = s.get("ftab")
            if s["_defer"]:
                key = (M, n, c, hw_out if ftab is not None else hw_dw, ntaps, s["hw_src"] if ftab is not None else hw_dw,
                       ftab.data_ptr() if ftab is not None else 0)
                self._dw_pending.setdefault(key, []).append(
                    dict(what=f"{what}:dW{si}", dpl=dpl, d_ld=npad, planes=planes, col_off=s.get("col_off", 0), wg=wg, out_ld=k,
                         ftab=ftab))
                continue
            if s["_dw"]:
                self._dw(ops, f"{what}:dW{si}", dpl, npad, planes, s.get("col_off", 0), c, ftab, ntaps,
                         hw_out if ftab is not None else hw_dw, s["hw_src"] if ftab is not None else hw_dw, M, n, wg, k,
                         self._pacc(wg))
                continue
            xT = self._scratch("xT", 2 * self._max_xT, torch.bfloat16)[: 2 * k * mpad].view(2, k, mpad)
            ops.append((lib.wd_transpose_planes,
                        (planes[0].data_ptr() + 2 * s.get("col_off", 0), planes[1].data_ptr() + 2 * s.get("col_off", 0), 0,
                         planes.shape[2], c, _ptr(ftab), ntaps, hw_out if ftab is not None else 0,
                         s["hw_src"] if ftab is not None else 0, M, mpad, 1 if wgp is None else 0, xT[0].data_ptr(),
                         xT[1].data_ptr()), f"{what}:T(x{si})"))
            if wgp is not None:
                packed, dst, cin = wgp
                self._gemm_dw(ops, f"{what}:dW{si}", doutT, n, xT, k, mpad, packed, k, 0)
                ops.append((lib.wd_permute_dw, (packed.data_ptr(), k, n, cin, 9, dst.data_ptr()), f"{what}:dW{si}:oihw"))
                self._pacc(dst)
            else:
                self._gemm_dw(ops, f"{what}:dW{si}", doutT, n, xT, k, mpad, wg, k, self._pacc(wg))
        if film_off is not None:
            # per-sample column sums are the FiLM gradient; summing those over the batch is the bias gradient
            B = M // hw_out
            if fuse_cs:
                ops.append((lib.wd_colsum_finish, (colpart.data_ptr(), hw_out // 64, n, B, self._dfilm.data_ptr() + 4 * film_off,
                                                   self.film_total, 0, 1.0), what + ":dfilm"))
            else:
                self._colsum(ops, what + ":dfilm", dout.data_ptr(), ldd, M, n, hw_out,
                             self._dfilm.data_ptr() + 4 * film_off, self.film_total, 0)
            for b in bias:
                if fuse_cs:
                    self._bias_finish(ops, what + ":dbias", colpart, mpad // 64, n, b)
                else:
                    self._colsum(ops, what + ":dbias", self._dfilm.data_ptr() + 4 * film_off, self.film_total, B, n, B,
                                 b.data_ptr(), n, self._pacc(b))
        else:
            for b in bias:
                if fuse_cs:
                    self._bias_finish(ops, what + ":dbias", colpart, mpad // 64, n, b)
                else:
                    self._colsum(ops, what + ":dbias", dout.data_ptr(), ldd, M, n, M, b.data_ptr(), n, self._pacc(b))

    def _gn_bwd(self, P, what, srcs: List[TAct], gn: torch.nn.GroupNorm, eps, silu, dz: torch.Tensor):
        ops = P.bwd
        lib = self.lib
        B = self._B
        hw = srcs[0].h * srcs[0].w
        ctot = sum(s.c for s in srcs)
        cpg = ctot // 32
        gam, bet = gn.weight, gn.bias
        gw, gb = self._w[self._gn_names[id(gn)] + ".g"], self._w[self._gn_names[id(gn)] + ".b"]
        pair = self._ppair(bet, gam)  # [d beta | d gamma], the order of the planar sums
        dbet, dgam = pair[0], pair[1]
        # one pass (the workgroup keeps its tile of dy / xhat in LDS) where the shape allows, else statistics pass + apply pass
        fused = self.fuse_gn_bwd and all(lib.wd_gn_bwd_fused_supported(hw, s.c, cpg) for s in srcs)
        nb = 1 if fused else lib.wd_gn_bwd_nchunk(hw)
        off = 0
        accp = self._pacc(pair)
        for s in srcs:
            part, nchunk, pc = s.stats
            sums = self._f32(P, B, nb, 2, s.c)
            common = (s.t.data_ptr(), s.c, dz.data_ptr(), ctot, off, B, hw, s.c, cpg, part.data_ptr(), nchunk, pc,
                      gw.data_ptr(), gb.data_ptr(), off, eps, int(silu), sums.data_ptr())
            g, acc = self._gacc(P, s)
            if fused:
                ops.append((lib.wd_gn_bwd_fused, common + (g.data_ptr(), s.c, acc), what + ":bwd"))
            else:
                ops.append((lib.wd_gn_bwd_stats, common, what + ":stats"))
                ops.append((lib.wd_gn_bwd_apply, common + (g.data_ptr(), s.c, acc), what + ":apply"))
            if len(srcs) == 1:
                self._param_colsum(ops, what + ":dbeta|dgamma", sums.data_ptr(), 2 * s.c, B * nb, 2 * s.c, pair.data_ptr(), accp)
            else:
                self._param_colsum(ops, what + ":dbeta", sums.data_ptr(), 2 * s.c, B * nb, s.c, dbet.data_ptr() + 4 * off, accp)
                self._param_colsum(ops, what + ":dgamma", sums.data_ptr() + 4 * s.c, 2 * s.c, B * nb, s.c,
                                   dgam.data_ptr() + 4 * off, accp)
            off += s.c

    def _ln_bwd(self, P, what, x: torch.Tensor, rows, c, ln: torch.nn.LayerNorm, name, dy: torch.Tensor, dx: torch.Tensor,
                acc: int):
        ops = P.bwd
        lib = self.lib
        nblk = lib.wd_layernorm_bwd_nblk(rows)
        colpart = self._f32(P, nblk, 2, c)
        ops.append((lib.wd_layernorm_bwd, (x.data_ptr(), c, dy.data_ptr(), c, rows, c, self._w[name + ".g"].data_ptr(), 1e-5,
                                           dx.data_ptr(), c, int(acc), colpart.data_ptr()), what))
        pair = self._ppair(ln.weight, ln.bias)  # [d gamma | d beta], the order of colpart
        self._param_colsum(ops, what + ":dgamma|dbeta", colpart.data_ptr(), 2 * c, nblk, 2 * c, pair.data_ptr(), self._pacc(pair))

    def _attn_bwd(self, P, what, q_ptr, ldq, k_ptr, ldk, v_ptr, ldv, dO: torch.Tensor, heads, nq, nk, d, scale, dq_ptr, lddq,
                  dkv_ptr, dkv_pitch_floats):
        """dq -> dq_ptr; [dK | dV] rows (b, j) -> dkv_ptr with the given row pitch (2*inner floats wide)."""
        ops = P.bwd
        lib = self.lib
        B = self._B
        inner = heads * d
        if nk > 16:
            # long key sets (spatial self-attention, PHOSC context): dK | dV land in the destination rows directly
            need = lib.wd_attention_bwd_scratch_floats(B, heads, nq, nk)
            scr = self._scratch("attn_bwd", max(need, self._max_attn_scr), torch.float32)
            ops.append((lib.wd_attention_bwd, (q_ptr, ldq, k_ptr, ldk, v_ptr, ldv, dO.data_ptr(), dO.shape[1], B, heads, nq, nk, d,
                                               float(scale), dq_ptr, lddq, dkv_ptr, dkv_pitch_floats, dkv_ptr + 4 * inner,
                                               dkv_pitch_floats, scr.data_ptr(), scr.numel()), what))
            return
        nwg = lib.wd_attention_bwd_small_nwg(heads, nq, nk, d)
        if nwg <= 0:
            raise NotImplementedError(f"attention backward shape heads={heads} nq={nq} nk={nk} d={d}")
        part = self._f32(P, B, nwg, nk, 2, inner)
        ops.append((lib.wd_attention_bwd_small, (q_ptr, ldq, k_ptr, ldk, v_ptr, ldv, dO.data_ptr(), dO.shape[1], B, heads, nq, nk,
                                                 d, float(scale), dq_ptr, lddq, part.data_ptr(), None), what))
        row = nk * 2 * inner
        tmp = self._f32(P, B, row)
        self._colsum(ops, what + ":dkv", part.data_ptr(), row, B * nwg, row, nwg, tmp.data_ptr(), row, 0)
        ops.append((lib.wd_copy2d, (dkv_ptr, 4 * dkv_pitch_floats, tmp.data_ptr(), 8 * inner, 8 * inner, B * nk),
                    what + ":dkv->slice"))

    # ------------------------------------------------------------------------------------------ blocks (fwd + tape)
    def _resblock(self, P, name, mod: ResBlockParams, srcs: List[TAct]) -> TAct:
        ops = P.step
        B = self._B
        h, w = srcs[0].h, srcs[0].w
        hw, M = h * w, B * h * w
        cin, cout = mod.cin, mod.cout
        tab, _, _ = self._table(h, w, "same")
        need_raw = cin != cout
        if mod.out_layers[2].p != 0:
            raise NotImplementedError("dropout > 0 in the HIP training step (train.py builds the UNet with dropout 0)")
        cpg = cin // 32
        parts = None
        if any(s.c % cpg for s in srcs):
            # GroupNorm groups straddle the skip concat (never at 320+320): materialise it (unet.py:1750) and hand its
            # gradient back to the two halves at the end of the block's backward
            parts = srcs
            catt = self._f32(P, M, cin)
            coff = 0
            for s in parts:
                ops.append((self.lib.wd_copy2d, (catt.data_ptr() + 4 * coff, 4 * cin, s.t.data_ptr(), 4 * s.c, 4 * s.c, M),
                            name + ":concat"))
                coff += s.c
            srcs = [TAct(catt, cin, h, w)]
        a1, raw = self._gn(P, ops, name + ".gn1", srcs, name + ".gn1", 1e-5, True, want_raw=need_raw)
        h1t = self._f32(P, M, cout)
        g1 = self._gemm(ops, name + ".conv1", [self._src(a1, cin, 9, tab, hw)], name + ".c1.w", M, hw,
                        bias=self._w[name + ".c1.b"], rowvec=self._film.data_ptr() + 4 * self.film_off[name],
                        rowvec_ld=self.film_total, out_f32=h1t, out_ld=cout, want_stats=True)
        h1 = TAct(h1t, cout, h, w, g1._stats)
        a2, _ = self._gn(P, ops, name + ".gn2", [h1], name + ".gn2", 1e-5, True)
        outt = self._f32(P, M, cout)
        if need_raw:
            g2 = self._gemm(ops, name + ".conv2+skip", [self._src(a2, cout, 9, tab, hw), self._src(raw, cin)],
                            name + ".c2.w", M, hw, bias=self._w[name + ".c2.b"], out_f32=outt, out_ld=cout,
                            want_stats=True)
        else:
            g2 = self._gemm(ops, name + ".conv2", [self._src(a2, cout, 9, tab, hw)], name + ".c2.w", M, hw,
                            bias=self._w[name + ".c2.b"], resid=srcs[0].t.data_ptr(), resid_ld=cout, out_f32=outt,
                            out_ld=cout, want_stats=True)
        out = TAct(outt, cout, h, w, g2._stats)
        self._gn_names[id(mod.in_layers[0])] = name + ".gn1"
        self._gn_names[id(mod.out_layers[0])] = name + ".gn2"

        def bwd():
            bops = P.bwd
            assert out.gw, name
            dO = out.g
            btab = self._btable(h, w, "same")
            da2 = self._f32(P, M, cout)
            segs = [dict(planes=a2, c=cout, ntaps=9, ftab=tab, btab=btab, hw_src=hw, wb="B:" + name + ".c2.w",
                         wgrad=self._pgrad(mod.out_layers[3].weight).view(cout, -1), dx=[(da2, cout, 0, 0, cout)],
                         dx_rows=M, dx_hw=hw)]
            biases = [self._pgrad(mod.out_layers[3].bias)]
            if need_raw:
                dxs, coff = [], 0
                for s in srcs:
                    g, acc = self._gacc(P, s)
                    dxs.append((g, s.c, acc, coff, s.c))
                    coff += s.c
                segs.append(dict(planes=raw, c=cin, ntaps=1, hw_src=hw, wb="B:" + name + ".skip.w",
                                 wgrad=self._pgrad(mod.skip_connection.weight).view(cout, cin), dx=dxs, dx_rows=M, dx_hw=hw))
                biases.append(self._pgrad(mod.skip_connection.bias))
            else:
                g, acc = self._gacc(P, srcs[0])
                if acc:
                    bops.append((self.lib.wd_add, (g.data_ptr(), dO.data_ptr(), g.numel()), name + ":dresid"))
                else:
                    bops.append((self.lib.wd_copy2d, (g.data_ptr(), 4 * cout, dO.data_ptr(), 4 * cout, 4 * cout, M),
                                 name + ":dresid"))
            self._bwd_linear(P, name + ".conv2", dO, M, cout, hw, segs, bias=biases)
            self._gn_bwd(P, name + ".gn2", [h1], mod.out_layers[0], 1e-5, True, da2)
            da1 = self._f32(P, M, cin)
            self._bwd_linear(P, name + ".conv1", h1.g, M, cout, hw,
                             [dict(planes=a1, c=cin, ntaps=9, ftab=tab, btab=btab, hw_src=hw, wb="B:" + name + ".c1.w",
                                   wgrad=self._pgrad(mod.in_layers[2].weight).view(cout, -1), dx=[(da1, cin, 0, 0, cin)],
                                   dx_rows=M, dx_hw=hw)],
                             bias=[self._pgrad(mod.in_layers[2].bias)], film_off=self.film_off[name])
            self._gn_bwd(P, name + ".gn1", srcs, mod.in_layers[0], 1e-5, True, da1)
            if parts is not None:
                coff = 0
                for s in parts:
                    g, acc = self._gacc(P, s)
                    dst = self._f32(P, M, s.c) if acc else g
                    bops.append((self.lib.wd_copy2d, (dst.data_ptr(), 4 * s.c, srcs[0].g.data_ptr() + 4 * coff, 4 * cin, 4 * s.c,
                                                      M), name + ":d(concat)"))
                    if acc:
                        bops.append((self.lib.wd_add, (g.data_ptr(), dst.data_ptr(), g.numel()), name + ":d(concat)+"))
                    coff += s.c

        self._tape.append(bwd)
        return out

    def _resample(self, P, name, mod, x: TAct, mode: str) -> TAct:
        ops = P.step
        B = self._B
        tab, ho, wo = self._table(x.h, x.w, mode)
        hw_in, M_in = x.h * x.w, B * x.h * x.w
        pl = self._planes(P, M_in, x.c)
        ops.append((self.lib.wd_split, (x.t.data_ptr(), x.c, M_in, x.c, 0, pl[0].data_ptr(),
                                        pl[1].data_ptr() if self.npass == 3 else None, x.c), name + ":split"))
        outt = self._f32(P, B * ho * wo, mod.cout)
        gg = self._gemm(ops, name + ".conv", [self._src(pl, x.c, 9, tab, hw_in)], name + ".w", B * ho * wo, ho * wo,
                        bias=self._w[name + ".b"], out_f32=outt, out_ld=mod.cout, want_stats=True)
        out = TAct(outt, mod.cout, ho, wo, gg._stats)
        conv = mod.op if mode == "down" else mod.conv

        def bwd():
            bops = P.bwd
            assert out.gw, name
            M_out, hw_out = B * ho * wo, ho * wo
            g, acc = self._gacc(P, x)
            if mode == "down":
                btab = self._btable(x.h, x.w, "down")
                dx, dx_rows, dx_hw = [(g, x.c, acc, 0, x.c)], M_in, hw_in
            else:  # nearest x2: data gradient at the upsampled resolution, then 2x2 sums
                btab = self._btable(ho, wo, "same")
                du = self._f32(P, M_out, x.c)
                dx, dx_rows, dx_hw = [(du, x.c, 0, 0, x.c)], M_out, hw_out
            self._bwd_linear(P, name + ".conv", out.g, M_out, mod.cout, hw_out,
                             [dict(planes=pl, c=x.c, ntaps=9, ftab=tab, btab=btab, hw_src=hw_in, wb="B:" + name + ".w",
                                   wgrad=self._pgrad(conv.weight).view(mod.cout, -1), dx=dx, dx_rows=dx_rows, dx_hw=dx_hw)],
                             bias=[self._pgrad(conv.bias)])
            if mode == "up":
                if acc:
                    tmp = self._f32(P, M_in, x.c)
                    bops.append((self.lib.wd_pool2x2_sum, (du.data_ptr(), B, x.h, x.w, x.c, tmp.data_ptr()), name + ":pool"))
                    bops.append((self.lib.wd_add, (g.data_ptr(), tmp.data_ptr(), g.numel()), name + ":pool+"))
                else:
                    bops.append((self.lib.wd_pool2x2_sum, (du.data_ptr(), B, x.h, x.w, x.c, g.data_ptr()), name + ":pool"))

        self._tape.append(bwd)
        return out

    def _transformer(self, P, name, mod: SpatialTransformerParams, x: TAct) -> TAct:
        ops = P.step
        lib = self.lib
        B = self._B
        h, w, c = x.h, x.w, x.c
        hw, M = h * w, B * h * w
        heads, d = mod.heads, mod.d_head
        inner = heads * d
        L = self._ctx_len
        scale = d ** -0.5
        lo_ok = self.npass == 3
        gpl, _ = self._gn(P, ops, name + ".gn", [x], name + ".gn", 1e-6, False)
        self._gn_names[id(mod.norm)] = name + ".gn"
        tok = self._f32(P, M, inner)
        self._gemm(ops, name + ".proj_in", [self._src(gpl, c)], name + ".pi.w", M, hw, bias=self._w[name + ".pi.b"],
                   out_f32=tok, out_ld=inner)
        saved = []
        cur = tok
        for di, tb in enumerate(mod.transformer_blocks):
            p = f"{name}.tb{di}"
            rec = dict(tb=tb, p=p, tok=cur)
            # base model: both attentions are cross-attentions reading norm2 (unet.py:337-345); PHOSC model: attn1 is the
            # spatial self-attention behind norm1 (unetPhosc.py:241-246)
            for tag, at in (("a1", tb.attn1), ("a2", tb.attn2)):
                self_attn = tag == "a1" and self.variant == "phosc"
                ln = "norm1" if self_attn else "norm2"
                n_pl = self._ln(P, ops, f"{p}.{ln}({tag})", cur, M, inner, f"{p}.{ln}")
                o = self._planes(P, M, inner)
                if self_attn:
                    qkv = self._f32(P, M, 3 * inner)
                    self._gemm(ops, p + ".a1.qkv", [self._src(n_pl, inner)], p + ".a1.qkv.w", M, hw, out_f32=qkv,
                               out_ld=3 * inner)
                    self._attention(ops, p + ".a1", qkv.data_ptr(), 3 * inner, qkv.data_ptr() + 4 * inner, 3 * inner,
                                    qkv.data_ptr() + 8 * inner, 3 * inner, heads, hw, hw, d, scale, o)
                    rec[tag] = dict(at=at, x=cur, n=n_pl, qkv=qkv, o=o, self_attn=True, ln=ln)
                else:
                    q = self._f32(P, M, inner)
                    self._gemm(ops, f"{p}.{tag}.q", [self._src(n_pl, inner)], f"{p}.{tag}.q.w", M, hw, out_f32=q, out_ld=inner)
                    ko = self.kv_off[f"{p}.{tag}"]
                    self._attention(ops, f"{p}.{tag}", q.data_ptr(), inner, self._kv.data_ptr() + 4 * ko, self.kv_total,
                                    self._kv.data_ptr() + 4 * (ko + inner), self.kv_total, heads, hw, L, d, scale, o)
                    rec[tag] = dict(at=at, x=cur, n=n_pl, q=q, o=o, ko=ko, self_attn=False, ln=ln)
                nxt = self._f32(P, M, inner)
                self._gemm(ops, f"{p}.{tag}.out", [self._src(o, inner)], f"{p}.{tag}.o.w", M, hw,
                           bias=self._w[f"{p}.{tag}.o.b"], resid=cur.data_ptr(), resid_ld=inner, out_f32=nxt, out_ld=inner)
                cur = nxt
            ffi = tb.ff.net[2].in_features
            n3 = self._ln(P, ops, p + ".norm3", cur, M, inner, p + ".norm3")
            u = self._f32(P, M, 2 * ffi)
            self._gemm(ops, p + ".ff1", [self._src(n3, inner)], p + ".ff1u.w", M, hw, bias=self._w[p + ".ff1u.b"], out_f32=u,
                       out_ld=2 * ffi)
            ffh = self._planes(P, M, ffi)
            ops.append((lib.wd_geglu_fwd, (u.data_ptr(), 2 * ffi, M, ffi, ffh[0].data_ptr(), ffh[1].data_ptr() if lo_ok else None,
                                           ffi), p + ".geglu"))
            nxt = self._f32(P, M, inner)
            xpl = self._planes(P, M, inner)
            self._gemm(ops, p + ".ff2", [self._src(ffh, ffi)], p + ".ff2.w", M, hw, bias=self._w[p + ".ff2.b"],
                       resid=cur.data_ptr(), resid_ld=inner, out_f32=nxt, out_ld=inner, out_pl=xpl)
            rec.update(x3=cur, n3=n3, u=u, ffh=ffh, ffi=ffi, out=nxt, xpl=xpl)
            saved.append(rec)
            cur = nxt
        outt = self._f32(P, M, c)
        gg = self._gemm(ops, name + ".proj_out", [self._src(saved[-1]["xpl"], inner)], name + ".po.w", M, hw,
                        bias=self._w[name + ".po.b"], resid=x.t.data_ptr(), resid_ld=c, out_f32=outt, out_ld=c,
                        want_stats=True)
        out = TAct(outt, c, h, w, gg._stats)

        def bwd():
            bops = P.bwd
            assert out.gw, name
            dO = out.g
            g, acc = self._gacc(P, x)
            if acc:
                bops.append((lib.wd_add, (g.data_ptr(), dO.data_ptr(), g.numel()), name + ":dresid"))
            else:
                bops.append((lib.wd_copy2d, (g.data_ptr(), 4 * c, dO.data_ptr(), 4 * c, 4 * c, M), name + ":dresid"))
            dcur = self._f32(P, M, inner)  # gradient of the running token stream (residual adds share it)
            self._bwd_linear(P, name + ".proj_out", dO, M, c, hw,
                             [dict(planes=saved[-1]["xpl"], c=inner, ntaps=1, hw_src=hw, wb="B:" + name + ".po.w",
                                   wgrad=self._pgrad(mod.proj_out.weight).view(c, inner), dx=[(dcur, inner, 0, 0, inner)],
                                   dx_rows=M, dx_hw=hw)], bias=[self._pgrad(mod.proj_out.bias)])
            for rec in reversed(saved):
                tb, p, ffi = rec["tb"], rec["p"], rec["ffi"]
                # ---- feed-forward: out = ff2(geglu(ff1(LN3(x3)))) + x3
                dffh = self._f32(P, M, ffi)
                self._bwd_linear(P, p + ".ff2", dcur, M, inner, hw,
                                 [dict(planes=rec["ffh"], c=ffi, ntaps=1, hw_src=hw, wb="B:" + p + ".ff2.w",
                                       wgrad=self._pgrad(tb.ff.net[2].weight), dx=[(dffh, ffi, 0, 0, ffi)], dx_rows=M,
                                       dx_hw=hw)], bias=[self._pgrad(tb.ff.net[2].bias)])
                dn3 = self._f32(P, M, inner)
                ff1_seg = [dict(planes=rec["n3"], c=inner, ntaps=1, hw_src=hw, wb="B:" + p + ".ff1.w",
                                wgrad=self._pgrad(tb.ff.net[0].proj.weight), dx=[(dn3, inner, 0, 0, inner)], dx_rows=M, dx_hw=hw)]
                if self.fuse_geglu_bwd and ffi % 64 == 0:
                    # d(GEGLU pre-activation) goes straight to operand planes + bias column sums (never written as fp32)
                    self._bwd_linear(P, p + ".ff1", None, M, 2 * ffi, hw, ff1_seg, bias=[self._pgrad(tb.ff.net[0].proj.bias)],
                                     geglu=(rec["u"], dffh, ffi))
                else:
                    du = self._f32(P, M, 2 * ffi)
                    bops.append((lib.wd_geglu_bwd, (rec["u"].data_ptr(), 2 * ffi, dffh.data_ptr(), ffi, M, ffi, du.data_ptr(), 2 * ffi),
                                 p + ".geglu:bwd"))
                    self._bwd_linear(P, p + ".ff1", du, M, 2 * ffi, hw, ff1_seg, bias=[self._pgrad(tb.ff.net[0].proj.bias)])
                self._ln_bwd(P, p + ".norm3:bwd", rec["x3"], M, inner, tb.norm3, p + ".norm3", dn3, dcur, 1)
                # ---- the two cross-attentions, last first
                for tag in ("a2", "a1"):
                    r = rec[tag]
                    at = r["at"]
                    do = self._f32(P, M, inner)
                    self._bwd_linear(P, f"{p}.{tag}.out", dcur, M, inner, hw,
                                     [dict(planes=r["o"], c=inner, ntaps=1, hw_src=hw, wb=f"B:{p}.{tag}.o.w",
                                           wgrad=self._pgrad(at.to_out[0].weight), dx=[(do, inner, 0, 0, inner)], dx_rows=M,
                                           dx_hw=hw)], bias=[self._pgrad(at.to_out[0].bias)])
                    dn = self._f32(P, M, inner)
                    if r["self_attn"]:
                        qkv = r["qkv"]
                        dqkv = self._f32(P, M, 3 * inner)
                        self._attn_bwd(P, f"{p}.a1:bwd", qkv.data_ptr(), 3 * inner, qkv.data_ptr() + 4 * inner, 3 * inner,
                                       qkv.data_ptr() + 8 * inner, 3 * inner, do, heads, hw, hw, d, scale, dqkv.data_ptr(),
                                       3 * inner, dqkv.data_ptr() + 4 * inner, 3 * inner)
                        self._bwd_linear(P, p + ".a1.qkv", dqkv, M, 3 * inner, hw,
                                         [dict(planes=r["n"], c=inner, ntaps=1, hw_src=hw, wb=f"B:{p}.a1.qkv.w",
                                               wgrad=self._pgroup([at.to_q.weight, at.to_k.weight, at.to_v.weight]),
                                               dx=[(dn, inner, 0, 0, inner)], dx_rows=M, dx_hw=hw)])
                    else:
                        dq = self._f32(P, M, inner)
                        ko = r["ko"]
                        self._attn_bwd(P, f"{p}.{tag}:bwd", r["q"].data_ptr(), inner, self._kv.data_ptr() + 4 * ko,
                                       self.kv_total, self._kv.data_ptr() + 4 * (ko + inner), self.kv_total, do, heads, hw, L, d,
                                       scale, dq.data_ptr(), inner, self._dkv.data_ptr() + 4 * ko, self.kv_total)
                        self._bwd_linear(P, f"{p}.{tag}.q", dq, M, inner, hw,
                                         [dict(planes=r["n"], c=inner, ntaps=1, hw_src=hw, wb=f"B:{p}.{tag}.q.w",
                                               wgrad=self._pgrad(at.to_q.weight), dx=[(dn, inner, 0, 0, inner)], dx_rows=M,
                                               dx_hw=hw)])
                    self._ln_bwd(P, f"{p}.{r['ln']}({tag}):bwd", r["x"], M, inner, getattr(tb, r["ln"]), f"{p}.{r['ln']}", dn,
                                 dcur, 1)
            dg = self._f32(P, M, c)
            self._bwd_linear(P, name + ".proj_in", dcur, M, inner, hw,
                             [dict(planes=gpl, c=c, ntaps=1, hw_src=hw, wb="B:" + name + ".pi.w",
                                   wgrad=self._pgrad(mod.proj_in.weight).view(inner, c), dx=[(dg, c, 0, 0, c)], dx_rows=M,
                                   dx_hw=hw)], bias=[self._pgrad(mod.proj_in.bias)])
            self._gn_bwd(P, name + ".gn", [x], mod.norm, 1e-6, False, dg)

        self._tape.append(bwd)
        return out

    # ------------------------------------------------------------------------------------------ plan
    def _size_scratch(self, B, H, W, L, ctx_len=0, phosc_len=0):
        m = self.model
        mc = m.model_channels
        cmax = mc * max(m.channel_mult)
        M = max(B * H * W, B * L)
        mpad = _rup(M, 64)
        nmax = max(8 * cmax, self.film_total, self.kv_total, 4 * mc)        # widest d(output): GEGLU pre-activation
        kmax = max(9 * 2 * cmax, 4 * cmax, 4 * mc, self.kpad_in)           # longest (tap, channel) list: 3x3 over a concat
        self._max_dpl = M * nmax
        self._max_doutT = nmax * mpad
        self._max_xT = kmax * mpad
        heads_max = max([mod.heads for _, mod in self._walk() if isinstance(mod, SpatialTransformerParams)] or [1])
        scr = 0
        if self.variant == "phosc" and H * W > 16:
            scr = max(scr, 2 * B * heads_max * (H * W) ** 2)
        if L > 16:
            scr = max(scr, 2 * B * heads_max * H * W * L)
        for n_tok in (ctx_len, phosc_len):
            if n_tok > 16:
                scr = max(scr, 2 * B * n_tok * n_tok)
        self._max_attn_scr = scr
        if scr:
            self._scratch("attn_bwd", scr, torch.float32)
        self._max_colpart = (mpad // 64) * nmax
        self._scratch("colpart", self._max_colpart, torch.float32)
        self._scratch("dpl", 2 * self._max_dpl, torch.bfloat16)
        self._scratch("doutT", 2 * self._max_doutT, torch.bfloat16)
        self._scratch("xT", 2 * self._max_xT, torch.bfloat16)
        self._cs_scratch = self._scratch("colsum", max(1 << 22, (M // 64 + 1) * nmax), torch.float32)
        if self._ws is None or self._ws.numel() < 8 * 2 * cmax * 9 * cmax:
            self._ws = torch.empty(max(128 * 128 * 160 * 8, 8 * 2 * cmax * 9 * cmax), dtype=torch.float32, device=self.device)

    def plan_train(self, B: int, H: int, W: int, ctx_len: int, phosc_len: int = 0) -> TrainPlan:
        key = (B, H, W, ctx_len, phosc_len, self.npass)
        if key in self._tplans:
            return self._tplans[key]
        if self._tplans:
            # scratch buffers and the gradient-accumulate flags are sized / decided per plan: one live shape at a time
            self._tplans.clear()
            self._scr = {k: v for k, v in self._scr.items() if isinstance(k, tuple)}
        m = self.model
        lib = self.lib
        if ctx_len == 0:
            raise NotImplementedError("context=None: every reference script conditions on the word (unet.py:1605)")
        if phosc_len and self.variant != "phosc":
            raise ValueError("phoscLabels are an input of UNetModelPhosc only")
        P = TrainPlan()
        self._cur_plan = P
        self._B = B
        L = ctx_len + phosc_len
        self._ctx_len = L
        self._tape = []
        self._pw = set()
        self._deferred, self._deferred_outs = [], {}
        self._gn_names = {}
        self._done_at, self._closure = {}, 0
        dev = self.device
        mc = m.model_channels
        ted = 4 * mc
        cd = m.context_dim
        lo_ok = self.npass == 3
        self._size_scratch(B, H, W, L, ctx_len, phosc_len)
        step = P.step

        P.x_in = torch.zeros((B, m.in_channels, H, W), dtype=torch.float32, device=dev)
        P.t_in = torch.zeros((B,), dtype=torch.int64, device=dev)
        P.y_in = torch.zeros((B,), dtype=torch.int64, device=dev)
        P.ctx_in = torch.zeros((B, ctx_len), dtype=torch.int64, device=dev)
        P.phosc_in = torch.zeros((B, max(phosc_len, 1)), dtype=torch.int32, device=dev)

        # ---- conditioning path (differentiated, so it is part of every step): CharacterEncoder + all K/V projections
        we = m.word_emb
        msl = m.max_seq_len
        ctx_pl = self._planes(P, B * L, cd)
        groups = []
        for (ids, n_tok, row0, i64) in ((P.ctx_in, ctx_len, 0, 1), (P.phosc_in, phosc_len, ctx_len, 0)):
            if n_tok == 0:
                continue
            use_pe = (self.variant == "base") or (n_tok <= msl)  # unetPhosc.py:726-729 skips the table for PHOSC vectors
            if use_pe and n_tok > msl:
                raise ValueError(f"context length {n_tok} exceeds max_seq_len {msl} (the reference fails too)")
            e = self._planes(P, B * n_tok, cd)
            step.append((lib.wd_embed_tokens, (ids.data_ptr(), i64, B * n_tok, n_tok, self._w["we.table"].data_ptr(),
                                               self._w["we.table"].shape[0], cd, self._w["pe"].data_ptr() if use_pe else None,
                                               e[0].data_ptr(), e[1].data_ptr() if lo_ok else None, cd), "word_emb.embedding"))
            qkv = self._f32(P, B * n_tok, 3 * cd)
            self._gemm(step, "word_emb.qkv", [self._src(e, cd)], "we.qkv.w", B * n_tok, n_tok, bias=self._w["we.qkv.b"],
                       out_f32=qkv, out_ld=3 * cd)
            # Word_Attention: softmax(q k^T) v without 1/sqrt(d) (unet.py:831-835)
            self._attention(step, "word_emb.attention", qkv.data_ptr(), 3 * cd, qkv.data_ptr() + 4 * cd, 3 * cd,
                            qkv.data_ptr() + 8 * cd, 3 * cd, 1, n_tok, n_tok, cd, 1.0, ctx_pl, out_rows=L, out_row0=row0)
            groups.append((ids, n_tok, row0, i64, e, qkv))
        self._kv = self._f32(P, B * L, self.kv_total)
        self._dkv = self._f32(P, B * L, self.kv_total)
        self._gemm(step, "cross.kv", [self._src(ctx_pl, cd)], "kv.w", B * L, L, out_f32=self._kv, out_ld=self.kv_total)

        # ---- time / writer embedding with the SiLU pre-activations kept
        te = self._planes(P, B, mc)
        step.append((lib.wd_timestep_embedding, (P.t_in.data_ptr(), B, self._w["freqs"].data_ptr(), mc // 2, te[0].data_ptr(),
                                                 te[1].data_ptr() if lo_ok else None, mc), "timestep_embedding"))
        pre1 = self._f32(P, B, ted)
        self._gemm(step, "time_embed.0", [self._src(te, mc)], "te0.w", B, 1, bias=self._w["te0.b"], out_f32=pre1, out_ld=ted)
        e1 = self._planes(P, B, ted)
        step.append((lib.wd_split, (pre1.data_ptr(), ted, B, ted, 1, e1[0].data_ptr(), e1[1].data_ptr() if lo_ok else None, ted),
                     "time_embed.1(SiLU)"))
        has_lab = m.num_classes is not None
        pre2 = self._f32(P, B, ted)
        self._gemm(step, "time_embed.2+label", [self._src(e1, ted)], "te2.w", B, 1, bias=self._w["te2.b"],
                   resid=self._w["label"].data_ptr() if has_lab else None, resid_ld=ted if has_lab else 0,
                   resid_rows=P.y_in.data_ptr() if has_lab else None, out_f32=pre2, out_ld=ted)
        e2 = self._planes(P, B, ted)
        step.append((lib.wd_split, (pre2.data_ptr(), ted, B, ted, 1, e2[0].data_ptr(), e2[1].data_ptr() if lo_ok else None, ted),
                     "emb_layers.0(SiLU)"))
        self._film = self._f32(P, B, self.film_total)
        self._dfilm = self._f32(P, B, self.film_total)
        self._gemm(step, "emb_layers(all)", [self._src(e2, ted)], "film.w", B, 1, bias=self._w["film.b"], out_f32=self._film,
                   out_ld=self.film_total)

        # ---- trunk
        xin = self._planes(P, B * H * W, self.kpad_in)
        step.append((lib.wd_im2col3x3, (P.x_in.data_ptr(), B, m.in_channels, H, W, xin[0].data_ptr(),
                                        xin[1].data_ptr() if lo_ok else None, self.kpad_in), "im2col"))
        h0 = self._f32(P, B * H * W, mc)
        g0 = self._gemm(step, "input_blocks.0", [self._src(xin, self.kpad_in)], "in.w", B * H * W, H * W, bias=self._w["in.b"],
                        out_f32=h0, out_ld=mc, want_stats=True)
        first = TAct(h0, mc, H, W, g0._stats)
        cur = first
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
        last = cur
        self._gn_names[id(m.out[0])] = "out.gn"
        gpl, _ = self._gn(P, step, "out.gn", [last], "out.gn", 1e-5, True)
        tab, _, _ = self._table(last.h, last.w, "same")
        oc = m.out_channels
        Mo, hwo = B * last.h * last.w, last.h * last.w
        otok = self._f32(P, Mo, oc)
        self._gemm(step, "out.conv", [self._src(gpl, last.c, 9, tab, hwo)], "out.w", Mo, hwo, bias=self._w["out.b"],
                   out_f32=otok, out_ld=oc)
        P.out = torch.empty((B, oc, last.h, last.w), dtype=torch.float32, device=dev)
        step.append((lib.wd_tokens_to_nchw, (otok.data_ptr(), oc, B, oc, hwo, P.out.data_ptr()), "tokens_to_nchw"))

        # ================================ backward list ================================
        bops = P.bwd
        P.dout = torch.zeros((B, oc, last.h, last.w), dtype=torch.float32, device=dev)
        if oc > 32:
            raise NotImplementedError("out_channels > 32")
        dtok = torch.zeros((Mo, 32), dtype=torch.float32, device=dev)  # columns >= oc stay zero
        P.keep.append(dtok)
        bops.append((lib.wd_nchw_to_tokens, (P.dout.data_ptr(), B, oc, hwo, dtok.data_ptr(), 32), "d(out):tokens"))
        dg = self._f32(P, Mo, last.c)
        self._bwd_linear(P, "out.conv", dtok, Mo, oc, hwo,
                         [dict(planes=gpl, c=last.c, ntaps=9, ftab=tab, btab=self._btable(last.h, last.w, "same"), hw_src=hwo,
                               wb="B:out.w", wgrad=self._pgrad(m.out[2].weight).view(oc, -1), dx=[(dg, last.c, 0, 0, last.c)],
                               dx_rows=Mo, dx_hw=hwo)], bias=[self._pgrad(m.out[2].bias)], npad=32)
        self._gn_bwd(P, "out.gn", [last], m.out[0], 1e-5, True, dg)
        # closures = units of the backward list (a layer each); bucket boundaries fall between closures, where the deferred
        # column sums collected so far are finished, so that every gradient written up to there is final
        # (bytes the backward pass will write: the dead heads of the reference - res.*, wrd_proj, attnc, to_kv, norm1 of the base
        # model - never get a gradient; this only balances the buckets, correctness does not depend on it)
        def _live(name):
            return not (name.startswith("res.") or name.startswith("wrd_proj.") or ".attnc." in name or ".to_kv." in name or
                        (self.variant == "base" and ".norm1." in name))
        total = sum(_rup(p.numel(), 64) for n_, p in m.named_parameters() if _live(n_))
        first_plan = self._cut_closures is None
        cuts_seen: List[int] = []
        P.bwd_cuts = []  # [(index into P.bwd where the segment ends, closure index)]
        self._closure = 1
        for fn in reversed(self._tape):
            fn()
            self._flush_dw(P)
            want = (first_plan and self.nbuckets > 1 and len(cuts_seen) < self.nbuckets - 1 and
                    self._arena_used >= (len(cuts_seen) + 1) * total // self.nbuckets) or \
                   (not first_plan and self._closure in self._cut_closures)
            if want:
                self._flush_deferred(P)
                cuts_seen.append(self._closure)
                P.bwd_cuts.append((len(P.bwd), self._closure))
            self._closure += 1
        if first_plan:
            self._cut_closures = cuts_seen
        # ---- input convolution: weight / bias gradient only
        assert first.gw
        cin0 = m.in_channels
        packed = self._f32(P, mc, self.kpad_in)
        self._bwd_linear(P, "input_blocks.0", first.g, B * H * W, mc, H * W,
                         [dict(planes=xin, c=self.kpad_in, ntaps=1, hw_src=H * W,
                               wgrad_packed=(packed, self._pgrad(m.input_blocks[0][0].weight), cin0))],
                         bias=[self._pgrad(m.input_blocks[0][0].bias)])
        # ---- FiLM projections -> time embedding MLP / writer embedding
        film_w = self._pgroup([l.weight for l in self._film_mods])
        film_b = self._pgroup([l.bias for l in self._film_mods])
        de2 = self._f32(P, B, ted)
        self._bwd_linear(P, "emb_layers(all)", self._dfilm, B, self.film_total, 1,
                         [dict(planes=e2, c=ted, ntaps=1, hw_src=1, wb="B:film.w", wgrad=film_w, dx=[(de2, ted, 0, 0, ted)],
                               dx_rows=B, dx_hw=1)], bias=[film_b])
        dpre2 = self._f32(P, B, ted)
        bops.append((lib.wd_silu_bwd, (pre2.data_ptr(), de2.data_ptr(), B * ted, dpre2.data_ptr()), "emb SiLU:bwd"))
        if has_lab:
            dl = self._pgrad(m.label_emb.weight)
            bops.append((lib.wd_embedding_bwd, (P.y_in.data_ptr(), 1, B, dpre2.data_ptr(), ted, m.num_classes, ted, dl.data_ptr(),
                                                self._pacc(dl)), "label_emb:bwd"))
        de1 = self._f32(P, B, ted)
        self._bwd_linear(P, "time_embed.2", dpre2, B, ted, 1,
                         [dict(planes=e1, c=ted, ntaps=1, hw_src=1, wb="B:te2.w", wgrad=self._pgrad(m.time_embed[2].weight),
                               dx=[(de1, ted, 0, 0, ted)], dx_rows=B, dx_hw=1)], bias=[self._pgrad(m.time_embed[2].bias)])
        dpre1 = self._f32(P, B, ted)
        bops.append((lib.wd_silu_bwd, (pre1.data_ptr(), de1.data_ptr(), B * ted, dpre1.data_ptr()), "time SiLU:bwd"))
        self._bwd_linear(P, "time_embed.0", dpre1, B, ted, 1,
                         [dict(planes=te, c=mc, ntaps=1, hw_src=1, wgrad=self._pgrad(m.time_embed[0].weight))],
                         bias=[self._pgrad(m.time_embed[0].bias)])
        # ---- K/V projections -> word encoder
        kv_params = []
        for at in self._kv_mods:
            kv_params += [at.to_k.weight, at.to_v.weight]
        kv_w = self._pgroup(kv_params)
        dctx = self._f32(P, B * L, cd)
        self._bwd_linear(P, "cross.kv", self._dkv, B * L, self.kv_total, L,
                         [dict(planes=ctx_pl, c=cd, ntaps=1, hw_src=L, wb="B:kv.w", wgrad=kv_w, dx=[(dctx, cd, 0, 0, cd)],
                               dx_rows=B * L, dx_hw=L)])
        at = we.attention
        qkv_w = self._pgroup([at.linear_query.weight, at.linear_key.weight, at.linear_value.weight])
        qkv_b = self._pgroup([at.linear_query.bias, at.linear_key.bias, at.linear_value.bias])
        dtab = self._pgrad(we.embedding.weight)
        for (ids, n_tok, row0, i64, e, qkv) in groups:
            if n_tok == L:
                dgrp = dctx
            else:  # this group's rows of every sample, made contiguous
                dgrp = self._f32(P, B * n_tok, cd)
                bops.append((lib.wd_copy2d, (dgrp.data_ptr(), 4 * n_tok * cd, dctx.data_ptr() + 4 * row0 * cd, 4 * L * cd,
                                             4 * n_tok * cd, B), "d(context):group rows"))
            dqkv = self._f32(P, B * n_tok, 3 * cd)
            self._attn_bwd(P, "word_emb.attention:bwd", qkv.data_ptr(), 3 * cd, qkv.data_ptr() + 4 * cd, 3 * cd,
                           qkv.data_ptr() + 8 * cd, 3 * cd, dgrp, 1, n_tok, n_tok, cd, 1.0, dqkv.data_ptr(), 3 * cd,
                           dqkv.data_ptr() + 4 * cd, 3 * cd)
            de = self._f32(P, B * n_tok, cd)
            self._bwd_linear(P, "word_emb.qkv", dqkv, B * n_tok, 3 * cd, n_tok,
                             [dict(planes=e, c=cd, ntaps=1, hw_src=n_tok, wb="B:we.qkv.w", wgrad=qkv_w,
                                   dx=[(de, cd, 0, 0, cd)], dx_rows=B * n_tok, dx_hw=n_tok)], bias=[qkv_b])
            bops.append((lib.wd_embedding_bwd, (ids.data_ptr(), i64, B * n_tok, de.data_ptr(), cd, we.embedding.weight.shape[0], cd,
                                                dtab.data_ptr(), self._pacc(dtab)), "word_emb.embedding:bwd"))
        self._flush_dw(P)
        self._flush_deferred(P)
        self._tape = []
        # arena prefix that is final at each cut: the longest prefix (in carve order) of buffers whose last writer is a closure
        # <= the cut's (buffers never written by this plan - dead heads - count as final)
        P.bwd_segments = []  # [(bwd_begin, bwd_end, arena_begin, arena_end)]
        lo_op, lo_ar = 0, 0
        for (op_end, closure) in P.bwd_cuts:
            hi_ar = lo_ar
            for (ptr, off, n) in self._carve_order:
                if off < lo_ar:
                    continue
                if self._done_at.get(ptr, 0) > closure:
                    break
                hi_ar = off + n
            P.bwd_segments.append((lo_op, op_end, lo_ar, hi_ar))
            lo_op, lo_ar = op_end, hi_ar
        P.bwd_segments.append((lo_op, len(P.bwd), lo_ar, self._arena_used))
        self._tplans[key] = P
        return P

    # ------------------------------------------------------------------------------------------ run
    def forward_train(self, x, t, context, y, phosc=None):
        """Training forward: keeps every intermediate for ``backward``.  Returns the plan's output buffer (no copy)."""
        self.refresh_weights()
        B, _, H, W = x.shape
        if context is None:
            raise NotImplementedError("context=None")
        self.check_ids(context, y, phosc)
        P = self.plan_train(B, H, W, context.shape[1], 0 if phosc is None else phosc.shape[1])
        P.x_in.copy_(x, non_blocking=True)
        P.t_in.copy_(t, non_blocking=True)
        P.ctx_in.copy_(context, non_blocking=True)
        if phosc is not None:
            P.phosc_in.copy_(phosc.to(torch.int32) if phosc.dtype != torch.int32 else phosc, non_blocking=True)
        if y is not None:
            P.y_in.copy_(y, non_blocking=True)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        P.run_step(stream)
        self._live = P
        return P.out

    def backward(self, dout: torch.Tensor, assign: bool = True):
        """Runs the backward list for d loss / d out (NCHW) and points ``param.grad`` at the gradient buffers (adding to
        a ``.grad`` the caller kept, like autograd's accumulation)."""
        P = self._live
        kept = []
        if assign:
            for k, p in self._params.items():
                if p.grad is not None:
                    # the caller did not reset this gradient to None (zero_grad(set_to_none=False) / accumulation)
                    kept.append((k, p.grad.clone() if p.grad.data_ptr() == self._grad[k].data_ptr() else p.grad))
        P.dout.copy_(dout, non_blocking=True)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        P.run_bwd(stream)
        if assign:
            for k, old in kept:
                self._grad[k].add_(old)
            self.assign_grads()

    def assign_grads(self):
        for k, p in self._params.items():
            p.grad = self._grad[k]

    def grads(self) -> Dict[str, torch.Tensor]:
        names = {id(p): n for n, p in self.model.named_parameters()}
        return {names[k]: g for k, g in self._grad.items()}
