"""GPU parity of each C-ABI kernel (include/wdiff_hip.h) against the CPU oracle / plain fp32-fp64 torch
references of the same op, on seeded inputs.  Tolerances: split-bf16 (npass=3) GEMMs 2e-5 relative to the
fp64 result (operands carry 16 mantissa bits); fp32 streaming kernels 1e-6; integer indexing and the DDPM update
bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ddpm_oracle as D  # noqa: E402
from oracle import unet_oracle as U  # noqa: E402
from tests._common import max_rel, rel_err  # noqa: E402
from worddiffusion_amd import _native as N  # noqa: E402
from worddiffusion_amd.engine import conv_gather_table, geglu_interleave, slab_order, slab_span  # noqa: E402

DEV = "cuda:0"


def _st():
    return torch.cuda.current_stream(torch.device(DEV)).cuda_stream


def planes_of(x: torch.Tensor) -> torch.Tensor:
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], 0).contiguous()


def unplanes(p: torch.Tensor) -> torch.Tensor:
    return p[0].float() + p[1].float()


SLAB = [False, True]


def run_gemm(a_list, w, m, hw_out, npass=3, bias=None, rowvec=None, resid=None, resid_rows=None, act=0,
             want_f32=True, want_planes=False, tile=0, n=None, slab=False, ksplit=1, same_w=0, dbg=0, tickets=None, wdirect=False,
             stat_part=None, stat_cpg=0):
    """a_list: list of (planes[2,rows,ld], c, ntaps, gather(int32 tensor|None), hw_src).
    slab=True: weights in slab order + the LDS-resident-slab kernel (w_layout 1)."""
    lib = N.lib()
    args = N.WdGemmArgs()
    keep = []
    for i, (pl, c, ntaps, gather, hw_src) in enumerate(a_list):
        s = N.WdSrc()
        s.hi, s.lo = pl[0].data_ptr(), pl[1].data_ptr()
        s.gather = gather.data_ptr() if gather is not None else None
        s.ld, s.c, s.ntaps, s.hw_src = pl.shape[2], c, ntaps, hw_src
        args.src[i] = s
    args.nsrc, args.npass = len(a_list), npass
    wp = planes_of(w)
    if slab:
        if not lib.wd_gemm_experimental():
            pytest.skip("slab kernel: library built without WDIFF_EXPERIMENTAL")
        g0 = a_list[0][3]
        span = slab_span(g0.cpu().numpy() if g0 is not None else None, hw_out, a_list[0][4], m)
        if span > 192 or tile == 64064:
            pytest.skip("slab kernel not applicable (span / tile)")
        wp = slab_order(wp, a_list[0][2], a_list[0][1], a_list[1][1] if len(a_list) > 1 else 0)
        args.w_layout, args.slab_rows = 1, span
    if same_w and not wdirect:  # row-shared taps kernel for 3x3 / pad 1 / stride 1 (w_layout 2)
        args.w_layout, args.slab_rows = 2, same_w
    if wdirect:  # fragment-major weights, loaded straight into the MFMA operand registers (w_layout 3; tile 64320 or 128160)
        wf = torch.empty_like(wp)
        N.check(lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), wp.shape[1], wp.shape[2], wf[0].data_ptr(), wf[1].data_ptr(),
                                   _st()), "wd_gemm_pack_w")
        wp = wf
        args.w_layout, args.slab_rows = 3, same_w
    if stat_part is not None:
        args.stat_part, args.stat_cpg = stat_part.data_ptr(), stat_cpg
    keep.append(wp)
    args.w_hi, args.w_lo = wp[0].data_ptr(), wp[1].data_ptr()
    n = w.shape[0] if n is None else n
    args.m, args.n, args.ktot, args.hw_out = m, n, w.shape[1], hw_out
    args.bias = bias.data_ptr() if bias is not None else None
    if rowvec is not None:
        args.rowvec, args.rowvec_ld = rowvec.data_ptr(), rowvec.shape[1]
    if resid is not None:
        args.resid, args.resid_ld = resid.data_ptr(), resid.shape[1]
    if resid_rows is not None:
        args.resid_rows = resid_rows.data_ptr()
    args.act = act
    n_out = n // 2 if act == N.ACT_GEGLU else n
    out = torch.full((m, n_out), float("nan"), device=DEV) if want_f32 else None
    opl = torch.zeros((2, m, n_out), dtype=torch.bfloat16, device=DEV) if want_planes else None
    if out is not None:
        args.out_f32, args.out_ld = out.data_ptr(), n_out
    if opl is not None:
        args.out_hi, args.out_lo, args.out_pl_ld = opl[0].data_ptr(), opl[1].data_ptr(), n_out
    args.tile = tile
    args.ksplit = ksplit
    args.dbg = dbg
    if ksplit != 1:
        ws = torch.empty(max(ksplit, 8) * m * n, device=DEV)
        keep.append(ws)
        args.ws, args.ws_floats = ws.data_ptr(), ws.numel()
    if tickets is not None:  # in-launch split-K combine: zeroed arrival counters, left zeroed
        args.tickets, args.ntickets = tickets.data_ptr(), tickets.numel()
    N.check(lib.wd_gemm(C.byref(args), _st()), "wd_gemm")
    torch.cuda.synchronize()
    return out, opl


@pytest.mark.parametrize("m,n,k,tile", [(256, 320, 320, 0), (70, 4, 64, 0), (128, 64, 32, 128064), (300, 160, 96, 128160),
                                        (257, 320, 640, 128160), (130, 100, 64, 64064), (64, 2560, 1280, 0),
                                        (4096, 320, 2880, 0)])
@pytest.mark.parametrize("slab", SLAB)
def test_gemm_linear(m, n, k, tile, slab):
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    out, opl = run_gemm([(planes_of(a.to(DEV)), k, 1, None, 0)], w.to(DEV), m, 1, bias=b.to(DEV), tile=tile,
                        want_planes=True, slab=slab)
    assert rel_err(out.cpu(), ref) < 2e-5
    assert rel_err(unplanes(opl).cpu(), ref) < 2e-5
    out1, _ = run_gemm([(planes_of(a.to(DEV)), k, 1, None, 0)], w.to(DEV), m, 1, npass=1, bias=b.to(DEV), tile=tile,
                       slab=slab)
    assert rel_err(out1.cpu(), ref) < 1e-2
    # exactness of the split: 3-pass result equals fp64 product of the bf16x2-rounded operands to fp32 accuracy
    a2, w2 = unplanes(planes_of(a)).double(), unplanes(planes_of(w)).double()
    assert rel_err(out.cpu(), a2 @ w2.t() + b.double()) < 5e-6


@pytest.mark.parametrize("mode", ["same", "down", "up"])
@pytest.mark.parametrize("B,C,h,w,Co", [(3, 64, 8, 32, 320), (2, 32, 4, 16, 64), (1, 96, 5, 7, 33), (5, 320, 8, 32, 320)])
@pytest.mark.parametrize("slab", SLAB)
def test_gemm_conv_gather(mode, B, C, h, w, Co, slab):
    g = torch.Generator().manual_seed(B * 1000 + C + h)
    x = torch.randn(B, C, h, w, generator=g)
    wt = torch.randn(Co, C, 3, 3, generator=g) / (9 * C) ** 0.5
    b = torch.randn(Co, generator=g)
    if mode == "same":
        ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    elif mode == "down":
        ref = F.conv2d(x.double(), wt.double(), b.double(), stride=2, padding=1)
    else:
        ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), wt.double(), b.double(), padding=1)
    tab, ho, wo = conv_gather_table(h, w, mode)
    tok = x.permute(0, 2, 3, 1).reshape(B * h * w, C).contiguous()
    wp = wt.permute(0, 2, 3, 1).reshape(Co, 9 * C).contiguous()
    out, _ = run_gemm([(planes_of(tok.to(DEV)), C, 9, torch.from_numpy(tab).to(DEV), h * w)], wp.to(DEV),
                      B * ho * wo, ho * wo, bias=b.to(DEV), slab=slab)
    got = out.cpu().reshape(B, ho, wo, Co).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("slab", SLAB)
def test_gemm_two_sources_film_residual_and_label_rows(slab):
    g = torch.Generator().manual_seed(5)
    B, hw, c1, c2, n = 3, 64, 64, 128, 160
    m = B * hw
    a1, a2 = torch.randn(m, c1, generator=g), torch.randn(m, c2, generator=g)
    w = torch.randn(n, c1 + c2, generator=g) / 14
    bias, film, res = torch.randn(n, generator=g), torch.randn(B, n + 7, generator=g), torch.randn(m, n, generator=g)
    ref = torch.cat([a1, a2], 1).double() @ w.double().t() + bias.double() + \
        film[:, :n].double().repeat_interleave(hw, 0) + res.double()
    out, _ = run_gemm([(planes_of(a1.to(DEV)), c1, 1, None, 0), (planes_of(a2.to(DEV)), c2, 1, None, 0)], w.to(DEV), m,
                      hw, bias=bias.to(DEV), rowvec=film.to(DEV), resid=res.to(DEV), slab=slab)
    assert rel_err(out.cpu(), ref) < 2e-5
    # 3x3 taps on the first source + identity second source (ResBlock conv2 | 1x1 skip, unet.py:621,632,671)
    hh, ww = 8, 8
    tab, _, _ = conv_gather_table(hh, ww, "same")
    wcat = torch.randn(n, 9 * c1 + c2, generator=g) / 30
    x1 = a1.reshape(B, hh, ww, c1).permute(0, 3, 1, 2)
    ref3 = (F.conv2d(x1.double(), wcat[:, :9 * c1].reshape(n, 3, 3, c1).permute(0, 3, 1, 2).double(), padding=1)
            .permute(0, 2, 3, 1).reshape(m, n) + a2.double() @ wcat[:, 9 * c1:].double().t() + bias.double())
    out3, _ = run_gemm([(planes_of(a1.to(DEV)), c1, 9, torch.from_numpy(tab).to(DEV), hw),
                        (planes_of(a2.to(DEV)), c2, 1, None, 0)], wcat.to(DEV), m, hw, bias=bias.to(DEV), slab=slab)
    assert rel_err(out3.cpu(), ref3) < 2e-5
    # label_emb gather + SiLU epilogue (time_embed.2 + label_emb[y] -> SiLU), unet.py:1551,1581,610
    table = torch.randn(11, n, generator=g)
    y = torch.tensor([3, 0, 10], dtype=torch.int64)
    a = torch.randn(3, c1, generator=g)
    w2 = torch.randn(n, c1, generator=g) / 8
    ref2 = F.silu(a.double() @ w2.double().t() + bias.double() + table[y].double())
    out2, pl2 = run_gemm([(planes_of(a.to(DEV)), c1, 1, None, 0)], w2.to(DEV), 3, 1, bias=bias.to(DEV),
                         resid=table.to(DEV), resid_rows=y.to(DEV), act=N.ACT_SILU, want_planes=True, slab=slab)
    assert rel_err(out2.cpu(), ref2) < 2e-5 and rel_err(unplanes(pl2).cpu(), ref2) < 2e-5


@pytest.mark.parametrize("slab", SLAB)
@pytest.mark.parametrize("m,dim,inner,tile", [(200, 64, 256, 128064), (300, 320, 1280, 128160), (130, 64, 128, 64064)])
def test_gemm_geglu_epilogue(slab, m, dim, inner, tile):
    g = torch.Generator().manual_seed(6 + inner)
    a = torch.randn(m, dim, generator=g)
    w = torch.randn(2 * inner, dim, generator=g) / dim ** 0.5
    b = torch.randn(2 * inner, generator=g)
    h = a.double() @ w.double().t() + b.double()
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    gr = (tile % 1000) // 2
    out, pl = run_gemm([(planes_of(a.to(DEV)), dim, 1, None, 0)], geglu_interleave(w, gr).to(DEV), m, 1,
                       bias=geglu_interleave(b, gr).to(DEV), act=N.ACT_GEGLU, want_planes=True, tile=tile, slab=slab)
    assert rel_err(out.cpu(), ref) < 2e-5 and rel_err(unplanes(pl).cpu(), ref) < 2e-5


@pytest.mark.parametrize("ksplit", [0, 2, 5])
def test_gemm_split_k(ksplit):
    """K cut across workgroups + fixed-order reduce == the unsplit result (same epilogue), 3x3 + skip source included."""
    g = torch.Generator().manual_seed(40 + ksplit)
    B, hh, ww, c1, c2, n = 4, 4, 16, 320, 640, 320
    hw, m = hh * ww, B * hh * ww
    a1, a2 = torch.randn(m, c1, generator=g), torch.randn(m, c2, generator=g)
    tab, _, _ = conv_gather_table(hh, ww, "same")
    wcat = torch.randn(n, 9 * c1 + c2, generator=g) / 60
    bias, film, res = torch.randn(n, generator=g), torch.randn(B, n, generator=g), torch.randn(m, n, generator=g)
    x1 = a1.reshape(B, hh, ww, c1).permute(0, 3, 1, 2)
    ref = (F.conv2d(x1.double(), wcat[:, :9 * c1].reshape(n, 3, 3, c1).permute(0, 3, 1, 2).double(), padding=1)
           .permute(0, 2, 3, 1).reshape(m, n) + a2.double() @ wcat[:, 9 * c1:].double().t() + bias.double() +
           film.double().repeat_interleave(hw, 0) + res.double())
    srcs = [(planes_of(a1.to(DEV)), c1, 9, torch.from_numpy(tab).to(DEV), hw), (planes_of(a2.to(DEV)), c2, 1, None, 0)]
    out, pl = run_gemm(srcs, wcat.to(DEV), m, hw, bias=bias.to(DEV), rowvec=film.to(DEV), resid=res.to(DEV),
                       want_planes=True, ksplit=ksplit)
    assert rel_err(out.cpu(), ref) < 2e-5 and rel_err(unplanes(pl).cpu(), ref) < 2e-5
    out2, _ = run_gemm(srcs, wcat.to(DEV), m, hw, bias=bias.to(DEV), rowvec=film.to(DEV), resid=res.to(DEV),
                       ksplit=ksplit)
    assert torch.equal(out, out2)  # deterministic
    # the in-launch combine (arrival tickets; the last workgroup of a tile sums the slabs in slice order inside the epilogue)
    # gives the same bits as the separate combine launch, whichever slice arrives last, and leaves its counters zeroed
    tk = torch.zeros(4096, dtype=torch.int32, device=DEV)
    for _ in range(3):
        out3, pl3 = run_gemm(srcs, wcat.to(DEV), m, hw, bias=bias.to(DEV), rowvec=film.to(DEV), resid=res.to(DEV),
                             want_planes=True, ksplit=ksplit, tickets=tk, dbg=0x2000)
        assert torch.equal(out3, out) and torch.equal(pl3, pl)
        assert int(tk.abs().sum()) == 0


@pytest.mark.parametrize("B,hh,ww,c1,c2,n,ksplit", [(4, 4, 16, 320, 640, 320, 1), (3, 8, 32, 64, 0, 160, 1), (5, 5, 7, 128, 64, 200, 1),
                                                      (4, 4, 16, 320, 640, 320, 3), (64, 8, 32, 64, 0, 320, 1)])
def test_gemm_two_workgroups_per_cu_kernel(B, hh, ww, c1, c2, n, ksplit):
    """wd_gemm4_kernel (32-deep stages, four waves, 64-row half epilogues; dbg 0x400 forces it): 3x3 gather source + optional
    identity skip source, FiLM row vector, residual, planes out, ragged M / N, split-K - vs fp64, and vs the v2 kernel."""
    g = torch.Generator().manual_seed(B + hh + ww + c1 + n)
    hw, m = hh * ww, B * hh * ww
    a1 = torch.randn(m, c1, generator=g)
    tab, _, _ = conv_gather_table(hh, ww, "same")
    wcat = torch.randn(n, 9 * c1 + c2, generator=g) / (9 * c1 + c2) ** 0.5
    bias, film, res = torch.randn(n, generator=g), torch.randn(B, n, generator=g), torch.randn(m, n, generator=g)
    x1 = a1.reshape(B, hh, ww, c1).permute(0, 3, 1, 2)
    ref = (F.conv2d(x1.double(), wcat[:, :9 * c1].reshape(n, 3, 3, c1).permute(0, 3, 1, 2).double(), padding=1)
           .permute(0, 2, 3, 1).reshape(m, n) + bias.double() + film.double().repeat_interleave(hw, 0) + res.double())
    srcs = [(planes_of(a1.to(DEV)), c1, 9, torch.from_numpy(tab).to(DEV), hw)]
    if c2:
        a2 = torch.randn(m, c2, generator=g)
        ref = ref + a2.double() @ wcat[:, 9 * c1:].double().t()
        srcs.append((planes_of(a2.to(DEV)), c2, 1, None, 0))
    kw = dict(bias=bias.to(DEV), rowvec=film.to(DEV), resid=res.to(DEV), want_planes=True, ksplit=ksplit, tile=128160)
    out, pl = run_gemm(srcs, wcat.to(DEV), m, hw, dbg=0x400, **kw)
    assert rel_err(out.cpu(), ref) < 2e-5 and rel_err(unplanes(pl).cpu(), ref) < 2e-5
    out0, _ = run_gemm(srcs, wcat.to(DEV), m, hw, **kw)
    assert max_rel(out.cpu(), out0.cpu()) < 5e-6


@pytest.mark.parametrize("B,hh,ww,c1,c2,n,ksplit,npass,tile", [
    (4, 4, 16, 320, 640, 320, 1, 3, 64320), (4, 4, 16, 320, 640, 320, 3, 3, 64320), (3, 8, 32, 64, 0, 320, 1, 3, 64320),
    (5, 5, 7, 128, 64, 320, 1, 3, 64320), (5, 5, 7, 128, 64, 320, 2, 1, 64320), (64, 8, 32, 64, 0, 320, 1, 3, 64320),
    (2, 8, 32, 64, 0, 640, 1, 3, 64320), (4, 4, 16, 320, 640, 320, 1, 3, 128160), (5, 5, 7, 128, 64, 160, 2, 3, 128160),
    (3, 8, 32, 64, 64, 480, 1, 1, 128160), (2, 8, 32, 320, 0, 320, 0, 3, 64320)])
def test_gemm_weights_to_registers_kernel(B, hh, ww, c1, c2, n, ksplit, npass, tile):
    """wd_gemmw_kernel (w_layout 3: fragment-major weights from wd_gemm_pack_w, loaded straight into the MFMA operand registers;
    64 x 320 and 128 x 160 tiles): 3x3 gather source (gather table and the computed table) + optional identity skip source, FiLM
    row vector, residual, planes out, ragged M, odd and even stage counts, split-K, single-pass bf16 - vs fp64 and vs the
    LDS-staged kernel (same K-half split and order: the same bits without a K cut); repeated launches give the same bits."""
    g = torch.Generator().manual_seed(B + hh + ww + c1 + n + tile)
    hw, m = hh * ww, B * hh * ww
    a1 = torch.randn(m, c1, generator=g)
    tab, _, _ = conv_gather_table(hh, ww, "same")
    wcat = torch.randn(n, 9 * c1 + c2, generator=g) / (9 * c1 + c2) ** 0.5
    bias, film, res = torch.randn(n, generator=g), torch.randn(B, n, generator=g), torch.randn(m, n, generator=g)
    x1 = a1.reshape(B, hh, ww, c1).permute(0, 3, 1, 2)
    ref = (F.conv2d(x1.double(), wcat[:, :9 * c1].reshape(n, 3, 3, c1).permute(0, 3, 1, 2).double(), padding=1)
           .permute(0, 2, 3, 1).reshape(m, n) + bias.double() + film.double().repeat_interleave(hw, 0) + res.double())
    srcs = [(planes_of(a1.to(DEV)), c1, 9, torch.from_numpy(tab).to(DEV), hw)]
    if c2:
        a2 = torch.randn(m, c2, generator=g)
        ref = ref + a2.double() @ wcat[:, 9 * c1:].double().t()
        srcs.append((planes_of(a2.to(DEV)), c2, 1, None, 0))
    kw = dict(bias=bias.to(DEV), rowvec=film.to(DEV), resid=res.to(DEV), want_planes=True, ksplit=ksplit, npass=npass)
    tol = 2e-5 if npass == 3 else 2e-2
    for same_w in (0, ww):
        out, pl = run_gemm(srcs, wcat.to(DEV), m, hw, wdirect=True, tile=tile, same_w=same_w, **kw)
        assert rel_err(out.cpu(), ref) < tol and rel_err(unplanes(pl).cpu(), ref) < (tol if npass == 3 else 3e-2)
        out2, _ = run_gemm(srcs, wcat.to(DEV), m, hw, wdirect=True, tile=tile, same_w=same_w, **kw)
        assert torch.equal(out, out2)
    out0, _ = run_gemm(srcs, wcat.to(DEV), m, hw, tile=128160 if n % 160 == 0 else 0, **kw)
    assert max_rel(out.cpu(), out0.cpu()) < (5e-6 if npass == 3 else 1e-2)


@pytest.mark.parametrize("B,hh,ww,cin,c2,n,taps,silu,eps,ksplit", [(3, 8, 32, 64, 0, 320, 9, 1, 1e-5, 1), (2, 8, 32, 320, 0, 320, 1, 0, 1e-6, 1),
                                                                  (2, 8, 32, 128, 64, 320, 9, 1, 1e-5, 1), (5, 4, 16, 320, 0, 320, 9, 1, 1e-5, 2),
                                                                  (64, 8, 32, 64, 0, 320, 9, 1, 1e-5, 1), (2, 8, 32, 320, 0, 320, 9, 1, 1e-5, 1),
                                                                  (3, 4, 16, 128, 0, 320, 9, 0, 1e-6, 1)])
def test_gemm_groupnorm_of_the_input_while_staging(B, hh, ww, cin, c2, n, taps, silu, eps, ksplit):
    """wd_gemm_args.a32*: src[0] is the fp32 map and the consumer's GroupNorm (+ SiLU) is applied while the weights-to-registers
    kernel stages its rows - vs F.group_norm -> SiLU -> conv in fp64 (zero padding applies to the NORMALISED map), and vs the
    two-launch form (wd_gn_apply planes -> wd_gemm); 3x3 and 1x1, an identity skip source from planes beside it, K cut."""
    lib = N.lib()
    g = torch.Generator().manual_seed(B + cin + taps + c2)
    hw, m = hh * ww, B * hh * ww
    x = torch.randn(m, cin, generator=g) * 1.5 + 0.3
    gam, bet = torch.randn(cin, generator=g) * 0.3 + 1, torch.randn(cin, generator=g) * 0.2
    w = torch.randn(n, taps * cin + c2, generator=g) / (taps * cin + c2) ** 0.5
    bias = torch.randn(n, generator=g)
    xn = F.group_norm(x.double().reshape(B, hw, cin).permute(0, 2, 1), 32, gam.double(), bet.double(), eps)
    if silu:
        xn = F.silu(xn)
    if taps == 9:
        ref = F.conv2d(xn.reshape(B, cin, hh, ww), w[:, :9 * cin].reshape(n, 3, 3, cin).permute(0, 3, 1, 2).double(), padding=1)
        ref = ref.permute(0, 2, 3, 1).reshape(m, n)
    else:
        ref = xn.permute(0, 2, 1).reshape(m, cin) @ w[:, :cin].double().t()
    ref = ref + bias.double()
    xd = x.to(DEV)
    cpg = cin // 32
    nchunk = lib.wd_gn_nchunk(hw)
    part = torch.zeros(B, nchunk, 32, 2, dtype=torch.float64, device=DEV)
    N.check(lib.wd_gn_stats(xd.data_ptr(), cin, B, hw, cin, cpg, part.data_ptr(), _st()), "stats")
    tab, _, _ = conv_gather_table(hh, ww, "same")
    tabd = torch.from_numpy(tab).to(DEV) if taps == 9 else None
    a = N.WdGemmArgs()
    s0 = N.WdSrc()
    s0.gather = tabd.data_ptr() if tabd is not None else None
    s0.ld, s0.c, s0.ntaps, s0.hw_src = cin, cin, taps, hw
    a.src[0] = s0
    a.nsrc = 1
    keep = []
    if c2:
        a2 = torch.randn(m, c2, generator=g)
        ref = ref + a2.double() @ w[:, taps * cin:].double().t()
        p2 = planes_of(a2.to(DEV))
        keep.append(p2)
        s1 = N.WdSrc()
        s1.hi, s1.lo, s1.ld, s1.c, s1.ntaps = p2[0].data_ptr(), p2[1].data_ptr(), c2, c2, 1
        a.src[1] = s1
        a.nsrc = 2
    a.npass = 3
    wp = planes_of(w.to(DEV))
    wf = torch.empty_like(wp)
    N.check(lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), n, w.shape[1], wf[0].data_ptr(), wf[1].data_ptr(), _st()), "pack")
    a.w_hi, a.w_lo, a.w_layout, a.tile = wf[0].data_ptr(), wf[1].data_ptr(), 3, 64320
    a.slab_rows = ww if taps == 9 else 0
    a.m, a.n, a.ktot, a.hw_out = m, n, w.shape[1], hw
    bd, gd, btd = bias.to(DEV), gam.to(DEV), bet.to(DEV)
    a.bias = bd.data_ptr()
    out = torch.full((m, n), float("nan"), device=DEV)
    a.out_f32, a.out_ld = out.data_ptr(), n
    a.a32, a.a32_ld, a.a32_part = xd.data_ptr(), cin, part.data_ptr()
    a.a32_nchunk, a.a32_pcpg, a.a32_cpg = nchunk, cpg, cpg
    a.a32_gamma, a.a32_beta, a.a32_eps, a.a32_silu = gd.data_ptr(), btd.data_ptr(), eps, silu
    ws = torch.empty(4 * m * n, device=DEV)
    a.ksplit = ksplit
    if ksplit != 1:
        a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
    N.check(lib.wd_gemm(C.byref(a), _st()), "wd_gemm + input GroupNorm")
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < 3e-5
    first = out.clone()
    N.check(lib.wd_gemm(C.byref(a), _st()), "wd_gemm + input GroupNorm")
    torch.cuda.synchronize()
    assert torch.equal(first, out)
    # the two-launch form: wd_gn_apply writes the normalised planes, the same kernel multiplies them
    pl = torch.zeros(2, m, cin, dtype=torch.bfloat16, device=DEV)
    N.check(lib.wd_gn_apply(xd.data_ptr(), cin, B, hw, cin, cpg, part.data_ptr(), nchunk, cpg, gd.data_ptr(), btd.data_ptr(), eps, silu,
                            pl[0].data_ptr(), pl[1].data_ptr(), cin, 0, None, None, _st()), "apply")
    a.a32 = None
    s0.hi, s0.lo = pl[0].data_ptr(), pl[1].data_ptr()
    a.src[0] = s0
    out2 = torch.full((m, n), float("nan"), device=DEV)
    a.out_f32 = out2.data_ptr()
    N.check(lib.wd_gemm(C.byref(a), _st()), "wd_gemm over planes")
    torch.cuda.synchronize()
    assert max_rel(out.cpu(), out2.cpu()) < 2e-5
    # refused where a tile would straddle samples, or without the fragment-major layout
    a.a32 = xd.data_ptr()
    a.hw_out = 100
    assert lib.wd_gemm(C.byref(a), _st()) != 0
    a.hw_out, a.w_layout = hw, 0
    assert lib.wd_gemm(C.byref(a), _st()) != 0


@pytest.mark.parametrize("m,k,resid", [(200, 320, True), (4096, 640, False), (64, 64, True)])
def test_gemm_layernorm_of_the_result_rows(m, k, resid):
    """wd_gemm_args.ln_*: the 64 x 320 weights-to-registers tiles hold whole rows, so the epilogue emits LayerNorm(result) * gamma +
    beta as the operand planes of the next GEMM (fp32 result still written) - vs F.layer_norm of the fp64 result; refused where
    rows are not whole (n != 320, K cut, LDS-staged layout)."""
    lib = N.lib()
    g = torch.Generator().manual_seed(m + k)
    n = 320
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    bias, res = torch.randn(n, generator=g), torch.randn(m, n, generator=g)
    gam, bet = torch.randn(n, generator=g) * 0.3 + 1, torch.randn(n, generator=g) * 0.2
    ref = x.double() @ w.double().t() + bias.double() + (res.double() if resid else 0)
    refn = F.layer_norm(ref, (n,), gam.double(), bet.double(), 1e-5)
    xp, wp = planes_of(x.to(DEV)), planes_of(w.to(DEV))
    wf = torch.empty_like(wp)
    N.check(lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), n, k, wf[0].data_ptr(), wf[1].data_ptr(), _st()), "pack")
    a = N.WdGemmArgs()
    s0 = N.WdSrc()
    s0.hi, s0.lo, s0.ld, s0.c, s0.ntaps = xp[0].data_ptr(), xp[1].data_ptr(), k, k, 1
    a.src[0] = s0
    a.nsrc, a.npass = 1, 3
    a.w_hi, a.w_lo, a.w_layout = wf[0].data_ptr(), wf[1].data_ptr(), 3
    a.m, a.n, a.ktot, a.hw_out = m, n, k, 1
    bd, rd, gd, btd = bias.to(DEV), res.to(DEV), gam.to(DEV), bet.to(DEV)
    a.bias = bd.data_ptr()
    if resid:
        a.resid, a.resid_ld = rd.data_ptr(), n
    out = torch.full((m, n), float("nan"), device=DEV)
    opl = torch.zeros(2, m, n, dtype=torch.bfloat16, device=DEV)
    a.out_f32, a.out_ld = out.data_ptr(), n
    a.out_hi, a.out_lo, a.out_pl_ld = opl[0].data_ptr(), opl[1].data_ptr(), n
    a.ln_gamma, a.ln_beta, a.ln_eps = gd.data_ptr(), btd.data_ptr(), 1e-5
    N.check(lib.wd_gemm(C.byref(a), _st()), "wd_gemm + LayerNorm")
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < 2e-5
    assert max_rel(unplanes(opl).cpu(), refn) < 5e-5
    a.w_layout = 0
    a.w_hi, a.w_lo = wp[0].data_ptr(), wp[1].data_ptr()
    assert lib.wd_gemm(C.byref(a), _st()) != 0  # the LDS-staged tiles (128 x 160) do not hold whole rows


@pytest.mark.parametrize("m,inner,npass,planes", [(64, 1280, 3, True), (200, 1280, 3, False), (4096, 1280, 3, True), (130, 256, 1, True), (70, 128, 3, True),
                                                   (64, 384, 3, False)])
def test_fused_geglu_feed_forward(m, inner, npass, planes):
    """wd_ff_fused: x + GEGLU(LN(x) W1^T + b1) W2^T + b2 (unet.py:122-149, 343-344) in one launch - hidden activations never
    stored - vs fp64, at ragged token counts, as fp32 and as split-bf16 planes; repeated launches give the same bits."""
    lib = N.lib()
    c = 320
    assert lib.wd_ff_supported(c, inner)
    g = torch.Generator().manual_seed(m + inner)
    x = torch.randn(m, c, generator=g)
    w1 = torch.randn(2 * inner, c, generator=g) / c ** 0.5
    b1 = torch.randn(2 * inner, generator=g)
    w2 = torch.randn(c, inner, generator=g) / inner ** 0.5
    b2, res = torch.randn(c, generator=g), torch.randn(m, c, generator=g)
    xd, w1d, w2d = x.double(), w1.double(), w2.double()
    hid = (xd @ w1d[:inner].t() + b1[:inner].double()) * F.gelu(xd @ w1d[inner:].t() + b1[inner:].double())
    ref = res.double() + hid @ w2d.t() + b2.double()

    def pack(w):
        wp = planes_of(w.to(DEV))
        wf = torch.empty_like(wp)
        N.check(lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), wp.shape[1], wp.shape[2], wf[0].data_ptr(), wf[1].data_ptr(),
                                   _st()), "wd_gemm_pack_w")
        return wf

    w1f, w2f = pack(geglu_interleave(w1, 16)), pack(w2)
    b1d, b2d, resd = geglu_interleave(b1, 16).to(DEV), b2.to(DEV), res.to(DEV)
    xp = planes_of(x.to(DEV))
    a = N.WdFfArgs()
    a.x_hi, a.x_lo, a.x_ld = xp[0].data_ptr(), xp[1].data_ptr(), c
    a.m, a.c, a.inner = m, c, inner
    a.w1_hi, a.w1_lo, a.b1 = w1f[0].data_ptr(), w1f[1].data_ptr(), b1d.data_ptr()
    a.w2_hi, a.w2_lo, a.b2 = w2f[0].data_ptr(), w2f[1].data_ptr(), b2d.data_ptr()
    a.resid, a.resid_ld = resd.data_ptr(), c
    out = torch.full((m, c), float("nan"), device=DEV)
    opl = torch.zeros(2, m, c, dtype=torch.bfloat16, device=DEV)
    a.out_f32, a.out_ld = out.data_ptr(), c
    if planes:
        a.out_hi, a.out_lo, a.out_pl_ld = opl[0].data_ptr(), opl[1].data_ptr(), c
    a.hw_out, a.npass = 1, npass
    N.check(lib.wd_ff_fused(C.byref(a), _st()), "wd_ff_fused")
    torch.cuda.synchronize()
    tol = 3e-5 if npass == 3 else 3e-2
    assert rel_err(out.cpu(), ref) < tol
    if planes:
        assert rel_err(unplanes(opl).cpu(), ref) < tol
    first = out.clone()
    N.check(lib.wd_ff_fused(C.byref(a), _st()), "wd_ff_fused")
    torch.cuda.synchronize()
    assert torch.equal(first, out)
    a.inner = 100
    assert lib.wd_ff_fused(C.byref(a), _st()) == N.WD_EINVAL
    a.inner = inner
    # ---- with the proj_out tail: out = resid3 + (resid + FF) W3^T + b3, + GroupNorm statistics of the result (64-row panels)
    w3 = torch.randn(c, c, generator=g) / c ** 0.5
    b3, res3 = torch.randn(c, generator=g), torch.randn(m, c, generator=g)
    ref3 = res3.double() + ref @ w3.double().t() + b3.double()
    w3f, b3d, res3d = pack(w3), b3.to(DEV), res3.to(DEV)
    a.w3_hi, a.w3_lo, a.b3 = w3f[0].data_ptr(), w3f[1].data_ptr(), b3d.data_ptr()
    a.resid3, a.resid3_ld = res3d.data_ptr(), c
    hw = 64 if m % 64 == 0 else 0
    part = None
    if hw:
        part = torch.full((m // hw, 1, 32, 2), float("nan"), dtype=torch.float64, device=DEV)
        a.stat_part, a.stat_cpg, a.hw_out = part.data_ptr(), c // 32, hw
    out.fill_(float("nan"))
    N.check(lib.wd_ff_fused(C.byref(a), _st()), "wd_ff_fused + proj_out")
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref3) < tol
    if planes:
        assert rel_err(unplanes(opl).cpu(), ref3) < tol
    if part is not None:
        o = out.cpu().double().reshape(m // hw, hw, 32, c // 32)
        assert max_rel(part.cpu()[:, 0, :, 0], o.sum(dim=(1, 3))) < 1e-5 and max_rel(part.cpu()[:, 0, :, 1], (o * o).sum(dim=(1, 3))) < 1e-5


@pytest.mark.parametrize("B,hh,ww,cin,n,tile,ksplit", [(3, 8, 32, 64, 320, 64320, 1), (4, 4, 16, 320, 320, 64320, 1), (4, 4, 16, 320, 320, 64320, 0),
                                                        (3, 8, 32, 64, 320, 128160, 1), (5, 8, 8, 64, 640, 64320, 1)])
def test_gemm_weights_to_registers_kernel_statistics(B, hh, ww, cin, n, tile, ksplit):
    """Fused GroupNorm statistics of the weights-to-registers kernel: kept per row panel of the tile (nchunk = hw / 64 for the 64-row
    tile), equal to sums over the finished output."""
    g = torch.Generator().manual_seed(n + hh + tile)
    hw, m = hh * ww, B * hh * ww
    a = torch.randn(m, cin, generator=g)
    w = torch.randn(n, 9 * cin, generator=g) / (9 * cin) ** 0.5
    bias, film = torch.randn(n, generator=g), torch.randn(B, n, generator=g)
    tab, _, _ = conv_gather_table(hh, ww, "same")
    cpg = n // 32
    nchunk = max(1, hw // (tile // 1000))
    part = torch.full((B, nchunk, 32, 2), float("nan"), dtype=torch.float64, device=DEV)
    srcs = [(planes_of(a.to(DEV)), cin, 9, torch.from_numpy(tab).to(DEV), hw)]
    out, _ = run_gemm(srcs, w.to(DEV), m, hw, wdirect=True, tile=tile, bias=bias.to(DEV), rowvec=film.to(DEV), ksplit=ksplit,
                      stat_part=part, stat_cpg=cpg)
    o = out.cpu().double().reshape(B, hw, 32, cpg)
    got = part.cpu().sum(dim=1)
    assert torch.isfinite(got).all()
    assert max_rel(got[..., 0], o.sum(dim=(1, 3))) < 1e-5 and max_rel(got[..., 1], (o * o).sum(dim=(1, 3))) < 1e-5


@pytest.mark.parametrize("B,hh,ww,c1,c2,n,ksplit,npass", [(4, 4, 16, 320, 640, 320, 1, 3), (3, 8, 32, 64, 0, 160, 1, 3), (5, 5, 7, 128, 64, 200, 1, 3),
                                                            (4, 4, 16, 320, 640, 320, 3, 3), (64, 8, 32, 64, 0, 320, 1, 3), (2, 8, 32, 32, 0, 320, 1, 3),
                                                            (3, 8, 32, 96, 32, 160, 2, 1), (64, 8, 32, 320, 0, 320, 1, 3)])
@pytest.mark.parametrize("sel", [0x80000])
def test_gemm_deep_pipeline_kernel(B, hh, ww, c1, c2, n, ksplit, npass, sel):
    """wd_gemm8_kernel (dbg 0x80000 forces it): 32-deep stages in a ring of four LDS buffers, counted vmcnt, four dedicated loader
    waves + eight compute waves with two fragment sets -
    3x3 gather source + optional identity skip source, FiLM row vector, residual, planes out, ragged M / N, split-K, short K loops
    (1 .. 3 stages: the prologue / drain paths), single-pass bf16 - vs fp64 and vs the v2 kernel; repeated launches give the same bits
    (a race between the LDS-DMA ring and the fragment reads would not)."""
    if not N.lib().wd_gemm_experimental():
        pytest.skip("library built without WDIFF_EXPERIMENTAL")
    g = torch.Generator().manual_seed(B + hh + ww + c1 + n)
    hw, m = hh * ww, B * hh * ww
    a1 = torch.randn(m, c1, generator=g)
    tab, _, _ = conv_gather_table(hh, ww, "same")
    wcat = torch.randn(n, 9 * c1 + c2, generator=g) / (9 * c1 + c2) ** 0.5
    bias, film, res = torch.randn(n, generator=g), torch.randn(B, n, generator=g), torch.randn(m, n, generator=g)
    x1 = a1.reshape(B, hh, ww, c1).permute(0, 3, 1, 2)
    ref = (F.conv2d(x1.double(), wcat[:, :9 * c1].reshape(n, 3, 3, c1).permute(0, 3, 1, 2).double(), padding=1)
           .permute(0, 2, 3, 1).reshape(m, n) + bias.double() + film.double().repeat_interleave(hw, 0) + res.double())
    srcs = [(planes_of(a1.to(DEV)), c1, 9, torch.from_numpy(tab).to(DEV), hw)]
    if c2:
        a2 = torch.randn(m, c2, generator=g)
        ref = ref + a2.double() @ wcat[:, 9 * c1:].double().t()
        srcs.append((planes_of(a2.to(DEV)), c2, 1, None, 0))
    tol = 2e-5 if npass == 3 else 2e-2
    kw = dict(bias=bias.to(DEV), rowvec=film.to(DEV), resid=res.to(DEV), want_planes=True, ksplit=ksplit, tile=128160, npass=npass)
    out, pl = run_gemm(srcs, wcat.to(DEV), m, hw, dbg=sel, **kw)
    assert rel_err(out.cpu(), ref) < tol and rel_err(unplanes(pl).cpu(), ref) < tol
    if c1 % 64 == 0 and c2 % 64 == 0:
        out0, _ = run_gemm(srcs, wcat.to(DEV), m, hw, **kw)
        assert max_rel(out.cpu(), out0.cpu()) < (5e-6 if npass == 3 else 2e-2)
    for _ in range(5):
        out2, pl2 = run_gemm(srcs, wcat.to(DEV), m, hw, dbg=sel, **kw)
        assert torch.equal(out, out2) and torch.equal(pl, pl2)


@pytest.mark.parametrize("k,sel", [(32, 0x80000), (64, 0x80000), (96, 0x80000), (128, 0x80000), (160, 0x80000), (320, 0x80000)])
def test_gemm_deep_pipeline_kernel_short_loops(k, sel):
    """1 .. 10 stages of a plain linear through wd_gemm8_kernel: every length of the prologue / steady state / drain."""
    if not N.lib().wd_gemm_experimental():
        pytest.skip("library built without WDIFF_EXPERIMENTAL")
    g = torch.Generator().manual_seed(k)
    m, n = 300, 320
    a, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    out, _ = run_gemm([(planes_of(a.to(DEV)), k, 1, None, 0)], w.to(DEV), m, 1, bias=b.to(DEV), tile=128160, dbg=sel)
    assert rel_err(out.cpu(), ref) < 2e-5


def test_gemm_two_workgroups_per_cu_kernel_geglu():
    g = torch.Generator().manual_seed(77)
    m, dim, inner = 300, 320, 640
    a = torch.randn(m, dim, generator=g)
    w = torch.randn(2 * inner, dim, generator=g) / dim ** 0.5
    b = torch.randn(2 * inner, generator=g)
    h = a.double() @ w.double().t() + b.double()
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    for npass, tol in ((3, 2e-5), (1, 2e-2)):
        out, pl = run_gemm([(planes_of(a.to(DEV)), dim, 1, None, 0)], geglu_interleave(w, 80).to(DEV), m, 1, npass=npass,
                           bias=geglu_interleave(b, 80).to(DEV), act=N.ACT_GEGLU, want_planes=True, tile=128160, dbg=0x400)
        assert rel_err(out.cpu(), ref) < tol and rel_err(unplanes(pl).cpu(), ref) < tol


@pytest.mark.parametrize("B,h,w,cin,cout", [(4, 4, 16, 64, 320), (2, 8, 8, 128, 320)])
def test_gemm_weight_groups_upsample_as_four_phases(B, h, w, cin, cout):
    """wd_gemm_args.w_ngroups: nearest x2 + conv3x3 (Upsample.forward, unet.py:488-499) as four 2x2 convolutions of the source map in
    one launch - a tile picks the weight image of its output phase - and wd_gn_apply2's perm_a reading the phase-major rows back in
    raster order; against F.interpolate + F.conv2d in fp64.  Bad combinations are rejected."""
    from worddiffusion_amd.engine import upsample_phase_tables, upsample_phase_weights
    lib = N.lib()
    g = torch.Generator().manual_seed(B * 100 + cin)
    x = torch.randn(B, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), wt.double(), bias.double(), padding=1)
    ref_tok = ref.permute(0, 2, 3, 1).reshape(B * 4 * h * w, cout)          # raster rows
    hw = h * w
    tab, perm = upsample_phase_tables(h, w)
    wph = upsample_phase_weights(wt)                                         # [phase][tap][n][c]
    tok = planes_of(x.permute(0, 2, 3, 1).reshape(B * hw, cin).to(DEV))
    imgs = torch.empty(2, 4, cout, 4 * cin, dtype=torch.bfloat16, device=DEV)
    for ph in range(4):
        wp = planes_of(wph[ph].permute(1, 0, 2).reshape(cout, 4 * cin).contiguous().to(DEV))
        N.check(lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), cout, 4 * cin, imgs[0, ph].data_ptr(), imgs[1, ph].data_ptr(),
                                   _st()), "wd_gemm_pack_w")
    tabd, permd, bd = torch.from_numpy(tab).to(DEV), torch.from_numpy(perm).to(DEV), bias.to(DEV)
    m = B * 4 * hw
    out = torch.full((m, cout), float("nan"), device=DEV)
    part = torch.zeros(B, 4 * hw // 64, 32, 2, dtype=torch.float64, device=DEV)
    a = N.WdGemmArgs()
    s0 = N.WdSrc()
    s0.hi, s0.lo, s0.gather = tok[0].data_ptr(), tok[1].data_ptr(), tabd.data_ptr()
    s0.ld, s0.c, s0.ntaps, s0.hw_src = cin, cin, 4, hw
    a.src[0] = s0
    a.nsrc, a.npass = 1, 3
    a.w_hi, a.w_lo = imgs[0].data_ptr(), imgs[1].data_ptr()
    a.w_layout, a.tile, a.ksplit = 3, 64320, 1
    a.w_ngroups, a.w_group_stride = 4, cout * 4 * cin
    a.m, a.n, a.ktot, a.hw_out = m, cout, 4 * cin, 4 * hw
    a.bias = bd.data_ptr()
    a.out_f32, a.out_ld = out.data_ptr(), cout
    a.stat_part, a.stat_cpg = part.data_ptr(), cout // 32
    N.check(lib.wd_gemm(C.byref(a), _st()), "wd_gemm (weight groups)")
    torch.cuda.synchronize()
    raster = out.reshape(B, 4 * hw, cout)[:, permd.long()].reshape(m, cout)
    assert rel_err(raster.cpu(), ref_tok) < 2e-5
    # the statistics of the phase-major map are those of the raster map (sums over a sample)
    sums = part.sum(1).cpu()                                                  # [B][32][2]
    grp = ref_tok.reshape(B, 4 * hw, 32, cout // 32)
    assert torch.allclose(sums[..., 0], grp.sum((1, 3)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(sums[..., 1], (grp * grp).sum((1, 3)), rtol=1e-4, atol=1e-3)
    # GroupNorm over [this map | a raster-ordered second source] reads the first through the permutation
    skip = torch.randn(m, cout, generator=g)
    gam, bet = torch.randn(2 * cout, generator=g), torch.randn(2 * cout, generator=g)
    cat = torch.cat([ref_tok, skip.double()], 1).reshape(B, 4 * hw, 2 * cout).permute(0, 2, 1)
    gref = F.silu(F.group_norm(cat, 32, gam.double(), bet.double(), 1e-5)).permute(0, 2, 1).reshape(m, 2 * cout)
    skd = skip.to(DEV)
    nchunk = lib.wd_gn_nchunk(4 * hw)
    part_b = torch.zeros(B, nchunk, 32, 2, dtype=torch.float64, device=DEV)
    N.check(lib.wd_gn_stats(skd.data_ptr(), cout, B, 4 * hw, cout, cout // 32, part_b.data_ptr(), _st()), "stats")
    pl, raw = torch.zeros(2, m, 2 * cout, dtype=torch.bfloat16, device=DEV), torch.zeros(2, m, 2 * cout, dtype=torch.bfloat16, device=DEV)
    gd, btd = gam.to(DEV), bet.to(DEV)
    N.check(lib.wd_gn_apply2(out.data_ptr(), cout, cout, part.data_ptr(), 4 * hw // 64, cout // 32, 0,
                             skd.data_ptr(), cout, cout, part_b.data_ptr(), nchunk, cout // 32, cout,
                             B, 4 * hw, 2 * cout // 32, gd.data_ptr(), btd.data_ptr(), 1e-5, 1, pl[0].data_ptr(), pl[1].data_ptr(), 2 * cout,
                             raw[0].data_ptr(), raw[1].data_ptr(), permd.data_ptr(), _st()), "wd_gn_apply2 (perm)")
    torch.cuda.synchronize()
    assert max_rel(unplanes(pl).cpu(), gref) < 5e-5
    assert max_rel(unplanes(raw).cpu()[:, :cout], ref_tok) < 3e-5
    # rejected: another tile, a K cut, runs that are not whole 64-row tiles
    for field, val in (("tile", 128160), ("ksplit", 2), ("w_ngroups", 3), ("w_group_stride", 0)):
        b = N.WdGemmArgs.from_buffer_copy(a)
        setattr(b, field, val)
        assert lib.wd_gemm(C.byref(b), _st()) == N.WD_EINVAL, field


def test_gemm_rejects_bad_arguments():
    lib = N.lib()
    a = N.WdGemmArgs()
    assert lib.wd_gemm(C.byref(a), _st()) == N.WD_EINVAL
    assert lib.wd_gemm(None, _st()) == N.WD_EINVAL


@pytest.mark.parametrize("B,hw,cs,silu,eps", [(3, 256, (320,), 1, 1e-5), (2, 64, (320, 320), 1, 1e-5),
                                              (2, 32, (64,), 0, 1e-6), (1, 100, (64, 32), 1, 1e-5)])
def test_groupnorm_silu_planes(B, hw, cs, silu, eps):
    lib = N.lib()
    g = torch.Generator().manual_seed(hw + sum(cs))
    xs = [torch.randn(B * hw, c, generator=g) * 2 + 0.5 for c in cs]
    ctot = sum(cs)
    gamma, beta = torch.randn(ctot, generator=g), torch.randn(ctot, generator=g)
    cat = torch.cat(xs, 1).reshape(B, hw, ctot).permute(0, 2, 1)
    ref = F.group_norm(cat.double(), 32, gamma.double(), beta.double(), eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(B * hw, ctot)
    cpg = ctot // 32
    if any(c % cpg for c in cs):
        pytest.skip("straddling groups are materialised by the engine (wd_copy2d)")
    nchunk = lib.wd_gn_nchunk(hw)
    pl = torch.zeros(2, B * hw, ctot, dtype=torch.bfloat16, device=DEV)
    raw = torch.zeros_like(pl)
    xd = [x.to(DEV) for x in xs]
    gd, bd = gamma.to(DEV), beta.to(DEV)
    parts = []
    for x, c in zip(xd, cs):  # every tensor keeps its own 32-group partials (c/32 channels each)
        part = torch.zeros(B, nchunk, 32, 2, dtype=torch.float64, device=DEV)
        N.check(lib.wd_gn_stats(x.data_ptr(), c, B, hw, c, c // 32, part.data_ptr(), _st()), "stats")
        parts.append(part)
    off = 0
    for x, c, part in zip(xd, cs, parts):
        N.check(lib.wd_gn_apply(x.data_ptr(), c, B, hw, c, cpg, part.data_ptr(), nchunk, c // 32, gd.data_ptr(),
                                bd.data_ptr(), eps, silu, pl[0].data_ptr(), pl[1].data_ptr(), ctot, off,
                                raw[0].data_ptr(), raw[1].data_ptr(), _st()), "apply")
        off += c
    torch.cuda.synchronize()
    assert max_rel(unplanes(pl).cpu(), ref) < 3e-5
    assert max_rel(unplanes(raw).cpu(), torch.cat(xs, 1)) < 1e-5


@pytest.mark.parametrize("B,hh,ww,cin,n,ksplit", [(3, 8, 32, 64, 320, 1), (4, 4, 16, 320, 320, 1), (4, 4, 16, 320, 320, 0),
                                                   (5, 8, 8, 64, 64, 1), (2, 8, 16, 64, 640, 1)])
def test_gemm_fused_groupnorm_statistics(B, hh, ww, cin, n, ksplit):
    """wd_gemm's epilogue statistics (stat_part) == sums over the finished output, and feed wd_gn_apply directly
    (also as the 2x-coarser groups of a concat norm)."""
    lib = N.lib()
    g = torch.Generator().manual_seed(n + hh)
    hw, m = hh * ww, B * hh * ww
    a = torch.randn(m, cin, generator=g)
    w = torch.randn(n, 9 * cin, generator=g) / (9 * cin) ** 0.5
    bias, film = torch.randn(n, generator=g), torch.randn(B, n, generator=g)
    tab, _, _ = conv_gather_table(hh, ww, "same")
    args_keep = {}
    cpg = n // 32
    nchunk = max(1, hw // 128)
    part = torch.full((B, nchunk, 32, 2), float("nan"), dtype=torch.float64, device=DEV)

    lib_args = N.WdGemmArgs()
    pl = planes_of(a.to(DEV))
    tabd = torch.from_numpy(tab).to(DEV)
    s0 = N.WdSrc()
    s0.hi, s0.lo, s0.gather = pl[0].data_ptr(), pl[1].data_ptr(), tabd.data_ptr()
    s0.ld, s0.c, s0.ntaps, s0.hw_src = cin, cin, 9, hw
    lib_args.src[0] = s0
    lib_args.nsrc, lib_args.npass = 1, 3
    wp = planes_of(w.to(DEV))
    lib_args.w_hi, lib_args.w_lo = wp[0].data_ptr(), wp[1].data_ptr()
    lib_args.m, lib_args.n, lib_args.ktot, lib_args.hw_out = m, n, 9 * cin, hw
    bd, fd = bias.to(DEV), film.to(DEV)
    lib_args.bias, lib_args.rowvec, lib_args.rowvec_ld = bd.data_ptr(), fd.data_ptr(), n
    out = torch.zeros(m, n, device=DEV)
    lib_args.out_f32, lib_args.out_ld = out.data_ptr(), n
    lib_args.stat_part, lib_args.stat_cpg = part.data_ptr(), cpg
    ws = torch.empty(8 * m * n, device=DEV)
    lib_args.ksplit, lib_args.ws, lib_args.ws_floats = ksplit, ws.data_ptr(), ws.numel()
    N.check(lib.wd_gemm(C.byref(lib_args), _st()), "wd_gemm+stats")
    torch.cuda.synchronize()
    if ksplit != 1:  # the same launch with the in-launch split-K combine: identical output and statistics
        out_sep, part_sep = out.clone(), part.clone()
        tk = torch.zeros(1024, dtype=torch.int32, device=DEV)
        lib_args.tickets, lib_args.ntickets, lib_args.dbg = tk.data_ptr(), tk.numel(), 0x2000
        out.zero_()
        part.fill_(float("nan"))
        N.check(lib.wd_gemm(C.byref(lib_args), _st()), "wd_gemm+stats+tickets")
        torch.cuda.synchronize()
        # (same output bits; the statistics are fixed-order sums over a different tile shape - 128 x 160 here, 64 x 40 in the
        # combine launch - so they agree to fp32 rounding of the per-thread column sums, not bit for bit)
        assert torch.equal(out, out_sep) and max_rel(part, part_sep) < 1e-5 and int(tk.abs().sum()) == 0
    o = out.cpu().double().reshape(B, hw, 32, cpg)
    ref_sum = o.sum(dim=(1, 3))
    ref_sq = (o * o).sum(dim=(1, 3))
    got = part.cpu().sum(dim=1)
    assert torch.isfinite(got).all()
    assert max_rel(got[..., 0], ref_sum) < 1e-5 and max_rel(got[..., 1], ref_sq) < 1e-5
    # consumer: GroupNorm + SiLU of this tensor as the first half of a 2n-channel concat norm (groups twice as wide)
    other = torch.randn(m, n, generator=g)
    gamma, beta = torch.randn(2 * n, generator=g), torch.randn(2 * n, generator=g)
    cat = torch.cat([out.cpu(), other], 1).reshape(B, hw, 2 * n).permute(0, 2, 1)
    ref = F.silu(F.group_norm(cat.double(), 32, gamma.double(), beta.double(), 1e-5)).permute(0, 2, 1).reshape(m, 2 * n)
    opl = torch.zeros(2, m, 2 * n, dtype=torch.bfloat16, device=DEV)
    gd, bd2, od = gamma.to(DEV), beta.to(DEV), other.to(DEV)
    N.check(lib.wd_gn_apply(out.data_ptr(), n, B, hw, n, 2 * cpg, part.data_ptr(), nchunk, cpg, gd.data_ptr(),
                            bd2.data_ptr(), 1e-5, 1, opl[0].data_ptr(), opl[1].data_ptr(), 2 * n, 0, None, None, _st()),
            "apply")
    nck = lib.wd_gn_nchunk(hw)
    part2 = torch.zeros(B, nck, 32, 2, dtype=torch.float64, device=DEV)
    N.check(lib.wd_gn_stats(od.data_ptr(), n, B, hw, n, cpg, part2.data_ptr(), _st()), "stats")
    N.check(lib.wd_gn_apply(od.data_ptr(), n, B, hw, n, 2 * cpg, part2.data_ptr(), nck, cpg, gd.data_ptr(),
                            bd2.data_ptr(), 1e-5, 1, opl[0].data_ptr(), opl[1].data_ptr(), 2 * n, n, None, None, _st()),
            "apply2")
    torch.cuda.synchronize()
    assert max_rel(unplanes(opl).cpu(), ref) < 3e-5


def test_layernorm_and_split():
    lib = N.lib()
    g = torch.Generator().manual_seed(9)
    for rows, c in ((1000, 320), (37, 64), (5, 1280)):
        x = torch.randn(rows, c, generator=g) * 3 + 1
        ga, be = torch.randn(c, generator=g), torch.randn(c, generator=g)
        ref = F.layer_norm(x.double(), (c,), ga.double(), be.double(), 1e-5)
        pl = torch.zeros(2, rows, c, dtype=torch.bfloat16, device=DEV)
        xd, gd, bd = x.to(DEV), ga.to(DEV), be.to(DEV)
        N.check(lib.wd_layernorm(xd.data_ptr(), c, rows, c, gd.data_ptr(), bd.data_ptr(), 1e-5, pl[0].data_ptr(),
                                 pl[1].data_ptr(), c, _st()), "ln")
        torch.cuda.synchronize()
        assert max_rel(unplanes(pl).cpu(), ref) < 3e-5
        N.check(lib.wd_split(xd.data_ptr(), c, rows, c, 1, pl[0].data_ptr(), pl[1].data_ptr(), c, _st()), "split")
        torch.cuda.synchronize()
        assert max_rel(unplanes(pl).cpu(), F.silu(x.double())) < 2e-5


@pytest.mark.parametrize("B,H,nq,nk,d,scale", [(3, 4, 256, 10, 80, 80 ** -0.5), (2, 4, 64, 64, 16, 0.25),
                                               (2, 1, 10, 10, 320, 1.0), (2, 4, 70, 779, 80, 80 ** -0.5),
                                               (1, 1, 769, 769, 64, 1.0), (2, 4, 256, 256, 80, 80 ** -0.5),
                                               (2, 2, 100, 33, 32, 0.3), (1, 3, 40, 200, 96, 0.1), (1, 2, 129, 65, 48, 0.2)])
def test_attention(B, H, nq, nk, d, scale):
    lib = N.lib()
    g = torch.Generator().manual_seed(nq + nk + d)
    inner = H * d
    q = torch.randn(B * nq, inner, generator=g) * 0.5
    kv = torch.randn(B * nk, 2 * inner + 8, generator=g) * 0.5  # k at col 0, v at col inner (a wider pitch on purpose)
    k, v = kv[:, :inner], kv[:, inner:2 * inner]

    def heads(t, n):
        return t.reshape(B, n, H, d).permute(0, 2, 1, 3).double()

    att = torch.softmax(heads(q, nq) @ heads(k, nk).transpose(-1, -2) * scale, -1)
    ref = (att @ heads(v, nk)).permute(0, 2, 1, 3).reshape(B * nq, inner)
    qd, kvd = q.to(DEV), kv.to(DEV)
    out = torch.zeros(B * (nq + 3), inner, device=DEV)
    pl = torch.zeros(2, B * (nq + 3), inner, dtype=torch.bfloat16, device=DEV)
    N.check(lib.wd_attention(qd.data_ptr(), inner, kvd.data_ptr(), kv.shape[1], kvd.data_ptr() + 4 * inner, kv.shape[1],
                             B, H, nq, nk, d, scale, out.data_ptr(), pl[0].data_ptr(), pl[1].data_ptr(), inner, nq + 3,
                             2, _st()), "attn")
    torch.cuda.synchronize()
    got = out.cpu().reshape(B, nq + 3, inner)[:, 2:2 + nq].reshape(B * nq, inner)
    assert max_rel(got, ref) < 2e-5
    gotp = unplanes(pl).cpu().reshape(B, nq + 3, inner)[:, 2:2 + nq].reshape(B * nq, inner)
    assert max_rel(gotp, ref) < 3e-5


def test_timestep_embedding_tokens_im2col_layout():
    lib = N.lib()
    t = torch.tensor([0, 1, 2, 17, 500, 998, 999], dtype=torch.int64)
    for dim in (320, 64):
        half = dim // 2
        freqs = torch.exp(-np.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
        pl = torch.zeros(2, 7, dim, dtype=torch.bfloat16, device=DEV)
        td, fd = t.to(DEV), freqs.to(DEV)
        N.check(lib.wd_timestep_embedding(td.data_ptr(), 7, fd.data_ptr(), half, pl[0].data_ptr(), pl[1].data_ptr(),
                                          dim, _st()), "temb")
        torch.cuda.synchronize()
        # |arg| up to 999 rad: device sincos vs host agree to a few 1e-7 absolute; planes add 2^-17 relative
        assert float((unplanes(pl).cpu() - U.timestep_embedding(t, dim)).abs().max()) < 2e-5
    # embedding + PE
    g = torch.Generator().manual_seed(3)
    table = torch.randn(53, 64, generator=g)
    pe = U.positional_encoding(10, 64)
    ids = torch.randint(0, 53, (4, 10), generator=g)
    ref = table[ids] + pe[:10]
    pl = torch.zeros(2, 40, 64, dtype=torch.bfloat16, device=DEV)
    idd, tabd, ped = ids.to(DEV), table.to(DEV), pe.to(DEV)
    N.check(lib.wd_embed_tokens(idd.data_ptr(), 1, 40, 10, tabd.data_ptr(), 53, 64, ped.data_ptr(), pl[0].data_ptr(),
                                pl[1].data_ptr(), 64, _st()), "embed")
    torch.cuda.synchronize()
    assert max_rel(unplanes(pl).cpu().reshape(4, 10, 64), ref) < 1e-5
    id32 = ids.to(torch.int32).to(DEV)
    N.check(lib.wd_embed_tokens(id32.data_ptr(), 0, 40, 10, tabd.data_ptr(), 53, 64, None, pl[0].data_ptr(),
                                pl[1].data_ptr(), 64, _st()), "embed32")
    torch.cuda.synchronize()
    assert max_rel(unplanes(pl).cpu().reshape(4, 10, 64), table[ids]) < 1e-5
    # im2col of the 4-channel latent (unet.py:1251) == unfold
    x = torch.randn(3, 4, 8, 32, generator=g)
    pl = torch.zeros(2, 3 * 256, 64, dtype=torch.bfloat16, device=DEV)
    xd = x.to(DEV)
    N.check(lib.wd_im2col3x3(xd.data_ptr(), 3, 4, 8, 32, pl[0].data_ptr(), pl[1].data_ptr(), 64, _st()), "im2col")
    torch.cuda.synchronize()
    unf = F.unfold(x, 3, padding=1).reshape(3, 4, 9, 256).permute(0, 3, 2, 1).reshape(3 * 256, 36)  # [tap][ci]
    got = unplanes(pl).cpu()
    assert max_rel(got[:, :36], unf) < 1e-5 and float(got[:, 36:].abs().max()) == 0.0
    # layout round trip
    tok = torch.zeros(3 * 256, 4, device=DEV)
    back = torch.zeros_like(xd)
    N.check(lib.wd_nchw_to_tokens(xd.data_ptr(), 3, 4, 256, tok.data_ptr(), 4, _st()), "to_tok")
    N.check(lib.wd_tokens_to_nchw(tok.data_ptr(), 4, 3, 4, 256, back.data_ptr(), _st()), "to_nchw")
    torch.cuda.synchronize()
    assert torch.equal(tok.cpu(), x.permute(0, 2, 3, 1).reshape(3 * 256, 4)) and torch.equal(back.cpu(), x)


def test_ddpm_step_bit_exact_and_noise_stream():
    lib = N.lib()
    T = 1000
    beta, alpha, ah = D.schedule(T)
    ca, cb, cs = 1 / torch.sqrt(alpha), (1 - alpha) / torch.sqrt(1 - ah), torch.sqrt(beta)
    g = torch.Generator().manual_seed(11)
    x, eps, z = (torch.randn(5, 4, 8, 32, generator=g) for _ in range(3))
    cad, cbd, csd = ca.to(DEV), cb.to(DEV), cs.to(DEV)
    for i in (999, 500, 2, 1):
        ref = D.reverse_step(beta, alpha, ah, x, eps, i, z)
        xd, ed, zd = x.to(DEV), eps.to(DEV), z.to(DEV)
        t_dev = torch.tensor([i], dtype=torch.int32, device=DEV)
        t64 = torch.zeros(5, dtype=torch.int64, device=DEV)
        N.check(lib.wd_ddpm_step(xd.data_ptr(), ed.data_ptr(), 5, 1024, cad.data_ptr(), cbd.data_ptr(), csd.data_ptr(),
                                 t_dev.data_ptr(), zd.data_ptr(), 0, 0, _st()), "ddpm")
        N.check(lib.wd_advance_timestep(t_dev.data_ptr(), -1, t64.data_ptr(), 5, _st()), "adv")
        torch.cuda.synchronize()
        assert torch.equal(xd.cpu(), ref), i  # same fp32 op order as train.py:236
        assert int(t_dev.item()) == i - 1 and torch.equal(t64.cpu(), torch.full((5,), i - 1, dtype=torch.int64))
    # device Philox noise: N(0,1), reproducible, and keyed by the GLOBAL sample index (shard-invariant)
    a = torch.zeros(64, 1024, device=DEV)
    N.check(lib.wd_randn(a.data_ptr(), 64, 1024, 1234, 0, 0, _st()), "randn")
    b = torch.zeros(32, 1024, device=DEV)
    N.check(lib.wd_randn(b.data_ptr(), 32, 1024, 1234, 32, 0, _st()), "randn")
    c = torch.zeros(64, 1024, device=DEV)
    N.check(lib.wd_randn(c.data_ptr(), 64, 1024, 1235, 0, 0, _st()), "randn")
    torch.cuda.synchronize()
    assert torch.equal(a[32:], b) and not torch.equal(a, c)
    assert abs(float(a.mean())) < 0.02 and abs(float(a.std()) - 1) < 0.02
    assert abs(float((a ** 4).mean()) - 3.0) < 0.15 and torch.isfinite(a).all()
    # z drawn inside the step: x_new - deterministic part == cs[t] * z with z ~ N(0,1)
    xd, ed = x.to(DEV), eps.to(DEV)
    t_dev = torch.tensor([700], dtype=torch.int32, device=DEV)
    N.check(lib.wd_ddpm_step(xd.data_ptr(), ed.data_ptr(), 5, 1024, cad.data_ptr(), cbd.data_ptr(), csd.data_ptr(),
                             t_dev.data_ptr(), None, 77, 0, _st()), "ddpm")
    torch.cuda.synchronize()
    zrec = (xd.cpu() - D.reverse_step(beta, alpha, ah, x, eps, 700, torch.zeros_like(x))) / cs[700]
    assert abs(float(zrec.mean())) < 0.05 and abs(float(zrec.std()) - 1) < 0.05


def test_noise_images_and_ema_exact():
    lib = N.lib()
    _, _, ah = D.schedule(1000)
    g = torch.Generator().manual_seed(13)
    x, eps = torch.randn(6, 4, 8, 32, generator=g), torch.randn(6, 4, 8, 32, generator=g)
    t = torch.tensor([1, 5, 300, 999, 2, 640], dtype=torch.int64)
    ref = D.noise_images(ah, x, t, eps)
    out = torch.zeros_like(x, device=DEV)
    xd, ed, td, sa, sb = x.to(DEV), eps.to(DEV), t.to(DEV), torch.sqrt(ah).to(DEV), torch.sqrt(1 - ah).to(DEV)
    N.check(lib.wd_noise_images(xd.data_ptr(), ed.data_ptr(), td.data_ptr(), sa.data_ptr(), sb.data_ptr(), 6, 1024,
                                out.data_ptr(), _st()), "noise_images")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)
    ema, p = torch.randn(100003, generator=g), torch.randn(100003, generator=g)
    ref = ema * 0.995 + (1 - 0.995) * p
    ed, pd = ema.to(DEV), p.to(DEV)
    N.check(lib.wd_ema_update(ed.data_ptr(), pd.data_ptr(), ema.numel(), 0.995, _st()), "ema")
    torch.cuda.synchronize()
    assert float((ed.cpu() - ref).abs().max()) <= 1e-9 + 2 ** -24 * float(ref.abs().max())


@pytest.mark.parametrize("B,hw,c,heads,L", [(3, 256, 320, 4, 10), (2, 70, 320, 4, 10), (2, 32, 64, 4, 7), (2, 64, 64, 2, 10),
                                            (2, 48, 320, 8, 5), (2, 20, 64, 2, 5), (1, 5, 320, 4, 9)])
def test_folded_cross_attention(B, hw, c, heads, L):
    """wd_xattn_fold + wd_xattn_fused == x + to_out(attention(to_q(LN(x)), K, V)) (unet.py:164-279,337-345) and the LayerNorm
    that follows, against fp64 torch."""
    lib = N.lib()
    assert lib.wd_xattn_supported(c, heads, L) == 1
    g = torch.Generator().manual_seed(hw + c + L)
    d = c // heads
    x = torch.randn(B * hw, c, generator=g) * 1.5 + 0.3
    kv = torch.randn(B * L, 2 * c + 8, generator=g) * 0.5
    wq = torch.randn(c, c, generator=g) / c ** 0.5
    wo = torch.randn(c, c, generator=g) / c ** 0.5
    bo = torch.randn(c, generator=g) * 0.1
    ga, be, ga2, be2 = (torch.randn(c, generator=g) * 0.2 + (1.0 if i % 2 == 0 else 0.0) for i in range(4))
    scale = d ** -0.5
    xd = x.double()
    n1 = F.layer_norm(xd, (c,), ga.double(), be.double(), 1e-5)
    q = n1 @ wq.double().t()
    k, v = kv[:, :c].double(), kv[:, c:2 * c].double()

    def hd(t, n):
        return t.reshape(B, n, heads, d).permute(0, 2, 1, 3)

    att = torch.softmax(hd(q, hw) @ hd(k, L).transpose(-1, -2) * scale, -1)
    o = (att @ hd(v, L)).permute(0, 2, 1, 3).reshape(B * hw, c)
    ref = o @ wo.double().t() + bo.double() + xd
    ref_n = F.layer_norm(ref, (c,), ga2.double(), be2.double(), 1e-5)
    dev = lambda t: t.contiguous().to(DEV)  # noqa: E731
    xg, kvg, wqg, wog, bog, gag, beg, ga2g, be2g = map(dev, (x, kv, wq, wo, bo, ga, be, ga2, be2))
    mq = torch.zeros(B, heads * L, c, device=DEV)
    mo = torch.zeros(B, heads * L, c, device=DEV)
    mq_pl = torch.zeros(B, 2, 64, c, dtype=torch.bfloat16, device=DEV)
    mot_pl = torch.zeros(B, 2, c, 64, dtype=torch.bfloat16, device=DEV)
    N.check(lib.wd_xattn_fold(kvg.data_ptr(), kv.shape[1], kvg.data_ptr() + 4 * c, kv.shape[1], B, heads, L, d, scale,
                              wqg.data_ptr(), wog.data_ptr(), c, mq.data_ptr(), mo.data_ptr(), mq_pl.data_ptr(), mot_pl.data_ptr(),
                              _st()), "fold")
    for planes in ((None, None), (mq_pl.data_ptr(), mot_pl.data_ptr())):  # fp32 VALU form, MFMA form
        out = torch.zeros(B * hw, c, device=DEV)
        pl = torch.zeros(2, B * hw, c, dtype=torch.bfloat16, device=DEV)
        N.check(lib.wd_xattn_fused(xg.data_ptr(), c, B, hw, c, gag.data_ptr(), beg.data_ptr(), 1e-5, mq.data_ptr(), mo.data_ptr(),
                                   heads, L, bog.data_ptr(), out.data_ptr(), c, ga2g.data_ptr(), be2g.data_ptr(), 1e-5,
                                   pl[0].data_ptr(), pl[1].data_ptr(), c, planes[0], planes[1], _st()), "fused")
        out2 = torch.zeros(B * hw, c, device=DEV)
        N.check(lib.wd_xattn_fused(xg.data_ptr(), c, B, hw, c, gag.data_ptr(), beg.data_ptr(), 1e-5, mq.data_ptr(), mo.data_ptr(),
                                   heads, L, bog.data_ptr(), out2.data_ptr(), c, None, None, 0.0, None, None, 0, planes[0],
                                   planes[1], _st()), "fused (no LN)")
        torch.cuda.synchronize()
        assert max_rel(out.cpu(), ref) < 2e-5 and torch.equal(out, out2)
        assert max_rel(unplanes(pl).cpu(), ref_n) < 3e-5


def test_folded_cross_attention_pair():
    """wd_xattn_pair == two wd_xattn_fused launches chained (attn1 -> attn2 -> norm3 planes), bit for bit."""
    lib = N.lib()
    B, hw, c, heads, L = 3, 100, 320, 4, 10
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(B * hw, c, generator=g) * 1.5).to(DEV)
    lay = []
    for _ in range(2):
        mq_pl = (torch.randn(B, 2, 64, c, generator=g) * 0.05).to(torch.bfloat16)
        mq_pl[:, :, heads * L:] = 0
        mot_pl = (torch.randn(B, 2, c, 64, generator=g) * 0.05).to(torch.bfloat16)
        mot_pl[:, :, :, heads * L:] = 0
        lay.append(dict(ga=(torch.randn(c, generator=g) * 0.2 + 1).to(DEV), be=(torch.randn(c, generator=g) * 0.2).to(DEV),
                        mq=mq_pl.to(DEV), mo=mot_pl.to(DEV), bi=(torch.randn(c, generator=g) * 0.1).to(DEV)))
    ga3, be3 = (torch.randn(c, generator=g) * 0.2 + 1).to(DEV), (torch.randn(c, generator=g) * 0.2).to(DEV)
    dummy = torch.zeros(B, heads * L, c, device=DEV)  # fp32 matrices: unused by the MFMA form

    def single(xin, ly, nxt):
        out = torch.zeros(B * hw, c, device=DEV)
        pl = torch.zeros(2, B * hw, c, dtype=torch.bfloat16, device=DEV)
        N.check(lib.wd_xattn_fused(xin.data_ptr(), c, B, hw, c, ly["ga"].data_ptr(), ly["be"].data_ptr(), 1e-5, dummy.data_ptr(),
                                   dummy.data_ptr(), heads, L, ly["bi"].data_ptr(), out.data_ptr(), c,
                                   ga3.data_ptr() if nxt else None, be3.data_ptr() if nxt else None, 1e-5,
                                   pl[0].data_ptr() if nxt else None, pl[1].data_ptr() if nxt else None, c, ly["mq"].data_ptr(),
                                   ly["mo"].data_ptr(), _st()), "fused")
        return out, pl

    t1, _ = single(x, lay[0], False)
    t2, pl2 = single(t1, lay[1], True)
    out = torch.zeros(B * hw, c, device=DEV)
    pl = torch.zeros(2, B * hw, c, dtype=torch.bfloat16, device=DEV)
    a, b = lay
    N.check(lib.wd_xattn_pair(x.data_ptr(), c, B, hw, c, 1e-5, heads, L, a["ga"].data_ptr(), a["be"].data_ptr(), a["mq"].data_ptr(),
                              a["mo"].data_ptr(), a["bi"].data_ptr(), b["ga"].data_ptr(), b["be"].data_ptr(), b["mq"].data_ptr(),
                              b["mo"].data_ptr(), b["bi"].data_ptr(), out.data_ptr(), c, ga3.data_ptr(), be3.data_ptr(), 1e-5,
                              pl[0].data_ptr(), pl[1].data_ptr(), c, _st()), "pair")
    torch.cuda.synchronize()
    assert torch.equal(out, t2) and torch.equal(pl, pl2)


@pytest.mark.parametrize("B,h,w,cin,cout,skip,ksplit", [(3, 8, 32, 64, 160, 0, 1), (2, 8, 32, 128, 320, 64, 1), (3, 5, 7, 64, 96, 0, 1),
                                                          (5, 4, 16, 64, 320, 128, 0), (2, 3, 200, 64, 64, 0, 1),
                                                          (64, 8, 32, 320, 320, 0, 1)])
@pytest.mark.parametrize("form", [0x1000])
def test_conv3x3_row_shared_taps_kernel(B, h, w, cin, cout, skip, ksplit, form):
    """wd_gemm with w_layout 2 (wd_conv3_kernel, forced by dbg 0x1000: the three taps of a kernel row share one A tile, 16x16x32 MFMA,
    stagger) vs F.conv2d (+ 1x1 skip over a second source), incl. panels that straddle samples, narrow / wide images, split-K, and
    the headline shape."""
    if not N.lib().wd_gemm_experimental():
        pytest.skip("library built without WDIFF_EXPERIMENTAL")
    g = torch.Generator().manual_seed(B * 100 + h * 10 + w + cin)
    x = torch.randn(B, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    tab, _, _ = conv_gather_table(h, w, "same")
    hw, m = h * w, B * h * w
    tok = x.permute(0, 2, 3, 1).reshape(m, cin).contiguous()
    a_list = [(planes_of(tok.to(DEV)), cin, 9, torch.from_numpy(tab).to(DEV), hw)]
    wp = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin)
    if skip:
        x2 = torch.randn(B, skip, h, w, generator=g)
        ws = torch.randn(cout, skip, generator=g) / skip ** 0.5
        ref = ref + F.conv2d(x2.double(), ws.double()[:, :, None, None])
        a_list.append((planes_of(x2.permute(0, 2, 3, 1).reshape(m, skip).contiguous().to(DEV)), skip, 1, None, 0))
        wp = torch.cat([wp, ws], 1)
    out, _ = run_gemm(a_list, wp.to(DEV), m, hw, bias=b.to(DEV), same_w=w, ksplit=ksplit, dbg=form)
    got = out.cpu().reshape(B, h, w, cout).permute(0, 3, 1, 2)
    assert max_rel(got, ref) < 2e-5
    out_again, _ = run_gemm(a_list, wp.to(DEV), m, hw, bias=b.to(DEV), same_w=w, ksplit=ksplit, dbg=form)
    assert torch.equal(out, out_again)
    # same result as the generic kernel up to summation order
    out0, _ = run_gemm(a_list, wp.to(DEV), m, hw, bias=b.to(DEV), ksplit=ksplit)
    assert max_rel(out.cpu(), out0.cpu()) < 5e-6


@pytest.mark.parametrize("B,h,w,c,oc,silu", [(3, 5, 7, 64, 4, 1), (2, 8, 32, 320, 4, 1), (2, 4, 40, 128, 3, 1), (1, 3, 64, 64, 1, 0)])
def test_gn_silu_conv3x3_few_output_channels(B, h, w, c, oc, silu):
    """wd_gn_conv3x3_few (the UNet's last layer in one fp32 launch) vs GroupNorm -> SiLU -> conv2d in fp64; statistics both from
    wd_gn_stats chunks and narrow / wide images (32 and 64 pixel lanes)."""
    lib = N.lib()
    g = torch.Generator().manual_seed(B + h + w + c)
    x = torch.randn(B, c, h, w, generator=g) * 2 + 0.3
    gam, bet = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    wt = torch.randn(oc, c, 3, 3, generator=g) / (9 * c) ** 0.5
    bias = torch.randn(oc, generator=g)
    y = F.group_norm(x.double(), 32, gam.double(), bet.double(), eps=1e-5)
    y = F.silu(y) if silu else y
    ref = F.conv2d(y, wt.double(), bias.double(), padding=1)
    hw = h * w
    tok = x.permute(0, 2, 3, 1).reshape(B * hw, c).contiguous().to(DEV)
    nchunk = lib.wd_gn_nchunk(hw)
    part = torch.zeros(B, nchunk, 32, 2, dtype=torch.float64, device=DEV)
    N.check(lib.wd_gn_stats(tok.data_ptr(), c, B, hw, c, c // 32, part.data_ptr(), _st()), "stats")
    out = torch.full((B, oc, h, w), float("nan"), device=DEV)
    assert lib.wd_gn_conv3x3_few_supported(c, w, oc)
    gd, bd, wd, bid = gam.to(DEV), bet.to(DEV), wt.to(DEV), bias.to(DEV)
    N.check(lib.wd_gn_conv3x3_few(tok.data_ptr(), c, B, h, w, c, c // 32, part.data_ptr(), nchunk, c // 32, gd.data_ptr(),
                                  bd.data_ptr(), 1e-5, silu, wd.data_ptr(), bid.data_ptr(), oc, out.data_ptr(), _st()), "gn_conv")
    torch.cuda.synchronize()
    assert max_rel(out.cpu(), ref) < 2e-6
    assert not lib.wd_gn_conv3x3_few_supported(c, 65, oc) and not lib.wd_gn_conv3x3_few_supported(c, w, 5)


@pytest.mark.parametrize("B,cin,cout,cpg,silu,film,resid,w", [(64, 320, 320, 10, 1, True, False, 16), (5, 64, 320, 20, 0, False, True, 16),
                                                               (16, 640, 320, 10, 1, True, False, 16), (3, 128, 160, 40, 1, False, False, 32),
                                                               (7, 384, 480, 10, 0, False, True, 16)])
def test_gemm_small_maps_whole_k_in_the_workgroup(B, cin, cout, cpg, silu, film, resid, w):
    _small_maps_case(B, cin, cout, cpg, silu, film, resid, w, 0)


@pytest.mark.parametrize("B,cin,cskip,cout", [(64, 320, 640, 320), (3, 64, 128, 160)])
def test_gemm_small_maps_conv_plus_identity_skip_source(B, cin, cskip, cout):
    _small_maps_case(B, cin, cout, 10, 1, True, False, 16, cskip)


@pytest.mark.parametrize("m,k,n,resid", [(4096, 1280, 320, True), (4096, 320, 320, True), (640, 64, 160, False)])
def test_gemm_small_maps_linear_layers(m, k, n, resid):
    """tile 64080 with an identity source only (1x1 / linear layers of the 4 x 16 level): 64 x 80 tiles, K over the waves."""
    lib = N.lib()
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g)
    wt = torch.randn(n, k, generator=g) / k ** 0.5
    bias = torch.randn(n, generator=g)
    rs = torch.randn(m, n, generator=g) if resid else None
    ref = x.double() @ wt.double().t() + bias.double() + (rs.double() if resid else 0)
    pl, wp = planes_of(x.to(DEV)), planes_of(wt.to(DEV))
    wf = torch.empty_like(wp)
    N.check(lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), n, k, wf[0].data_ptr(), wf[1].data_ptr(), _st()), "pack")
    a = N.WdGemmArgs()
    s0 = N.WdSrc()
    s0.hi, s0.lo, s0.ld, s0.c, s0.ntaps = pl[0].data_ptr(), pl[1].data_ptr(), k, k, 1
    a.src[0] = s0
    a.nsrc, a.npass = 1, 3
    a.w_hi, a.w_lo, a.w_layout, a.tile = wf[0].data_ptr(), wf[1].data_ptr(), 3, 64080
    a.m, a.n, a.ktot, a.hw_out = m, n, k, 64
    bd, out, opl = bias.to(DEV), torch.full((m, n), float("nan"), device=DEV), torch.zeros(2, m, n, dtype=torch.bfloat16, device=DEV)
    a.bias, a.out_f32, a.out_ld = bd.data_ptr(), out.data_ptr(), n
    a.out_hi, a.out_lo, a.out_pl_ld = opl[0].data_ptr(), opl[1].data_ptr(), n
    if resid:
        rd = rs.to(DEV)
        a.resid, a.resid_ld = rd.data_ptr(), n
    N.check(lib.wd_gemm(C.byref(a), _st()), "wd_gemm 64080 linear")
    torch.cuda.synchronize()
    assert max_rel(out.cpu(), ref) < 2e-5
    assert max_rel(unplanes(opl).cpu(), ref) < 2e-5


@pytest.mark.parametrize("B,cin,cout", [(64, 320, 320), (3, 64, 160)])
def test_gemm_small_maps_stride_two_convolution(B, cin, cout):
    """tile 64080 over the Downsample convolution (3x3, stride 2, pad 1: 8x32 -> 4x16): the slab holds the nine source rows."""
    lib = N.lib()
    g = torch.Generator().manual_seed(B + cin)
    h, w = 8, 32
    x = torch.randn(B, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), bias.double(), stride=2, padding=1)
    tab, ho, wo = conv_gather_table(h, w, "down")
    m = B * ho * wo
    pl = planes_of(x.permute(0, 2, 3, 1).reshape(B * h * w, cin).contiguous().to(DEV))
    wp = planes_of(wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).to(DEV))
    wf = torch.empty_like(wp)
    N.check(lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), cout, 9 * cin, wf[0].data_ptr(), wf[1].data_ptr(), _st()), "pack")
    tabd = torch.from_numpy(tab).to(DEV)
    a = N.WdGemmArgs()
    s0 = N.WdSrc()
    s0.hi, s0.lo, s0.gather = pl[0].data_ptr(), pl[1].data_ptr(), tabd.data_ptr()
    s0.ld, s0.c, s0.ntaps, s0.hw_src = cin, cin, 9, h * w
    a.src[0] = s0
    a.nsrc, a.npass = 1, 3
    a.w_hi, a.w_lo, a.w_layout, a.tile, a.slab_rows = wf[0].data_ptr(), wf[1].data_ptr(), 3, 64080, wo
    a.m, a.n, a.ktot, a.hw_out = m, cout, 9 * cin, ho * wo
    bd, out = bias.to(DEV), torch.full((m, cout), float("nan"), device=DEV)
    a.bias, a.out_f32, a.out_ld = bd.data_ptr(), out.data_ptr(), cout
    N.check(lib.wd_gemm(C.byref(a), _st()), "wd_gemm 64080 stride 2")
    torch.cuda.synchronize()
    assert max_rel(out.cpu(), ref.permute(0, 2, 3, 1).reshape(m, cout)) < 2e-5


def _small_maps_case(B, cin, cout, cpg, silu, film, resid, w, cskip):
    """wd_gemm_args.tile = 64080 (wd_gemmq_kernel): 3x3 convolution over 64-position samples, 64 x 80 tiles, the input rows kept in
    LDS, the eight waves splitting K - plain epilogue (bias / FiLM / residual / statistics / planes of the result) and with the
    consumer's GroupNorm (+SiLU) in the same launch, vs fp64 torch and vs the default kernel."""
    lib = N.lib()
    g = torch.Generator().manual_seed(B + cin + cout + w)
    h = 64 // w
    hw, m = h * w, B * h * w
    x = torch.randn(B, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    fl = torch.randn(B, cout, generator=g) * 0.3 if film else None
    rs = torch.randn(m, cout, generator=g) if resid else None
    gam, bet = torch.randn(cout, generator=g) * 0.2 + 1, torch.randn(cout, generator=g) * 0.2
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    xs = torch.randn(B, cskip, h, w, generator=g) if cskip else None     # the 1x1 skip convolution of a decoder block as a second source
    ws = torch.randn(cout, cskip, generator=g) / max(cskip, 1) ** 0.5 if cskip else None
    if cskip:
        ref = ref + F.conv2d(xs.double(), ws.double()[:, :, None, None])
    if film:
        ref = ref + fl.double()[:, :, None, None]
    if resid:
        ref = ref + rs.double().reshape(B, h, w, cout).permute(0, 3, 1, 2)
    refn = F.group_norm(ref, cout // cpg, gam.double(), bet.double(), 1e-5)
    if silu:
        refn = F.silu(refn)
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(m, -1)  # noqa: E731
    tab, _, _ = conv_gather_table(h, w, "same")
    pl = planes_of(tok(x).contiguous().to(DEV))
    pls = planes_of(tok(xs).contiguous().to(DEV)) if cskip else None
    wcat = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin)
    if cskip:
        wcat = torch.cat([wcat, ws], 1)
    wp = planes_of(wcat.contiguous().to(DEV))
    wf = torch.empty_like(wp)
    N.check(lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), cout, wcat.shape[1], wf[0].data_ptr(), wf[1].data_ptr(), _st()), "pack")
    tabd = torch.from_numpy(tab).to(DEV)
    pc = 10 if cpg % 10 == 0 else cpg

    def args_for(gn: bool, small: bool):
        a = N.WdGemmArgs()
        s = N.WdSrc()
        s.hi, s.lo, s.gather = pl[0].data_ptr(), pl[1].data_ptr(), tabd.data_ptr()
        s.ld, s.c, s.ntaps, s.hw_src = cin, cin, 9, hw
        a.src[0] = s
        a.nsrc, a.npass = 1, 3
        if cskip:
            s1 = N.WdSrc()
            s1.hi, s1.lo, s1.ld, s1.c, s1.ntaps = pls[0].data_ptr(), pls[1].data_ptr(), cskip, cskip, 1
            a.src[1] = s1
            a.nsrc = 2
        if small:
            a.w_hi, a.w_lo, a.w_layout, a.tile, a.slab_rows = wf[0].data_ptr(), wf[1].data_ptr(), 3, 64080, w
        else:
            a.w_hi, a.w_lo = wp[0].data_ptr(), wp[1].data_ptr()
        a.m, a.n, a.ktot, a.hw_out = m, cout, 9 * cin + cskip, hw
        keep = dict(bias=bias.to(DEV), out=torch.full((m, cout), float("nan"), device=DEV),
                    opl=torch.zeros(2, m, cout, dtype=torch.bfloat16, device=DEV), ws=torch.empty(8 * m * cout, device=DEV),
                    part=torch.zeros(B, 1, cout // pc, 2, dtype=torch.float64, device=DEV), gam=gam.to(DEV), bet=bet.to(DEV))
        a.bias = keep["bias"].data_ptr()
        if film:
            keep["fl"] = fl.to(DEV)
            a.rowvec, a.rowvec_ld = keep["fl"].data_ptr(), cout
        if resid:
            keep["rs"] = rs.to(DEV)
            a.resid, a.resid_ld = keep["rs"].data_ptr(), cout
        a.out_f32, a.out_ld = keep["out"].data_ptr(), cout
        a.out_hi, a.out_lo, a.out_pl_ld = keep["opl"][0].data_ptr(), keep["opl"][1].data_ptr(), cout
        a.ws, a.ws_floats = keep["ws"].data_ptr(), keep["ws"].numel()
        a.stat_part, a.stat_cpg = keep["part"].data_ptr(), pc
        if gn:
            a.gn_gamma, a.gn_beta, a.gn_eps, a.gn_silu, a.gn_cpg = keep["gam"].data_ptr(), keep["bet"].data_ptr(), 1e-5, silu, cpg
        return a, keep

    # plain epilogue: result, planes of the result, statistics - against fp64 and against the default kernel
    a, k = args_for(False, True)
    N.check(lib.wd_gemm(C.byref(a), _st()), "wd_gemm 64080")
    torch.cuda.synchronize()
    assert max_rel(k["out"].cpu(), tok(ref)) < 2e-5
    assert max_rel(unplanes(k["opl"]).cpu(), tok(ref)) < 2e-5
    a0, k0 = args_for(False, False)
    N.check(lib.wd_gemm(C.byref(a0), _st()), "wd_gemm")
    torch.cuda.synchronize()
    assert max_rel(k["out"].cpu(), k0["out"].cpu()) < 5e-6
    assert torch.allclose(k["part"], k0["part"], rtol=2e-4, atol=2e-3)  # (fp32 partial sums in a different order)
    # replay: bit-identical (fixed summation order over the eight waves)
    a2, k2 = args_for(False, True)
    N.check(lib.wd_gemm(C.byref(a2), _st()), "wd_gemm 64080 again")
    torch.cuda.synchronize()
    assert torch.equal(k["out"], k2["out"])
    # the consumer's GroupNorm in the same launch (160-multiple widths only, as the combine-launch form)
    ag, kg = args_for(True, True)
    rc = lib.wd_gemm(C.byref(ag), _st())
    if cout % 160:
        assert rc != 0
        return
    N.check(rc, "wd_gemm 64080 + GroupNorm")
    torch.cuda.synchronize()
    assert torch.equal(kg["out"], k["out"])
    assert max_rel(unplanes(kg["opl"]).cpu(), tok(refn)) < 3e-5
    assert torch.allclose(kg["part"], k["part"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("B,cin,cout,cpg,silu,film,resid", [(64, 320, 320, 10, 1, True, False), (5, 64, 320, 20, 0, False, True),
                                                             (16, 128, 160, 40, 1, False, False)])
def test_gemm_groupnorm_in_the_combine_launch(B, cin, cout, cpg, silu, film, resid):
    """wd_gemm_args.gn_*: a K-cut 3x3 convolution over 4x16 images whose combine launch also applies the consumer's GroupNorm
    (+SiLU) and writes it as operand planes - vs conv -> (+FiLM / +residual) -> F.group_norm -> SiLU in fp64; the fp32 result
    and the (sample, group) statistics are still written.  Requests the shapes cannot serve are errors, not skipped norms."""
    lib = N.lib()
    g = torch.Generator().manual_seed(B + cin + cout)
    h, w = 4, 16
    hw, m = h * w, B * h * w
    x = torch.randn(B, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    fl = torch.randn(B, cout, generator=g) * 0.3 if film else None
    rs = torch.randn(m, cout, generator=g) if resid else None
    gam, bet = torch.randn(cout, generator=g) * 0.2 + 1, torch.randn(cout, generator=g) * 0.2
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    if film:
        ref = ref + fl.double()[:, :, None, None]
    if resid:
        ref = ref + rs.double().reshape(B, h, w, cout).permute(0, 3, 1, 2)
    refn = F.group_norm(ref, cout // cpg, gam.double(), bet.double(), 1e-5)
    if silu:
        refn = F.silu(refn)
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(m, -1)  # noqa: E731
    tab, _, _ = conv_gather_table(h, w, "same")
    pl = planes_of(tok(x).contiguous().to(DEV))
    wp = planes_of(wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).to(DEV))
    tabd = torch.from_numpy(tab).to(DEV)
    pc = 10 if cpg % 10 == 0 else cpg  # the statistics groups may be finer than the consumer's groups

    def args_for(gn: bool, ksplit=0):
        a = N.WdGemmArgs()
        s = N.WdSrc()
        s.hi, s.lo, s.gather = pl[0].data_ptr(), pl[1].data_ptr(), tabd.data_ptr()
        s.ld, s.c, s.ntaps, s.hw_src = cin, cin, 9, hw
        a.src[0] = s
        a.nsrc, a.npass = 1, 3
        a.w_hi, a.w_lo = wp[0].data_ptr(), wp[1].data_ptr()
        a.m, a.n, a.ktot, a.hw_out = m, cout, 9 * cin, hw
        a.ksplit = ksplit
        keep = dict(bias=bias.to(DEV), out=torch.full((m, cout), float("nan"), device=DEV),
                    opl=torch.zeros(2, m, cout, dtype=torch.bfloat16, device=DEV), ws=torch.empty(8 * m * cout, device=DEV),
                    part=torch.zeros(B, 1, cout // pc, 2, dtype=torch.float64, device=DEV), gam=gam.to(DEV), bet=bet.to(DEV))
        a.bias = keep["bias"].data_ptr()
        if film:
            keep["fl"] = fl.to(DEV)
            a.rowvec, a.rowvec_ld = keep["fl"].data_ptr(), cout
        if resid:
            keep["rs"] = rs.to(DEV)
            a.resid, a.resid_ld = keep["rs"].data_ptr(), cout
        a.out_f32, a.out_ld = keep["out"].data_ptr(), cout
        a.out_hi, a.out_lo, a.out_pl_ld = keep["opl"][0].data_ptr(), keep["opl"][1].data_ptr(), cout
        a.ws, a.ws_floats = keep["ws"].data_ptr(), keep["ws"].numel()
        a.stat_part, a.stat_cpg = keep["part"].data_ptr(), pc
        if gn:
            a.gn_gamma, a.gn_beta, a.gn_eps, a.gn_silu, a.gn_cpg = keep["gam"].data_ptr(), keep["bet"].data_ptr(), 1e-5, silu, cpg
        return a, keep

    cut = lib.wd_gemm_auto_ksplit(m, cout, 9 * cin, 8 * m * cout)
    a, k = args_for(True)
    rc = lib.wd_gemm(C.byref(a), _st())
    if cut <= 1:  # (B = 64 at cout = 320 fills the chip: 64 tiles x 4; a tiny batch may not be cut at all)
        assert rc != 0
        return
    N.check(rc, "wd_gemm + GroupNorm")
    torch.cuda.synchronize()
    assert max_rel(k["out"].cpu(), tok(ref)) < 2e-5
    assert max_rel(unplanes(k["opl"]).cpu(), tok(refn)) < 3e-5
    # statistics as the ordinary combine writes them
    a0, k0 = args_for(False)
    N.check(lib.wd_gemm(C.byref(a0), _st()), "wd_gemm")
    torch.cuda.synchronize()
    assert torch.equal(k["out"], k0["out"])
    assert torch.allclose(k["part"], k0["part"], rtol=1e-12, atol=0)
    assert max_rel(unplanes(k0["opl"]).cpu(), tok(ref)) < 2e-5   # without gn_*: the planes hold the result itself
    # not a K-cut launch -> error
    a1, _k1 = args_for(True, ksplit=1)
    assert lib.wd_gemm(C.byref(a1), _st()) != 0


@pytest.mark.parametrize("B,H,nq,nk,d", [(3, 4, 256, 779, 80), (2, 4, 64, 779, 80), (2, 2, 100, 70, 32), (1, 1, 17, 130, 96)])
def test_attention_over_prepacked_keys_and_values(B, H, nq, nk, d):
    """wd_attention_pack_kv + wd_attention_packed (the K / V images of a fixed context built once) == wd_attention on the same
    K / V: the same LDS images, the same products, the same bits - for 128-query tiles and for the key-split 64-query tiles."""
    lib = N.lib()
    g = torch.Generator().manual_seed(nq + nk + d)
    inner = H * d
    q = (torch.randn(B * nq, inner, generator=g)).to(DEV)
    kv = (torch.randn(B * nk, 2 * inner + 8, generator=g)).to(DEV)
    scale = d ** -0.5
    nimg = lib.wd_attention_packed_elems(B, H, nk, d)
    assert nimg > 0 and lib.wd_attention_packed_elems(B, H, 10, d) == 0
    img = torch.empty(nimg, dtype=torch.bfloat16, device=DEV)
    N.check(lib.wd_attention_pack_kv(kv.data_ptr(), kv.shape[1], kv.data_ptr() + 4 * inner, kv.shape[1], B, H, nk, d, img.data_ptr(),
                                     _st()), "pack")
    ref = torch.full((B * nq, inner), float("nan"), device=DEV)
    N.check(lib.wd_attention(q.data_ptr(), inner, kv.data_ptr(), kv.shape[1], kv.data_ptr() + 4 * inner, kv.shape[1], B, H, nq, nk, d,
                             scale, ref.data_ptr(), None, None, inner, nq, 0, _st()), "attention")
    out = torch.full((B * nq, inner), float("nan"), device=DEV)
    pl = torch.zeros(2, B * nq, inner, dtype=torch.bfloat16, device=DEV)
    N.check(lib.wd_attention_packed(q.data_ptr(), inner, img.data_ptr(), B, H, nq, nk, d, scale, out.data_ptr(), pl[0].data_ptr(),
                                    pl[1].data_ptr(), inner, nq, 0, _st()), "packed")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert max_rel(unplanes(pl).cpu(), ref.cpu()) < 1e-5
    # and against fp64
    qd, kd, vd = q.double().cpu(), kv[:, :inner].double().cpu(), kv[:, inner:2 * inner].double().cpu()
    hd = lambda t, n: t.reshape(B, n, H, d).permute(0, 2, 1, 3)  # noqa: E731
    att = torch.softmax(hd(qd, nq) @ hd(kd, nk).transpose(-1, -2) * scale, -1)
    o = (att @ hd(vd, nk)).permute(0, 2, 1, 3).reshape(B * nq, inner)
    assert max_rel(out.cpu(), o) < 2e-5
