"""GPU parity of the backward building blocks (include/wdiff_hip.h, "backward building blocks") against torch autograd of
the same op on the CPU (fp64 where it matters).  Tolerances as in test_gpu_kernels.py."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests._common import max_rel, rel_err  # noqa: E402
from tests.test_gpu_kernels import DEV, _st, planes_of, run_gemm, unplanes  # noqa: E402
from worddiffusion_amd import _native as N  # noqa: E402
from worddiffusion_amd.backward import conv_bwd_table, pack_dx_weight, unpack_dw  # noqa: E402
from worddiffusion_amd.engine import conv_gather_table  # noqa: E402


def transpose_planes(lib, src, is_f32, c, m, mpad, gather=None, ntaps=1, hw_out=0, hw_src=0, tap_minor=0):
    out = torch.zeros(2, ntaps * c, mpad, dtype=torch.bfloat16, device=DEV)
    if is_f32:
        hi, lo, ld = src.data_ptr(), None, src.shape[1]
    else:
        hi, lo, ld = src[0].data_ptr(), src[1].data_ptr(), src.shape[2]
    N.check(lib.wd_transpose_planes(hi, lo, int(is_f32), ld, c, gather.data_ptr() if gather is not None else None, ntaps,
                                    hw_out, hw_src, m, mpad, tap_minor, out[0].data_ptr(), out[1].data_ptr(), _st()), "transpose")
    return out


@pytest.mark.parametrize("mode,B,Ci,h,w,Co", [("same", 3, 64, 8, 32, 128), ("down", 2, 64, 8, 16, 64), ("same", 1, 64, 5, 7, 64)])
def test_conv_data_and_weight_gradients_through_wd_gemm(mode, B, Ci, h, w, Co):
    lib = N.lib()
    g = torch.Generator().manual_seed(B + Ci + h)
    x = torch.randn(B, Ci, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = (torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64) / (9 * Ci) ** 0.5).requires_grad_(True)
    stride = 2 if mode == "down" else 1
    y = F.conv2d(x, wt, None, stride=stride, padding=1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    ftab, ho, wo = conv_gather_table(h, w, mode)
    btab, _, _ = conv_bwd_table(h, w, mode)
    hw_in, hw_out, m_in, m_out = h * w, ho * wo, B * h * w, B * ho * wo
    dy_tok = dy.permute(0, 2, 3, 1).reshape(m_out, Co).float().contiguous()
    x_tok = x.detach().permute(0, 2, 3, 1).reshape(m_in, Ci).float().contiguous()
    # ---- data gradient: tap-gather GEMM over d out with the inverse table and [C_in][tap][C_out] weights
    dx, _ = run_gemm([(planes_of(dy_tok.to(DEV)), Co, 9, torch.from_numpy(btab).to(DEV), hw_out)],
                     pack_dx_weight(wt.detach().float()).to(DEV), m_in, hw_in)
    ref_dx = x.grad.permute(0, 2, 3, 1).reshape(m_in, Ci)
    assert rel_err(dx.cpu(), ref_dx) < 2e-5
    # ---- weight gradient: GEMM over the tokens with transposed operands
    mpad = (m_out + 63) // 64 * 64
    dyT = transpose_planes(lib, dy_tok.to(DEV), True, Co, m_out, mpad)
    xcolT = transpose_planes(lib, planes_of(x_tok.to(DEV)), False, Ci, m_out, mpad, torch.from_numpy(ftab).to(DEV), 9, hw_out,
                             hw_in, tap_minor=1)
    args = N.WdGemmArgs()
    s0 = N.WdSrc()
    s0.hi, s0.lo, s0.ld, s0.c, s0.ntaps = dyT[0].data_ptr(), dyT[1].data_ptr(), mpad, mpad, 1
    args.src[0] = s0
    args.nsrc, args.npass = 1, 3
    args.w_hi, args.w_lo = xcolT[0].data_ptr(), xcolT[1].data_ptr()
    args.m, args.n, args.ktot, args.hw_out = Co, 9 * Ci, mpad, 1
    dwp = torch.zeros(Co, 9 * Ci, device=DEV)
    args.out_f32, args.out_ld = dwp.data_ptr(), 9 * Ci
    ws = torch.empty(8 * Co * 9 * Ci, device=DEV)
    args.ksplit, args.ws, args.ws_floats = 0, ws.data_ptr(), ws.numel()
    N.check(lib.wd_gemm(C.byref(args), _st()), "dW gemm")
    torch.cuda.synchronize()
    assert rel_err(dwp.cpu().reshape(wt.shape), wt.grad) < 2e-5  # tap-minor rows: the GEMM output IS the OIHW gradient
    packed = torch.randn(Co, 9 * Ci + 4, device=DEV)
    perm = torch.zeros(Co, Ci, 3, 3, device=DEV)
    N.check(lib.wd_permute_dw(packed.data_ptr(), 9 * Ci + 4, Co, Ci, 9, perm.data_ptr(), _st()), "permute")
    acc = perm.clone()
    N.check(lib.wd_add(acc.data_ptr(), perm.data_ptr(), acc.numel(), _st()), "add")
    torch.cuda.synchronize()
    assert torch.equal(perm.cpu(), unpack_dw(packed.cpu()[:, :9 * Ci], wt.shape).float())
    assert torch.equal(acc, 2 * perm)


def run_dw(lib, dpl, xpl, n, c, m, gather=None, ntaps=1, hw_out=0, hw_src=0, npass=3, nslice=0, acc=None, x_col=0):
    """wd_dw over planes dpl [2][m][d_ld], xpl [2][rows][x_ld] (channels x_col .. x_col + c) -> grad [n][c * ntaps]."""
    grad = torch.full((n, c * ntaps), float("nan"), device=DEV) if acc is None else acc.clone()
    ws = torch.empty(max(nslice, lib.wd_dw_slices(m, n, c, ntaps), 1) * n * c * ntaps, device=DEV)
    a = N.WdDwArgs()
    a.d_hi, a.d_lo = dpl[0].data_ptr(), (dpl[1].data_ptr() if npass == 3 else None)
    a.x_hi, a.x_lo = xpl[0].data_ptr() + 2 * x_col, (xpl[1].data_ptr() + 2 * x_col if npass == 3 else None)
    a.gather = gather.data_ptr() if gather is not None else None
    a.grad, a.grad_ld, a.ws, a.ws_floats = grad.data_ptr(), c * ntaps, ws.data_ptr(), ws.numel()
    a.d_ld, a.x_ld = dpl.shape[2], xpl.shape[2]
    a.ntaps, a.hw_out, a.hw_src = ntaps, hw_out, hw_src
    a.m, a.n, a.c, a.npass, a.accumulate, a.nslice = m, n, c, npass, int(acc is not None), nslice
    N.check(lib.wd_dw(C.byref(a), _st()), "wd_dw")
    torch.cuda.synchronize()
    return grad


@pytest.mark.parametrize("mode,B,Ci,h,w,Co,nslice", [("same", 4, 320, 8, 32, 320, 0), ("same", 2, 160, 8, 32, 160, 1),
                                                      ("same", 3, 320, 4, 16, 160, 3), ("down", 2, 160, 8, 32, 320, 2),
                                                      ("up", 2, 160, 4, 16, 160, 0), ("same", 16, 640, 8, 32, 320, 0)])
def test_weight_gradient_from_row_major_planes_conv(mode, B, Ci, h, w, Co, nslice):
    """wd_dw (transposed LDS reads, no transposed copies) against autograd's conv weight gradient, fp64."""
    lib = N.lib()
    g = torch.Generator().manual_seed(B * 7 + Ci + h)
    x = torch.randn(B, Ci, h, w, generator=g, dtype=torch.float64)
    wt = (torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64) / (9 * Ci) ** 0.5).requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if mode == "up" else x
    y = F.conv2d(xin, wt, None, stride=2 if mode == "down" else 1, padding=1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    ftab, ho, wo = conv_gather_table(h, w, mode)
    m = B * ho * wo
    if not lib.wd_dw_supported(m, Co, Ci, 9, ho * wo):
        pytest.skip("shape outside wd_dw_supported")
    tok = x.permute(0, 2, 3, 1).reshape(B * h * w, Ci).float().contiguous()
    dtok = dy.permute(0, 2, 3, 1).reshape(m, Co).float().contiguous()
    xpl, dpl = planes_of(tok.to(DEV)), planes_of(dtok.to(DEV))
    got = run_dw(lib, dpl, xpl, Co, Ci, m, torch.from_numpy(ftab).to(DEV), 9, ho * wo, h * w, nslice=nslice)
    # reference on the operands the kernel sees (the bf16x2-rounded values), fp64
    x2 = unplanes(xpl).double().cpu().reshape(B, h, w, Ci).permute(0, 3, 1, 2)
    d2 = unplanes(dpl).double().cpu().reshape(B, ho, wo, Co).permute(0, 3, 1, 2)
    w2 = wt.detach().clone().requires_grad_(True)
    x2in = F.interpolate(x2, scale_factor=2, mode="nearest") if mode == "up" else x2
    F.conv2d(x2in, w2, None, stride=2 if mode == "down" else 1, padding=1).backward(d2)
    assert rel_err(got.cpu().reshape(Co, Ci, 3, 3), w2.grad) < 5e-6
    assert rel_err(got.cpu().reshape(Co, Ci, 3, 3), wt.grad) < 2e-5
    # accumulation into a gradient that holds a value; one-pass (bf16) mode
    base = torch.randn(Co, Ci * 9, generator=g).to(DEV)
    got2 = run_dw(lib, dpl, xpl, Co, Ci, m, torch.from_numpy(ftab).to(DEV), 9, ho * wo, h * w, nslice=nslice, acc=base)
    assert rel_err((got2 - base).cpu().reshape(Co, Ci, 3, 3), w2.grad) < 2e-5
    got1 = run_dw(lib, dpl, xpl, Co, Ci, m, torch.from_numpy(ftab).to(DEV), 9, ho * wo, h * w, nslice=nslice, npass=1)
    x1 = xpl[0].double().cpu().reshape(B, h, w, Ci).permute(0, 3, 1, 2)
    d1 = dpl[0].double().cpu().reshape(B, ho, wo, Co).permute(0, 3, 1, 2)
    w1 = wt.detach().clone().requires_grad_(True)
    x1in = F.interpolate(x1, scale_factor=2, mode="nearest") if mode == "up" else x1
    F.conv2d(x1in, w1, None, stride=2 if mode == "down" else 1, padding=1).backward(d1)
    assert rel_err(got1.cpu().reshape(Co, Ci, 3, 3), w1.grad) < 5e-6


@pytest.mark.parametrize("m,n,c,ld_extra,x_col,nslice", [(16384, 320, 320, 0, 0, 0), (4096, 2560, 320, 0, 0, 0), (1024, 320, 1280, 64, 0, 5),
                                                         (512, 160, 160, 32, 160, 4), (128, 160, 160, 0, 0, 0)])
def test_weight_gradient_from_row_major_planes_linear(m, n, c, ld_extra, x_col, nslice):
    lib = N.lib()
    g = torch.Generator().manual_seed(m + n + c)
    d = torch.randn(m, n + ld_extra, generator=g)
    x = torch.randn(m, x_col + c + ld_extra, generator=g)
    dpl, xpl = planes_of(d.to(DEV)), planes_of(x.to(DEV))
    got = run_dw(lib, dpl, xpl, n, c, m, hw_out=min(m, 256), hw_src=min(m, 256), nslice=nslice, x_col=x_col)
    ref = unplanes(dpl).double().cpu()[:, :n].t() @ unplanes(xpl).double().cpu()[:, x_col:x_col + c]
    assert rel_err(got.cpu(), ref) < 5e-6
    assert rel_err(got.cpu(), d.double()[:, :n].t() @ x.double()[:, x_col:x_col + c]) < 2e-5


@pytest.mark.parametrize("m,n,c,nitems,npass", [(4096, 320, 320, 6, 3), (1024, 160, 320, 3, 1), (16384, 320, 320, 8, 3)])
def test_weight_gradient_group_of_layers_in_one_launch(m, n, c, nitems, npass):
    """wd_dw_group: several same-shape layers (different operands, leading dimensions, accumulate flags) in one launch."""
    lib = N.lib()
    g = torch.Generator().manual_seed(m + n + nitems)
    arr = (N.WdDwItem * nitems)()
    keep, refs, grads, bases = [], [], [], []
    for i in range(nitems):
        d = torch.randn(m, n + 8 * (i % 3), generator=g)
        x = torch.randn(m, c + 16 * (i % 2), generator=g)
        dpl, xpl = planes_of(d.to(DEV)), planes_of(x.to(DEV))
        acc = i % 2
        grad = torch.randn(n, c, generator=g).to(DEV) if acc else torch.full((n, c), float("nan"), device=DEV)
        bases.append(grad.clone() if acc else None)
        arr[i].d_hi, arr[i].d_lo = dpl[0].data_ptr(), (dpl[1].data_ptr() if npass == 3 else None)
        arr[i].x_hi, arr[i].x_lo = xpl[0].data_ptr(), (xpl[1].data_ptr() if npass == 3 else None)
        arr[i].grad, arr[i].grad_ld, arr[i].d_ld, arr[i].x_ld, arr[i].accumulate = grad.data_ptr(), c, dpl.shape[2], xpl.shape[2], acc
        dd = (unplanes(dpl) if npass == 3 else dpl[0].float()).double().cpu()[:, :n]
        xx = (unplanes(xpl) if npass == 3 else xpl[0].float()).double().cpu()[:, :c]
        refs.append(dd.t() @ xx)
        grads.append(grad)
        keep += [dpl, xpl]
    dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    ws = torch.empty(lib.wd_dw_group_slices(m, n, c, 1, nitems) * nitems * n * c, device=DEV)
    a = N.WdDwArgs()
    a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
    a.ntaps, a.hw_out, a.hw_src, a.m, a.n, a.c, a.npass = 1, 256, 256, m, n, c, npass
    N.check(lib.wd_dw_group(C.byref(a), C.cast(arr, C.c_void_p), dev.data_ptr(), nitems, _st()), "wd_dw_group")
    torch.cuda.synchronize()
    for i in range(nitems):
        got = grads[i] - bases[i] if bases[i] is not None else grads[i]
        assert rel_err(got.cpu(), refs[i]) < (2e-5 if bases[i] is not None else 5e-6), i


def test_weight_gradient_kernel_rejects_bad_arguments():
    lib = N.lib()
    assert not lib.wd_dw_supported(100, 320, 320, 9, 50)
    assert not lib.wd_dw_supported(256, 300, 320, 9, 256)
    assert not lib.wd_dw_supported(256, 320, 64, 1, 256)
    assert lib.wd_dw_supported(16384, 320, 640, 9, 256)
    a = N.WdDwArgs()
    assert lib.wd_dw(C.byref(a), _st()) == N.WD_EINVAL


@pytest.mark.parametrize("m,inner,extra", [(512, 128, 0), (4096, 1280, 0), (200, 64, 8)])
def test_geglu_backward_inside_the_dout_preparation(m, inner, extra):
    """wd_dout_prep_geglu = wd_geglu_bwd followed by wd_dout_prep, without the fp32 intermediate: same planes, transposed planes and
    64-row column sums."""
    lib = N.lib()
    g = torch.Generator().manual_seed(m + inner)
    u = torch.randn(m, 2 * inner + extra, generator=g).to(DEV)
    dh = torch.randn(m, inner + extra, generator=g).to(DEV)
    n, mpad = 2 * inner, (m + 63) // 64 * 64
    du = torch.empty(m, n, device=DEV)
    N.check(lib.wd_geglu_bwd(u.data_ptr(), u.shape[1], dh.data_ptr(), dh.shape[1], m, inner, du.data_ptr(), n, _st()), "geglu_bwd")
    outs = []
    for fused in (0, 1):
        pl = torch.full((2, m, n), -1, dtype=torch.bfloat16, device=DEV)
        tp = torch.zeros((2, n, mpad), dtype=torch.bfloat16, device=DEV)
        cp = torch.full((mpad // 64, n), float("nan"), device=DEV)
        if fused:
            N.check(lib.wd_dout_prep_geglu(u.data_ptr(), u.shape[1], dh.data_ptr(), dh.shape[1], m, inner, mpad, pl[0].data_ptr(),
                                           pl[1].data_ptr(), tp[0].data_ptr(), tp[1].data_ptr(), cp.data_ptr(), _st()), "prep geglu")
        else:
            N.check(lib.wd_dout_prep(du.data_ptr(), n, m, n, n, mpad, pl[0].data_ptr(), pl[1].data_ptr(), tp[0].data_ptr(),
                                     tp[1].data_ptr(), cp.data_ptr(), _st()), "prep")
        torch.cuda.synchronize()
        outs.append((pl.clone(), tp.clone(), cp.clone()))
    # (not bit for bit: the two kernels contract their multiplications differently - a last-place difference in fp32)
    (pl0, tp0, cp0), (pl1, tp1, cp1) = outs
    assert max_rel(unplanes(pl1).cpu(), unplanes(pl0).cpu()) < 1e-5
    assert max_rel(unplanes(tp1).cpu(), unplanes(tp0).cpu()) < 1e-5
    assert torch.equal(unplanes(tp1)[:, :m].t(), unplanes(pl1))
    assert max_rel(cp1.cpu(), cp0.cpu()) < 1e-5
    # and against autograd
    uu = u[:, :n].double().cpu().requires_grad_(True)
    (uu[:, :inner] * F.gelu(uu[:, inner:])).backward(dh[:, :inner].double().cpu())
    assert max_rel(unplanes(outs[1][0]).cpu(), uu.grad) < 1e-5


def test_transpose_planes_and_colsum():
    lib = N.lib()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(300, 100, generator=g)
    t = transpose_planes(lib, x.to(DEV), True, 100, 300, 320)
    torch.cuda.synchronize()
    got = unplanes(t).cpu()
    assert max_rel(got[:, :300], x.t()) < 1e-5 and float(got[:, 300:].abs().max()) == 0.0
    out = torch.full((5, 128), 7.0, device=DEV)
    scratch = torch.empty(1 << 16, device=DEV)
    xd = x.to(DEV)
    N.check(lib.wd_colsum(xd.data_ptr(), 100, 300, 100, 60, out.data_ptr(), 128, 0, 0.5, scratch.data_ptr(), scratch.numel(),
                          _st()), "colsum")
    N.check(lib.wd_colsum(xd.data_ptr(), 100, 300, 100, 60, out.data_ptr(), 128, 1, 0.5, scratch.data_ptr(), scratch.numel(),
                          _st()), "colsum+")
    torch.cuda.synchronize()
    ref = x.double().reshape(5, 60, 100).sum(1)
    assert max_rel(out.cpu()[:, :100], ref) < 1e-6 and float((out.cpu()[:, 100:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("B,hw,cs,silu,eps", [(3, 256, (320,), 1, 1e-5), (2, 64, (320, 320), 1, 1e-5), (2, 32, (64,), 0, 1e-6),
                                               (2, 256, (320, 320), 0, 1e-6)])
def test_groupnorm_backward(B, hw, cs, silu, eps):
    lib = N.lib()
    g = torch.Generator().manual_seed(hw + sum(cs))
    ctot = sum(cs)
    xs = [(torch.randn(B * hw, c, generator=g) * 2 + 0.5) for c in cs]
    gamma = torch.randn(ctot, generator=g, dtype=torch.float64, requires_grad=True)
    beta = torch.randn(ctot, generator=g, dtype=torch.float64, requires_grad=True)
    xr = [x.double().requires_grad_(True) for x in xs]
    cat = torch.cat(xr, 1).reshape(B, hw, ctot).permute(0, 2, 1)
    y = F.group_norm(cat, 32, gamma, beta, eps)
    if silu:
        y = F.silu(y)
    dz = torch.randn(B * hw, ctot, generator=g)
    y.permute(0, 2, 1).reshape(B * hw, ctot).backward(dz.double())
    cpg = ctot // 32
    nck = lib.wd_gn_nchunk(hw)
    nb = lib.wd_gn_bwd_nchunk(hw)
    gd, bd, dzd = gamma.detach().float().to(DEV), beta.detach().float().to(DEV), dz.to(DEV)
    scratch = torch.empty(1 << 20, device=DEV)
    dgam = torch.zeros(2, ctot, device=DEV)
    off = 0
    for x, c, xref in zip(xs, cs, xr):
        xd = x.to(DEV)
        part = torch.zeros(B, nck, 32, 2, dtype=torch.float64, device=DEV)
        N.check(lib.wd_gn_stats(xd.data_ptr(), c, B, hw, c, c // 32, part.data_ptr(), _st()), "stats")
        sums = torch.zeros(B, nb, 2, c, device=DEV)
        N.check(lib.wd_gn_bwd_stats(xd.data_ptr(), c, dzd.data_ptr(), ctot, off, B, hw, c, cpg, part.data_ptr(), nck, c // 32,
                                    gd.data_ptr(), bd.data_ptr(), off, eps, silu, sums.data_ptr(), _st()), "bwd stats")
        dx = torch.ones(B * hw, c, device=DEV)
        N.check(lib.wd_gn_bwd_apply(xd.data_ptr(), c, dzd.data_ptr(), ctot, off, B, hw, c, cpg, part.data_ptr(), nck, c // 32,
                                    gd.data_ptr(), bd.data_ptr(), off, eps, silu, sums.data_ptr(), dx.data_ptr(), c, 1, _st()),
                "bwd apply")
        N.check(lib.wd_colsum(sums.data_ptr(), 2 * c, B * nb, 2 * c, B * nb, dgam.data_ptr() + 4 * off, 0, 0, 1.0,
                              scratch.data_ptr(), scratch.numel(), _st()), "param grads")
        torch.cuda.synchronize()
        assert max_rel(dx.cpu() - 1.0, xref.grad) < 3e-5
        # colsum wrote [d beta (c) | d gamma (c)] contiguously at dgam.flat[off : off + 2c]
        flat = dgam.reshape(-1).cpu()
        assert max_rel(flat[off:off + c], beta.grad[off:off + c]) < 3e-5
        assert max_rel(flat[off + c:off + 2 * c], gamma.grad[off:off + c]) < 3e-5
        # the one-pass form: same dx, the same per-channel sums with one chunk per sample
        if lib.wd_gn_bwd_fused_supported(hw, c, cpg):
            sums1 = torch.full((B, 1, 2, c), float("nan"), device=DEV)
            dx1 = torch.ones(B * hw, c, device=DEV)
            N.check(lib.wd_gn_bwd_fused(xd.data_ptr(), c, dzd.data_ptr(), ctot, off, B, hw, c, cpg, part.data_ptr(), nck, c // 32,
                                        gd.data_ptr(), bd.data_ptr(), off, eps, silu, sums1.data_ptr(), dx1.data_ptr(), c, 1, _st()),
                    "bwd fused")
            torch.cuda.synchronize()
            assert max_rel(dx1.cpu() - 1.0, xref.grad) < 3e-5
            assert max_rel(sums1.sum((0, 1)).cpu(), sums.sum((0, 1)).cpu()) < 1e-5
            dx0 = torch.full((B * hw, c), float("nan"), device=DEV)
            N.check(lib.wd_gn_bwd_fused(xd.data_ptr(), c, dzd.data_ptr(), ctot, off, B, hw, c, cpg, part.data_ptr(), nck, c // 32,
                                        gd.data_ptr(), bd.data_ptr(), off, eps, silu, sums1.data_ptr(), dx0.data_ptr(), c, 0, _st()),
                    "bwd fused (assign)")
            torch.cuda.synchronize()
            assert max_rel(dx0.cpu(), xref.grad) < 3e-5
        else:
            assert c % 40 != 0 or 40 % cpg != 0
        off += c
        dgam.zero_()
        if len(cs) > 1:
            break  # (offset bookkeeping of the flat buffer above is only meaningful for the first source)


def test_layernorm_backward():
    lib = N.lib()
    g = torch.Generator().manual_seed(4)
    for rows, c in ((1000, 320), (37, 64)):
        x = (torch.randn(rows, c, generator=g) * 3 + 1)
        xr = x.double().requires_grad_(True)
        ga = torch.randn(c, generator=g, dtype=torch.float64, requires_grad=True)
        be = torch.randn(c, generator=g, dtype=torch.float64, requires_grad=True)
        dy = torch.randn(rows, c, generator=g)
        F.layer_norm(xr, (c,), ga, be, 1e-5).backward(dy.double())
        nblk = lib.wd_layernorm_bwd_nblk(rows)
        xd, dyd, gd = x.to(DEV), dy.to(DEV), ga.detach().float().to(DEV)
        dx = torch.zeros(rows, c, device=DEV)
        colpart = torch.zeros(nblk, 2, c, device=DEV)
        N.check(lib.wd_layernorm_bwd(xd.data_ptr(), c, dyd.data_ptr(), c, rows, c, gd.data_ptr(), 1e-5, dx.data_ptr(), c, 0,
                                     colpart.data_ptr(), _st()), "ln bwd")
        out = torch.zeros(2, c, device=DEV)
        scratch = torch.empty(1 << 18, device=DEV)
        N.check(lib.wd_colsum(colpart.data_ptr(), 2 * c, nblk, 2 * c, nblk, out.data_ptr(), 2 * c, 0, 1.0, scratch.data_ptr(),
                              scratch.numel(), _st()), "colsum")
        torch.cuda.synchronize()
        assert max_rel(dx.cpu(), xr.grad) < 3e-5
        assert max_rel(out.cpu()[0], ga.grad) < 3e-5 and max_rel(out.cpu()[1], be.grad) < 3e-5


@pytest.mark.parametrize("B,H,nq,nk,d,scale", [(3, 4, 256, 10, 80, 80 ** -0.5), (2, 1, 10, 10, 320, 1.0), (2, 4, 70, 7, 16, 0.25)])
def test_attention_backward_small(B, H, nq, nk, d, scale):
    lib = N.lib()
    g = torch.Generator().manual_seed(nq + nk)
    inner = H * d
    q = (torch.randn(B * nq, inner, generator=g) * 0.5)
    k = (torch.randn(B * nk, inner, generator=g) * 0.5)
    v = (torch.randn(B * nk, inner, generator=g) * 0.5)
    do = torch.randn(B * nq, inner, generator=g)
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))

    def heads(t, n):
        return t.reshape(B, n, H, d).permute(0, 2, 1, 3)

    att = torch.softmax(heads(qr, nq) @ heads(kr, nk).transpose(-1, -2) * scale, -1)
    o = (att @ heads(vr, nk)).permute(0, 2, 1, 3).reshape(B * nq, inner)
    o.backward(do.double())
    qd, kd, vd, dod = q.to(DEV), k.to(DEV), v.to(DEV), do.to(DEV)
    dq = torch.zeros(B * nq, inner, device=DEV)
    nwg = lib.wd_attention_bwd_small_nwg(H, nq, nk, d)
    assert nwg > 0
    part = torch.zeros(B, nwg, nk, 2, inner, device=DEV)
    nw = C.c_int(0)
    N.check(lib.wd_attention_bwd_small(qd.data_ptr(), inner, kd.data_ptr(), inner, vd.data_ptr(), inner, dod.data_ptr(), inner,
                                       B, H, nq, nk, d, scale, dq.data_ptr(), inner, part.data_ptr(), C.byref(nw), _st()),
            "attn bwd")
    torch.cuda.synchronize()
    assert nw.value == nwg
    assert max_rel(dq.cpu(), qr.grad) < 3e-5
    dkv = part.sum(1).cpu()
    assert max_rel(dkv[:, :, 0].reshape(B * nk, inner), kr.grad) < 3e-5
    assert max_rel(dkv[:, :, 1].reshape(B * nk, inner), vr.grad) < 3e-5


def test_geglu_silu_pool_embedding_backward():
    lib = N.lib()
    g = torch.Generator().manual_seed(8)
    rows, inner = 300, 128
    u = torch.randn(rows, 2 * inner, generator=g)
    ur = u.double().requires_grad_(True)
    a, gt = ur.chunk(2, -1)
    hh = a * F.gelu(gt)
    dh = torch.randn(rows, inner, generator=g)
    hh.backward(dh.double())
    ud, dhd = u.to(DEV), dh.to(DEV)
    pl = torch.zeros(2, rows, inner, dtype=torch.bfloat16, device=DEV)
    du = torch.zeros(rows, 2 * inner, device=DEV)
    N.check(lib.wd_geglu_fwd(ud.data_ptr(), 2 * inner, rows, inner, pl[0].data_ptr(), pl[1].data_ptr(), inner, _st()), "geglu")
    N.check(lib.wd_geglu_bwd(ud.data_ptr(), 2 * inner, dhd.data_ptr(), inner, rows, inner, du.data_ptr(), 2 * inner, _st()),
            "geglu bwd")
    torch.cuda.synchronize()
    assert max_rel(unplanes(pl).cpu(), hh.detach()) < 2e-5 and max_rel(du.cpu(), ur.grad) < 2e-5
    pre = torch.randn(5000, generator=g)
    pr = pre.double().requires_grad_(True)
    dact = torch.randn(5000, generator=g)
    F.silu(pr).backward(dact.double())
    dpre = torch.zeros(5000, device=DEV)
    pd, dd = pre.to(DEV), dact.to(DEV)
    N.check(lib.wd_silu_bwd(pd.data_ptr(), dd.data_ptr(), 5000, dpre.data_ptr(), _st()), "silu bwd")
    torch.cuda.synchronize()
    assert max_rel(dpre.cpu(), pr.grad) < 2e-5
    # nearest x2 upsampling backward
    small = torch.randn(2, 8, 4, 6, generator=g, dtype=torch.float64, requires_grad=True)  # NCHW
    up = F.interpolate(small, scale_factor=2, mode="nearest")
    dup = torch.randn(up.shape, generator=g)
    up.backward(dup.double())
    dtok = dup.permute(0, 2, 3, 1).contiguous().to(DEV)  # [B][2h][2w][c]
    outp = torch.zeros(2, 4, 6, 8, device=DEV)
    N.check(lib.wd_pool2x2_sum(dtok.data_ptr(), 2, 4, 6, 8, outp.data_ptr(), _st()), "pool")
    torch.cuda.synchronize()
    assert max_rel(outp.cpu().permute(0, 3, 1, 2), small.grad) < 1e-6
    # embedding backward with duplicate ids
    ids = torch.tensor([3, 0, 3, 7, 3, 0], dtype=torch.int64)
    dd2 = torch.randn(6, 40, generator=g)
    table = torch.zeros(11, 40, dtype=torch.float64, requires_grad=True)
    F.embedding(ids, table).backward(dd2.double())
    dt = torch.full((11, 40), 2.0, device=DEV)
    idd, ddd = ids.to(DEV), dd2.to(DEV)
    N.check(lib.wd_embedding_bwd(idd.data_ptr(), 1, 6, ddd.data_ptr(), 40, 11, 40, dt.data_ptr(), 1, _st()), "emb bwd")
    torch.cuda.synchronize()
    assert max_rel(dt.cpu() - 2.0, table.grad) < 1e-6


@pytest.mark.parametrize("B,H,nq,nk,d,scale,self_attn", [(2, 4, 32, 32, 16, 0.25, True), (2, 4, 256, 256, 80, 80 ** -0.5, True),
                                                          (1, 4, 70, 779, 80, 80 ** -0.5, False), (1, 1, 130, 130, 320, 1.0, True),
                                                          (2, 2, 50, 10, 32, 0.2, False)])
def test_attention_backward_generic(B, H, nq, nk, d, scale, self_attn):
    """wd_attention_bwd (any key count) vs autograd; self-attention reads q, k, v out of one [M, 3*inner] buffer and writes
    d(qkv) the same way (the PHOSC UNet's attn1, unetPhosc.py:241)."""
    lib = N.lib()
    g = torch.Generator().manual_seed(nq * 3 + nk)
    inner = H * d
    if self_attn:
        qkv = torch.randn(B * nq, 3 * inner, generator=g) * 0.5
        q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
    else:
        q = torch.randn(B * nq, inner, generator=g) * 0.5
        kv = torch.randn(B * nk, 2 * inner, generator=g) * 0.5
        k, v = kv[:, :inner], kv[:, inner:]
    do = torch.randn(B * nq, inner, generator=g)
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, k, v))

    def heads(t, n):
        return t.reshape(B, n, H, d).permute(0, 2, 1, 3)

    att = torch.softmax(heads(qr, nq) @ heads(kr, nk).transpose(-1, -2) * scale, -1)
    (att @ heads(vr, nk)).permute(0, 2, 1, 3).reshape(B * nq, inner).backward(do.double())
    dod = do.to(DEV)
    nscr = lib.wd_attention_bwd_scratch_floats(B, H, nq, nk)
    scr = torch.empty(nscr, device=DEV)
    if self_attn:
        src = qkv.to(DEV)
        dst = torch.zeros(B * nq, 3 * inner, device=DEV)
        ld = 3 * inner
        N.check(lib.wd_attention_bwd(src.data_ptr(), ld, src.data_ptr() + 4 * inner, ld, src.data_ptr() + 8 * inner, ld,
                                     dod.data_ptr(), inner, B, H, nq, nk, d, scale, dst.data_ptr(), ld,
                                     dst.data_ptr() + 4 * inner, ld, dst.data_ptr() + 8 * inner, ld, scr.data_ptr(), nscr, _st()),
                "attn bwd")
        torch.cuda.synchronize()
        got = dst.cpu()
        dq, dk, dv = got[:, :inner], got[:, inner:2 * inner], got[:, 2 * inner:]
    else:
        qd, kvd = q.contiguous().to(DEV), kv.to(DEV)
        dq = torch.zeros(B * nq, inner, device=DEV)
        dkv = torch.zeros(B * nk, 2 * inner, device=DEV)
        N.check(lib.wd_attention_bwd(qd.data_ptr(), inner, kvd.data_ptr(), 2 * inner, kvd.data_ptr() + 4 * inner, 2 * inner,
                                     dod.data_ptr(), inner, B, H, nq, nk, d, scale, dq.data_ptr(), inner, dkv.data_ptr(),
                                     2 * inner, dkv.data_ptr() + 4 * inner, 2 * inner, scr.data_ptr(), nscr, _st()), "attn bwd")
        torch.cuda.synchronize()
        dq, dk, dv = dq.cpu(), dkv.cpu()[:, :inner], dkv.cpu()[:, inner:]
    assert max_rel(dq, qr.grad) < 3e-5 and max_rel(dk, kr.grad) < 3e-5 and max_rel(dv, vr.grad) < 3e-5
