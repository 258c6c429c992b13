#!/usr/bin/env python3
"""Micro-benchmark of wd_ff_fused at the 8x32 level of the headline batch (run on the GPU box).

    python tools/ff_bench.py [--proj 1] [--iters 30]        WDIFF_LIB=<other .so> for A/B builds of the kernel
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from worddiffusion_amd import _native as N  # noqa: E402
from worddiffusion_amd.engine import geglu_interleave  # noqa: E402

DEV = "cuda:0"


def planes(x):
    hi = x.to(torch.bfloat16)
    return torch.stack([hi, (x - hi.float()).to(torch.bfloat16)], 0).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proj", type=int, default=1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--m", type=int, default=16384)
    a = ap.parse_args()
    lib = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    m, c, inner = a.m, 320, 1280
    g = torch.Generator().manual_seed(0)
    x = torch.randn(m, c, generator=g).to(DEV)
    w1 = (torch.randn(2 * inner, c, generator=g) / c ** 0.5).to(DEV)
    w2 = (torch.randn(c, inner, generator=g) / inner ** 0.5).to(DEV)
    w3 = (torch.randn(c, c, generator=g) / c ** 0.5).to(DEV)
    b1, b2, b3 = torch.randn(2 * inner, device=DEV), torch.randn(c, device=DEV), torch.randn(c, device=DEV)
    res, res3 = torch.randn(m, c, device=DEV), torch.randn(m, c, device=DEV)

    def pack(w):
        wp = planes(w)
        wf = torch.empty_like(wp)
        N.check(lib.wd_gemm_pack_w(wp[0].data_ptr(), wp[1].data_ptr(), wp.shape[1], wp.shape[2], wf[0].data_ptr(), wf[1].data_ptr(), st), "pack")
        return wf

    w1f, w2f, w3f = pack(geglu_interleave(w1, 16)), pack(w2), pack(w3)
    b1i = geglu_interleave(b1, 16).contiguous()
    xp = planes(x)
    out = torch.empty(m, c, device=DEV)
    f = N.WdFfArgs()
    f.x_hi, f.x_lo, f.x_ld = xp[0].data_ptr(), xp[1].data_ptr(), c
    f.m, f.c, f.inner = m, c, inner
    f.w1_hi, f.w1_lo, f.b1 = w1f[0].data_ptr(), w1f[1].data_ptr(), b1i.data_ptr()
    f.w2_hi, f.w2_lo, f.b2 = w2f[0].data_ptr(), w2f[1].data_ptr(), b2.data_ptr()
    f.resid, f.resid_ld = res.data_ptr(), c
    f.out_f32, f.out_ld = out.data_ptr(), c
    if a.proj:
        f.w3_hi, f.w3_lo, f.b3 = w3f[0].data_ptr(), w3f[1].data_ptr(), b3.data_ptr()
        f.resid3, f.resid3_ld = res3.data_ptr(), c
    f.hw_out, f.npass = 1, 3
    for _ in range(3):
        N.check(lib.wd_ff_fused(C.byref(f), st), "wd_ff_fused")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        lib.wd_ff_fused(C.byref(f), st)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / a.iters
    fl = 2.0 * m * (3.0 * inner * c + (c * c if a.proj else 0))
    print(f"wd_ff_fused m={m} proj={a.proj}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s algorithmic ({3 * fl / us / 1e6:6.1f} MFMA)   lib {N.LIB_PATH}")


if __name__ == "__main__":
    main()
