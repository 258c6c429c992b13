#!/usr/bin/env python3
"""Micro-benchmark of wd_gemm on the shapes of the UNet (run on the GPU box; used for tuning and PMC runs).

    python tools/gemm_bench.py [--shape conv8x32] [--tile 0] [--npass 3] [--iters 20]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from worddiffusion_amd import _native as N  # noqa: E402
from worddiffusion_amd.engine import conv_gather_table, slab_order, slab_span  # noqa: E402

DEV = "cuda:0"
B = 64
SHAPES = {
    # name: (h, w, cin, cout, kind)   kind: conv3 | lin | conv3cat (640 -> 320 + skip) | geglu
    "conv8x32": (8, 32, 320, 320, "conv3"),
    "conv8x32cat": (8, 32, 640, 320, "conv3"),
    "conv4x16": (4, 16, 320, 320, "conv3"),
    "conv4x16cat": (4, 16, 640, 320, "conv3"),
    "lin8x32": (8, 32, 320, 320, "lin"),
    "ff1": (8, 32, 320, 2560, "geglu"),
    "ff2": (8, 32, 1280, 320, "lin"),
    "lin2560": (8, 32, 320, 2560, "lin"),    # ff1 without the GEGLU epilogue: what that epilogue costs
    "lin4x16": (4, 16, 320, 320, "lin"),
    "temb": (1, 1, 1280, 1280, "lin"),
    "lin_k64": (8, 32, 64, 320, "lin"),      # one stage: the fixed cost of a launch (prologue + epilogue)
    "lin_k128": (8, 32, 128, 320, "lin"),
    "conv_k576": (8, 32, 64, 320, "conv3"),  # nine one-chunk taps
}


def main():
    global B
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="all")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--npass", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--slab", type=int, default=0)
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--ksplit", type=int, default=1)
    ap.add_argument("--tickets", type=int, default=0, help="1: in-launch split-K combine (arrival tickets) instead of the combine launch")
    ap.add_argument("--batch", type=int, default=0, help="batch size (default: the module's B)")
    ap.add_argument("--conv3", type=int, default=0, help="1: row-shared-taps kernel for the 3x3 shapes (w_layout 2)")
    ap.add_argument("--stamps", type=int, default=0, help="1: per-stage cycle stamps of workgroup 0 (s_memtime)")
    ap.add_argument("--wdirect", type=int, default=0, help="1 / 2: fragment-major weights, 128 x 160 / 64 x 320 weights-to-registers kernel (w_layout 3); 3: the 64 x 80 whole-K kernel for 64-position samples (tile 64080); "
                    "the result is compared with the default kernel's first")
    ap.add_argument("--cold", type=int, default=0, help="1: flush caches (1 GiB write) before every launch; "
                    "2: same, then read the weights once (emulates a prefetch) before the launch")
    ap.add_argument("--a32", type=int, default=0, help="1 (with --wdirect 2): src[0] is the fp32 map, GroupNorm + SiLU applied inside "
                    "the GEMM (wd_gemm_args.a32*; 3x3 single-source shapes take the slab kernel wd_gemms_kernel)")
    ap.add_argument("--mnk", default="", help="m,n,k of a plain product (no taps) instead of --shape, e.g. 320,320,16384: the "
                    "weight-gradient GEMMs of the training step (reduction over the tokens)")
    a = ap.parse_args()
    if a.mnk:
        mm, nn, kk = (int(v) for v in a.mnk.split(","))
        SHAPES["mnk"] = (1, mm, kk, nn, "lin")
        a.shape, a.batch = "mnk", 1
    if a.batch:
        B = a.batch
    lib = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    names = list(SHAPES) if a.shape == "all" else a.shape.split(",")
    for name in names:
        h, w, cin, cout, kind = SHAPES[name]
        hw, m = h * w, B * h * w
        ntaps = 9 if kind == "conv3" else 1
        ktot = ntaps * cin
        act = torch.randn(2, m, cin, device=DEV).to(torch.bfloat16)
        wt = (torch.randn(2, cout, ktot, device=DEV) / ktot ** 0.5).to(torch.bfloat16)
        bias = torch.randn(cout, device=DEV)
        tab_np = conv_gather_table(h, w, "same")[0] if ntaps == 9 else None
        tab = torch.from_numpy(tab_np).to(DEV) if ntaps == 9 else None
        if a.slab:
            wt = slab_order(wt, ntaps, cin, 0)
        n_out = cout // 2 if kind == "geglu" else cout
        out = torch.empty(m, n_out, device=DEV)
        g = N.WdGemmArgs()
        s = N.WdSrc()
        s.hi, s.lo = act[0].data_ptr(), act[1].data_ptr()
        s.gather = tab.data_ptr() if tab is not None else None
        s.ld, s.c, s.ntaps, s.hw_src = cin, cin, ntaps, hw
        g.src[0] = s
        g.nsrc, g.npass = 1, a.npass
        g.w_hi, g.w_lo = wt[0].data_ptr(), wt[1].data_ptr()
        g.m, g.n, g.ktot, g.hw_out = m, cout, ktot, hw
        g.bias = bias.data_ptr()
        g.act = N.ACT_GEGLU if kind == "geglu" else 0
        g.out_f32, g.out_ld = out.data_ptr(), n_out
        g.tile = a.tile if kind != "geglu" or a.tile else 0
        g.dbg = a.dbg
        g.ksplit = a.ksplit
        ws = torch.empty(max(8, a.ksplit) * m * cout if a.ksplit != 1 and m * cout * max(8, a.ksplit) < 2 ** 28 else 1, device=DEV)
        if a.ksplit != 1:
            g.ws, g.ws_floats = ws.data_ptr(), ws.numel()
        tk = torch.zeros(8192, dtype=torch.int32, device=DEV)
        if a.tickets:
            g.tickets, g.ntickets = tk.data_ptr(), tk.numel()
            g.dbg = g.dbg | 0x2000
        if a.conv3 and kind == "conv3":
            g.w_layout, g.slab_rows = 2, w
            g.dbg = a.dbg | 0x1000
        if a.wdirect and cout % 160 == 0 and kind != "geglu":
            N.check(lib.wd_gemm(C.byref(g), st), name)   # reference result from the default kernel
            ref = out.clone()
            wf = torch.empty_like(wt)
            N.check(lib.wd_gemm_pack_w(wt[0].data_ptr(), wt[1].data_ptr(), cout, ktot, wf[0].data_ptr(), wf[1].data_ptr(), st), "pack_w")
            g.w_hi, g.w_lo = wf[0].data_ptr(), wf[1].data_ptr()
            g.w_layout, g.slab_rows, g.tile = 3, (w if ntaps == 9 else 0), (64320 if a.wdirect == 2 else 64080 if a.wdirect == 3 else 128160)
            if a.a32:
                x32 = torch.randn(m, cin, device=DEV)
                nchunk = lib.wd_gn_nchunk(hw)
                part = torch.zeros(B, nchunk, 32, 2, dtype=torch.float64, device=DEV)
                N.check(lib.wd_gn_stats(x32.data_ptr(), cin, B, hw, cin, cin // 32, part.data_ptr(), st), "stats")
                gam, bet = torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV)
                g.a32, g.a32_ld, g.a32_part, g.a32_nchunk, g.a32_pcpg, g.a32_cpg = x32.data_ptr(), cin, part.data_ptr(), nchunk, cin // 32, cin // 32
                g.a32_gamma, g.a32_beta, g.a32_eps, g.a32_silu = gam.data_ptr(), bet.data_ptr(), 1e-5, 1
                keep_a32 = (x32, part, gam, bet)
            if a.ksplit != 1:
                g.ksplit = a.ksplit
            out.zero_()
            N.check(lib.wd_gemm(C.byref(g), st), name + " (w-direct)")
            torch.cuda.synchronize()
            print(f"{name}: w-direct vs default kernel max |diff| {float((out - ref).abs().max()):.3e}  (max |ref| {float(ref.abs().max()):.3f})", flush=True)
        if a.stamps and a.wdirect == 3:
            sb = torch.zeros(64, dtype=torch.int64, device=DEV)
            g.ws, g.ws_floats, g.dbg = sb.data_ptr(), 0, a.dbg | 0x100
            N.check(lib.wd_gemm(C.byref(g), st), name)
            torch.cuda.synchronize()
            v = sb.cpu().view(8, 8)
            for wv in range(8):
                t = v[wv]
                print(f"   wave {wv}: fill+barrier {int(t[1] - t[0])}  loop {int(t[4] - t[1])}  wait-for-others {int(t[5] - t[4])}  "
                      f"reduce {int(t[6] - t[5])}  epilogue {int(t[7] - t[6])}   total {int(t[7] - t[0])} cycles")
            g.dbg, g.ws, g.ws_floats = a.dbg, None, 0
        elif a.stamps:
            nk = (ktot // 64 + 1) & ~1 if a.wdirect else ktot // 64
            sb = torch.zeros(8 * nk * 4, dtype=torch.int64, device=DEV)
            g.ws, g.ws_floats, g.dbg = sb.data_ptr(), 0, a.dbg | 0x100
            N.check(lib.wd_gemm(C.byref(g), st), name)
            torch.cuda.synchronize()
            v = sb.cpu().view(8, nk, 4)
            for wv in (0, 4):
                w = v[wv]
                wait = (w[:, 1] - w[:, 0]).float()       # barrier + vmcnt wait
                rd = (w[:, 2] - w[:, 1]).float()         # address prep + ds_read of the fragments (until they landed)
                mf = (w[:, 3] - w[:, 2]).float()         # MFMA groups with the DMA issue interleaved
                tot = (w[1:, 0] - w[:-1, 0]).float()
                if a.stamps > 1:
                    print("   stage durations:", [int(x) for x in tot[3:21].tolist()])
                    print("   wait:", [int(x) for x in wait[3:21].tolist()])
                    print("   mfma:", [int(x) for x in mf[3:21].tolist()])
                print(f"{name} wave {wv}: per stage (s_memtime ticks, 100 MHz?) wait {wait[2:-1].mean():.1f}  frag-read {rd[2:-1].mean():.1f}  "
                      f"mfma+dma {mf[2:-1].mean():.1f}  stage {tot[2:-1].mean():.1f}   whole loop {int(w[-1, 3] - w[0, 0])}")
            g.dbg = a.dbg
            g.ws, g.ws_floats = (ws.data_ptr(), ws.numel()) if a.ksplit != 1 else (None, 0)
        if a.slab:
            g.w_layout, g.slab_rows = 1, slab_span(tab_np, hw, hw, m)
        if kind == "geglu" and not g.tile:
            g.tile = 128160
        for _ in range(3):
            N.check(lib.wd_gemm(C.byref(g), st), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if a.cold:
            junk = torch.empty(256 * 1024 * 1024, device=DEV)
            tot = 0.0
            for _ in range(a.iters):
                junk.fill_(1.0)
                if a.cold == 2:
                    wt.sum()
                if a.cold == 3:
                    wt.sum(); act.sum()
                e0.record()
                lib.wd_gemm(C.byref(g), st)
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            us = 1e3 * tot / a.iters
        else:
            e0.record()
            for _ in range(a.iters):
                lib.wd_gemm(C.byref(g), st)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / a.iters
        fl = 2.0 * m * cout * ktot
        print(f"{name:12s} m={m:6d} n={cout:5d} k={ktot:5d} tile={a.tile:6d} npass={a.npass} slab={a.slab} ksplit={a.ksplit} cold={a.cold}: {us:8.1f} us  "
              f"{fl / us / 1e6:7.1f} TF/s algorithmic ({a.npass * fl / us / 1e6:7.1f} MFMA)", flush=True)


if __name__ == "__main__":
    main()
