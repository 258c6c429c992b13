"""Times wd_dw (weight gradient from the row-major planes, csrc/wd_dw.hip) on the layer shapes of the training step at batch 64:
   python tools/dw_bench.py [--nslice N] [--npass 1|3] [--iters 20]
Rates are algorithmic (each multiply-add once); x3 for the MFMA work of the split-bf16 passes."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from worddiffusion_amd import _native as N  # noqa: E402
from worddiffusion_amd.engine import conv_gather_table  # noqa: E402

DEV = "cuda:0"
SHAPES = {  # name: (h, w, c_in, n_out, taps)
    "conv8x32": (8, 32, 320, 320, 9),
    "conv8x32cat": (8, 32, 640, 320, 9),
    "conv4x16": (4, 16, 320, 320, 9),
    "conv4x16cat": (4, 16, 640, 320, 9),
    "lin8x32": (8, 32, 320, 320, 1),
    "skip8x32": (8, 32, 640, 320, 1),
    "ff1": (8, 32, 320, 2560, 1),
    "ff2": (8, 32, 1280, 320, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="all")
    ap.add_argument("--nslice", type=int, default=0)
    ap.add_argument("--npass", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--group", type=int, default=0, help="N > 1: N layers of the shape in one wd_dw_group launch (time per layer)")
    ap.add_argument("--stamps", type=int, default=0, help="1: cycle sums of workgroup 0 (library built with -DWD_DW_STAMPS)")
    a = ap.parse_args()
    lib = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    for name in (list(SHAPES) if a.shape == "all" else a.shape.split(",")):
        h, w, c, n, taps = SHAPES[name]
        hw, m = h * w, a.batch * h * w
        x = torch.randn(2, m, c, device=DEV).to(torch.bfloat16)
        d = torch.randn(2, m, n, device=DEV).to(torch.bfloat16)
        tab = torch.from_numpy(conv_gather_table(h, w, "same")[0]).to(DEV) if taps == 9 else None
        ns = a.nslice or lib.wd_dw_slices(m, n, c, taps)
        ws = torch.empty(ns * n * c * taps, device=DEV)
        grad = torch.empty(n, c * taps, device=DEV)
        g = N.WdDwArgs()
        g.d_hi, g.d_lo, g.x_hi, g.x_lo = d[0].data_ptr(), d[1].data_ptr(), x[0].data_ptr(), x[1].data_ptr()
        g.gather = tab.data_ptr() if tab is not None else None
        g.grad, g.grad_ld, g.ws, g.ws_floats = grad.data_ptr(), c * taps, ws.data_ptr(), ws.numel()
        g.d_ld, g.x_ld, g.ntaps, g.hw_out, g.hw_src = n, c, taps, hw, hw
        g.m, g.n, g.c, g.npass, g.nslice = m, n, c, a.npass, ns
        if a.stamps:
            sb = torch.zeros(16, dtype=torch.int64, device=DEV)
            g.stamps, g.dbg = sb.data_ptr(), 0x100
            N.check(lib.wd_dw(C.byref(g), st), name)
            torch.cuda.synchronize()
            v = sb.cpu().view(2, 2, 4)
            units = (m // 64) // ns
            nh = max(1, units - 2)  # half-steps of each kind that were summed (k = 2 .. U - 3)
            for grp in (0, 1):
                for kind, kn in ((0, "read"), (1, "mfma")):
                    t = v[grp, kind].tolist()
                    print(f"   group {grp} {kn:4s} half-step: issue {t[0] / nh:7.1f}  work {t[1] / nh:7.1f}  dma wait {t[2] / nh:7.1f}  "
                          f"barrier {t[3] / nh:7.1f}   (s_memtime ticks, 100 MHz)")
            g.stamps, g.dbg = None, 0
        if a.group > 1:
            G = a.group
            xs = [torch.randn(2, m, c, device=DEV).to(torch.bfloat16) for _ in range(G)]
            ds = [torch.randn(2, m, n, device=DEV).to(torch.bfloat16) for _ in range(G)]
            gs = [torch.empty(n, c * taps, device=DEV) for _ in range(G)]
            arr = (N.WdDwItem * G)()
            for i in range(G):
                arr[i].d_hi, arr[i].d_lo, arr[i].x_hi, arr[i].x_lo = ds[i][0].data_ptr(), ds[i][1].data_ptr(), xs[i][0].data_ptr(), xs[i][1].data_ptr()
                arr[i].grad, arr[i].grad_ld, arr[i].d_ld, arr[i].x_ld = gs[i].data_ptr(), c * taps, n, c
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
            nsg = a.nslice or lib.wd_dw_group_slices(m, n, c, taps, G)
            wsg = torch.empty(nsg * G * n * c * taps, device=DEV)
            g.ws, g.ws_floats, g.nslice = wsg.data_ptr(), wsg.numel(), nsg
            run = lambda: N.check(lib.wd_dw_group(C.byref(g), C.cast(arr, C.c_void_p), dev.data_ptr(), G, st), name)  # noqa: E731
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters / G
            gf = 2.0 * m * n * c * taps * 1e-9
            print(f"{name:12s} group of {G}: slices={nsg:3d} wgs={(n // 160) * (c // 160) * taps * nsg * G:4d}: {us:8.1f} us per layer  {gf / us * 1e3:6.1f} TF/s algorithmic", flush=True)
            continue
        for _ in range(3):
            N.check(lib.wd_dw(C.byref(g), st), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            N.check(lib.wd_dw(C.byref(g), st), name)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        gf = 2.0 * m * n * c * taps * 1e-9
        tiles = (n // 160) * (c // 160) * taps
        print(f"{name:12s} m={m:6d} n={n:5d} c={c:5d} taps={taps} slices={ns:3d} wgs={tiles * ns:4d}: {us:8.1f} us  {gf / us * 1e3:6.1f} TF/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
