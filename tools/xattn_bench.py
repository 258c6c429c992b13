"""Timing of wd_xattn_fused at the headline shape (B=64, 8x32 tokens, 320 channels, 4 heads, 10 keys)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from worddiffusion_amd import _native as N  # noqa: E402

lib = N.lib()
DEV = "cuda:0"
B, hw, c, heads, L = 64, int(os.environ.get("HW", "256")), 320, 4, 10
x = torch.randn(B * hw, c, device=DEV)
mq = torch.randn(B, heads * L, c, device=DEV) * 0.05
mo = torch.randn(B, heads * L, c, device=DEV) * 0.05
g = torch.ones(c, device=DEV)
b = torch.zeros(c, device=DEV)
out = torch.empty(B * hw, c, device=DEV)
pl = torch.zeros(2, B * hw, c, dtype=torch.bfloat16, device=DEV)
mq_pl = (torch.randn(B, 2, 64, c, device=DEV) * 0.05).to(torch.bfloat16)
mot_pl = (torch.randn(B, 2, c, 64, device=DEV) * 0.05).to(torch.bfloat16)
MFMA = os.environ.get("MFMA", "1") == "1"
st = torch.cuda.current_stream().cuda_stream


def run(n_ln):
    N.check(lib.wd_xattn_fused(x.data_ptr(), c, B, hw, c, g.data_ptr(), b.data_ptr(), 1e-5, mq.data_ptr(), mo.data_ptr(), heads, L,
                               b.data_ptr(), out.data_ptr(), c, g.data_ptr() if n_ln else None, b.data_ptr() if n_ln else None, 1e-5,
                               pl[0].data_ptr() if n_ln else None, pl[1].data_ptr() if n_ln else None, c,
                               mq_pl.data_ptr() if MFMA else None, mot_pl.data_ptr() if MFMA else None, st), "fused")


for n_ln in (0, 1):
    for _ in range(3):
        run(n_ln)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        run(n_ln)
    torch.cuda.synchronize()
    print(f"dbg={os.environ.get('WDIFF_XATTN_DBG', '0')} next_ln={n_ln}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us")
