#!/usr/bin/env python3
"""Micro-benchmark of wd_attention_bwd_small at the training shape (B=64, 4 heads x 80, 256 queries, 10 keys)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from worddiffusion_amd import _native as N  # noqa: E402

DEV = "cuda:0"
B, H, nq, nk, d = int(os.environ.get("B", "64")), 4, int(os.environ.get("NQ", "256")), 10, 80
inner = H * d
lib = N.lib()
st = torch.cuda.current_stream().cuda_stream
q = torch.randn(B * nq, inner, device=DEV)
k = torch.randn(B * nk, inner, device=DEV)
v = torch.randn(B * nk, inner, device=DEV)
do = torch.randn(B * nq, inner, device=DEV)
dq = torch.zeros(B * nq, inner, device=DEV)
nwg = lib.wd_attention_bwd_small_nwg(H, nq, nk, d)
part = torch.zeros(B, nwg, nk, 2, inner, device=DEV)
nw = C.c_int(0)


def run():
    N.check(lib.wd_attention_bwd_small(q.data_ptr(), inner, k.data_ptr(), inner, v.data_ptr(), inner, do.data_ptr(), inner, B, H, nq,
                                       nk, d, d ** -0.5, dq.data_ptr(), inner, part.data_ptr(), C.byref(nw), st), "attn bwd")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f"attn_bwd_small B={B} nq={nq}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us  nwg={nwg}")
