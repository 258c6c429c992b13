"""Timing of wd_xattn_pair (both cross-attentions of a base-model block + norm3 planes) at the headline shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from worddiffusion_amd import _native as N  # noqa: E402

lib = N.lib()
DEV = "cuda:0"
B, hw, c, heads, L = 64, int(os.environ.get("HW", "256")), 320, 4, 10
x = torch.randn(B * hw, c, device=DEV)
g = torch.ones(c, device=DEV)
b = torch.zeros(c, device=DEV)
out = torch.empty(B * hw, c, device=DEV)
pl = torch.zeros(2, B * hw, c, dtype=torch.bfloat16, device=DEV)
mq = [(torch.randn(B, 2, 64, c, device=DEV) * 0.05).to(torch.bfloat16) for _ in range(2)]
mo = [(torch.randn(B, 2, c, 64, device=DEV) * 0.05).to(torch.bfloat16) for _ in range(2)]
st = torch.cuda.current_stream().cuda_stream


def run():
    N.check(lib.wd_xattn_pair(x.data_ptr(), c, B, hw, c, 1e-5, heads, L, g.data_ptr(), b.data_ptr(), mq[0].data_ptr(), mo[0].data_ptr(),
                              b.data_ptr(), g.data_ptr(), b.data_ptr(), mq[1].data_ptr(), mo[1].data_ptr(), b.data_ptr(), out.data_ptr(), c,
                              g.data_ptr(), b.data_ptr(), 1e-5, pl[0].data_ptr(), pl[1].data_ptr(), c, st), "pair")


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    run()
e1.record()
torch.cuda.synchronize()
print(f"wd_xattn_pair B={B} hw={hw} c={c}: {e0.elapsed_time(e1) * 10:.1f} us per launch")
