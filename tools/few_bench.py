"""Micro-benchmark of wd_gn_conv3x3_few (GroupNorm + SiLU + 3x3 convolution to 4 channels + NCHW) at the headline shape."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from worddiffusion_amd import _native as N
lib = N.lib(); DEV = "cuda:0"
B, h, w, c, oc = 64, 8, 32, 320, 4
st = torch.cuda.current_stream().cuda_stream
tok = torch.randn(B * h * w, c, device=DEV)
nchunk = lib.wd_gn_nchunk(h * w)
part = torch.zeros(B, nchunk, 32, 2, dtype=torch.float64, device=DEV)
N.check(lib.wd_gn_stats(tok.data_ptr(), c, B, h * w, c, c // 32, part.data_ptr(), st), "stats")
gam, bet = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
wt, bias = torch.randn(oc, c, 3, 3, device=DEV) * 0.02, torch.zeros(oc, device=DEV)
out = torch.empty(B, oc, h, w, device=DEV)
def run():
    N.check(lib.wd_gn_conv3x3_few(tok.data_ptr(), c, B, h, w, c, c // 32, part.data_ptr(), nchunk, c // 32, gam.data_ptr(), bet.data_ptr(),
                                  1e-5, 1, wt.data_ptr(), bias.data_ptr(), oc, out.data_ptr(), st), "k")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"gn_conv3x3_few B={B} {h}x{w} c={c}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us")
