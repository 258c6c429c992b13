"""AutoencoderKL decode of one batch at the reference's latent size (for rocprofv3 --kernel-trace --stats and timing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from worddiffusion_amd.synthetic import fill_module_  # noqa: E402
from worddiffusion_amd.vae import AutoencoderKL  # noqa: E402

B = int(os.environ.get("B", "64"))
dev = "cuda:0"
vae = AutoencoderKL()
fill_module_(vae, 0)
vae = vae.to(dev).eval()
vae.set_precision(os.environ.get("PREC", "bf16x3"))
z = torch.randn(B, 4, 8, 32, device=dev) / 0.18215
for _ in range(2):
    img = vae.decode(z).sample
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    img = vae.decode(z).sample
e1.record()
torch.cuda.synchronize()
print(f"vae decode B={B}: {e0.elapsed_time(e1) / 3:.2f} ms  finite={bool(torch.isfinite(img).all())}")
