"""One full Diffusion.sampling() call at the headline configuration through the PRODUCT API (not bench.py's step runner):
64 words x 999 steps, base UNet, synthetic weights -> wall time per call, images/s, finite check."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from worddiffusion_amd import Diffusion  # noqa: E402

dev = "cuda:0"
model, args = bench.build_model(dev, os.environ.get("PREC", "bf16x3"), "base")
diff = Diffusion(noise_steps=1000, img_size=(64, 256), args=args)
B = int(os.environ.get("B", "64"))
words = [("word%d" % i)[:8].replace("0", "a").replace("1", "b").replace("2", "c").replace("3", "d").replace("4", "e")
         .replace("5", "f").replace("6", "g").replace("7", "h").replace("8", "i").replace("9", "j") for i in range(B)]
labels = torch.arange(B) % 339
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = diff.sampling(model, None, B, words, labels, args, seed=5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"call {it}: {dt:.3f} s for {B} images x 999 steps = {B / dt:.2f} images/s ({1e3 * dt / 999:.3f} ms/step incl. setup), "
          f"finite={bool(torch.isfinite(x).all())}, stats={diff.last_stats}")
