"""TrainStep iterations at the headline shape (for rocprofv3 --kernel-trace --stats).  STEPS (default 25): the one-time work of
the first step - plan building, ~700 buffer fills, weight uploads - is in the trace as well; divide the per-step kernels by STEPS
and read FillFunctor / copyBuffer as set-up, not as per-step cost."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from worddiffusion_amd import Diffusion  # noqa: E402
from worddiffusion_amd.optim import FusedAdamW  # noqa: E402
from worddiffusion_amd.synthetic import synthetic_inputs  # noqa: E402
from worddiffusion_amd.training import TrainStep  # noqa: E402

B = int(os.environ.get("B", "64"))
steps = int(os.environ.get("STEPS", "25"))
dev = "cuda:0"
model, args = bench.build_model(dev, os.environ.get("PREC", "bf16x3"), "base")
model.train()
ema = copy.deepcopy(model).eval().requires_grad_(False)
opt = FusedAdamW(model.parameters(), lr=1e-4, ema_model=ema)
step = TrainStep(model, Diffusion(noise_steps=1000, img_size=(64, 256), args=args), opt, use_graph=os.environ.get("GRAPH", "1") == "1")
inp = synthetic_inputs(B, seed=7, hw=(8, 32), num_classes=339)
x, c, y = inp["x"].to(dev), inp["context"].to(dev), inp["y"].to(dev)
for _ in range(steps):
    loss = step(x, c, y)
torch.cuda.synchronize()
print("loss", float(loss))
