"""200 TrainStep iterations on one fixed synthetic batch at the headline shape, both precision modes: the loss must fall (the
model overfits the batch) and the EMA copy must stay finite."""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from worddiffusion_amd import Diffusion
from worddiffusion_amd.optim import FusedAdamW
from worddiffusion_amd.synthetic import synthetic_inputs
from worddiffusion_amd.training import TrainStep
dev = "cuda:0"
for prec in ("bf16x3", "bf16"):
    model, args = bench.build_model(dev, prec, "base")
    model.train()
    ema = copy.deepcopy(model).eval().requires_grad_(False)
    opt = FusedAdamW(model.parameters(), lr=1e-4, ema_model=ema)
    step = TrainStep(model, Diffusion(noise_steps=1000, img_size=(64, 256), args=args), opt, seed=3)
    inp = synthetic_inputs(64, seed=7, hw=(8, 32), num_classes=339)
    x, c, y = inp["x"].to(dev), inp["context"].to(dev), inp["y"].to(dev)
    ls = []
    for i in range(200):
        loss = step(x, c, y)
        if i % 25 == 0 or i == 199:
            ls.append(round(float(loss), 4))
    print(prec, ls, "ema finite", all(torch.isfinite(p).all().item() for p in ema.parameters()))
