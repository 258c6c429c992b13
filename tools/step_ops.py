"""Per-launch table of one denoising step at the headline shape (B = 64, base UNet): every op of the plan's step list timed on its own
(hipEvent pair around 20 back-to-back launches, after warm passes of the whole step), with the algorithmic rate of the GEMMs.
Launches run eagerly and alone here (warm L2, no neighbours), so their sum differs a little from the graph's step time."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from worddiffusion_amd import _native as N  # noqa: E402

dev = torch.device("cuda:0")
batch = int(os.environ.get("BATCH", "64"))
variant = os.environ.get("VARIANT", "base")  # "phosc": UNetModelPhosc with the 769-int PHOSC vector
model, args = B.build_model(dev, "bf16x3", variant)
run = B.StepRunner(model, args, dev, batch, 0, 0, phosc_len=769 if variant == "phosc" else 0)
P, st = run.P, run.stream.cuda_stream
REP = 20
rows = []
with torch.cuda.stream(run.stream):
    for _ in range(3):
        P.run_step(st)
    run.stream.synchronize()
    for fn, a, what in P.step:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(run.stream)
        for _ in range(REP):
            N.check(fn(*a, st), what)
        e1.record(run.stream)
        run.stream.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / REP
        info = ""
        obj = getattr(a[0], "_obj", None) if a else None
        if isinstance(obj, N.WdGemmArgs):
            gf = 2.0 * obj.m * obj.n * obj.ktot * 1e-9
            info = f"m={obj.m:6d} n={obj.n:5d} k={obj.ktot:5d} {gf:7.2f} GF {gf / us * 1e3:6.1f} TF/s"
        if isinstance(obj, N.WdFfArgs):
            gf = 2.0 * obj.m * (3.0 * obj.inner * obj.c + (obj.c * obj.c if obj.w3_hi else 0)) * 1e-9
            info = f"m={obj.m:6d} c={obj.c:5d} h={obj.inner:5d} {gf:7.2f} GF {gf / us * 1e3:6.1f} TF/s"
        rows.append((what, getattr(fn, "__name__", str(fn)), us, info))
tot = sum(r[2] for r in rows)
for what, name, us, info in rows:
    print(f"{what[:46]:46s} {name[:22]:22s} {us:8.1f} us  {info}")
print(f"{len(rows)} launches, sum {tot:.1f} us")
