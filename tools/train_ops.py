"""Per-launch table of one TRAINING step at the configs[2] shape (B = 64, base UNet): every op of the forward and backward launch
lists timed on its own (hipEvent pair around REP back-to-back launches, after warm passes), summed per kind at the end.  The ops run
eagerly and alone here, so their sum differs a little from the replayed graph's step time."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from worddiffusion_amd import _native as N  # noqa: E402
from worddiffusion_amd.synthetic import synthetic_inputs  # noqa: E402

B = int(os.environ.get("B", "64"))
REP = int(os.environ.get("REP", "10"))
dev = torch.device("cuda:0")
model, args = bench.build_model(dev, os.environ.get("PREC", "bf16x3"), "base")
model.train()
eng = model.train_engine
inp = synthetic_inputs(B, seed=7, hw=(8, 32), num_classes=339)
x, c, y = inp["x"].to(dev), inp["context"].to(dev), inp["y"].to(dev)
t = torch.randint(1, 1000, (B,), device=dev)
stream = torch.cuda.Stream(dev)
st = stream.cuda_stream
with torch.cuda.stream(stream):
    for _ in range(2):
        out = eng.forward_train(x, t, c, y)
        eng.backward(torch.randn_like(out), assign=False)
    stream.synchronize()
    P = eng._live
    rows = []
    for phase, ops in (("fwd", list(P.cond) + list(getattr(P, "film", [])) + list(P.step)), ("bwd", list(P.bwd))):
        for fn, a, what in ops:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(REP):
                N.check(fn(*a, st), what)
            e1.record(stream)
            stream.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / REP
            info = ""
            obj = getattr(a[0], "_obj", None) if a else None
            if isinstance(obj, N.WdGemmArgs):
                gf = 2.0 * obj.m * obj.n * obj.ktot * 1e-9
                info = f"m={obj.m:6d} n={obj.n:5d} k={obj.ktot:6d} ks={obj.ksplit:3d} {gf:7.2f} GF {gf / us * 1e3:6.1f} TF/s"
            rows.append((phase, what, getattr(fn, "__name__", str(fn)), us, info))
kinds = collections.OrderedDict()
for phase, what, name, us, info in rows:
    print(f"{phase} {what[:52]:52s} {name[:20]:20s} {us:8.1f} us  {info}")
    tag = what.split(":")
    kind = name
    if name == "wd_gemm":
        kind = "wd_gemm:" + ("dW" if any(p.startswith("dW") for p in tag) else "dX" if any(p.startswith("dX") for p in tag) else phase)
    k = kinds.setdefault((phase, kind), [0, 0.0])
    k[0] += 1
    k[1] += us
tot = sum(r[3] for r in rows)
print()
for (phase, kind), (n, us) in kinds.items():
    print(f"{phase} {kind:32s} {n:4d} launches {us:9.1f} us  {100 * us / tot:5.1f} %")
print(f"{len(rows)} launches, sum {tot:.1f} us")
