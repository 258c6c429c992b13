"""wd_repack_multi alone: the parameters of the base UNet -> every packed operand of the TRAINING engine (forward planes, data-gradient
planes, fragment-major images, vectors), timed over REP launches.   python tools/repack_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from worddiffusion_amd import _native as N  # noqa: E402

REP = int(os.environ.get("REP", "50"))
dev = torch.device("cuda:0")
model, args = bench.build_model(dev, "bf16x3", "base")
for which in ("train_engine", "engine"):
    model.train(which == "train_engine")
    eng = getattr(model, which)
    eng.refresh_weights(force=True)
    _, table, chunks, n = eng._pack[:4]
    st = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        N.check(eng.lib.wd_repack_multi(table.data_ptr(), n, chunks, st), "wd_repack_multi")
    e1.record()
    torch.cuda.synchronize()
    nbytes = sum(p.numel() for p in model.parameters()) * 4
    print(f"{which}: {n} entries, {chunks} chunks, {e0.elapsed_time(e1) * 1e3 / REP:.1f} us per launch ({nbytes / 1e6:.1f} MB of parameters)")
