"""The training leg of bench.py on its own (configs[2] shape: batch 64, base UNet): ms per step in split-bf16 and single-pass bf16.
   python tools/train_time.py [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from worddiffusion_amd import dist as wdist  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
out = bench.train_leg("cuda:0", "bf16x3", int(os.environ.get("B", "64")), steps, 5, 0, 1, torch.cuda.synchronize, wdist)
print(json.dumps({k: out[k] for k in ("ms_per_step", "images_per_sec", "loss_finite", "bf16_single_pass") if k in out}))
kc = out.get("kernel_classes")
if kc:
    print(json.dumps({k: v for k, v in kc.items() if v.get("launches_per_step")}))
