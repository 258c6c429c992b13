"""Shared helpers for the tests: configs of the golden vectors and state_dict construction."""
import os
import types

import numpy as np
import torch

from worddiffusion_amd.synthetic import synthetic_tensor

FULL = dict(image_size=(64, 256), in_channels=4, model_channels=320, out_channels=4, num_res_blocks=1,
            attention_resolutions=(1, 1), channel_mult=(1, 1), num_heads=4, num_classes=339,
            context_dim=320, vocab_size=53, max_seq_len=10)
SMALL = dict(image_size=(32, 64), in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1,
             attention_resolutions=(1, 1), channel_mult=(1, 1), num_heads=4, num_classes=11,
             context_dim=64, vocab_size=53, max_seq_len=10)
DEEP = dict(image_size=(32, 64), in_channels=4, model_channels=32, out_channels=4, num_res_blocks=2,
            attention_resolutions=(2,), channel_mult=(1, 2, 2), num_heads=2, num_classes=5,
            context_dim=64, vocab_size=53, max_seq_len=10)

# tag -> (cfg, variant, phosc_on)
FWD_CASES = {
    "fwd_base_small": (SMALL, "base", False),
    "fwd_phosc_small_nophosc": (SMALL, "phosc", False),
    "fwd_phosc_small": (SMALL, "phosc", True),
    "fwd_base_deep": (DEEP, "base", False),
    "fwd_phosc_deep": (DEEP, "phosc", True),
    "fwd_base_full": (FULL, "base", False),
    "fwd_phosc_full_nophosc": (FULL, "phosc", False),
    "fwd_phosc_full": (FULL, "phosc", True),
}


def load_golden(golden_dir, tag):
    return np.load(os.path.join(golden_dir, tag + ".npz"), allow_pickle=False)


def golden_state_dict(g, seed=None):
    """Rebuild the state_dict the reference model was filled with from (keys, shapes, seed)."""
    seed = int(g["seed"]) if seed is None else seed
    sd = {}
    for k, s in zip(g["keys"], g["shapes"]):
        shape = tuple(int(v) for v in str(s).split(",")) if str(s) else ()
        sd[str(k)] = torch.from_numpy(synthetic_tensor(str(k), shape, seed))
    return sd


def make_args(**kw):
    base = dict(device="cpu", interpolation=False, charLevelEmb=0, charImages=0, attentionMaps=0,
                ocrTraining=0, imgConditioned=0, wrdChrWrStyl=0, phosc=0, phos=0, latent=True)
    base.update(kw)
    return types.SimpleNamespace(**base)


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def join_all(procs, timeout):
    """Joins every spawned rank within ``timeout`` seconds in total; a rank still alive after that is killed (so a hung child
    never keeps the GPU context or the port for the tests that follow) and the exit codes are asserted afterwards."""
    import time
    deadline = time.monotonic() + timeout
    try:
        for p in procs:
            p.join(max(0.0, deadline - time.monotonic()))
    finally:
        hung = [p for p in procs if p.is_alive()]
        for p in hung:
            p.kill()
        for p in hung:
            p.join()
    assert not hung, f"{len(hung)} rank(s) did not finish within {timeout} s and were killed"
    codes = [p.exitcode for p in procs]
    assert all(c == 0 for c in codes), f"rank exit codes {codes}"
