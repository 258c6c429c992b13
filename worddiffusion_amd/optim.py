"""Optimiser side of the reference's training step (``train.py:289-294``): ``mse_loss`` and ``FusedAdamW`` - one HIP
launch updates every parameter (and the EMA copy) instead of the reference's ~260 x 3 elementwise launches per step for
``AdamW.step`` + ``EMA.step_ema``.  Gradients come from the caller (``param.grad``); the backward kernels of the UNet are
the next row of the plan."""
from __future__ import annotations

import struct
from typing import Optional

import torch

from . import _native as N


def mse_loss(pred: torch.Tensor, target: torch.Tensor, want_grad: bool = True):
    """nn.MSELoss()(target, pred) (train.py:289) -> (loss [1] device tensor, d loss / d pred or None)."""
    if not pred.is_cuda:
        raise N.NativeError("mse_loss runs on the GPU only (no CPU fallback)")
    lib = N.lib()
    pred, target = pred.contiguous().float(), target.contiguous().float()
    n = pred.numel()
    grad = torch.empty_like(pred) if want_grad else None
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    scratch = torch.empty(1024, dtype=torch.float64, device=pred.device)
    st = torch.cuda.current_stream(pred.device).cuda_stream
    N.check(lib.wd_mse_loss(pred.data_ptr(), target.data_ptr(), n, grad.data_ptr() if want_grad else None, loss.data_ptr(),
                            scratch.data_ptr(), 1024, st), "wd_mse_loss")
    return loss, grad


class FusedAdamW:
    """``optim.AdamW(model.parameters(), lr=1e-4)`` (train.py:405) + ``EMA(0.995).step_ema`` (train.py:294,410) in one
    multi-tensor launch.  ``ema_model`` is optional; with it ``step()`` reproduces ``ema.step_ema(ema_model, model)``
    (plain copy for the first ``step_start_ema`` steps, then the moving average)."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ema_model=None, ema_beta=0.995,
                 step_start_ema=2000):
        self.params = [p for p in params]
        if not self.params or not self.params[0].is_cuda:
            raise N.NativeError("FusedAdamW needs CUDA parameters (no CPU fallback)")
        # one group in torch.optim's layout: the source of truth for the hyper-parameters (LR schedulers write ``lr`` here)
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                                  params=self.params)]
        self.ema_params = list(ema_model.parameters()) if ema_model is not None else None
        if self.ema_params is not None and len(self.ema_params) != len(self.params):
            raise ValueError("ema_model does not match the parameter list")
        self.ema_beta, self.step_start_ema = ema_beta, step_start_ema
        self.step_count = 0
        self.exp_avg = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self._table = None
        self._grads = None
        self._grad_ptrs = None
        self._has_state = [False] * len(self.params)  # parameters that have been stepped (torch keeps state only for those)

    lr = property(lambda self: self.param_groups[0]["lr"])
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    # ---- checkpointing: ``torch.save(optimizer.state_dict(), "models/optim.pt")`` (train.py:316) / resume -------------------
    def state_dict(self):
        """``torch.optim.AdamW.state_dict()`` layout - ``state[i] = {step, exp_avg, exp_avg_sq}`` for every parameter that has
        received a gradient, ``param_groups`` with parameter indices - so the file loads into either optimiser.  The extra
        ``wdiff`` entry (ignored by torch) carries the step counter that also positions the EMA warm-up."""
        state = {}
        if self.step_count > 0:
            for i, p in enumerate(self.params):
                if not self._has_state[i]:
                    continue  # never stepped (no gradient): torch.optim.AdamW holds no state for it either
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=self.exp_avg[i].clone(),
                                exp_avg_sq=self.exp_avg_sq[i].clone())
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return dict(state=state, param_groups=[group],
                    wdiff=dict(step_count=self.step_count, ema_beta=self.ema_beta, step_start_ema=self.step_start_ema))

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        for k, v in groups[0].items():
            if k != "params":
                self.param_groups[0][k] = tuple(v) if k == "betas" else v
        steps = set()
        with torch.no_grad():
            for i, p in enumerate(self.params):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                self._has_state[i] = st is not None
                if st is None:
                    self.exp_avg[i].zero_()
                    self.exp_avg_sq[i].zero_()
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {i} has shape {tuple(st['exp_avg'].shape)}, parameter {tuple(p.shape)}")
                self.exp_avg[i].copy_(st["exp_avg"])
                self.exp_avg_sq[i].copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): one fused launch uses one bias correction")
        extra = sd.get("wdiff", {})
        self.step_count = int(extra.get("step_count", steps.pop() if steps else 0))
        self.ema_beta = extra.get("ema_beta", self.ema_beta)
        self.step_start_ema = extra.get("step_start_ema", self.step_start_ema)

    def _build_table(self):
        lib = N.lib()
        chunk = lib.wd_adamw_chunk()
        assert lib.wd_adamw_table_entry_bytes() == 56
        recs, c0 = [], 0
        self._grads = []
        for i, p in enumerate(self.params):
            g = p.grad.contiguous() if p.grad is not None else None  # None: torch.optim.AdamW skips the parameter
            self._grads.append(g)
            ema = self.ema_params[i].data_ptr() if self.ema_params is not None else 0
            recs.append(struct.pack("<QQQQQqq", p.data_ptr(), g.data_ptr() if g is not None else 0, self.exp_avg[i].data_ptr(),
                                    self.exp_avg_sq[i].data_ptr(), ema, p.numel(), c0))
            c0 += (p.numel() + chunk - 1) // chunk
        raw = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8)
        self._table = raw.to(self.params[0].device)
        self._chunks = c0
        self._grad_ptrs = [g.data_ptr() if g is not None else 0 for g in self._grads]
        self._has_state = [h or g is not None for h, g in zip(self._has_state, self._grads)]

    def step(self):
        lib = N.lib()
        cur = [p.grad.data_ptr() if p.grad is not None else 0 for p in self.params]
        if self._table is None or cur != self._grad_ptrs:
            self._build_table()
        self.step_count += 1
        if self.ema_params is None:
            mode = 0
        else:
            mode = 1 if (self.step_count - 1) < self.step_start_ema else 2
        st = torch.cuda.current_stream(self.params[0].device).cuda_stream
        N.check(lib.wd_adamw_multi(self._table.data_ptr(), len(self.params), self._chunks, float(self.lr),
                                   float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                                   self.step_count, mode, float(self.ema_beta), st), "wd_adamw_multi")
        # the kernel changed the parameters (and the EMA copy) behind autograd's back: the engines that packed operands from
        # them must repack before their next use
        from .engine import note_native_write
        note_native_write()
