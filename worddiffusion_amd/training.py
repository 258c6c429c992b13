"""The body of the reference's training loop (``train.py:277-294``) as a handful of device launches:

    t = diffusion.sample_timesteps(B); x_t, noise = diffusion.noise_images(latents, t)        train.py:281-282
    predicted_noise = model(x_t, ..., timesteps=t, context=text_features, y=labels)            train.py:287
    loss = mse(noise, predicted_noise); optimizer.zero_grad(); loss.backward()                 train.py:289-291
    optimizer.step(); ema.step_ema(ema_model, model)                                           train.py:293-294

``TrainStep`` runs noise_images + forward + loss + backward as ONE hipGraph, then one multi-tensor AdamW+EMA launch and one
weight-repack launch.  Data-parallel (world > 1): the backward list is cut into a few segments (``WDIFF_GRAD_BUCKETS``, default 3)
after each of which a PREFIX of the flat gradient arena is final - the gradients of the layers nearest the output, which the
backward pass finishes first; every segment is its own hipGraph and the all-reduce of its prefix (RCCL over xGMI) runs on a
second stream while the next segment computes, so only the last bucket's collective is exposed.  The same arithmetic is
reachable piecewise through the reference's own call surface (``model(...)``, ``loss.backward()``, ``FusedAdamW.step``);
this class only removes the per-step host work.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _native as N
from .optim import FusedAdamW


class TrainStep:
    def __init__(self, model, diffusion, optimizer: FusedAdamW, seed: int = 0, use_graph: bool = True,
                 process_group=None, bucketed: Optional[bool] = None):
        """``bucketed``: run the backward pass in segments with the gradient all-reduce of each segment's arena prefix
        overlapped with the next segment (default: whenever world > 1; ``True`` at world 1 only cuts the graph - tests)."""
        self.model, self.diffusion, self.opt = model, diffusion, optimizer
        self.seed, self.use_graph = seed, use_graph
        self._bucketed_arg = bucketed
        self.lib = N.lib()
        self.eng = model.train_engine
        self.step_index = 0
        self._key = None
        self._graph = None
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        # data-parallel ranks must not draw the same (eps, t): the noise rows are keyed by the GLOBAL row of the step
        # (``(step * world + rank) * B + b``, so world ranks of batch B draw what one process of batch world*B would), and the
        # timesteps come from a per-rank host generator (a single process keeps the reference's global host RNG, train.py:281)
        self._tgen = torch.Generator().manual_seed((int(seed) * 1000003 + self.rank) & 0x7FFFFFFFFFFFFFFF) if self.world > 1 \
            else None

    def _prepare(self, B, H, W, L, dev, phosc_len=0):
        eng = self.eng
        eng.refresh_weights()
        P = eng.plan_train(B, H, W, L, phosc_len)
        ah = self.diffusion.alpha_hat.cpu()
        self._sa, self._sb = torch.sqrt(ah).to(dev).contiguous(), torch.sqrt(1 - ah).to(dev).contiguous()
        self._x0 = torch.zeros_like(P.x_in)
        self._eps = torch.zeros_like(P.x_in)
        self._loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scratch = torch.zeros(1024, dtype=torch.float64, device=dev)
        self._P = P
        self._stream = torch.cuda.Stream(device=dev)
        self._comm = torch.cuda.Stream(device=dev)
        for g in ([self._graph] if self._graph is not None else []) + list(getattr(self, "_seg_graphs", []) or []):
            if g is not None:
                self.lib.wd_graph_destroy(g)
        self._graph = None
        self._seg_graphs = None
        self.bucketed = (self.world > 1) if self._bucketed_arg is None else bool(self._bucketed_arg)
        self.segments = list(getattr(P, "bwd_segments", [])) if self.bucketed else []
        if len(self.segments) <= 1:
            self.segments = []

    def _body(self, st, seg: Optional[int] = None):
        """The whole step body (seg None), or segment ``seg`` of it: segment 0 = noise + forward + loss + the first part of the
        backward list, the others = the following parts."""
        lib, P = self.lib, self._P
        if seg is None or seg == 0:
            n = P.x_in[0].numel()
            B = P.x_in.shape[0]
            N.check(lib.wd_noise_images(self._x0.data_ptr(), self._eps.data_ptr(), P.t_in.data_ptr(), self._sa.data_ptr(),
                                        self._sb.data_ptr(), B, n, P.x_in.data_ptr(), st), "wd_noise_images")
            P.run_step(st)
            N.check(lib.wd_mse_loss(P.out.data_ptr(), self._eps.data_ptr(), P.out.numel(), P.dout.data_ptr(),
                                    self._loss.data_ptr(), self._scratch.data_ptr(), 1024, st), "wd_mse_loss")
        if seg is None:
            P.run_bwd(st)
        else:
            b0, b1, _, _ = self.segments[seg]
            P.run_bwd(st, b0, b1)

    def _capture(self, st, seg=None):
        lib = self.lib
        N.check(lib.wd_graph_begin(st), "wd_graph_begin")
        g = C.c_void_p()
        try:
            self._body(st, seg)
        finally:
            rc = lib.wd_graph_end(st, C.byref(g))
        N.check(rc, "wd_graph_end")
        return g

    def _run_bucketed(self, st):
        """Backward in segments; after segment j the arena prefix [a0, a1) is final: its all-reduce goes to the comm stream
        behind an event, the next segment runs meanwhile.  Returns after queueing everything; the compute stream waits for
        the collectives before the optimiser."""
        lib = self.lib
        arena = self.eng.grad_arena()
        if self.use_graph and self._seg_graphs is None:
            self._seg_graphs = [self._capture(st, j) for j in range(len(self.segments))]
        for j, (_, _, a0, a1) in enumerate(self.segments):
            if self.use_graph:
                N.check(lib.wd_graph_launch(self._seg_graphs[j], st), "wd_graph_launch")
            else:
                self._body(st, j)
            if self.world > 1 and a1 > a0:
                ev = torch.cuda.Event()
                ev.record(self._stream)
                self._comm.wait_event(ev)
                with torch.cuda.stream(self._comm):
                    torch.distributed.all_reduce(arena[a0:a1], group=self.pg)  # RCCL over xGMI (gloo in the one-GPU tests)
        if self.world > 1:
            self._stream.wait_stream(self._comm)
            arena.mul_(1.0 / self.world)

    def __call__(self, latents: torch.Tensor, text_features: torch.Tensor, labels: Optional[torch.Tensor],
                 t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                 phoscLabels: Optional[torch.Tensor] = None, check: bool = True) -> torch.Tensor:
        """One optimisation step on a batch of latents [B, C, H, W]; returns the loss as a device tensor [1] (the
        reference's ``loss.item()`` per step, train.py:295, is the caller's choice)."""
        if not latents.is_cuda:
            raise N.NativeError("TrainStep runs on the GPU only (no CPU fallback)")
        dev = latents.device
        B, _, H, W = latents.shape
        phosc_len = 0 if phoscLabels is None else phoscLabels.shape[1]
        key = (B, H, W, text_features.shape[1], phosc_len, str(dev))
        if key != self._key:
            self._prepare(B, H, W, text_features.shape[1], dev, phosc_len)
            self._key = key
        lib, P = self.lib, self._P
        if check:  # out-of-range ids raise here instead of faulting a kernel (device tensors: one small sync; a loader that has
            self.eng.check_ids(text_features, labels, phoscLabels)  # checked its host tensors passes check=False)
        if t is None:
            if self._tgen is None:
                t = self.diffusion.sample_timesteps(B)  # host RNG like the reference (train.py:281)
            else:
                t = torch.randint(low=1, high=self.diffusion.noise_steps, size=(B,), generator=self._tgen)
        # the legacy default stream cannot be captured: the step runs on its own stream, ordered after the caller's work
        # and before whatever the caller enqueues next
        cur = torch.cuda.current_stream(dev)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            st = self._stream.cuda_stream
            P.t_in.copy_(t, non_blocking=True)
            P.ctx_in.copy_(text_features, non_blocking=True)
            if phoscLabels is not None:  # UNetModelPhosc: PHOSC vector appended to the context (unetPhosc.py:1119-1131)
                P.phosc_in.copy_(phoscLabels.to(torch.int32) if phoscLabels.dtype != torch.int32 else phoscLabels,
                                 non_blocking=True)
            if labels is not None:
                P.y_in.copy_(labels, non_blocking=True)
            self._x0.copy_(latents, non_blocking=True)
            if noise is not None:
                self._eps.copy_(noise, non_blocking=True)
            else:
                N.check(lib.wd_randn(self._eps.data_ptr(), B, self._eps[0].numel(), self.seed,
                                     (self.step_index * self.world + self.rank) * B, 2, st), "wd_randn")
            if self.segments:
                self._run_bucketed(st)
            else:
                if self.use_graph:
                    if self._graph is None:
                        self._graph = self._capture(st)
                    N.check(lib.wd_graph_launch(self._graph, st), "wd_graph_launch")
                else:
                    self._body(st)
                if self.world > 1:
                    arena = self.eng.grad_arena()
                    torch.distributed.all_reduce(arena, group=self.pg)  # one all-reduce of the whole arena (WDIFF_GRAD_BUCKETS=1)
                    arena.mul_(1.0 / self.world)
            self.eng.assign_grads()
            self.opt.step()
            self.eng.refresh_weights()
        cur.wait_stream(self._stream)
        for tns in (latents, text_features, labels, noise):
            if tns is not None and tns.is_cuda:
                tns.record_stream(self._stream)
        self.step_index += 1
        return self._loss
