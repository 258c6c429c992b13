#!/usr/bin/env python3
"""Benchmark of the denoising hot path: images/s of 1000-step DDPM sampling (999 executed steps,
reference train.py:221) of 64x256 word images = [4,8,32] latents, base UNet (unet.py semantics), batch 64 per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one denoising step of the whole batch: UNet forward + x <- update with on-device noise + timestep
decrement, replayed from one hipGraph with x resident in HBM.  value = N * B / (999 * ms_per_step): the images a
full sampling() call of B images per GPU delivers per second at the measured step time (K defaults to 999 = one
full call).  Ranks are independent (weak scaling, no collective in the timed loop): batch shards + per-sample
Philox streams keyed by the global sample index.

Prints ONE JSON line on rank 0 with the driver's fields plus
  roofline     - the dominant kernel class (tap-gather MFMA GEMM): algorithmic FLOP / hipEvent-measured time;
  cpu_baseline - the CPU oracle (a port: the reference itself does not travel) on this box's host cores;
and, at N = 1, extras that never cost the line: full_call (one real Diffusion.sampling() of 64 words x 999 steps), train_step
(BASELINE configs[2]: the train.py loop, + one epoch over cached latents), phosc_variant (configs[4] on one GPU), vae_decode.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FULL = dict(image_size=(64, 256), in_channels=4, model_channels=320, out_channels=4, num_res_blocks=1,
            attention_resolutions=(1, 1), channel_mult=(1, 1), num_heads=4, num_classes=339,
            context_dim=320, vocab_size=53, max_seq_len=10)
BATCH = 64
T = 1000
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def _lib_digest():
    """(content hash of csrc/*.hip + headers + flags, is it the one the loaded libwdiff_hip.so was built from)"""
    try:
        from worddiffusion_amd import build as B
        want = B.source_digest()
        with open(os.path.join(B.CSRC, "build", "manifest.json")) as f:
            man = json.load(f)
        ok = all(man.get(src.replace(".hip", ".o")) == B._digest([os.path.join(B.CSRC, src)] + B.HEADERS) for src in B.SOURCES)
        return dict(sources=want[:16], library_built_from_these_sources=bool(ok))
    except Exception as e:
        return dict(error=f"{type(e).__name__}: {e}")


def build_model(dev, precision, variant="base"):
    from worddiffusion_amd import UNetModel, UNetModelPhosc
    from worddiffusion_amd.synthetic import fill_module_
    args = types.SimpleNamespace(device=dev, interpolation=False, latent=True, phosc=1 if variant == "phosc" else 0, phos=0)
    cls = UNetModel if variant == "base" else UNetModelPhosc
    m = fill_module_(cls(args=args, **FULL), 0).to(dev).eval()
    m.set_precision(precision)
    return m, args


class StepRunner:
    """The loop body of Diffusion.sampling, set up once so that K steps can be timed."""

    def __init__(self, model, args, dev, batch, seed, sample_offset, phosc_len=0):
        from worddiffusion_amd import Diffusion
        from worddiffusion_amd import _native as N
        from worddiffusion_amd.synthetic import synthetic_inputs
        self.N, self.lib = N, N.lib()
        self.dev, self.batch, self.seed, self.off = dev, batch, seed, sample_offset
        self.diff = Diffusion(noise_steps=T, img_size=(64, 256), args=args)
        eng = model.engine
        eng.refresh_weights()
        inp = synthetic_inputs(batch, seed=2 + sample_offset, phosc_len=phosc_len)
        self.P = P = eng.plan(batch, 8, 32, 10, phosc_len, film_steps=T if self.diff.tabulate_film else 0)
        self.stream = torch.cuda.Stream(device=dev)
        self.ca, self.cb, self.cs = self.diff._step_tables(dev)
        with torch.cuda.stream(self.stream):
            st = self.stream.cuda_stream
            N.check(self.lib.wd_randn(P.x_in.data_ptr(), batch, P.x_in[0].numel(), seed, sample_offset, 0, st), "randn")
            eng.load_inputs(P, None, None, inp["context"].to(dev), inp["y"].to(dev),
                            inp["phosc"].to(dev) if phosc_len else None)
            self.t_dev = P.t_dev
            self.reset_t()
            # per-sampling-call work (word encoder, K/V, folded attention matrices, FiLM table of all T steps): outside the
            # K timed steps, measured here once (second run: warm) and reported as config.per_call_setup_ms
            P.run_cond(st)
            P.run_film(st)
            P.film_prepare(T - 1, st)
            self.stream.synchronize()
            t0 = time.perf_counter()
            P.run_cond(st)
            P.run_film(st)
            P.film_prepare(T - 1, st)
            self.stream.synchronize()
            self.setup_ms = 1e3 * (time.perf_counter() - t0)
        self.stream.synchronize()
        self.graph = None

    def reset_t(self):
        self.t_dev.fill_(T - 1)
        self.P.t_in.fill_(T - 1)
        self.P.film_prepare(T - 1, self.stream.cuda_stream)  # (callers hold torch's stream context of self.stream)

    def one_step(self, st):
        P, lib, N = self.P, self.lib, self.N
        P.run_step(st)
        N.check(lib.wd_ddpm_step(P.x_in.data_ptr(), P.out.data_ptr(), self.batch, P.x_in[0].numel(), self.ca.data_ptr(),
                                 self.cb.data_ptr(), self.cs.data_ptr(), self.t_dev.data_ptr(), None, self.seed, self.off,
                                 st), "ddpm_step")
        N.check(lib.wd_advance_timestep(self.t_dev.data_ptr(), -1, P.t_in.data_ptr(), self.batch, st), "advance")

    def capture(self):
        st = self.stream.cuda_stream
        with torch.cuda.stream(self.stream):
            self.N.check(self.lib.wd_graph_begin(st), "graph_begin")
            self.one_step(st)
            g = C.c_void_p()
            self.N.check(self.lib.wd_graph_end(st, C.byref(g)), "graph_end")
        self.graph = g

    def run(self, k):
        """k graph replays; the timestep wraps back to T-1 after reaching 0 (index 0 is never used)."""
        st = self.stream.cuda_stream
        with torch.cuda.stream(self.stream):
            left = int(self.t_dev.item())
            done = 0
            while done < k:
                if left <= 0:
                    self.reset_t()
                    left = T - 1
                n = min(left, k - done)
                for j in range(n):
                    # the FiLM rows of a step come from the resident chunk of timesteps: entering the next chunk launches
                    # its two kernels here, inside the timed loop (as Diffusion.sampling does)
                    self.P.film_prepare(left - j, st)
                    self.N.check(self.lib.wd_graph_launch(self.graph, st), "graph_launch")
                left -= n
                done += n


def cpu_info():
    """CPU model name, physical cores (distinct (package, core id) pairs) and logical CPUs from /proc/cpuinfo."""
    model, cores, logical, phys, core = "unknown", set(), 0, None, None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "processor":
                    logical += 1
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                    cores.add((phys, core))
    except OSError:
        pass
    return dict(cpu_model=model, physical_cores=len(cores) or None, logical_cpus=logical or os.cpu_count(),
                usable_cpus=len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())


def cpu_baseline_config1(threads, iters=50):
    """SURVEY.md section 8d config 1 / BASELINE configs[0]: the reference's CPU-runnable case - ``UNetModelPhosc`` (latent
    config, no PHOSC vector), B = 1, 50 denoising iterations (Diffusion(noise_steps=51)), word "MOVE", writer 3 - through the
    CPU oracle (the reference itself does not travel to this box)."""
    from oracle import ddpm_oracle as D
    from oracle import unet_oracle as U
    from worddiffusion_amd.synthetic import synthetic_tensor
    torch.set_num_threads(threads)
    sd = {k: torch.from_numpy(synthetic_tensor(k, s, 0)) for k, s in U.state_dict_shapes(FULL, "phosc")}
    orc = U.UNetOracle(FULL, sd, "phosc", False)
    ctx = torch.tensor([D.label_padding("MOVE")], dtype=torch.int64)
    y = torch.tensor([3], dtype=torch.int64)
    Tn = iters + 1
    beta, alpha, ah = D.schedule(Tn)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 8, 32, generator=g)
    with torch.no_grad():
        orc(x, torch.tensor([Tn - 1]), ctx, y)  # warm-up
        t0 = time.perf_counter()
        for i in reversed(range(1, Tn)):
            eps = orc(x, torch.full((1,), i, dtype=torch.int64), ctx, y)
            x = D.reverse_step(beta, alpha, ah, x, eps, i, torch.randn(1, 4, 8, 32, generator=g))
        dt = (time.perf_counter() - t0) / iters
    return dict(workload="config 1: UNetModelPhosc latent config, B=1, %d DDPM iterations (oracle forward fp32 + update), "
                         "1 warm-up" % iters, s_per_forward=dt, images_per_sec_at_999_steps=1.0 / (dt * (T - 1)),
                iterations=iters, threads=threads, output_finite=bool(torch.isfinite(x).all()))


def cpu_baseline(threads, config1=True):
    """The CPU oracle (port of the reference forward, fp32 torch ops) on the host cores: B=64 forwards + update."""
    from oracle import ddpm_oracle as D
    from oracle import unet_oracle as U
    from worddiffusion_amd.synthetic import synthetic_inputs, synthetic_tensor
    torch.set_num_threads(threads)
    sd = {k: torch.from_numpy(synthetic_tensor(k, s, 0)) for k, s in U.state_dict_shapes(FULL, "base")}
    orc = U.UNetOracle(FULL, sd, "base")
    inp = synthetic_inputs(BATCH, seed=2)
    beta, alpha, ah = D.schedule(T)
    x = inp["x"]
    nfw = 6
    with torch.no_grad():
        orc(x, inp["t"], inp["context"], inp["y"])  # warm-up
        t0 = time.perf_counter()
        for i in range(nfw):
            tt = torch.full((BATCH,), T - 1 - i, dtype=torch.int64)
            eps = orc(x, tt, inp["context"], inp["y"])
            x = D.reverse_step(beta, alpha, ah, x, eps, T - 1 - i, torch.randn_like(x))
        dt = (time.perf_counter() - t0) / nfw
    out = dict(value=BATCH / (dt * (T - 1)), unit="images/s", cores=threads, kind="port",
               sample=f"{nfw} denoising steps (oracle UNet forward fp32 + update) of the B={BATCH} batch after 1 warm-up, "
                      f"{dt:.3f} s/step, extrapolated to {T - 1} steps", s_per_step=dt)
    out.update(cpu_info())
    if config1:
        try:
            out["config1"] = cpu_baseline_config1(threads)
        except Exception as e:  # an extra: never costs the line
            out["config1"] = dict(error=f"{type(e).__name__}: {e}")
    return out


def vae_leg(dev, precision, B):
    """the decode that follows the 999 steps (train.py:239-247) - SD-v1.5 AutoencoderKL decoder, synthetic
    weights, [B,4,8,32] latents -> [B,3,64,256] images.  An extra object beside the headline (the headline metric excludes the
    VAE decode by definition, section 8d)."""
    from worddiffusion_amd.synthetic import fill_module_
    from worddiffusion_amd.vae import AutoencoderKL
    vae = AutoencoderKL()
    fill_module_(vae, 0)
    vae = vae.to(dev).eval()
    vae.set_precision(precision)
    z = torch.randn(B, 4, 8, 32, device=dev) / 0.18215
    for _ in range(2):
        img = vae.decode(z).sample
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        img = vae.decode(z).sample
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return dict(workload="AutoencoderKL decode of the batch ([B,4,8,32] -> [B,3,64,256]), SD-v1.5 decoder, synthetic weights",
                ms_per_batch=ms, images_per_sec=B * 1e3 / ms, output_finite=bool(torch.isfinite(img).all().item()),
                share_of_a_sampling_call=None)


def epoch_leg(step, dev, B, n_items=1000):
    """BASELINE configs[2] / SURVEY section 8d config 3: ONE epoch of the train.py batch loop (train.py:253-295: loader batch ->
    TrainStep; the 30-batch break of train.py:263-264) over a synthetic IAM-shaped set of N = 1000 cached latents
    ([4,8,32] ~ N(0,1) * 0.18215, 339 writers, random words) read from the tensor-only container of latents.py - host batching,
    pinned copies and the H2D included, wall-clock."""
    import tempfile
    import numpy as np
    from worddiffusion_amd.latents import CachedLatentDataset, LatentCache, save_latent_cache, train_epoch
    rs = np.random.RandomState(0)
    letters = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxy"
    rows = [(f"{int(rs.randint(0, 339)):03d}", f"img{i:05d}", "".join(letters[int(k)] for k in rs.randint(0, 51, size=int(rs.randint(1, 11)))))
            for i in range(n_items)]
    wr = {f"{i:03d}": i for i in range(339)}
    lat = {r[1] + ".png": torch.from_numpy((rs.standard_normal((4, 8, 32)) * 0.18215).astype(np.float32)) for r in rows}
    with tempfile.TemporaryDirectory() as td:
        cache = LatentCache(save_latent_cache(os.path.join(td, "latents.safetensors"), lat))
        data = CachedLatentDataset(rows, wr, cache)
        train_epoch(step, data, B, dev, epoch=0, seed=1, max_batches=2)  # warm-up (page cache, pinned pool)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = train_epoch(step, data, B, dev, epoch=1, seed=1, max_batches=30)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dict(workload="one epoch of the train.py loop over %d cached latents (latents.py container -> host batches -> TrainStep), "
                         "batch %d, 30-batch break as train.py:263" % (n_items, B),
                batches=res["batches"], images=res["images"], seconds=dt, images_per_sec=res["images"] / dt,
                ms_per_batch=1e3 * dt / max(res["batches"], 1), mean_loss=res["mean_loss"])


def train_leg(dev, precision, B, steps, warmup, rank, world, barrier, wdist):
    """BASELINE configs[2]/[3]: the train.py batch loop (noise_images -> UNet -> MSE -> backward -> AdamW -> EMA) on the same
    model and latent shape as the headline, synthetic batch resident in HBM, one gradient all-reduce per step when world > 1.
    Reported as an extra object next to the headline metric (the headline stays the sampling throughput)."""
    from worddiffusion_amd import Diffusion
    from worddiffusion_amd import _native as N
    from worddiffusion_amd.optim import FusedAdamW
    from worddiffusion_amd.synthetic import synthetic_inputs
    from worddiffusion_amd.training import TrainStep
    import copy
    model, args = build_model(dev, precision, "base")
    model.train()
    ema_model = copy.deepcopy(model).eval().requires_grad_(False)
    opt = FusedAdamW(model.parameters(), lr=1e-4, ema_model=ema_model, ema_beta=0.995, step_start_ema=2000)
    diff = Diffusion(noise_steps=T, img_size=(64, 256), args=args)
    step = TrainStep(model, diff, opt, seed=99 + rank)
    inp = synthetic_inputs(B, seed=7 + rank, hw=(8, 32), num_classes=339)
    x, ctx, y = inp["x"].to(dev), inp["context"].to(dev), inp["y"].to(dev)
    for _ in range(warmup):
        step(x, ctx, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(x, ctx, y)
    torch.cuda.synchronize()
    barrier()
    elapsed = wdist.max_over_ranks(time.perf_counter() - t0, device=dev)
    ms = 1e3 * elapsed / steps
    out = dict(workload="train.py batch loop, base UNetModel, batch %d per GPU of [4,8,32] latents, AdamW(lr 1e-4) + EMA(0.995), "
                        "forward+loss+backward in one hipGraph" % B,
               ms_per_step=ms, images_per_sec=world * B * 1e3 / ms, steps=steps, warmup=warmup,
               loss_finite=bool(torch.isfinite(loss).all().item()), grad_allreduce_mb=step.eng.grad_arena().numel() * 4 / 1e6)
    # the same loop with single-pass bf16 MFMA operands (what BASELINE configs[2] literally names: "bf16"); fp32 master
    # weights, fp32 accumulation, fp32 optimiser - outside the 1e-3 parity bar, reported as an extra
    try:
        model.set_precision("bf16")
        step16 = TrainStep(model, diff, opt, seed=199 + rank)
        for _ in range(warmup):
            step16(x, ctx, y)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss16 = step16(x, ctx, y)
        torch.cuda.synchronize()
        barrier()
        el16 = wdist.max_over_ranks(time.perf_counter() - t0, device=dev)
        out["bf16_single_pass"] = dict(ms_per_step=1e3 * el16 / steps, images_per_sec=world * B * steps / el16,
                                       loss_finite=bool(torch.isfinite(loss16).all().item()))
    except Exception as e:
        out["bf16_single_pass"] = dict(error=f"{type(e).__name__}: {e}")
    model.set_precision("bf16x3")
    if world == 1:
        try:
            out["train_epoch"] = epoch_leg(step, dev, B)
        except Exception as e:  # an extra: never costs the line
            out["train_epoch"] = dict(error=f"{type(e).__name__}: {e}")
    if rank == 0:
        lib = N.lib()
        eager = TrainStep(model, diff, opt, seed=5, use_graph=False)
        eager.world = 1  # rank-0-only profiling pass: no collective (the other ranks are not in it)
        eager(x, ctx, y)
        torch.cuda.synchronize()
        lib.wd_prof_enable(1)
        eager(x, ctx, y)
        torch.cuda.synchronize()
        pms = (C.c_double * N.NCLASS)()
        cnt = (C.c_int64 * N.NCLASS)()
        fls = (C.c_double * N.NCLASS)()
        lib.wd_prof_collect_flops(pms, cnt, fls)
        lib.wd_prof_enable(0)
        kc = {}
        for i in range(N.NCLASS):
            d = dict(ms_per_step=pms[i], launches_per_step=int(cnt[i]))
            if i in CONTRACTION_CLASSES and pms[i] > 0 and fls[i] > 0:
                tf = fls[i] / (pms[i] * 1e-3) / 1e12
                d.update(gflop_per_step=fls[i] / 1e9, tflops=tf, mfma_frac=tf / PEAK_BF16_TFLOPS)
            kc[N.CLASS_NAMES[i]] = d
        out["kernel_classes"] = kc
        out["gemm_tflops_algorithmic"] = fls[0] / (pms[0] * 1e-3) / 1e12 if pms[0] > 0 else None
        # roofline of the training step's contractions (eager pass, hipEvent pair per launch; algorithmic FLOPs, each multiply-add
        # once - the split-bf16 path issues three MFMAs per product)
        tot_fl = sum(fls[i] for i in CONTRACTION_CLASSES)
        tot_ms = sum(pms[i] for i in CONTRACTION_CLASSES)
        if tot_ms > 0:
            dwc = kc.get("weight_gradient", {})
            out["roofline"] = dict(bound="mfma", unit="TFLOP/s", peak=PEAK_BF16_TFLOPS, achieved=tot_fl / (tot_ms * 1e-3) / 1e12,
                                   frac=tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, mfma_issue_factor=3,
                                   gflop_per_step=tot_fl / 1e9, contraction_ms_per_step=tot_ms,
                                   weight_gradient_tflops=dwc.get("tflops"),
                                   note="all contraction classes of one training step (forward, data gradients, weight gradients); "
                                        "profiles/rNN_pmc_mfma_util_train.json holds the MFMA-busy counters per kernel")
    return out


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: start N fresh worker processes of this script, one per GPU, with the
    torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), BEFORE anything in this process has
    touched the GPU (children are new processes, not a re-exec).  Rank 0's stdout (the one JSON line) is passed through;
    the exit code is the first non-zero child code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p in procs:
            code = p.wait()
            rc = rc or code
            if code != 0:  # a dead rank would leave the others waiting at the next barrier
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def launch_check(a):
    """``--launch-check``: rendezvous + barrier + max-over-ranks only (no GPU work) - rehearses the N-rank control flow of
    this script on a machine without GPUs (tests/test_dist_gloo.py).  Its line is NOT a measurement."""
    from worddiffusion_amd import dist as wdist
    rank, world, local = wdist.init_process_group("gloo")
    assert world == a.gpus, f"launched with WORLD_SIZE={world} but --gpus {a.gpus}"
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    slowest = wdist.max_over_ranks(float(rank + 1))
    train = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if a.train_steps != 0:  # the multi-GPU training leg's process handling (fresh children, wall-clock limit), without kernels
            train = train_leg_in_children(a, rank, world, local)
    if rank == 0:
        print(json.dumps(dict(launch_check=True, n_gpus=world, max_over_ranks=slowest, value=None, train_step=train,
                              note="control-flow rehearsal only: no kernel ran")))


def full_call_leg(model, args, dev, B, rank, world, barrier, wdist):
    """One real ``Diffusion.sampling()`` call per rank - B words x 999 steps through the product API (word encoder, K/V, folds,
    FiLM table, graph capture, 999 replays, read-back of the latents) - timed wall-clock between barriers, max over ranks.
    The un-extrapolated companion of the headline ``value``."""
    from worddiffusion_amd import Diffusion
    from worddiffusion_amd.synthetic import synthetic_inputs
    diff = Diffusion(noise_steps=T, img_size=(64, 256), args=args)
    inp = synthetic_inputs(B, seed=2 + rank * B)
    words = ["".join("ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxy"[(7 * i + 3 * j) % 51] for j in range(1 + i % 10))
             for i in range(B)]
    barrier()
    t0 = time.perf_counter()
    lat = diff.sampling(model, None, B, words, inp["y"], args, seed=1234, sample_offset=rank * B)
    torch.cuda.synchronize()
    barrier()
    sec = wdist.max_over_ranks(time.perf_counter() - t0, device=dev)
    model.eval()
    return dict(workload="one Diffusion.sampling() call per GPU: %d words x %d executed steps, vae=None (latents returned)"
                         % (B, T - 1), seconds=sec, images_per_sec=world * B / sec, executed_steps=diff.last_stats["steps"],
                graph=diff.last_stats["graph"], output_finite=bool(torch.isfinite(lat).all().item()))


CONTRACTION_CLASSES = {
    0: "wd_gemm2_kernel<128,160,3,2,M16> (LDS-staged tap-gather split-bf16 MFMA GEMM: the 4x16-level convolutions, K cut over workgroups)",
    6: "wd_gemm tiles other than 128x160",
    8: "wd_gemm4_kernel (two workgroups per CU)",
    9: "wd_gemmw_kernel<3,1> (64x320 tiles, A rows through LDS, fragment-major weights straight into the MFMA operand registers: "
       "the 3x3 convolutions and 1x1 projections of the 8x32 level)",
    10: "wd_ff_kernel<3> (GEGLU feed-forward + residual per 64-token panel, hidden activations on chip)",
    11: "wd_dw_kernel<3> (weight gradients of the training step from the row-major planes, transposed LDS reads)",
    12: "wd_gemmq_kernel (64x80 tiles, all of K inside the workgroup, input rows kept in LDS, weights straight to registers: the 3x3 and "
        "linear layers of the 4x16 level)",
}
PMC_KEYS = {0: "wd_gemm2_kernel<128, 160, 3, 2, false, true>", 9: "wd_gemmw_kernel<3, 1, false>", 10: "wd_ff_kernel<3, true>"}


def streaming_bytes_per_step(P, lib, npl):
    """Algorithmic HBM bytes per step of the streaming classes, from the launch plan's own arguments: GroupNorm apply (fp32 map in,
    operand planes out), the folded cross-attention pair (tokens in, tokens + next-LayerNorm planes out) and the split-K combine
    (slabs in, result + planes out)."""
    by = {"gn_apply": 0.0, "attention": 0.0, "gemm_splitk_reduce": 0.0}
    for fn, args, what in P.step:
        name = getattr(fn, "__name__", "")
        if name == "wd_gn_apply":
            rows, c = args[2] * args[3], args[4]
            by["gn_apply"] += rows * c * (4 + 2 * npl) + (rows * c * 2 * npl if args[17] else 0)
        elif name == "wd_gn_apply2":
            rows, c = args[14] * args[15], args[2] + args[9]
            by["gn_apply"] += rows * c * (4 + 2 * npl) + (rows * c * 2 * npl if args[24] else 0)
        elif name in ("wd_xattn_pair", "wd_xattn_fused"):
            rows, c = args[2] * args[3], args[4]
            by["attention"] += rows * c * (4 + 4 + 2 * npl)
        elif name == "wd_gemm":
            g = args[0]._obj
            if g.w_layout != 3 and g.ws and g.ksplit == 0 and g.act == 0:
                ks = lib.wd_gemm_auto_ksplit(g.m, g.n, g.ktot, g.ws_floats)
                if ks > 1:
                    by["gemm_splitk_reduce"] += g.m * g.n * (4 * ks + (4 if g.out_f32 else 0) + (2 * npl if g.out_hi else 0) +
                                                               (4 if g.resid else 0))
    return by


def roofline_leg(runner, precision):
    """Eager pass with a hipEvent pair around every launch on the launch stream (wd_prof_*): per-class time, algorithmic FLOP/s
    against the dense bf16 MFMA peak for the contraction classes, algorithmic bytes/s against the HBM peak for the streaming ones.
    The `roofline` object describes the class that takes the most time."""
    from worddiffusion_amd import _native as N
    lib = N.lib()
    nprof = 3
    with torch.cuda.stream(runner.stream):
        st = runner.stream.cuda_stream
        runner.reset_t()
        runner.one_step(st)
        torch.cuda.synchronize()
        lib.wd_prof_enable(1)
        for _ in range(nprof):
            runner.one_step(st)
        ms = (C.c_double * N.NCLASS)()
        cnt = (C.c_int64 * N.NCLASS)()
        fl = (C.c_double * N.NCLASS)()
        lib.wd_prof_collect_flops(ms, cnt, fl)
        lib.wd_prof_enable(0)
    npl = 2 if precision == "bf16x3" else 1
    sbytes = streaming_bytes_per_step(runner.P, lib, npl)
    classes = {}
    for i in range(N.NCLASS):
        name = N.CLASS_NAMES[i]
        d = dict(ms_per_step=ms[i] / nprof, launches_per_step=int(cnt[i]) // nprof)
        if i in CONTRACTION_CLASSES and ms[i] > 0:
            tf = fl[i] / (ms[i] * 1e-3) / 1e12
            d.update(gflop_per_step=fl[i] / nprof / 1e9, tflops=tf, mfma_frac=tf / PEAK_BF16_TFLOPS)
        if name in sbytes and ms[i] > 0:
            gbs = sbytes[name] / (ms[i] / nprof * 1e-3) / 1e9
            d.update(mb_per_step=sbytes[name] / 1e6, gb_per_s=gbs, hbm_frac=gbs / PEAK_HBM_GBS)
        classes[name] = d
    dom = max(CONTRACTION_CLASSES, key=lambda i: ms[i])
    ach = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
    traffic, traffic_note = None, "no PMC summary for this kernel under profiles/"
    try:  # HBM bytes per launch of that kernel from the separate rocprofv3 --pmc passes (profiles/, per round)
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        name = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_hbm_traffic.json"))[-1]  # the latest round's
        k = json.load(open(os.path.join(pdir, name)))[PMC_KEYS[dom]]
        traffic = (k["fetch_mb_corrected"] + k["write_mb"]) * 1e6
        traffic_note = ("HBM bytes per launch of %s: %.1f MB fetched (FETCH_SIZE, gfx950 x2 correction) + %.1f MB written (WRITE_SIZE); "
                        "separate rocprofv3 --pmc passes over this bench command, profiles/%s"
                        % (PMC_KEYS[dom], k["fetch_mb_corrected"], k["write_mb"], name))
    except (OSError, KeyError, ValueError, IndexError):
        pass
    mfma_util = None
    try:  # MFMA-pipe busy fraction of that kernel from its own PMC pass (tools/pmc_mfma.py)
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        name = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_mfma_util.json"))[-1]
        mfma_util = json.load(open(os.path.join(pdir, name)))[PMC_KEYS[dom]]["mfma_util_percent"] / 100.0
    except (OSError, KeyError, ValueError, IndexError):
        pass
    total_fl = sum(fl[i] for i in CONTRACTION_CLASSES)
    total_ms = sum(ms[i] for i in range(N.NCLASS))
    roof = dict(bound="mfma", kernel=CONTRACTION_CLASSES[dom] + "; hipEvent pair around every launch", achieved=ach,
                peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_BF16_TFLOPS, traffic=traffic, traffic_note=traffic_note,
                launches_per_step=int(cnt[dom]) // nprof, avg_launch_us=1e3 * ms[dom] / max(int(cnt[dom]), 1),
                algorithmic_gflop_per_step=fl[dom] / nprof / 1e9, share_of_step_kernel_time=ms[dom] / total_ms if total_ms else None,
                all_contractions=dict(gflop_per_step=total_fl / nprof / 1e9,
                                      tflops=total_fl / (sum(ms[i] for i in CONTRACTION_CLASSES) * 1e-3) / 1e12),
                mfma_issue_factor=3 if precision == "bf16x3" else 1, mfma_pipe_busy_pmc=mfma_util,
                note="achieved counts each multiply-add once; the split-bf16 path issues 3 MFMAs per product; mfma_pipe_busy_pmc = "
                     "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) of that kernel from a separate rocprofv3 --pmc pass "
                     "(profiles/rNN_pmc_mfma_util.json)")
    return roof, classes



def train_worker(a):
    """Child process of one rank for the multi-GPU training leg (``--train-worker``): its own process group on MASTER_PORT (the
    parent passes port + 1), ``train_leg`` with a barrier either side, rank 0 prints the object."""
    from worddiffusion_amd import dist as wdist
    if os.environ.get("WDIFF_BENCH_TEST_HANG") == "1":  # (tests/test_dist_gloo.py: the timeout path)
        time.sleep(3600)
    if a.launch_check:  # CPU rehearsal of the control flow: rendezvous, barrier, one object - no kernel
        rank, world, local = wdist.init_process_group("gloo")
        import torch.distributed as dist
        dist.barrier()
        if rank == 0:
            print(json.dumps(dict(launch_check=True, n_gpus=world, ms_per_step=None)))
        dist.destroy_process_group()
        return
    rehearse = os.environ.get("WDIFF_BENCH_REHEARSE", "0") == "1"
    rank, world, local = wdist.init_process_group("gloo" if rehearse else "nccl")
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"

    def barrier():
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()

    out = train_leg(dev, a.precision, a.batch, a.train_steps, 3, rank, world, barrier, wdist)
    if rank == 0:
        print(json.dumps(out))
    import torch.distributed as dist
    dist.destroy_process_group()


def train_leg_in_children(a, rank, world, local):
    """Every rank starts its own child (same RANK / WORLD_SIZE, MASTER_PORT + 1) and waits for it with a wall-clock limit; rank
    0 returns the child's object, or ``{"error": ...}``.  The children are new processes (subprocess, no exec of this one)."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "29500")) + 1
    env = dict(os.environ, MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.abspath(__file__), "--train-worker", "--gpus", str(world), "--batch", str(a.batch),
           "--precision", a.precision, "--train-steps", str(a.train_steps)] + (["--launch-check"] if a.launch_check else [])
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True)
    try:
        out, _ = p.communicate(timeout=a.train_timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        p.communicate()
        return dict(error="timeout", limit_s=a.train_timeout)
    if rank != 0:
        return None
    if p.returncode != 0:
        return dict(error=f"child exit code {p.returncode}")
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    try:
        return json.loads(lines[-1])
    except (IndexError, ValueError):
        return dict(error="no JSON object from the training child")



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=T - 1)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16"])
    ap.add_argument("--variant", default="base", choices=["base", "phosc"],
                    help="base = BASELINE configs[1] (the headline); phosc = UNetModelPhosc with a 769-int PHOSC vector "
                         "(configs[4] model: 779-token context, 256-token self-attention), reported as an extra")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE-decode extra (section 8f-2)")
    ap.add_argument("--train-steps", type=int, default=-1,
                    help="also time this many train.py-style steps (0 = skip; default: 30 on one GPU, 0 on several - the "
                         "multi-GPU run is the scaling measurement of the headline metric, pass a count to add the data-parallel "
                         "training leg with its gradient all-reduce)")
    ap.add_argument("--no-full-call", action="store_true", help="skip the un-extrapolated full sampling() call")
    ap.add_argument("--no-phosc", action="store_true", help="skip the PHOSC-variant extra (BASELINE configs[4] on one GPU)")
    ap.add_argument("--train-worker", action="store_true",
                    help="(internal) this process is the per-rank child of the multi-GPU training leg: rendezvous, train_leg, one JSON "
                         "object from rank 0")
    ap.add_argument("--train-timeout", type=float, default=240.0, help="wall-clock limit of the multi-GPU training leg (seconds)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous / barrier / max-over-ranks only, no GPU work (CPU rehearsal of the N-rank control flow)")
    a = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        # no launcher around us: become the launcher (nothing in this process has touched the GPU yet)
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != a.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={env_world} but --gpus {a.gpus}")
    if a.train_worker:
        return train_worker(a)
    if a.launch_check:
        return launch_check(a)

    from worddiffusion_amd import dist as wdist
    # WDIFF_BENCH_REHEARSE=1: every rank on cuda:0 with gloo - rehearses the N>1 control flow on a one-GPU box (numbers from
    # such a run mean nothing; RCCL refuses two ranks on one device)
    rehearse = os.environ.get("WDIFF_BENCH_REHEARSE", "0") == "1"
    rank, world, local = wdist.init_process_group("gloo" if rehearse else "nccl")  # no-op for a single process
    if world == 1 or rehearse:
        torch.cuda.set_device(0)
        local = 0
    dev = f"cuda:{local if world > 1 else 0}"
    B = a.batch

    model, args = build_model(dev, a.precision, a.variant)
    runner = StepRunner(model, args, dev, B, seed=1234, sample_offset=rank * B,
                        phosc_len=769 if a.variant == "phosc" else 0)
    setup_ms = runner.setup_ms
    runner.capture()
    runner.run(a.warmup)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    runner.run(a.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = wdist.max_over_ranks(elapsed, device=dev)
    ms_per_step = 1e3 * elapsed / a.steps
    finite = bool(torch.isfinite(runner.P.x_in).all().item())

    roof = None
    prof_extra = None
    if rank == 0 and not a.no_roofline:
        roof, prof_extra = roofline_leg(runner, a.precision)

    phosc_extra = None
    if rank == 0 and world == 1 and a.variant == "base" and not a.no_phosc:
        # BASELINE configs[4] (PHOSC-conditioned sampling; sharded over the ranks like the headline, no collective): one GPU's share -
        # the same loop with UNetModelPhosc and the 769-int PHOSC vector (self-attention + 779-key cross-attention per block)
        try:
            pm, pargs = build_model(dev, a.precision, "phosc")
            pr = StepRunner(pm, pargs, dev, B, seed=4321, sample_offset=0, phosc_len=769)
            pr.capture()
            pr.run(10)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pr.run(200)
            torch.cuda.synchronize()
            pms = 1e3 * (time.perf_counter() - t0) / 200
            phosc_extra = dict(workload="configs[4] on one GPU: UNetModelPhosc (args.phosc = 1), batch %d, 10 text + 769 PHOSC tokens, "
                                        "200 graph-replayed steps" % B, ms_per_step=pms,
                               images_per_sec=B * 1e3 / (pms * (T - 1)), per_call_setup_ms=pr.setup_ms,
                               output_finite=bool(torch.isfinite(pr.P.x_in).all().item()), launches_per_step=len(pr.P.step) + 2)
            if not a.no_roofline:
                proof, pclasses = roofline_leg(pr, a.precision)
                # the attention class of this variant is MFMA work: 2 * 2 * B * heads * nq * nk * d per attention (QK^T and PV),
                # self-attention over the map's own positions + cross-attention over the 779-token context, per block
                att_fl = sum(4.0 * B * 4 * hw * (hw + 779) * 80 for hw in (256, 64, 256, 256))
                ac = pclasses.get("attention", {})
                if ac.get("ms_per_step"):
                    tf = att_fl / (ac["ms_per_step"] * 1e-3) / 1e12
                    ac.update(gflop_per_step=att_fl / 1e9, tflops=tf, mfma_frac=tf / PEAK_BF16_TFLOPS)
                phosc_extra["roofline"] = proof
                phosc_extra["kernel_classes"] = pclasses
            del pr, pm
            torch.cuda.empty_cache()
        except Exception as e:  # an extra: never costs the headline line
            phosc_extra = dict(error=f"{type(e).__name__}: {e}")

    full_call = None
    if not a.no_full_call and a.variant == "base" and B == BATCH:
        try:
            full_call = full_call_leg(model, args, dev, B, rank, world, barrier, wdist)
        except Exception as e:  # an extra: never costs the headline line (every rank fails alike: same code, same sizes)
            full_call = dict(error=f"{type(e).__name__}: {e}")

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.variant == "base":
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        phys = cpu_info().get("physical_cores") or usable
        cpu = cpu_baseline(max(1, min(usable, phys)))  # N = physical cores (as far as the affinity mask allows)
        if cpu["cores"] > 64:
            # the oracle's many small fp32 ops stop scaling (and get slower) beyond ~64 threads on this class of host: the same
            # sample at 64 threads beside it, so that the stated baseline is not an artefact of oversubscription
            try:
                c64 = cpu_baseline(64, config1=False)
                cpu["at_64_threads"] = dict(value=c64["value"], cores=64, s_per_step=c64["s_per_step"])
            except Exception as e:
                cpu["at_64_threads"] = dict(error=f"{type(e).__name__}: {e}")

    train = None
    if a.train_steps < 0:
        a.train_steps = 30 if world == 1 else 10
    if a.train_steps > 0 and a.variant == "base" and world > 1:
        # BASELINE configs[3] (data-parallel training over RCCL): every rank hands the leg to a FRESH child process with a wall-clock
        # limit - its rendezvous, its collectives and its GPU context are its own, so a stuck all-reduce costs `train_step:
        # {"error": "timeout"}` and never the headline line, which is already measured
        del runner
        torch.cuda.empty_cache()
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        train = train_leg_in_children(a, rank, world, local)
    elif a.train_steps > 0 and a.variant == "base":
        del runner
        torch.cuda.empty_cache()
        try:
            train = train_leg(dev, a.precision, B, a.train_steps, 5, rank, world, barrier, wdist)
        except Exception as e:  # the extra leg must never cost the headline line
            train = dict(error=f"{type(e).__name__}: {e}")

    vae = None
    if rank == 0 and world == 1 and not a.no_vae and a.variant == "base":
        try:
            if train is None:
                del runner
            torch.cuda.empty_cache()
            vae = vae_leg(dev, a.precision, B)
            vae["share_of_a_sampling_call"] = vae["ms_per_batch"] / (ms_per_step * (T - 1) + vae["ms_per_batch"])
        except Exception as e:
            vae = dict(error=f"{type(e).__name__}: {e}")

    if rank == 0:
        value = world * B * 1e3 / (ms_per_step * (T - 1))
        line = dict(metric="denoised 64x256 word images/sec (1000-step DDPM), whole job", value=value, unit="images/s",
                    n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=ms_per_step, higher_is_better=True,
                    scaling="weak", vs_baseline=None,
                    dtype="bf16x3" if a.precision == "bf16x3" else "bf16", data="synthetic",
                    config=dict(workload=("%s: batch %d per GPU of 64x256 crops = [%d,4,8,32] latents, 1000-step DDPM (999 executed "
                                          "steps), %s (320 ch, mult (1,1), 4 heads, 339 writers), random-init synthetic weights"
                                          % ("BASELINE configs[1]" if (a.variant == "base" and B == BATCH) else
                                             ("BASELINE configs[4], one GPU's share" if a.variant == "phosc" else "configs[1] model at another batch"),
                                             B, B, "base unet.py UNetModel" if a.variant == "base" else
                                             "unetPhosc.py UNetModelPhosc (args.phosc = 1, 10 word ids + 769-int PHOSC vector)")),
                                variant=a.variant, batch_per_gpu=B, noise_steps=T, executed_steps_per_image=T - 1, forwards_per_step=1,
                                precision=("split-bf16 MFMA x3, fp32 accumulate (<=1e-4 of the fp32 reference)"
                                           if a.precision == "bf16x3" else "bf16 MFMA single pass (outside 1e-3 parity)"),
                                images_per_sec_per_gpu=value / world, output_finite=finite,
                                per_call_setup_ms=setup_ms, kernel_sources_sha256=_lib_digest()),
                    roofline=roof, cpu_baseline=cpu, full_call=full_call, kernel_classes=prof_extra, train_step=train,
                    vae_decode=vae, phosc_variant=phosc_extra)
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
