"""GPU parity of the sampler LOOPS of the reference's variant scripts, through the public Python surface (C-ABI underneath):

  * ``Diffusion.sampling(..., phoscLabels=...)`` / ``sampling_phosc`` (trainGWModifyCondition.py:249-275) against the
    reference's recorded-noise trajectories with ``UNetModelPhosc(args.phosc = 1)`` - 37-int vector on the small model, the
    769-int PHOSC vector on the full 320-channel config (779-key MFMA attention + FiLM table inside the captured graph);
  * ``sampling_modify_condition`` (trainModifyCondition.py:545-611: s_id = ones, '_' alphabet, T = 600 default);
  * ``args.charLevelEmb = 1`` (unet.py:855-866) - bit-identical to charLevelEmb = 0 in the reference, and here;
  * the model's train / eval state after a sampling call (train.py:201,238).

Goldens: ``oracle/make_golden_samplers.py`` (outputs of the reference's own modules).  Tolerance: 1e-3 max-norm relative
(BASELINE.json north_star); the split-bf16 path is asserted at 1e-4 on the short trajectories."""
import copy
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as U  # noqa: E402
from tests._common import FULL, SMALL, golden_state_dict, load_golden, make_args, max_rel  # noqa: E402
from worddiffusion_amd import EMA, Diffusion, UNetModel, UNetModelPhosc  # noqa: E402
from worddiffusion_amd.synthetic import fill_module_, synthetic_inputs, synthetic_tensor  # noqa: E402

DEV = "cuda:0"


class IdentityVAE:
    def decode(self, z):
        return types.SimpleNamespace(sample=z)


def _phosc_model(cfg, seed, phosc=1):
    m = UNetModelPhosc(args=make_args(device=DEV, phosc=phosc), **cfg)
    return fill_module_(m, seed).to(DEV).eval()


@pytest.mark.parametrize("tag,cfg,hw", [("ddpm_traj_phosc_small", SMALL, (32, 64)), ("ddpm_traj_phosc_full", FULL, (64, 256))])
@pytest.mark.parametrize("use_graph", [True, False])
def test_phosc_sampling_loop_matches_reference(golden_dir, tag, cfg, hw, use_graph):
    g = load_golden(golden_dir, tag)
    T, n = int(g["T"]), g["labels"].shape[0]
    m = _phosc_model(cfg, int(g["seed"]))
    args = make_args(device=DEV, phosc=1)
    diff = Diffusion(noise_steps=T, img_size=hw, args=args)
    noise = torch.from_numpy(g["noise"])
    labels, phosc = torch.from_numpy(g["labels"]), torch.from_numpy(g["phosc"])
    kw = dict(x_T=noise[0], noise=list(noise[1:]))
    # per-step states (the recording loop runs eager launches)
    rec = []
    img = diff.sampling(m, IdentityVAE(), n, str(g["word"]), labels, args, phoscLabels=phosc, record=rec, **kw)
    xs = torch.stack([r.cpu() for r in rec])
    assert xs.shape == tuple(g["x_per_step"].shape)
    assert max_rel(xs, g["x_per_step"]) < 1e-4
    assert float((img - torch.from_numpy(g["image"])).abs().max()) < 1e-3
    # the product call forms: keyword form and the trainGWModifyCondition.py:249 argument order, graph or eager
    a = diff.sampling(m, IdentityVAE(), n, str(g["word"]), labels, args, phoscLabels=phosc, use_graph=use_graph, **kw)
    b = diff.sampling_phosc(m, IdentityVAE(), n, str(g["word"]), phosc, labels, args, use_graph=use_graph, **kw)
    assert torch.equal(a, b) and torch.equal(a, img)  # graph replay == eager launches, bit for bit
    assert diff.last_stats["graph"] == use_graph and diff.last_stats["steps"] == T - 1
    assert float((a - torch.from_numpy(g["image"])).abs().max()) < 1e-3
    with pytest.raises(ValueError):
        diff.sampling(m, None, n, str(g["word"]), labels, args)  # args.phosc = 1 without a PHOSC vector


@pytest.mark.parametrize("tag", ["skip", "full", "phosc"])
def test_step_skipping_sampler_matches_reference(golden_dir, tag):
    """Diffusion.sampling3 (same argument order as regenerateFromtrain2.py:465) against the trajectories recorded from the
    reference's own class on its hard-coded 600-step schedule: the UNet on 120 of 599 iterations + deterministic update,
    fullSampling (every step, recorded draws), the PHOSC call form from a given x_t at epoch 12.  Graph replay == eager."""
    g = load_golden(golden_dir, "ddpm_traj_sampling3")
    full, epoch = bool(g[tag + "_full"]), int(g[tag + "_epoch"])
    words = [str(w) for w in g[tag + "_words"]]
    n = len(words)
    phosc_on = (tag + "_phosc") in g.files
    m = _phosc_model(SMALL, int(g[tag + "_seed"]), phosc=1 if phosc_on else 0)
    args = make_args(device=DEV, phosc=1 if phosc_on else 0)
    args.fullSampling = full
    diff = Diffusion(noise_steps=600, img_size=(32, 64), args=args)
    labels = torch.from_numpy(g[tag + "_labels"])
    phosc = torch.from_numpy(g[tag + "_phosc"]) if phosc_on else None
    noise = torch.from_numpy(g[tag + "_noise"])
    ni = int(g[tag + "_noise_input"])
    x_t = torch.from_numpy(g[tag + "_x_t"])
    kw = dict(x_T=noise[0], noise=list(noise[1:]) if full else None)
    rec = []
    x0 = diff.sampling3(epoch, x_t, words, phosc, m, m, None, 0, ni, n, words, labels, args, record=rec, **kw)
    assert diff.last_stats["model_calls"] == int(g[tag + "_calls"])
    # the reference records the x handed to the model on every 10th model call
    called = [i for i in reversed(range(1, 600)) if full or Diffusion.sampling3_calls_model(i, 600, epoch)]
    steps = list(reversed(range(1, 600)))
    xs = torch.stack([rec[steps.index(i)].cpu() for i in called[::10]])
    ref = g[tag + "_x_every10calls"]
    assert xs.shape == tuple(ref.shape)
    for i in range(xs.shape[0]):
        assert max_rel(xs[i], ref[i]) < 1e-3, (tag, i)
    assert max_rel(x0.cpu(), g[tag + "_x_final"]) < 1e-3
    zero, imgs, img = diff.sampling3(epoch, x_t, words, phosc, m, m, IdentityVAE(), 0, ni, n, words, labels, args, **kw)
    assert zero == 0 and float((img.cpu() - torch.from_numpy(g[tag + "_image"])).abs().max()) < 5e-3
    x1 = diff.sampling3(epoch, x_t, words, phosc, m, m, None, 0, ni, n, words, labels, args, use_graph=False, **kw)
    x2 = diff.sampling3(epoch, x_t, words, phosc, m, m, None, 0, ni, n, words, labels, args, use_graph=True, **kw)
    assert torch.equal(x1, x2) and torch.equal(x1, x0)


def test_modify_condition_sampling_loop_matches_reference(golden_dir):
    g = load_golden(golden_dir, "ddpm_traj_modcond")
    cfg = dict(SMALL, vocab_size=int(g["vocab_size"]))
    m = fill_module_(UNetModel(args=make_args(device=DEV), **cfg), int(g["seed"])).to(DEV).eval()
    args = make_args(device=DEV)
    word, labels = str(g["word"]), torch.from_numpy(g["labels"])
    # T = 8, every step
    diff = Diffusion(noise_steps=8, img_size=(32, 64), args=args)
    noise = torch.from_numpy(g["T8_noise"])
    rec = []
    img = diff.sampling_modify_condition(m, IdentityVAE(), None, word, None, 3, labels, args, x_T=noise[0],
                                         noise=list(noise[1:]), record=rec)
    xs = torch.stack([r.cpu() for r in rec])
    assert max_rel(xs, g["T8_x_per_step"]) < 1e-4
    assert float((img - torch.from_numpy(g["T8_image"])).abs().max()) < 1e-3
    for use_graph in (True, False):
        again = diff.sampling_modify_condition(m, IdentityVAE(), None, word, None, 3, labels, args, x_T=noise[0],
                                               noise=list(noise[1:]), use_graph=use_graph)
        assert torch.equal(again, img)
    # s_id = ones (trainModifyCondition.py:565): the labels argument must not matter
    other = diff.sampling_modify_condition(m, IdentityVAE(), None, word, None, 3, torch.tensor([0, 0, 0]), args,
                                           x_T=noise[0], noise=list(noise[1:]))
    assert torch.equal(other, img)
    # the script's default schedule: 600 noise steps -> 599 executed steps; x every 100 steps.  The random-weight UNet is
    # not a denoiser: |x| grows from 2.7 to 86 along this trajectory, and with it any rounding difference, hence the
    # looser bound on the late checkpoints (still max-norm relative to the fp32 reference).
    diff = Diffusion(noise_steps=600, img_size=(32, 64), args=args)
    noise = torch.from_numpy(g["T600_noise"])
    rec = []
    diff.sampling_modify_condition(m, None, None, word, None, 2, labels[:2], args, x_T=noise[0], noise=list(noise[1:]),
                                   record=rec)
    assert len(rec) == 599
    xs = torch.stack([rec[k].cpu() for k in range(0, 599, 100)])
    ref = g["T600_x_every100"]
    errs = [max_rel(xs[k], ref[k]) for k in range(len(ref))]
    assert errs[0] == 0.0 and errs[1] < 1e-4 and max(errs) < 1e-3, errs
    lat = diff.sampling_modify_condition(m, None, None, word, None, 2, labels[:2], args, x_T=noise[0], noise=list(noise[1:]))
    assert diff.last_stats["graph"] and diff.last_stats["steps"] == 599
    assert torch.equal(lat, diff.sampling_modify_condition(m, None, None, word, None, 2, labels[:2], args, x_T=noise[0],
                                                           noise=list(noise[1:]), use_graph=False))


def test_char_level_emb_flag_is_accepted_and_changes_nothing(golden_dir):
    g0 = load_golden(golden_dir, "fwd_base_full")
    g1 = load_golden(golden_dir, "fwd_base_full_charlevel")
    assert bool(g1["bit_identical_to_charLevelEmb0"]) and np.array_equal(g0["out"], g1["out"])
    outs = []
    for flag in (0, 1):
        m = UNetModel(args=make_args(device=DEV, charLevelEmb=flag), **FULL)
        m.load_state_dict(golden_state_dict(g0), strict=True)
        m = m.to(DEV).eval()
        with torch.no_grad():
            outs.append(m(torch.from_numpy(g0["x"]).to(DEV), None, original_images=None,
                          timesteps=torch.from_numpy(g0["t"]).to(DEV), context=torch.from_numpy(g0["context"]).to(DEV),
                          y=torch.from_numpy(g0["y"]).to(DEV)))
    assert torch.equal(outs[0], outs[1])
    assert max_rel(outs[1].cpu(), g1["out"]) < 1e-4
    with pytest.raises(ValueError):  # unet.py:864 views the embedding as (B, 10, 320)
        m(torch.from_numpy(g0["x"]).to(DEV), None, timesteps=torch.from_numpy(g0["t"]).to(DEV),
          context=torch.from_numpy(g0["context"][:, :7]).to(DEV), y=torch.from_numpy(g0["y"]).to(DEV))


def test_sampling_leaves_the_model_in_train_mode_like_the_reference():
    """train.py:201 ``model.eval()`` ... train.py:238 ``model.train()`` - unconditional, whatever mode the model came in;
    sampling3 (regenerateFromtrain2.py:622, ``#model.train()``) leaves it in eval mode."""
    m = fill_module_(UNetModelPhosc(args=make_args(device=DEV), **SMALL), 3).to(DEV)
    args = make_args(device=DEV)
    diff = Diffusion(noise_steps=4, img_size=(32, 64), args=args)
    labels = torch.tensor([1, 2], dtype=torch.int64)
    for start in (m.eval, m.train):
        start()
        diff.sampling(m, None, 2, "MOVE", labels, args, seed=1)
        assert m.training
    ema_model = copy.deepcopy(m).eval().requires_grad_(False)
    diff.sampling(ema_model, None, 2, "MOVE", labels, args, seed=1)
    assert ema_model.training
    # a frozen copy in train mode still takes the inference path (no gradient plan is built for it)
    out = ema_model(torch.zeros(2, 4, 4, 8, device=DEV), None, timesteps=torch.tensor([1, 2], device=DEV),
                    context=torch.full((2, 10), 52, device=DEV), y=labels.to(DEV))
    assert out.grad_fn is None and ema_model._train_engine is None
    args3 = make_args(device=DEV, fullSampling=False)
    diff.sampling3(0, None, ["MOVE", "a"], None, m, m, None, 0, 1, 2, ["MOVE", "a"], labels, args3, seed=1)
    assert not m.training


def test_ema_update_reaches_the_packed_weights():
    """``EMA.step_ema`` after the warm-up updates the EMA parameters with a raw kernel; the engine of
    the EMA model must notice and repack - a forward after the update equals a forward of a fresh model loaded from
    ``ema_model.state_dict()``."""
    cfg = SMALL
    m = fill_module_(UNetModelPhosc(args=make_args(device=DEV), **cfg), 1).to(DEV)
    ema_model = copy.deepcopy(m).eval().requires_grad_(False)
    inp = synthetic_inputs(2, seed=3, hw=(4, 8), num_classes=cfg["num_classes"])

    def fwd(model):
        with torch.no_grad():
            return model(inp["x"].to(DEV), None, timesteps=inp["t"].to(DEV), context=inp["context"].to(DEV), y=inp["y"].to(DEV))

    before = fwd(ema_model)
    fill_module_(m, 2)  # "training" moved the weights
    m.to(DEV)
    ema = EMA(0.5)
    ema.step = 2000  # past the warm-up: update_model_average, not the load_state_dict copy
    ema.step_ema(ema_model, m)
    after = fwd(ema_model)
    assert not torch.equal(before, after)
    fresh = UNetModelPhosc(args=make_args(device=DEV), **cfg)
    fresh.load_state_dict(ema_model.state_dict())
    fresh = fresh.to(DEV).eval()
    assert torch.equal(after, fwd(fresh))
    # and through the sampler (the reference loop: ema.step_ema(...) then diffusion.sampling(ema_model, ...))
    diff = Diffusion(noise_steps=5, img_size=(32, 64), args=make_args(device=DEV))
    lab = torch.tensor([1, 2], dtype=torch.int64)
    a = diff.sampling(ema_model, None, 2, "MOVE", lab, make_args(device=DEV), seed=4)
    b = diff.sampling(fresh, None, 2, "MOVE", lab, make_args(device=DEV), seed=4)
    assert torch.equal(a, b)


def test_out_of_range_ids_raise_instead_of_reading_past_the_tables():
    m = fill_module_(UNetModel(args=make_args(device=DEV), **SMALL), 3).to(DEV).eval()
    x = torch.zeros(2, 4, 4, 8, device=DEV)
    t = torch.tensor([1, 2], device=DEV)
    ctx = torch.full((2, 10), 52, dtype=torch.int64, device=DEV)
    y = torch.tensor([0, 1], device=DEV)
    with torch.no_grad():
        m(x, None, timesteps=t, context=ctx, y=y)
        with pytest.raises(IndexError):
            m(x, None, timesteps=t, context=ctx, y=torch.tensor([0, SMALL["num_classes"]], device=DEV))
        with pytest.raises(IndexError):
            m(x, None, timesteps=t, context=ctx, y=torch.tensor([-1, 0], device=DEV))
        bad = ctx.clone()
        bad[1, 3] = SMALL["vocab_size"]
        with pytest.raises(IndexError):
            m(x, None, timesteps=t, context=bad, y=y)
    args = make_args(device=DEV)
    diff = Diffusion(noise_steps=4, img_size=(32, 64), args=args)
    with pytest.raises(IndexError):
        diff.sampling(m, None, 2, "MOVE", torch.tensor([0, 99]), args, seed=1)
    with pytest.raises(ValueError):
        diff.sampling(m, None, 2, "MOVE", None, args, seed=1)  # class-conditional model without writer ids
    with pytest.raises(IndexError):  # '_' (id 53) does not fit a 53-row table
        diff.sampling(m, None, 2, "a_b", torch.tensor([0, 1]), args, seed=1, underscore=True)
    with pytest.raises(KeyError):  # and is not in the 52-letter alphabet of train.py
        diff.sampling(m, None, 2, "a_b", torch.tensor([0, 1]), args, seed=1)


def test_fused_adamw_state_dict_round_trip_resumes_bit_exactly():
    """``torch.save(optimizer.state_dict())`` (train.py:316) and a resumed run: save after 2 steps, load
    into a fresh optimiser + model, step twice more - equal to 4 uninterrupted steps, bit for bit; the file also loads into
    ``torch.optim.AdamW``."""
    from worddiffusion_amd.optim import FusedAdamW
    from worddiffusion_amd.training import TrainStep
    cfg, B = SMALL, 4
    diff = Diffusion(noise_steps=1000, img_size=(32, 64), args=make_args(device=DEV))
    rs = np.random.RandomState(11)
    batches = []
    for i in range(4):
        inp = synthetic_inputs(B, seed=200 + i, hw=(4, 8), num_classes=cfg["num_classes"])
        inp["t"] = torch.from_numpy(rs.randint(1, 1000, size=(B,))).long()
        inp["eps"] = torch.from_numpy(rs.standard_normal(tuple(inp["x"].shape)).astype(np.float32))
        batches.append(inp)

    def setup():
        m = fill_module_(UNetModel(args=make_args(device=DEV), **cfg), 17).to(DEV).train()
        ema_m = copy.deepcopy(m).eval().requires_grad_(False)
        opt = FusedAdamW(m.parameters(), lr=1e-4, ema_model=ema_m, ema_beta=0.9, step_start_ema=3)
        return m, ema_m, opt, TrainStep(m, diff, opt, seed=3)

    def run(step, bs):
        for b in bs:
            step(b["x"].to(DEV), b["context"].to(DEV), b["y"].to(DEV), t=b["t"], noise=b["eps"].to(DEV))
        torch.cuda.synchronize()

    m_a, ema_a, opt_a, step_a = setup()
    run(step_a, batches)
    m_b, ema_b, opt_b, step_b = setup()
    run(step_b, batches[:2])
    osd = opt_b.state_dict()
    assert hasattr(opt_b, "param_groups") and opt_b.param_groups[0]["lr"] == 1e-4
    assert osd["param_groups"][0]["params"] == list(range(len(list(m_b.parameters()))))
    n_grad = sum(1 for p in m_b.parameters() if p.grad is not None)
    assert len(osd["state"]) == n_grad < len(osd["param_groups"][0]["params"])  # dead heads carry no state, as in torch
    assert all(float(s["step"]) == 2.0 for s in osd["state"].values())
    msd = {k: v.clone() for k, v in m_b.state_dict().items()}
    esd = {k: v.clone() for k, v in ema_b.state_dict().items()}
    # resume in fresh objects
    m_c, ema_c, opt_c, step_c = setup()
    m_c.load_state_dict(msd)
    ema_c.load_state_dict(esd)
    opt_c.load_state_dict(osd)
    assert opt_c.step_count == 2
    run(step_c, batches[2:])
    for (k, a), c in zip(m_a.state_dict().items(), m_c.state_dict().values()):
        assert torch.equal(a, c), k
    for (k, a), c in zip(ema_a.state_dict().items(), ema_c.state_dict().values()):
        assert torch.equal(a, c), k  # the EMA warm-up position (copy until step 3, then average) was restored too
    # interoperability with torch.optim.AdamW's loader
    topt = torch.optim.AdamW(list(m_c.parameters()), lr=1e-4)
    topt.load_state_dict({k: v for k, v in osd.items() if k != "wdiff"})
    assert len(topt.state) == n_grad


def test_data_parallel_noise_rows_differ_by_rank():
    """with the default seed every rank used to draw the same eps rows and the same timesteps."""
    from worddiffusion_amd import _native as N
    lib = N.lib()
    B, n = 4, 4 * 4 * 8
    st = torch.cuda.current_stream().cuda_stream

    def rows(step_index, world, rank):
        e = torch.empty(B, n, device=DEV)
        N.check(lib.wd_randn(e.data_ptr(), B, n, 0, (step_index * world + rank) * B, 2, st), "wd_randn")
        return e.cpu()

    # two ranks of batch 4 draw what one process of batch 8 draws
    whole = torch.empty(2 * B, n, device=DEV)
    N.check(lib.wd_randn(whole.data_ptr(), 2 * B, n, 0, 0, 2, st), "wd_randn")
    assert torch.equal(torch.cat([rows(0, 2, 0), rows(0, 2, 1)]), whole.cpu())
    assert not torch.equal(rows(0, 2, 0), rows(0, 2, 1)) and not torch.equal(rows(0, 2, 1), rows(1, 2, 0))


def test_film_table_chunks_equal_the_per_step_path(monkeypatch):
    """The sampler tabulates the FiLM vectors per CHUNK of timesteps (engine.FILM_CHUNK_ROWS rows resident).  With chunks of 8
    timesteps a 40-step trajectory crosses five chunk boundaries: same result as one chunk holding every step, and as the
    per-step path (time MLP + emb_layers evaluated inside every step), graph and eager."""
    import worddiffusion_amd.engine as E
    args = make_args(device=DEV)
    labels = torch.tensor([1, 7, 3], dtype=torch.int64)
    words = ["MOVE", "a", "Zebra"]

    def run(chunk_rows, tabulate, use_graph=True):
        monkeypatch.setattr(E, "FILM_CHUNK_ROWS", chunk_rows)
        m = fill_module_(UNetModel(args=args, **SMALL), 5).to(DEV).eval()  # fresh engine: plans are cached per model
        diff = Diffusion(noise_steps=40, img_size=(32, 64), args=args)
        diff.tabulate_film = tabulate
        out = diff.sampling(m, None, 3, words, labels, args, seed=11, use_graph=use_graph)
        P = next(iter(m.engine._plans.values()))
        return out, getattr(P, "film_nchunks", 0)

    whole, n1 = run(1 << 20, True)
    chunked, n5 = run(16, True)
    assert n1 == 1 and n5 == 5
    assert torch.equal(whole, chunked)
    eager, _ = run(16, True, use_graph=False)
    assert torch.equal(chunked, eager)
    per_step, n0 = run(16, False)
    assert n0 == 0 and max_rel(per_step.cpu(), whole.cpu()) < 1e-5
