"""The oracle (oracle/*.py) against every golden vector produced by the reference itself."""
import numpy as np
import pytest
import torch

from oracle import ddpm_oracle as D
from oracle import unet_oracle as U
from tests._common import FWD_CASES, SMALL, golden_state_dict, load_golden, max_rel, rel_err


def test_primitives(golden_dir):
    g = load_golden(golden_dir, "primitives")
    t = torch.from_numpy(g["t"])
    assert torch.equal(U.timestep_embedding(t, 320), torch.from_numpy(g["temb320"]))
    assert torch.equal(U.timestep_embedding(t, 64), torch.from_numpy(g["temb64"]))
    assert torch.equal(U.positional_encoding(10, 320), torch.from_numpy(g["pe_10x320"]))
    assert torch.equal(U.positional_encoding(10, 64), torch.from_numpy(g["pe_10x64"]))
    for T in (1000, 600, 51):
        b, a, ah = D.schedule(T)
        assert torch.equal(b, torch.from_numpy(g[f"beta{T}"]))
        assert torch.equal(a, torch.from_numpy(g[f"alpha{T}"]))
        assert torch.equal(ah, torch.from_numpy(g[f"alpha_hat{T}"]))
    for w, ref in zip(g["words"], g["label_padding"]):
        assert D.label_padding(str(w)) == [int(v) for v in ref]
    assert int(g["num_tokens"]) == D.NUM_TOKENS and int(g["vocab_size"]) == 53


@pytest.mark.parametrize("tag", sorted(FWD_CASES))
def test_forward_matches_reference(golden_dir, tag):
    cfg, variant, phosc_on = FWD_CASES[tag]
    g = load_golden(golden_dir, tag)
    sd = golden_state_dict(g)
    # the key/shape list the oracle derives from the kwargs == the reference's state_dict
    shapes = U.state_dict_shapes(cfg, variant)
    assert [k for k, _ in shapes] == [str(k) for k in g["keys"]]
    assert [",".join(map(str, s)) for _, s in shapes] == [str(s) for s in g["shapes"]]
    orc = U.UNetOracle(cfg, sd, variant, phosc_on)
    taps = {}
    phosc = torch.from_numpy(g["phosc"]) if "phosc" in g.files else None
    with torch.no_grad():
        out = orc(torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["context"]),
                  torch.from_numpy(g["y"]), phosc, taps)
    assert max_rel(out, g["out"]) < 2e-5, tag
    for name in g.files:
        if not name.startswith("hook:"):
            continue
        key = name[5:]
        if key == "time_embed":
            continue  # before the label embedding is added; covered by the blocks that consume emb
        if key == "word_emb":
            ref = g[name]
            got = taps["context"][:, -ref.shape[1]:] if phosc_on else taps["context"]
            assert max_rel(got, ref) < 2e-5, (tag, key)
            continue
        assert key in taps, key
        assert max_rel(taps[key], g[name]) < 2e-5, (tag, key)


def test_ddpm_trajectory(golden_dir):
    g = load_golden(golden_dir, "ddpm_traj")
    T = int(g["T"])
    from tests._common import make_args  # noqa
    sd = {k: torch.from_numpy(v) for k, v in (
        (k, __import__("worddiffusion_amd.synthetic", fromlist=["x"]).synthetic_tensor(k, s, int(g["seed"])))
        for k, s in U.state_dict_shapes(SMALL, "phosc"))}
    orc = U.UNetOracle(SMALL, sd, "phosc", False)
    ctx = torch.tensor([D.label_padding(str(g["word"]))] * 3, dtype=torch.int64)
    y = torch.from_numpy(g["labels"])
    noise = torch.from_numpy(g["noise"])
    rec = []
    with torch.no_grad():
        x0 = D.sampling(lambda x, t: orc(x, t, ctx, y), noise[0], list(noise[1:]), T, rec)
    xs = torch.stack(rec)
    assert xs.shape == tuple(g["x_per_step"].shape)
    assert max_rel(xs, g["x_per_step"]) < 5e-5
    # sampling() returns x/0.18215 through the identity VAE then (x/2+.5).clamp(0,1)  (train.py:239-247)
    img = ((x0 / 0.18215) / 2 + 0.5).clamp(0, 1)
    assert float((img - torch.from_numpy(g["image"])).abs().max()) < 2e-4
    # noise_images
    _, _, ah = D.schedule(T)
    xt = D.noise_images(ah, torch.from_numpy(g["ni_x0"]), torch.from_numpy(g["ni_t"]), torch.from_numpy(g["ni_eps"]))
    assert torch.equal(xt, torch.from_numpy(g["ni_xt"]))


def _oracle(cfg, variant, seed, phosc_on=False):
    from worddiffusion_amd.synthetic import synthetic_tensor
    sd = {k: torch.from_numpy(synthetic_tensor(k, s, seed)) for k, s in U.state_dict_shapes(cfg, variant)}
    return U.UNetOracle(cfg, sd, variant, phosc_on)


@pytest.mark.parametrize("tag,cfgname", [("ddpm_traj_phosc_small", "SMALL"), ("ddpm_traj_phosc_full", "FULL")])
def test_phosc_sampling_loop(golden_dir, tag, cfgname):
    """The reference's train.Diffusion.sampling driving UNetModelPhosc(args.phosc=1) with a PHOSC vector
    (trainGWModifyCondition.py:272-273 call form): oracle loop == recorded reference trajectory."""
    import tests._common as TC
    cfg = getattr(TC, cfgname)
    g = load_golden(golden_dir, tag)
    T, n = int(g["T"]), g["labels"].shape[0]
    orc = _oracle(cfg, "phosc", int(g["seed"]), True)
    ctx = torch.tensor([D.label_padding(str(g["word"]))] * n, dtype=torch.int64)
    y, phosc = torch.from_numpy(g["labels"]), torch.from_numpy(g["phosc"])
    noise = torch.from_numpy(g["noise"])
    rec = []
    with torch.no_grad():
        x0 = D.sampling(lambda x, t: orc(x, t, ctx, y, phosc), noise[0], list(noise[1:]), T, rec)
    assert max_rel(torch.stack(rec), g["x_per_step"]) < 5e-5
    img = ((x0 / 0.18215) / 2 + 0.5).clamp(0, 1)
    assert float((img - torch.from_numpy(g["image"])).abs().max()) < 2e-4


def test_modify_condition_sampling_loop_and_alphabet(golden_dir):
    """trainModifyCondition.py: label_padding with the '_' alphabet (vocab 54), Diffusion.sampling with s_id = ones,
    T = 8 (every step) and the script's default T = 600 (checkpoints)."""
    p = load_golden(golden_dir, "primitives_modcond")
    for w, ref in zip(p["words"], p["label_padding"]):
        assert D.label_padding_underscore(str(w)) == [int(v) for v in ref]
    assert int(p["vocab_size"]) == len(D.C_CLASSES_UNDERSCORE) + D.NUM_TOKENS == 54
    assert str(p["c_classes"]) == D.C_CLASSES_UNDERSCORE and int(p["default_noise_steps"]) == 600
    g = load_golden(golden_dir, "ddpm_traj_modcond")
    cfg = dict(SMALL, vocab_size=int(g["vocab_size"]))
    orc = _oracle(cfg, "base", int(g["seed"]))
    for tag, n, every in (("T8", 3, 1), ("T600", 2, 100)):
        noise = torch.from_numpy(g[tag + "_noise"])
        T = noise.shape[0] + 1
        ctx = torch.tensor([D.label_padding_underscore(str(g["word"]))] * n, dtype=torch.int64)
        s_id = torch.ones(n, dtype=torch.int64)  # trainModifyCondition.py:565, whatever ``labels`` holds
        rec = []
        with torch.no_grad():
            x0 = D.sampling(lambda x, t: orc(x, t, ctx, s_id), noise[0], list(noise[1:]), T, rec)
        xs = torch.stack(rec[::every])
        ref = g["T8_x_per_step"] if tag == "T8" else g["T600_x_every100"]
        assert xs.shape == tuple(ref.shape)
        assert max_rel(xs, ref) < 5e-5, tag
        img = ((x0 / 0.18215) / 2 + 0.5).clamp(0, 1)
        assert float((img - torch.from_numpy(g[tag + "_image"])).abs().max()) < 2e-4, tag


@pytest.mark.parametrize("tag", ["skip", "full", "phosc"])
def test_step_skipping_sampler_matches_reference(golden_dir, tag):
    """regenerateFromtrain2.Diffusion.sampling3 (:465-648) as recorded from the reference's own class (600-step schedule; UNet on
    120 of the 599 iterations without fullSampling, deterministic update; every iteration and the stochastic update with it;
    PHOSC call form, start from a given x_t, epoch 12): oracle loop == recorded trajectory."""
    g = load_golden(golden_dir, "ddpm_traj_sampling3")
    assert str(g["lang"]) == "ENG" and int(g["max_chars"]) == D.MAX_CHARS and int(g["noise_steps"]) == 600
    for w, ref in zip(("MOVE", "a", "getting"), g["label_padding"]):
        assert D.label_padding(w) == [int(v) for v in ref]
    full, epoch = bool(g[tag + "_full"]), int(g[tag + "_epoch"])
    words = [str(w) for w in g[tag + "_words"]]
    phosc_on = (tag + "_phosc") in g.files
    orc = _oracle(SMALL, "phosc", int(g[tag + "_seed"]), phosc_on)
    ctx = torch.tensor([D.label_padding(w) for w in words], dtype=torch.int64)
    y = torch.from_numpy(g[tag + "_labels"])
    phosc = torch.from_numpy(g[tag + "_phosc"]) if phosc_on else None
    noise = torch.from_numpy(g[tag + "_noise"])
    assert int(g[tag + "_ndraws"]) == 599  # the start x + one draw per step i > 1 (made even where the update ignores it)
    x_start = torch.from_numpy(g[tag + "_x_t"]) if int(g[tag + "_noise_input"]) == 0 else noise[0]
    rec = []
    with torch.no_grad():
        x0, calls = D.sampling3(lambda x, t: orc(x, t, ctx, y, phosc) if phosc_on else orc(x, t, ctx, y), x_start, 600, epoch, full,
                                list(noise[1:]) if full else None, rec)
    assert calls == int(g[tag + "_calls"]) == (599 if full else 120)
    xs = torch.stack(rec[::10])
    ref = g[tag + "_x_every10calls"]
    assert xs.shape == tuple(ref.shape)
    # (|x| grows over the trajectory with random weights: compare each recorded state at its own scale)
    for i in range(xs.shape[0]):
        assert max_rel(xs[i], ref[i]) < 2e-4, (tag, i)
    assert max_rel(x0, g[tag + "_x_final"]) < 2e-4
    img = ((x0 / 0.18215) / 2 + 0.5).clamp(0, 1)
    assert float((img - torch.from_numpy(g[tag + "_image"])).abs().max()) < 2e-3


def test_train_step(golden_dir):
    from worddiffusion_amd.synthetic import synthetic_tensor
    g = load_golden(golden_dir, "train_step")
    shapes = U.state_dict_shapes(SMALL, "phosc")
    sd = {k: torch.from_numpy(synthetic_tensor(k, s, int(g["seed_model"]))).requires_grad_(True) for k, s in shapes}
    orc = U.UNetOracle(SMALL, sd, "phosc", False)
    _, _, ah = D.schedule(1000)
    x_t = D.noise_images(ah, torch.from_numpy(g["x0"]), torch.from_numpy(g["t"]), torch.from_numpy(g["eps"]))
    assert torch.equal(x_t, torch.from_numpy(g["x_t"]))
    pred = orc(x_t, torch.from_numpy(g["t"]), torch.from_numpy(g["context"]), torch.from_numpy(g["y"]))
    loss = torch.nn.functional.mse_loss(pred, torch.from_numpy(g["eps"]))
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    loss.backward()
    for name in g.files:
        if name.startswith("grad:"):
            k = name[5:]
            ref = torch.from_numpy(g[name]).double()
            # (the key-bias gradient of Word_Attention is analytically 0: absolute floor)
            assert float((sd[k].grad.double() - ref).norm()) < 1e-4 * float(ref.norm()) + 1e-7, k
    # EMA (train.py:151-159) and AdamW on a few tensors
    ema = {k: torch.from_numpy(synthetic_tensor(k, s, int(g["seed_ema"]))) for k, s in shapes}
    cur = {k: v.detach() for k, v in sd.items()}
    keys = [n[4:] for n in g.files if n.startswith("ema:")]
    D.ema_update(ema, cur, 0.995, keys)
    for k in keys:
        assert max_rel(ema[k], g["ema:" + k]) < 1e-6
    for name in g.files:
        if name.startswith("adamw:"):
            k = name[6:]
            p, _, _ = D.adamw_step(cur[k], sd[k].grad, torch.zeros_like(cur[k]), torch.zeros_like(cur[k]), 1)
            assert max_rel(p, g[name]) < 1e-6, k
