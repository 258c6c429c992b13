"""GPU tests of the rows either side of the hot path: the bulk sampling driver on the reference's
own gt lines - one ``regenerate()`` row against ``ddpm_oracle.sampling`` of that row with the device's own noise read back -
for the base model and for ``UNetModelPhosc`` with PHOSC vectors wired through ``phosc_of``; the cached-latent training
epoch (``latents.train_epoch`` around ``TrainStep``) against the same loop on the CPU oracle."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ddpm_oracle as D  # noqa: E402
from oracle import unet_oracle as U  # noqa: E402
from tests._common import SMALL, load_golden, make_args, max_rel  # noqa: E402
from worddiffusion_amd import Diffusion, UNetModel, UNetModelPhosc  # noqa: E402
from worddiffusion_amd import _native as N  # noqa: E402
from worddiffusion_amd.driver import make_phosc_of, read_gt, regenerate, writer_dict  # noqa: E402
from worddiffusion_amd.synthetic import fill_module_, synthetic_tensor  # noqa: E402

DEV = "cuda:0"


def device_noise_of_row(seed, row, T, n):
    """x_T and the per-step z of global row ``row`` exactly as the sampler draws them on the device: x_T = wd_randn stream 0;
    z_t = what wd_ddpm_step adds at timestep t, read back by running the step on x = eps = 0 with ca = 1, cb = 0, cs = 1."""
    lib = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    x_T = torch.empty(1, n, device=DEV)
    N.check(lib.wd_randn(x_T.data_ptr(), 1, n, seed, row, 0, st), "wd_randn")
    ones, zeros = torch.ones(T, device=DEV), torch.zeros(T, device=DEV)
    eps = torch.zeros(1, n, device=DEV)
    zs = []
    for t in range(T - 1, 1, -1):
        x = torch.zeros(1, n, device=DEV)
        t_dev = torch.tensor([t], dtype=torch.int32, device=DEV)
        N.check(lib.wd_ddpm_step(x.data_ptr(), eps.data_ptr(), 1, n, ones.data_ptr(), zeros.data_ptr(), ones.data_ptr(),
                                 t_dev.data_ptr(), None, seed, row, st), "wd_ddpm_step")
        zs.append(x.cpu())
    return x_T.cpu(), zs


def write_alphabet_csv(golden_dir, path, version="eng"):
    """The reference's shape-count table travels as data inside tests/golden/phosc.npz; written back in its csv form."""
    g = load_golden(golden_dir, "phosc")
    with open(path, "w") as f:
        for letter, row in zip(g[f"{version}:letters"], g[f"{version}:table"]):
            f.write(",".join([str(letter)] + [str(int(v)) for v in row]) + "\r\n")
    return path


@pytest.mark.parametrize("variant", ["base", "phosc"])
def test_regenerate_row_matches_oracle_on_reference_gt_lines(golden_dir, tmp_path, variant):
    rows = read_gt(os.path.join(golden_dir, "gt_samples.txt"))[:14]  # the IAM lines: 2 writers, words of the 52-letter alphabet
    wr = writer_dict(rows)
    assert wr == {"049": 0, "537": 1}
    T, seed = 7, 77
    phosc_of = None
    if variant == "phosc":
        phosc_of = make_phosc_of(write_alphabet_csv(golden_dir, str(tmp_path / "Alphabet.csv")))
        assert phosc_of("Members").shape == (769,) and phosc_of("Members").dtype == torch.int64
        args = make_args(device=DEV, phosc=1)
        m = fill_module_(UNetModelPhosc(args=args, **SMALL), 9).to(DEV).eval()
    else:
        args = make_args(device=DEV)
        m = fill_module_(UNetModel(args=args, **SMALL), 9).to(DEV).eval()
    diff = Diffusion(noise_steps=T, img_size=(32, 64), args=args)
    start, lat = regenerate(m, diff, rows, wr, args, batch=5, seed=seed, rank=0, world=1, phosc_of=phosc_of,
                            out_dir=str(tmp_path / "out"))
    assert start == 0 and lat.shape == (14, 4, 4, 8)
    assert sorted(os.listdir(tmp_path / "out")) == sorted(r[1] + ".npy" for r in rows)
    # two ranks, other batch size: same rows (global-row-indexed noise)
    parts = [regenerate(m, diff, rows, wr, args, batch=3, seed=seed, rank=r, world=2, phosc_of=phosc_of)[1] for r in range(2)]
    assert max_rel(torch.cat(parts), lat) < 1e-5
    # ---- rows 0, 6 and 11 against the oracle loop with the same noise
    sd = {k: torch.from_numpy(synthetic_tensor(k, s, 9)) for k, s in U.state_dict_shapes(SMALL, variant)}
    orc = U.UNetOracle(SMALL, sd, variant, variant == "phosc")
    for r in (0, 6, 11):
        s_id, image, word = rows[r]
        x_T, zs = device_noise_of_row(seed, r, T, 4 * 4 * 8)
        ctx = torch.tensor([D.label_padding(word)], dtype=torch.int64)
        y = torch.tensor([wr[s_id]], dtype=torch.int64)
        ph = phosc_of(word)[None] if phosc_of is not None else None
        with torch.no_grad():
            ref = D.sampling(lambda x, t: orc(x, t, ctx, y, ph), x_T.reshape(1, 4, 4, 8), [z.reshape(1, 4, 4, 8) for z in zs], T)
        assert max_rel(lat[r], ref[0]) < 1e-4, (variant, r, word)
        assert np.allclose(np.load(tmp_path / "out" / f"{image}.npy"), lat[r].numpy())


def test_driver_command_line_writes_what_regenerate_returns(golden_dir, tmp_path, monkeypatch):
    """``python -m worddiffusion_amd.driver`` end to end (flag set of full_sampling.py:40-66, checkpoint layout of :98-110): a
    saved ``models/ema_ckpt.pt`` is loaded with ``weights_only=True``, the gt file is read, the rows are sampled with the
    step-skipping sampler over the '_' alphabet and written as latents - the files equal what ``regenerate()`` returns for the
    same seed on a model loaded from the same checkpoint."""
    from worddiffusion_amd import driver
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    gt = tmp_path / "gt.txt"
    lines = [ln for ln in open(os.path.join(golden_dir, "gt_samples.txt")).read().splitlines() if ln.strip()]
    rows_txt = [ln for ln in lines if not ln.startswith("#")][:5]
    gt.write_text("\n".join(rows_txt) + "\n")
    rows = read_gt(str(gt))
    assert len(rows) == len(rows_txt)
    args = make_args(device=DEV)
    args.fullSampling, args.latent = False, True
    kw = dict(image_size=(64, 256), in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=(1, 1),
              channel_mult=(1, 1), num_heads=2, num_classes=339, context_dim=64, vocab_size=54, max_seq_len=10)
    m = fill_module_(UNetModel(args=args, **kw), 17)
    os.makedirs(tmp_path / "run" / "models")
    torch.save(m.state_dict(), tmp_path / "run" / "models" / "ema_ckpt.pt")
    out = tmp_path / "out"
    driver.main(["--gt_train", str(gt), "--models_path", str(tmp_path / "run"), "--save_path", str(out), "--writer_dict",
                 str(tmp_path / "writers.json"), "--batch_size", "2", "--emb_dim", "64", "--num_heads", "2", "--noise_steps", "11",
                 "--vocab_size", "54", "--skip_steps", "1", "--seed", "5"])
    m2 = UNetModel(args=args, **kw).to(DEV)
    m2.load_state_dict(torch.load(tmp_path / "run" / "models" / "ema_ckpt.pt", map_location=DEV, weights_only=True))
    m2 = m2.eval().requires_grad_(False)
    diff = Diffusion(noise_steps=11, img_size=(64, 256), args=args)
    wr = writer_dict(rows, str(tmp_path / "writers.json"))
    _, ref = regenerate(m2, diff, rows, wr, args, batch=2, seed=5, skip_steps=True, rank=0, world=1)
    assert ref.shape == (len(rows), 4, 8, 32) and torch.isfinite(ref).all()
    for (_, image, _), r in zip(rows, ref):
        got = np.load(out / "images" / f"{image}.npy")
        assert np.array_equal(got, r.numpy()), image


def test_cached_latent_epoch_matches_oracle_loop(golden_dir, tmp_path):
    """One epoch of the train.py batch loop fed from the cached-latent container (trainModifyCondition.py vaeFromDict=1:
    ``latents = images``) through ``latents.train_epoch`` -> ``TrainStep``; the same batches through the CPU oracle under
    autograd + torch.optim.AdamW.  Timesteps and noise are taken from the step's own streams (host RNG / device Philox) and
    replayed on the CPU."""
    from worddiffusion_amd.latents import CachedLatentDataset, LatentCache, save_latent_cache, train_epoch
    from worddiffusion_amd.optim import FusedAdamW
    from worddiffusion_amd.training import TrainStep
    rows = read_gt(os.path.join(golden_dir, "gt_samples.txt"))[:14]
    wr = writer_dict(rows)
    rs = np.random.RandomState(3)
    lat = {r[1] + ".png": torch.from_numpy((rs.standard_normal((1, 4, 4, 8)) * 0.18215 * 5).astype(np.float32)) for r in rows}
    cache = LatentCache(save_latent_cache(str(tmp_path / "lat.safetensors"), lat))
    ds = CachedLatentDataset(rows, wr, cache)
    cfg, seed, B = SMALL, 23, 4
    m = fill_module_(UNetModel(args=make_args(device=DEV), **cfg), seed).to(DEV).train()
    ema_m = copy.deepcopy(m).eval().requires_grad_(False)
    opt = FusedAdamW(m.parameters(), lr=1e-4, ema_model=ema_m, ema_beta=0.995, step_start_ema=2000)
    diff = Diffusion(noise_steps=1000, img_size=(32, 64), args=make_args(device=DEV))
    step = TrainStep(m, diff, opt, seed=5)
    seen = []
    orig_call = step.__call__

    class Spy:
        """records what the loop fed and what the step drew (t from the host RNG, eps from the device stream)"""

        def __call__(self, latents, words, s_id, phoscLabels=None):
            loss = orig_call(latents, words, s_id, phoscLabels=phoscLabels)
            torch.cuda.synchronize()
            seen.append(dict(x=latents.cpu(), ctx=words.cpu(), y=s_id.cpu(), t=step._P.t_in.cpu().clone(),
                             eps=step._eps.cpu().clone(), loss=float(loss.cpu())))
            return loss

    torch.manual_seed(1)
    res = train_epoch(Spy(), ds, B, DEV, epoch=0, seed=9, max_batches=30)
    assert res["batches"] == 3 and res["images"] == 12 and len(seen) == 3  # 14 rows, drop_last
    assert abs(res["mean_loss"] - np.mean([s["loss"] for s in seen])) < 1e-5
    # the batches are the cached tensors of the rows the loader says it took
    order = [n for b in ds.batches(B, seed=9, epoch=0, pin=False) for n in b["image_names"]]
    assert torch.equal(seen[0]["x"][0], lat[order[0]][0]) and torch.equal(seen[2]["x"][3], lat[order[11]][0])
    # ---- the same three steps on the CPU
    sd = {k: torch.from_numpy(synthetic_tensor(k, s, seed)).requires_grad_(True) for k, s in U.state_dict_shapes(cfg, "base")}
    init = {k: v.detach().clone() for k, v in sd.items()}
    orc = U.UNetOracle(cfg, sd, "base", False)
    topt = torch.optim.AdamW(list(sd.values()), lr=1e-4)
    _, _, ah = D.schedule(1000)
    for s in seen:
        x_t = D.noise_images(ah, s["x"], s["t"], s["eps"])
        loss = torch.nn.functional.mse_loss(orc(x_t, s["t"], s["ctx"], s["y"]), s["eps"])
        topt.zero_grad()
        loss.backward()
        topt.step()
        assert abs(float(loss.detach()) - s["loss"]) < 2e-4 * max(1.0, abs(s["loss"]))
    got = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if v.grad is None or v.numel() < 256:
            continue
        upd = (v.detach() - init[k]).double().norm()
        assert float((got[k].double() - v.detach().double()).norm() / (upd + 1e-30)) < 0.05, k
    # EMA warm-up (first 2000 steps: plain copy of the weights, train.py:161-165)
    for k, v in ema_m.state_dict().items():
        assert torch.equal(v.cpu(), got[k]), k
