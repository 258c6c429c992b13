"""``Diffusion`` / ``EMA`` / ``label_padding`` with the reference's call surface (``train.py:42-52,140-251``).

The reverse loop of ``Diffusion.sampling`` (``train.py:221-236``) runs entirely on the device: the step-invariant
conditioning (word embedding, cross-attention K/V) is computed once, one denoising step (UNet forward + the
``x <- 1/sqrt(a) (x - (1-a)/sqrt(1-ah) eps) + sqrt(b) z`` update with on-device Philox noise + timestep decrement)
is captured into a hipGraph and replayed ``noise_steps - 1`` times; there is no per-step host->device traffic.

Facts preserved from the reference: the method is ``sampling`` (``sample`` is provided as
an alias because ``sampling.py:119`` calls it); index 0 of the schedule is never used; with ``cfg_scale > 0``
the reference runs the UNet twice on identical inputs and ``lerp(a, a, w) == a`` bit-for-bit, so one forward per
step is executed (``forwards_per_step = 2`` re-enables the literal behaviour for timing comparisons).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _native as N

C_CLASSES = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
LETTER2INDEX = {c: i for i, c in enumerate(C_CLASSES)}
MAX_CHARS = 10
TOKENS = {"PAD_TOKEN": 52}
NUM_TOKENS = len(TOKENS)
VOCAB_SIZE = len(C_CLASSES) + NUM_TOKENS
# the alphabet of trainModifyCondition.py:68 / trainGWModifyCondition.py:53: the 52 letters + '_' (a space is written as
# '_', :169) -> 53 classes, vocab_size 54; PAD stays 52, so '_' is id 53 and 'z' (51 + 1) still collides with PAD as in
# train.py
C_CLASSES_UNDERSCORE = C_CLASSES + "_"
LETTER2INDEX_UNDERSCORE = {c: i for i, c in enumerate(C_CLASSES_UNDERSCORE)}
VOCAB_SIZE_UNDERSCORE = len(C_CLASSES_UNDERSCORE) + NUM_TOKENS


def _pad_ids(labels: str, table, num_tokens: int, max_len: int) -> List[int]:
    try:
        ll = [table[ch] + num_tokens for ch in labels]
    except KeyError as e:  # the reference raises KeyError from ``letter2index[i]`` as well (train.py:45)
        raise KeyError(f"character {e.args[0]!r} of {labels!r} is not in the label alphabet") from None
    if len(ll) > max_len:
        raise ValueError(f"word longer than {max_len} characters: {labels!r}")
    return ll + [TOKENS["PAD_TOKEN"]] * (max_len - len(ll))


def label_padding(labels: str, num_tokens: int = NUM_TOKENS, max_len: int = MAX_CHARS) -> List[int]:
    """``train.py:42-52``: letter indices shifted by ``num_tokens`` and right-padded with PAD (52) to 10."""
    return _pad_ids(labels, LETTER2INDEX, num_tokens, max_len)


def label_padding_underscore(labels: str, num_tokens: int = NUM_TOKENS, max_len: int = MAX_CHARS) -> List[int]:
    """``trainModifyCondition.py:166-180`` (same in ``trainGWModifyCondition.py:64-78``): ``labels.replace(" ", "_")``, then
    the 53-class alphabet (``'_'`` -> 52 + num_tokens = 53); models fed with it are built with ``vocab_size = 54``."""
    return _pad_ids(labels.replace(" ", "_"), LETTER2INDEX_UNDERSCORE, num_tokens, max_len)


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


class EMA:
    """``train.py:140-170``."""

    def __init__(self, beta):
        self.beta = beta
        self.step = 0

    def update_model_average(self, ma_model, current_model):
        lib = N.lib()
        for cur, ma in zip(current_model.parameters(), ma_model.parameters()):
            if not ma.is_cuda:
                raise N.NativeError("EMA.update_model_average runs on the GPU only (no CPU fallback)")
            assert ma.is_contiguous() and cur.is_contiguous() and ma.dtype == torch.float32
            N.check(lib.wd_ema_update(ma.data_ptr(), cur.data_ptr(), ma.numel(), float(self.beta),
                                      _stream_ptr(ma.device)), "wd_ema_update")
        # the kernel wrote the averaged parameters behind autograd's back: the engine of ``ma_model`` (which keys its packed
        # operands on the parameters' state, engine._signature) must repack before the next forward / sampling
        from .engine import note_native_write
        note_native_write()

    def step_ema(self, ema_model, model, step_start_ema=2000):
        if self.step < step_start_ema:
            self.reset_parameters(ema_model, model)
            self.step += 1
            return
        self.update_model_average(ema_model, model)
        self.step += 1

    def reset_parameters(self, ema_model, model):
        ema_model.load_state_dict(model.state_dict())


class Diffusion:
    """``train.py:174-251`` (T=1000) / ``trainModifyCondition.py:515-622`` (T=600)."""

    def __init__(self, noise_steps=1000, beta_start=1e-4, beta_end=0.02, img_size=(64, 128), args=None):
        self.noise_steps = noise_steps
        self.beta_start = beta_start
        self.beta_end = beta_end
        dev = getattr(args, "device", "cpu") if args is not None else "cpu"
        self.beta = self.prepare_noise_schedule().to(dev)
        self.alpha = 1. - self.beta
        self.alpha_hat = torch.cumprod(self.alpha, dim=0)
        self.img_size = img_size
        self.device = dev
        self._tables = None
        self._graphs = {}
        self.forwards_per_step = 1
        self.tabulate_film = os.environ.get("WDIFF_FILM_TABLE", "1") != "0"
        self.last_stats = {}

    def prepare_noise_schedule(self):
        return torch.linspace(self.beta_start, self.beta_end, self.noise_steps)

    def sample_timesteps(self, n):
        return torch.randint(low=1, high=self.noise_steps, size=(n,))

    # --------------------------------------------------------------------------------------------------
    def noise_images(self, x, t, eps=None, seed=None):
        """``train.py:190-194``; the noise comes from the device Philox stream (or ``eps`` if given)."""
        if not x.is_cuda:
            raise N.NativeError("Diffusion.noise_images runs on the GPU only (no CPU fallback)")
        lib = N.lib()
        st = _stream_ptr(x.device)
        x = x.contiguous().float()
        n = x[0].numel()
        if eps is None:
            eps = torch.empty_like(x)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else seed
            N.check(lib.wd_randn(eps.data_ptr(), x.shape[0], n, seed, 0, 1, st), "wd_randn")
        ah = self.alpha_hat.cpu()
        sa, sb = torch.sqrt(ah).to(x.device), torch.sqrt(1 - ah).to(x.device)
        out = torch.empty_like(x)
        t = t.to(x.device).long().contiguous()
        N.check(lib.wd_noise_images(x.data_ptr(), eps.data_ptr(), t.data_ptr(), sa.data_ptr(), sb.data_ptr(),
                                    x.shape[0], n, out.data_ptr(), st), "wd_noise_images")
        return out, eps

    def _step_tables(self, device):
        """ca = 1/sqrt(alpha), cb = (1-alpha)/sqrt(1-alpha_hat), cs = sqrt(beta): the three per-t scalars of
        ``train.py:236``, evaluated with the reference's fp32 op order."""
        if self._tables is None or self._tables[0].device != torch.device(device):
            a, ah, b = self.alpha.cpu(), self.alpha_hat.cpu(), self.beta.cpu()
            ca = 1 / torch.sqrt(a)
            cb = (1 - a) / (torch.sqrt(1 - ah))
            cs = torch.sqrt(b)
            self._tables = tuple(t.to(device).contiguous() for t in (ca, cb, cs))
        return self._tables

    # --------------------------------------------------------------------------------------------------
    def _denoise(self, model, n, text_features, labels, phosc, device, x_T=None, noise=None, seed=None,
                 sample_offset=0, record=None, use_graph=True, calls_model=None, deterministic=False):
        lib = N.lib()
        eng = model.engine
        eng.refresh_weights()
        eng.check_ids(text_features, labels, phosc)  # host tensors here: no device sync
        h, w = self.img_size[0] // 8, self.img_size[1] // 8
        ctx_len = text_features.shape[1]
        phosc_len = 0 if phosc is None else phosc.shape[1]
        T = self.noise_steps
        P = eng.plan(n, h, w, ctx_len, phosc_len, film_steps=T if self.tabulate_film else 0)
        ca, cb, cs = self._step_tables(device)
        if deterministic:  # regenerateFromtrain2.py:618 drops the sqrt(beta) * noise term
            cs = torch.zeros_like(cs)
        T = self.noise_steps
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
        npix = P.x_in[0].numel()

        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            st = side.cuda_stream
            if x_T is not None:
                P.x_in.copy_(x_T.to(device))
            else:
                N.check(lib.wd_randn(P.x_in.data_ptr(), n, npix, seed, sample_offset, 0, st), "wd_randn")
            eng.load_inputs(P, None, None, text_features.to(device), labels.to(device) if labels is not None else None,
                            phosc.to(device) if phosc is not None else None, check=False)
            t_dev = P.t_dev
            t_dev.fill_(T - 1)
            P.t_in.fill_(T - 1)
            zbuf = torch.zeros_like(P.x_in) if noise is not None else None
            P.run_cond(st)
            P.run_film(st)  # time MLP of every timestep; the FiLM rows follow per chunk of timesteps (P.film_prepare)
            P.film_prepare(T - 1, st)  # before the capture: the captured step only reads the table

            def one_step(stream, forward=True):
                if forward:
                    for _ in range(self.forwards_per_step):
                        P.run_step(stream)
                N.check(lib.wd_ddpm_step(P.x_in.data_ptr(), P.out.data_ptr(), n, npix, ca.data_ptr(), cb.data_ptr(),
                                         cs.data_ptr(), t_dev.data_ptr(), zbuf.data_ptr() if zbuf is not None else None,
                                         seed, sample_offset, stream), "wd_ddpm_step")
                N.check(lib.wd_advance_timestep(t_dev.data_ptr(), -1, P.t_in.data_ptr(), n, stream),
                        "wd_advance_timestep")

            def capture(forward):
                N.check(lib.wd_graph_begin(st), "wd_graph_begin")
                try:
                    one_step(st, forward)
                finally:
                    g = C.c_void_p()
                    rc = lib.wd_graph_end(st, C.byref(g))
                N.check(rc, "wd_graph_end")
                return g

            gexec = gskip = None
            if use_graph and record is None:
                gexec = capture(True)
                if calls_model is not None:
                    gskip = capture(False)  # steps that reuse the previous predicted noise: update only
            k = 0
            ncalls = 0
            for i in reversed(range(1, T)):
                if record is not None:
                    record.append(P.x_in.clone())
                if zbuf is not None and i > 1:
                    zbuf.copy_(noise[k].to(device))
                    k += 1
                fwd = calls_model is None or bool(calls_model(i))
                ncalls += int(fwd)
                if fwd:
                    P.film_prepare(i, st)  # FiLM rows of timestep i (computed per chunk of timesteps, see engine.plan)
                if gexec is not None:
                    N.check(lib.wd_graph_launch(gexec if fwd else gskip, st), "wd_graph_launch")
                else:
                    one_step(st, fwd)
            x = P.x_in.clone()
        torch.cuda.current_stream(device).wait_stream(side)
        if gexec is not None:
            side.synchronize()
            lib.wd_graph_destroy(gexec)
            if gskip is not None:
                lib.wd_graph_destroy(gskip)
        self.last_stats = dict(steps=T - 1, forwards_per_step=self.forwards_per_step, graph=gexec is not None,
                               seed=seed, sample_offset=sample_offset, model_calls=ncalls)
        return x

    def _text_features(self, x_text, n, underscore=False):
        words = [x_text] * n if isinstance(x_text, str) else list(x_text)
        if len(words) != n:
            raise ValueError("x_text must be one word or a list of n words")
        pad = label_padding_underscore if underscore else label_padding
        return torch.tensor(np.array([pad(w, NUM_TOKENS) for w in words], dtype="int64"))

    def _finish(self, x, vae, args):
        """``train.py:238-250``: latents / 0.18215 -> vae.decode -> [0,1] image (vae is duck-typed)."""
        latent = getattr(args, "latent", True)
        if latent == True:  # noqa: E712  (argparse type=bool quirk of the reference)
            if vae is None:
                return x
            latents = 1 / 0.18215 * x
            image = vae.decode(latents).sample
            image = (image / 2 + 0.5).clamp(0, 1)
            image = image.cpu().permute(0, 2, 3, 1).numpy()
            image = torch.from_numpy(image)
            return image.permute(0, 3, 1, 2)
        raise NotImplementedError("latent=False (pixel-space UNet) is dead code in the reference (SURVEY.md 0.5)")

    @torch.no_grad()
    def sampling(self, model, vae, n, x_text, labels, args, mix_rate=None, cfg_scale=3, phoscLabels=None,
                 noise=None, x_T=None, seed=None, sample_offset=0, record=None, use_graph=True, underscore=None):
        """``train.py:200`` signature; extra keyword-only style arguments (phoscLabels, noise, x_T, seed,
        sample_offset) serve the PHOSC variant (``trainGWModifyCondition.py:249``), the parity tests and
        rank-sharded sampling.  ``vae=None`` returns the denoised latents.  ``underscore``: word ids from the 53-class
        ``'_'`` alphabet of the ModifyCondition scripts (default: when the model's table has the 54 rows that alphabet
        needs).  Like the reference (``train.py:201,238``) the model is put in eval mode for the loop and in TRAIN mode
        afterwards, whatever mode it came in."""
        if mix_rate is not None:
            raise NotImplementedError("mix_rate interpolation (unet.py:1558-1573)")
        if underscore is None:
            underscore = int(model.word_emb.embedding.weight.shape[0]) == VOCAB_SIZE_UNDERSCORE
        model.eval()
        device = torch.device(getattr(args, "device", self.device))
        if device.type != "cuda":
            raise N.NativeError("Diffusion.sampling runs on an MI355X only (no CPU fallback)")
        if self.img_size is None or not (getattr(args, "latent", True) == True):  # noqa: E712
            raise NotImplementedError("latent=False")
        tf = self._text_features(x_text, n, underscore)
        phosc = None
        if getattr(args, "phosc", 0) == 1 or getattr(args, "phos", 0) == 1:
            if phoscLabels is None:
                raise ValueError("args.phosc/phos set but phoscLabels missing")
            phosc = phoscLabels.int()
        try:
            x = self._denoise(model, n, tf, labels, phosc, device, x_T=x_T, noise=noise, seed=seed,
                              sample_offset=sample_offset, record=record, use_graph=use_graph)
        finally:
            model.train()  # train.py:238 (unconditional)
        return self._finish(x, vae, args)

    sample = sampling  # sampling.py:119 / full_sampling.py:167 call .sample(...)

    @staticmethod
    def sampling3_calls_model(i, noise_steps, epoch=0):
        """The step-skipping predicate of ``regenerateFromtrain2.py:536`` (literal; its last clause ``epoch>50==0`` is a
        chained comparison that is always False): the UNet runs at ``i == T-1`` and whenever ``i % 5 == 0``."""
        return bool(i % 100 == 0 or i % 5 == 0 or i == noise_steps or i == noise_steps - 1 or (epoch > 3 and i % 25 == 0) or
                    (epoch > 5 and i % 15 == 0) or (epoch > 10 and i % 10 == 0) or (epoch > 50 == 0))

    @torch.no_grad()
    def sampling3(self, epoch, x_t, words, phoscLabels, model, model1, vae, emaOld, noiseInput, n, x_text, labels, args,
                  mix_rate=None, cfg_scale=3, seed=None, sample_offset=0, use_graph=True, x_T=None, noise=None, record=None):
        """Bulk-regeneration sampler of ``regenerateFromtrain2.py:465-648`` (same argument order): the predicted noise is
        refreshed only on the steps of ``sampling3_calls_model`` (1 in 5) and reused in between, and unless
        ``args.fullSampling`` the update is deterministic (no ``sqrt(beta) * noise`` term, ``:618``).  ``noiseInput == 0``
        starts from ``x_t`` instead of fresh noise (``:518-519``).  Returns ``(0, [images], images)`` like the reference when
        a ``vae`` is given, the denoised latents otherwise.  The per-step ``flagGen.txt`` poll (``:523-530``) is not
        reproduced.  ``x_T`` / ``noise`` / ``record`` (as in ``sampling``): the start latent, the per-step draws and a list that
        receives every step's x - how the tests replay the trajectories recorded from the reference's own loop
        (``tests/golden/ddpm_traj_sampling3.npz``, ``oracle/make_golden_sampling3.py``)."""
        if mix_rate is not None:
            raise NotImplementedError("mix_rate interpolation (unet.py:1558-1573)")
        if emaOld == 1:
            model = model1
        model.eval()  # and it stays in eval mode: ``#model.train()`` is commented out at regenerateFromtrain2.py:622
        device = torch.device(getattr(args, "device", self.device))
        if device.type != "cuda":
            raise N.NativeError("Diffusion.sampling3 runs on an MI355X only (no CPU fallback)")
        if isinstance(x_text, str) or len(x_text) <= 1:
            word_list = [x_text if isinstance(x_text, str) else x_text[0]] * n
        else:
            word_list = list(words)
        tf = self._text_features(word_list, n, int(model.word_emb.embedding.weight.shape[0]) == VOCAB_SIZE_UNDERSCORE)
        phosc = None
        if getattr(args, "phosc", 0) == 1 or getattr(args, "phos", 0) == 1:
            if phoscLabels is None:
                raise ValueError("args.phosc/phos set but phoscLabels missing")
            phosc = phoscLabels.int()
        full = bool(getattr(args, "fullSampling", False))
        T = self.noise_steps
        x = self._denoise(model, n, tf, labels, phosc, device, x_T=x_t if noiseInput == 0 else x_T, noise=noise, seed=seed,
                          sample_offset=sample_offset, record=record, use_graph=use_graph,
                          calls_model=None if full else (lambda i: self.sampling3_calls_model(i, T, epoch)),
                          deterministic=not full)
        if vae is None:
            return x
        image = self._finish(x, vae, args)
        return 0, [image], image

    def sampling_modify_condition(self, model, vae, latents, x_text, words, n, labels, args, **kw):
        """Argument order of ``trainModifyCondition.py:545``; that variant feeds writer id 1 for every sample
        (``s_id = torch.ones(...)``, ``:565``) whatever ``labels`` holds, reads its word ids from the ``'_'`` alphabet
        (``:166-180``) and never runs the second forward (``if 0:`` at ``:590``).  Its schedule is the caller's:
        ``Diffusion()`` of that script defaults to ``noise_steps = 600`` (``:516``) - pass it to the constructor."""
        s_id = torch.ones(n, dtype=torch.int64)
        kw.setdefault("underscore", True)
        return self.sampling(model, vae, n, x_text, s_id, args, cfg_scale=0, **kw)

    def sampling_phosc(self, model, vae, n, x_text, phoscLabels, labels, args, mix_rate=None, cfg_scale=3, **kw):
        """Argument order of ``trainGWModifyCondition.py:249``."""
        return self.sampling(model, vae, n, x_text, labels, args, mix_rate=mix_rate, cfg_scale=cfg_scale,
                             phoscLabels=phoscLabels, **kw)
