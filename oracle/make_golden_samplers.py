#!/usr/bin/env python3
"""Golden vectors for the sampler LOOPS the reference's variant scripts run, generated from the REFERENCE itself.

TEST INFRASTRUCTURE - not part of the product.  Run once, in the build container:

    python oracle/make_golden_samplers.py /root/reference tests/golden

Same import stand-ins as ``oracle/make_golden.py`` (plus empty ``torchvision.transforms`` / ``torchvision.utils`` modules
for ``trainModifyCondition.py``'s import lines).  Only tensors are written.

  ddpm_traj_phosc_small.npz / ddpm_traj_phosc_full.npz
      the reference's ``train.Diffusion.sampling`` (train.py:200-251) driving the reference's
      ``unetPhosc.UNetModelPhosc`` built with ``args.phosc = 1``, through a call adapter that supplies the PHOSC vector the
      way ``trainGWModifyCondition.py:272-273`` calls the model (``model(x, phoscLabels, timesteps=t, context=..., y=...)``) -
      that script itself cannot be imported (SyntaxError at its line 445), its loop body is train.py's.
      small: 64-channel model, 37-int vector, n = 3, T = 8;  full: the 320-channel latent config, 769 ints, n = 2, T = 5.
  ddpm_traj_modcond.npz
      ``trainModifyCondition.Diffusion.sampling`` (trainModifyCondition.py:545-611: keyword call of ``unet.UNetModel``,
      ``s_id = ones`` whatever ``labels`` holds, ``'_'`` label alphabet of 54 ids) - T = 8 with every step's x, and the
      script's default T = 600 (599 steps, checkpoints every 100 steps), recorded noise.
  fwd_base_full_charlevel.npz
      ``unet.UNetModel`` with ``args.charLevelEmb = 1`` (the default of ``unet.py:1871``) on the inputs of
      ``fwd_base_full.npz``: the output is asserted here to be bit-identical to ``charLevelEmb = 0``.
  primitives_modcond.npz
      ``trainModifyCondition.label_padding`` / ``vocab_size`` / ``c_classes`` (space -> '_', id 53).
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as MG  # noqa: E402
from worddiffusion_amd.synthetic import fill_module_, synthetic_inputs  # noqa: E402

np_ = MG.np_


class IdentityVAE:
    def decode(self, z):
        return types.SimpleNamespace(sample=z)


class NoiseRecorder:
    """Replaces torch.randn / randn_like by draws from one seeded generator and keeps every draw."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.rec = []
        self._randn, self._randn_like = torch.randn, torch.randn_like

    def __enter__(self):
        def randn(*size, **kw):
            shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list)) else size
            z = self._randn(tuple(shape), generator=self.g)
            self.rec.append(np_(z).copy())
            return z

        torch.randn = randn
        torch.randn_like = lambda x, **kw: randn(tuple(x.shape))
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def spy(model, keep_every=1):
    xs, n = [], [0]
    orig = model.forward

    def fwd(x, *a, **k):
        if n[0] % keep_every == 0:
            xs.append(np_(x).copy())
        n[0] += 1
        return orig(x, *a, **k)

    model.forward = fwd
    return xs, orig


def gen_phosc_traj(outdir, tag, ref_train, ref_phosc, cfg, seed, n, T, phosc_len, hw, word):
    args = MG.make_args(phosc=1)
    torch.manual_seed(0)
    model = ref_phosc.UNetModelPhosc(args=args, **cfg).eval()
    fill_module_(model, seed)
    inp = synthetic_inputs(n, seed=seed + 2, hw=hw, num_classes=cfg["num_classes"], phosc_len=phosc_len)
    phosc, labels = inp["phosc"], inp["y"]
    xs, orig = spy(model, keep_every=2)  # train.py:223-228 calls the model twice per step with identical inputs

    class Adapter:
        """train.py:223's positional call ``model(x, None, t, text_features, labels, mix_rate=...)`` -> the keyword call of
        trainGWModifyCondition.py:272-273 with the PHOSC vector."""

        def eval(self):
            model.eval()

        def train(self):
            model.train()

        def __call__(self, x, _none, t, text_features, y, mix_rate=None):
            return model.forward(x, phosc, timesteps=t, context=text_features, y=y)

    diff = ref_train.Diffusion(noise_steps=T, img_size=(hw[0] * 8, hw[1] * 8), args=args)
    with NoiseRecorder(seed * 7 + 1) as nr:
        img = diff.sampling(Adapter(), IdentityVAE(), n, word, labels, args)
    model.forward = orig
    model.eval()
    np.savez_compressed(os.path.join(outdir, tag + ".npz"), noise=np.stack(nr.rec), x_per_step=np.stack(xs),
                        labels=np_(labels), phosc=np_(phosc), word=np.array(word), T=np.int64(T), image=np_(img),
                        seed=np.int64(seed), phosc_len=np.int64(phosc_len))
    print(f"[golden] {tag}: {len(xs)} steps, {len(nr.rec)} noise draws, final |x|={np.abs(np_(img)).mean():.4f}")


def gen_modcond(outdir, ref_unet, ref_mc):
    cfg = dict(MG.SMALL, vocab_size=int(ref_mc.vocab_size))
    args = MG.make_args()
    torch.manual_seed(0)
    model = ref_unet.UNetModel(args=args, **cfg).eval()
    fill_module_(model, 51)
    n = 3
    labels_ignored = torch.tensor([5, 9, 2], dtype=torch.int64)  # the loop feeds s_id = ones instead (:565)
    word = "to be"  # the space becomes '_' (id 53): trainModifyCondition.py:169
    out = dict(seed=np.int64(51), word=np.array(word), labels=np_(labels_ignored), vocab_size=np.int64(ref_mc.vocab_size))
    # T = 8: every step's x
    xs, orig = spy(model)
    diff = ref_mc.Diffusion(noise_steps=8, img_size=(32, 64), args=args)
    with NoiseRecorder(4321) as nr:
        img = diff.sampling(model, IdentityVAE(), None, word, None, n, labels_ignored, args)
    model.forward = orig
    out.update(T8_noise=np.stack(nr.rec), T8_x_per_step=np.stack(xs), T8_image=np_(img))
    # the script's own default schedule: Diffusion() -> noise_steps = 600 (trainModifyCondition.py:516), 599 steps
    diff = ref_mc.Diffusion(img_size=(32, 64), args=args)
    assert diff.noise_steps == 600
    xs, orig = spy(model, keep_every=100)
    with NoiseRecorder(8765) as nr:
        img = diff.sampling(model, IdentityVAE(), None, word, None, 2, labels_ignored[:2], args)
    model.forward = orig
    out.update(T600_noise=np.stack(nr.rec), T600_x_every100=np.stack(xs), T600_image=np_(img))
    np.savez_compressed(os.path.join(outdir, "ddpm_traj_modcond.npz"), **out)
    print(f"[golden] ddpm_traj_modcond: T=8 final |x|={np.abs(out['T8_image']).mean():.4f}; "
          f"T=600 final |x|={np.abs(np_(img)).mean():.4f} ({len(nr.rec)} draws)")


def gen_charlevel(outdir, ref_unet):
    g0 = np.load(os.path.join(outdir, "fwd_base_full.npz"), allow_pickle=False)
    torch.manual_seed(0)
    model = ref_unet.UNetModel(args=MG.make_args(charLevelEmb=1), **MG.FULL).eval()
    fill_module_(model, int(g0["seed"]))
    with torch.no_grad():
        out = model(torch.from_numpy(g0["x"]), None, original_images=None, timesteps=torch.from_numpy(g0["t"]),
                    context=torch.from_numpy(g0["context"]).clone(), y=torch.from_numpy(g0["y"]))
    same = bool(np.array_equal(np_(out), g0["out"]))
    assert same, "charLevelEmb=1 changed the output of unet.UNetModel"
    np.savez_compressed(os.path.join(outdir, "fwd_base_full_charlevel.npz"), out=np_(out), seed=g0["seed"],
                        bit_identical_to_charLevelEmb0=np.bool_(same),
                        keys=np.array(list(model.state_dict().keys())))
    print(f"[golden] fwd_base_full_charlevel: bit-identical to charLevelEmb=0: {same}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("reference")
    ap.add_argument("outdir")
    a = ap.parse_args()
    outdir = os.path.abspath(a.outdir)
    sys.dont_write_bytecode = True
    MG._install_stubs()
    tv = sys.modules["torchvision"]
    for sub in ("transforms", "utils"):
        mod = types.ModuleType("torchvision." + sub)
        setattr(tv, sub, mod)
        sys.modules["torchvision." + sub] = mod
    sys.path.insert(0, os.path.abspath(a.reference))
    os.chdir(tempfile.mkdtemp())
    torch.set_num_threads(8)

    import unet as ref_unet  # noqa
    import unetPhosc as ref_phosc  # noqa
    import train as ref_train  # noqa
    import trainModifyCondition as ref_mc  # noqa

    words = ["to be", "a_b", "MOVE", "x y z", "Zz_", "getting"]
    np.savez_compressed(os.path.join(outdir, "primitives_modcond.npz"), words=np.array(words),
                        label_padding=np.array([[int(v) for v in ref_mc.label_padding(w, ref_mc.num_tokens)] for w in words],
                                               dtype=np.int64),
                        num_tokens=np.int64(ref_mc.num_tokens), vocab_size=np.int64(ref_mc.vocab_size),
                        c_classes=np.array(ref_mc.c_classes), max_chars=np.int64(ref_mc.OUTPUT_MAX_LEN),
                        default_noise_steps=np.int64(ref_mc.Diffusion(img_size=(64, 256), args=MG.make_args()).noise_steps))
    gen_phosc_traj(outdir, "ddpm_traj_phosc_small", ref_train, ref_phosc, MG.SMALL, 61, 3, 8, 37, (4, 8), "MOVE")
    gen_phosc_traj(outdir, "ddpm_traj_phosc_full", ref_train, ref_phosc, MG.FULL, 62, 2, 5, 769, (8, 32), "getting")
    gen_modcond(outdir, ref_unet, ref_mc)
    gen_charlevel(outdir, ref_unet)


if __name__ == "__main__":
    main()
