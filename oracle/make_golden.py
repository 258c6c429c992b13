#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

TEST INFRASTRUCTURE - not part of the product.  Run once, in the build
container, where the reference checkout is mounted:

    python oracle/make_golden.py /root/reference tests/golden

The reference's modules (``unet.py``, ``unetPhosc.py``, ``train.py``) are
imported unmodified from the path given on the command line; nothing from the
reference is copied into this repository - only tensors (inputs and the
reference's outputs) are written, as ``.npz``.  Import needs the stand-ins that
SURVEY.md section 8c lists (all are for modules/paths the path does not use):
a stub ``omegaconf.listconfig.ListConfig`` type, an in-memory pickle for the
hard-coded ``cropStyleDict_Numpy.pkl`` open() in ``unet.py:1159``, empty stub
modules for ``torchvision`` / ``diffusers`` / ``wandb``, and a writable CWD for
the two json files ``train.py:55-71`` writes at import.

Weights are not stored: every state_dict entry is filled by
``worddiffusion_amd.synthetic.synthetic_tensor(key, shape, seed)`` (the
reference zero-initialises several convolutions, so a fresh model would output
exactly 0 - SURVEY.md fact 0.8).
"""
from __future__ import annotations

import argparse
import builtins
import io
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from worddiffusion_amd.synthetic import fill_module_, synthetic_inputs  # noqa: E402


def _install_stubs():
    oc = types.ModuleType("omegaconf")
    lc = types.ModuleType("omegaconf.listconfig")

    class ListConfig(list):
        pass

    lc.ListConfig = ListConfig
    oc.listconfig = lc
    sys.modules["omegaconf"] = oc
    sys.modules["omegaconf.listconfig"] = lc
    for name in ("torchvision", "wandb"):
        sys.modules[name] = types.ModuleType(name)
    df = types.ModuleType("diffusers")
    df.AutoencoderKL = object
    sys.modules["diffusers"] = df

    real_open = builtins.open

    def patched_open(file, *a, **k):
        if isinstance(file, str) and file.endswith("cropStyleDict_Numpy.pkl"):
            return io.BytesIO(pickle.dumps({}))
        return real_open(file, *a, **k)

    builtins.open = patched_open


def make_args(**kw):
    base = dict(device="cpu", interpolation=False, charLevelEmb=0, charImages=0, attentionMaps=0,
                ocrTraining=0, imgConditioned=0, wrdChrWrStyl=0, phosc=0, phos=0, latent=True)
    base.update(kw)
    return types.SimpleNamespace(**base)


FULL = dict(image_size=(64, 256), in_channels=4, model_channels=320, out_channels=4, num_res_blocks=1,
            attention_resolutions=(1, 1), channel_mult=(1, 1), num_heads=4, num_classes=339,
            context_dim=320, vocab_size=53, max_seq_len=10)
SMALL = dict(image_size=(32, 64), in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1,
             attention_resolutions=(1, 1), channel_mult=(1, 1), num_heads=4, num_classes=11,
             context_dim=64, vocab_size=53, max_seq_len=10)
# a deeper shape to pin the generic constructor logic: 3 levels, attention only at ds=2, 2 res blocks
DEEP = dict(image_size=(32, 64), in_channels=4, model_channels=32, out_channels=4, num_res_blocks=2,
            attention_resolutions=(2,), channel_mult=(1, 2, 2), num_heads=2, num_classes=5,
            context_dim=64, vocab_size=53, max_seq_len=10)


def np_(t):
    return t.detach().cpu().numpy()


def hook_outputs(model, names):
    rec = {}
    handles = []
    mods = dict(model.named_modules())
    for n in names:
        def mk(n):
            def hook(_m, _inp, out):
                rec[n] = np_(out[0] if isinstance(out, tuple) else out).copy()
            return hook
        handles.append(mods[n].register_forward_hook(mk(n)))
    return rec, handles


def gen_forward(outdir, tag, cls, cfg, args, seed, batch, hw, base_variant, phosc_len=0, hooks=()):
    torch.manual_seed(0)
    model = cls(args=args, **cfg).eval()
    fill_module_(model, seed)
    inp = synthetic_inputs(batch, seed=seed + 2, hw=hw, in_ch=cfg["in_channels"],
                           num_classes=cfg["num_classes"], max_len=cfg["max_seq_len"], phosc_len=phosc_len)
    rec, handles = hook_outputs(model, hooks)
    with torch.no_grad():
        if base_variant:
            out = model(inp["x"], None, original_images=None, timesteps=inp["t"], context=inp["context"].clone(),
                        y=inp["y"])
        else:
            out = model(inp["x"], inp.get("phosc"), timesteps=inp["t"], context=inp["context"].clone(), y=inp["y"])
    for h in handles:
        h.remove()
    payload = dict(x=np_(inp["x"]), t=np_(inp["t"]), context=np_(inp["context"]), y=np_(inp["y"]),
                   out=np_(out), seed=np.int64(seed),
                   keys=np.array(list(model.state_dict().keys())),
                   shapes=np.array([",".join(map(str, v.shape)) for v in model.state_dict().values()]))
    if phosc_len:
        payload["phosc"] = np_(inp["phosc"])
    for k, v in rec.items():
        payload["hook:" + k] = v
    np.savez_compressed(os.path.join(outdir, tag + ".npz"), **payload)
    print(f"[golden] {tag}: out mean|.|={np.abs(np_(out)).mean():.4f} params={sum(p.numel() for p in model.parameters())}")
    return model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("reference")
    ap.add_argument("outdir")
    a = ap.parse_args()
    outdir = os.path.abspath(a.outdir)
    os.makedirs(outdir, exist_ok=True)
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, os.path.abspath(a.reference))
    os.chdir(tempfile.mkdtemp())  # train.py writes two json files into the CWD at import
    torch.set_num_threads(8)

    import unet as ref_unet  # noqa
    import unetPhosc as ref_phosc  # noqa
    import train as ref_train  # noqa

    # ---- (a1) timestep embedding, (a3) positional-encoding table, (a11) schedules ---------------
    t = torch.tensor([0, 1, 2, 17, 500, 998, 999], dtype=torch.int64)
    prim = dict(t=np_(t), temb320=np_(ref_unet.timestep_embedding(t, 320)), temb64=np_(ref_unet.timestep_embedding(t, 64)))
    ce = ref_phosc.CharacterEncoder(53, 320, 10)
    prim["pe_10x320"] = np_(ce.positional_encoding)
    ce64 = ref_phosc.CharacterEncoder(53, 64, 10)
    prim["pe_10x64"] = np_(ce64.positional_encoding)
    for T in (1000, 600, 51):
        d = ref_train.Diffusion(noise_steps=T, img_size=(64, 256), args=make_args())
        prim[f"beta{T}"] = np_(d.beta)
        prim[f"alpha{T}"] = np_(d.alpha)
        prim[f"alpha_hat{T}"] = np_(d.alpha_hat)
    words = ["MOVE", "a", "getting", "ABCDEFGHIJ", "text", "Zz"]
    prim["words"] = np.array(words)
    prim["label_padding"] = np.array([ref_train.label_padding(w, ref_train.num_tokens) for w in words], dtype=np.int64)
    prim["num_tokens"] = np.int64(ref_train.num_tokens)
    prim["vocab_size"] = np.int64(ref_train.vocab_size)
    np.savez_compressed(os.path.join(outdir, "primitives.npz"), **prim)

    # ---- (a4..a10) forwards --------------------------------------------------------------------------
    small_hooks_base = ("time_embed", "word_emb", "input_blocks.0", "input_blocks.1.0", "input_blocks.1.1",
                        "input_blocks.1.1.transformer_blocks.0", "input_blocks.2", "input_blocks.3",
                        "middle_block", "output_blocks.0", "output_blocks.1", "output_blocks.1.1",
                        "output_blocks.2", "output_blocks.3")
    gen_forward(outdir, "fwd_base_small", ref_unet.UNetModel, SMALL, make_args(), 11, 2, (4, 8), True,
                hooks=small_hooks_base)
    gen_forward(outdir, "fwd_phosc_small_nophosc", ref_phosc.UNetModelPhosc, SMALL, make_args(), 12, 2, (4, 8), False,
                hooks=small_hooks_base)
    gen_forward(outdir, "fwd_phosc_small", ref_phosc.UNetModelPhosc, SMALL, make_args(phosc=1), 13, 2, (4, 8), False,
                phosc_len=37, hooks=("word_emb",))
    gen_forward(outdir, "fwd_base_deep", ref_unet.UNetModel, DEEP, make_args(), 14, 2, (8, 16), True)
    gen_forward(outdir, "fwd_phosc_deep", ref_phosc.UNetModelPhosc, DEEP, make_args(phos=1), 15, 3, (8, 16), False,
                phosc_len=20)
    gen_forward(outdir, "fwd_base_full", ref_unet.UNetModel, FULL, make_args(), 21, 2, (8, 32), True)
    gen_forward(outdir, "fwd_phosc_full_nophosc", ref_phosc.UNetModelPhosc, FULL, make_args(), 22, 2, (8, 32), False)
    gen_forward(outdir, "fwd_phosc_full", ref_phosc.UNetModelPhosc, FULL, make_args(phosc=1), 23, 2, (8, 32), False,
                phosc_len=769)

    # ---- (a12/a13) DDPM: noise_images and a sampling trajectory with RECORDED noise ---------------------
    args = make_args()
    torch.manual_seed(0)
    model = ref_phosc.UNetModelPhosc(args=args, **SMALL).eval()
    fill_module_(model, 31)
    T = 8
    diff = ref_train.Diffusion(noise_steps=T, img_size=(32, 64), args=args)
    n = 3
    labels = torch.tensor([3, 0, 7], dtype=torch.int64)

    recorded = []
    real_randn, real_randn_like = torch.randn, torch.randn_like
    g = torch.Generator().manual_seed(1234)

    def rec_randn(*size, **kw):
        shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list)) else size
        z = real_randn(tuple(shape), generator=g)
        recorded.append(np_(z).copy())
        return z

    def rec_randn_like(x, **kw):
        return rec_randn(tuple(x.shape))

    class IdentityVAE:
        def decode(self, z):
            return types.SimpleNamespace(sample=z)

    xs = []
    orig_forward = model.forward

    def spy_forward(x, *a_, **k_):
        xs.append(np_(x).copy())
        return orig_forward(x, *a_, **k_)

    model.forward = spy_forward
    torch.randn, torch.randn_like = rec_randn, rec_randn_like
    try:
        img = diff.sampling(model, IdentityVAE(), n, "MOVE", labels, args)
        x0 = torch.from_numpy(recorded[0])
        tt = torch.tensor([1, 4, 7], dtype=torch.int64)
        n_before = len(recorded)
        xt, eps = diff.noise_images(x0, tt)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    model.forward = orig_forward
    # sampling calls the model twice per step with identical inputs (train.py:223-228): keep every 2nd x
    xs_step = np.stack(xs[0::2])
    np.savez_compressed(os.path.join(outdir, "ddpm_traj.npz"),
                        noise=np.stack(recorded[:n_before]),  # [T-1 (+init)]: recorded[0] = x_T, then z per step i>1
                        x_per_step=xs_step, labels=np_(labels), word=np.array("MOVE"), T=np.int64(T),
                        image=np_(img), seed=np.int64(31),
                        ni_x0=np_(x0), ni_t=np_(tt), ni_eps=np_(eps), ni_xt=np_(xt))
    print(f"[golden] ddpm_traj: {len(xs)} forwards, {n_before} noise draws, final |x|={np.abs(np_(img)).mean():.4f}")

    # ---- (a-T) EMA + one train step (loss and a few gradients; AdamW update of two tensors) ------------------
    torch.manual_seed(0)
    m1 = ref_phosc.UNetModelPhosc(args=args, **SMALL).train()
    fill_module_(m1, 41)
    m2 = ref_phosc.UNetModelPhosc(args=args, **SMALL)
    fill_module_(m2, 42)
    ema = ref_train.EMA(0.995)
    ema.step = 5
    ema.step_ema(m2, m1, step_start_ema=2)  # real EMA update
    sd2 = m2.state_dict()
    ema_keys = ["out.2.weight", "time_embed.0.bias", "input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight"]
    inp = synthetic_inputs(4, seed=43, hw=(4, 8), num_classes=SMALL["num_classes"])
    diff = ref_train.Diffusion(noise_steps=1000, img_size=(32, 64), args=args)
    eps = torch.from_numpy(np.random.RandomState(44).standard_normal(inp["x"].shape).astype(np.float32))
    tt = inp["t"]
    x_t = torch.sqrt(diff.alpha_hat[tt])[:, None, None, None] * inp["x"] + torch.sqrt(1 - diff.alpha_hat[tt])[:, None, None, None] * eps
    opt = torch.optim.AdamW(m1.parameters(), lr=1e-4)
    pred = m1(x_t, None, timesteps=tt, context=inp["context"].clone(), y=inp["y"])
    loss = torch.nn.MSELoss()(eps, pred)
    opt.zero_grad()
    loss.backward()
    grad_keys = ["out.2.weight", "out.2.bias", "time_embed.0.weight", "label_emb.weight",
                 "input_blocks.0.0.weight", "input_blocks.1.0.in_layers.2.weight",
                 "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight",
                 "input_blocks.1.1.transformer_blocks.0.ff.net.0.proj.weight",
                 "middle_block.1.proj_out.weight", "output_blocks.1.1.conv.weight",
                 "word_emb.embedding.weight", "word_emb.attention.linear_key.bias"]
    params = dict(m1.named_parameters())
    payload = dict(seed_model=np.int64(41), seed_ema=np.int64(42), x0=np_(inp["x"]), eps=np_(eps), t=np_(tt),
                   context=np_(inp["context"]), y=np_(inp["y"]), x_t=np_(x_t), loss=np_(loss), pred=np_(pred))
    for k in ema_keys:
        payload["ema:" + k] = np_(sd2[k])
    gsq = 0.0
    for k, p in params.items():
        if p.grad is not None:
            gsq += float((p.grad.double() ** 2).sum())
    payload["grad_norm"] = np.float64(gsq ** 0.5)
    for k in grad_keys:
        payload["grad:" + k] = np_(params[k].grad)
    opt.step()
    for k in grad_keys[:4]:
        payload["adamw:" + k] = np_(params[k])
    np.savez_compressed(os.path.join(outdir, "train_step.npz"), **payload)
    print(f"[golden] train_step: loss={float(loss):.6f} |g|={gsq ** 0.5:.4f}")


if __name__ == "__main__":
    main()
