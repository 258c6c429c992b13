"""Bulk sampling driver in the shape of the reference's ``full_sampling.py`` / ``sampling.py``.

The reference walks the ground-truth list one row at a time - one 999-step ``diffusion.sampling`` call with ``n = 1`` per
(writer, word) row (``full_sampling.py:147-170``).  Here the rows are cut into rank shards and batches, every batch is ONE
sampling call (B rows at once), and the on-device noise is keyed by the global row index, so the result of a row does not
depend on the batch size or the number of ranks.  Host-side pieces restated from the reference:

  * ``read_gt``       - ``writer,image word`` lines (``full_sampling.py:132-143``);
  * ``writer_dict``   - writer id -> class index in order of first appearance (``train.py:371-388``,
                        ``writers_dict_train.json``);
  * ``write_png``     - 8-bit RGB / grey PNG without PIL / cv2 (the reference saves through torchvision + PIL,
                        ``train.py:104-137``);
  * ``regenerate``    - the loop; ``vae=None`` writes latents (``.npy``), a duck-typed VAE (``vae.decode(z).sample``) PNGs.
"""
from __future__ import annotations

import argparse
import json
import os
import struct
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .dist import env_rank_world, shard_range


def read_gt(path: str) -> List[Tuple[str, str, str]]:
    """[(writer id, image name, transcription)] - ``full_sampling.py:132-143``: ``i.strip().split(' ')``, writer and image
    are the two comma-separated fields of the first token, the transcription is the second token."""
    rows = []
    with open(path, "r") as f:
        for line in f.readlines():
            parts = line.strip().split(" ")
            if len(parts) < 2 or "," not in parts[0]:
                continue
            s_id, image = parts[0].split(",")[0], parts[0].split(",")[1]
            rows.append((s_id, image, parts[1]))
    return rows


def writer_dict(rows: Sequence[Tuple[str, str, str]], path: Optional[str] = None) -> Dict[str, int]:
    """Writer id -> class index.  ``path``: an existing ``writers_dict_train.json`` (``full_sampling.py:152-153``); otherwise
    built like ``train.py:371-388`` (first appearance order)."""
    if path is not None and os.path.exists(path):
        with open(path, "r") as f:
            return {str(k): int(v) for k, v in json.load(f).items()}
    out: Dict[str, int] = {}
    for s_id, _, _ in rows:
        if s_id not in out:
            out[s_id] = len(out)
    return out


def make_phosc_of(alphabet_csv: str, version: str = "eng", phosc: int = 1, phos: int = 0):
    """word -> int64 PHOSC vector (``datasets.py:44-70``) with the reference's shape-count table read from ``alphabet_csv``
    (``ResPhoSCNetZSL/modules/utils/Alphabet.csv``; a data file of the reference, not shipped here).  One vector per
    distinct word is computed and kept (the reference recomputes or unpickles a ``wordPhosc`` dictionary,
    trainModifyCondition.py:268-292)."""
    from .phosc import load_alphabet, phosc_vector
    index, table = load_alphabet(alphabet_csv)
    memo: Dict[str, torch.Tensor] = {}

    def phosc_of(word: str) -> torch.Tensor:
        if word not in memo:
            memo[word] = torch.from_numpy(phosc_vector(word, index, table, version, phosc=phosc, phos=phos))
        return memo[word]

    return phosc_of


def write_png(path: str, img: np.ndarray) -> None:
    """uint8 [H, W] (grey) or [H, W, 3] (RGB) -> PNG (zlib-deflated, filter 0 scanlines)."""
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3):
        raise ValueError("write_png expects uint8 [H,W] or [H,W,3]")
    h, w = img.shape[:2]
    color = 0 if img.ndim == 2 else 2
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


@torch.no_grad()
def regenerate(model, diffusion, rows: Sequence[Tuple[str, str, str]], wr_dict: Dict[str, int], args, vae=None,
               batch: int = 64, out_dir: Optional[str] = None, seed: int = 0, rank: Optional[int] = None,
               world: Optional[int] = None, skip_steps: bool = False, phosc_of=None):
    """Samples every gt row once (this rank's shard of them), ``batch`` rows per ``sampling`` call.

    Returns ``(start, latents_or_images)`` for the shard.  With ``out_dir`` the results are written as
    ``<image>.png`` (a VAE was given) or ``<image>.npy`` (latents).  ``skip_steps`` selects the step-skipping sampler of
    ``regenerateFromtrain2.py`` (``Diffusion.sampling3``); ``phosc_of(word) -> int tensor [769]`` supplies PHOSC vectors for
    ``UNetModelPhosc`` (``args.phosc == 1``)."""
    r0, w0, _ = env_rank_world()
    rank = r0 if rank is None else rank
    world = w0 if world is None else world
    start, count = shard_range(len(rows), rank, world)
    mine = rows[start:start + count]
    outs = []
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
    for b0 in range(0, count, batch):
        chunk = mine[b0:b0 + batch]
        words = [t for _, _, t in chunk]
        labels = torch.tensor([wr_dict[s] for s, _, _ in chunk], dtype=torch.int64)
        phosc = torch.stack([phosc_of(wd) for wd in words]) if phosc_of is not None else None
        kw = dict(seed=seed, sample_offset=start + b0)
        if skip_steps:
            res = diffusion.sampling3(0, None, words, phosc, model, model, vae, 0, 1, len(chunk), words, labels, args, **kw)
            res = res if vae is None else res[2]
        else:
            res = diffusion.sampling(model, vae, len(chunk), words, labels, args, phoscLabels=phosc, **kw)
        res = res.detach().cpu()
        outs.append(res)
        if out_dir is not None:
            for (_, image, _), item in zip(chunk, res):
                if vae is None:
                    np.save(os.path.join(out_dir, f"{image}.npy"), item.numpy())
                else:
                    arr = (item.clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
                    write_png(os.path.join(out_dir, f"{image}.png"), arr if arr.shape[2] == 3 else arr[:, :, 0])
    return start, (torch.cat(outs) if outs else torch.empty(0))


def main(argv=None):
    """``python -m worddiffusion_amd.driver --gt_train gt.txt --models_path run/ --save_path out/`` (one process per GPU under
    torchrun; flags follow ``full_sampling.py:40-66`` where they exist)."""
    import copy
    import types
    from . import Diffusion, UNetModel, UNetModelPhosc
    ap = argparse.ArgumentParser()
    ap.add_argument("--gt_train", required=True)
    ap.add_argument("--models_path", default=None, help="directory holding models/ema_ckpt.pt (reference layout)")
    ap.add_argument("--save_path", required=True)
    ap.add_argument("--writer_dict", default="./writers_dict_train.json")
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--emb_dim", type=int, default=320)
    ap.add_argument("--num_heads", type=int, default=4)
    ap.add_argument("--num_res_blocks", type=int, default=1)
    ap.add_argument("--noise_steps", type=int, default=1000)
    ap.add_argument("--phosc", type=int, default=0)
    ap.add_argument("--alphabet_csv", default=None,
                    help="--phosc 1: the reference's shape-count table (ResPhoSCNetZSL/modules/utils/Alphabet.csv)")
    ap.add_argument("--vocab_size", type=int, default=53, choices=[53, 54],
                    help="53: the 52-letter alphabet of train.py; 54: the '_' alphabet of trainModifyCondition.py:68")
    ap.add_argument("--skip_steps", type=int, default=0, help="1: regenerateFromtrain2.py's step-skipping sampler")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--stable_dif_path", default=None,
                    help="local Stable-Diffusion checkout in diffusers layout (its vae/ subfolder is read; train.py:415): with it "
                         "the rows are decoded and written as PNGs, without it as latents (.npy)")
    a = ap.parse_args(argv)
    rank, world, local = env_rank_world()
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    args = types.SimpleNamespace(device=dev, interpolation=False, charLevelEmb=0, charImages=0, attentionMaps=0, ocrTraining=0,
                                 imgConditioned=0, wrdChrWrStyl=0, phosc=a.phosc, phos=0, latent=True, fullSampling=False)
    rows = read_gt(a.gt_train)
    wr = writer_dict(rows, a.writer_dict)
    cls = UNetModelPhosc if a.phosc else UNetModel
    unet = cls(image_size=(64, 256), in_channels=4, model_channels=a.emb_dim, out_channels=4, num_res_blocks=a.num_res_blocks,
               attention_resolutions=(1, 1), channel_mult=(1, 1), num_heads=a.num_heads, num_classes=max(339, len(wr)),
               context_dim=a.emb_dim, vocab_size=a.vocab_size, args=args, max_seq_len=10).to(dev)
    if a.models_path:
        unet.load_state_dict(torch.load(os.path.join(a.models_path, "models", "ema_ckpt.pt"), map_location=dev,
                                        weights_only=True))
    ema_model = copy.deepcopy(unet).eval().requires_grad_(False)
    diffusion = Diffusion(noise_steps=a.noise_steps, img_size=(64, 256), args=args)
    vae = None
    if a.stable_dif_path:
        from .vae import AutoencoderKL
        vae = AutoencoderKL.from_pretrained(a.stable_dif_path, subfolder="vae").to(dev)
    phosc_of = None
    if a.phosc:
        if not a.alphabet_csv:
            ap.error("--phosc 1 needs --alphabet_csv (the PHOS shape-count table)")
        phosc_of = make_phosc_of(a.alphabet_csv)
    start, res = regenerate(ema_model, diffusion, rows, wr, args, vae=vae, batch=a.batch_size,
                            out_dir=os.path.join(a.save_path, "images"), seed=a.seed, skip_steps=bool(a.skip_steps),
                            phosc_of=phosc_of)
    print(f"[rank {rank}/{world}] rows {start}..{start + len(res)} of {len(rows)} written to {a.save_path}/images")


if __name__ == "__main__":
    main()
