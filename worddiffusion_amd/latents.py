"""Cached VAE latents as the training input (second half) and the epoch loop around ``TrainStep``.

The reference's ``--vaeFromDict 1`` mode (``trainModifyCondition.py:300-325``) reads two pickled dictionaries,
``imageWordLineVae3.pkl`` (word crops) and ``imageWordLineVae3OnlyChar.pkl`` (character crops), of the form
``{image name: {"images": FloatTensor[1, 4, 8, 32], ...}}``; ``IAMDataset.__getitem__`` looks an image up in the first and
falls back to the second (``:452-456``), takes ``["images"].squeeze()`` (``:457-458``) and the training loop feeds it to
``noise_images`` as is (``latents = images``, ``:713-716``: the 0.18215 scale was applied when the cache was made).

A pickle executes code when it is read, so this package never opens one.  The on-disk form here is a TENSOR-ONLY
container with the same content:

    <name>.safetensors      one fp32 tensor [C, H, W] per image, keyed by the image name (e.g. ``a01-000u-00-00.png``)
    (or ``.npz`` with the same keys - read with ``allow_pickle=False``)

and a user converts their own pickles once, offline, in their own environment, with ``convert_latent_dict`` (it takes the
dictionary object they loaded themselves - see INTEGRATION.md).

``CachedLatentDataset`` mirrors the reference dataset's item (``trainModifyCondition.py:460-477``: image name, latent, word
ids through ``label_padding``, writer index, transcription[, PHOSC vector]) and batches it on the host; ``train_epoch`` is the
batch loop of ``train.py:261-295`` with the loop body replaced by one ``TrainStep`` call.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterator, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from .diffusion import NUM_TOKENS, label_padding, label_padding_underscore
from .dist import shard_range


def _as_chw(t) -> torch.Tensor:
    t = torch.as_tensor(t).detach().to(torch.float32).cpu()
    t = t.squeeze()  # trainModifyCondition.py:458
    if t.dim() != 3:
        raise ValueError(f"a cached latent must squeeze to [C, H, W], got {tuple(t.shape)}")
    return t.contiguous()


def save_latent_cache(path: str, latents: Mapping[str, torch.Tensor]) -> str:
    """Writes ``{image name: latent [C,H,W]}`` as ``.safetensors`` (default) or ``.npz`` (by extension)."""
    items = {str(k): _as_chw(v) for k, v in latents.items()}
    if path.endswith(".npz"):
        np.savez(path, **{k: v.numpy() for k, v in items.items()})
    else:
        from safetensors.torch import save_file
        save_file(items, path)
    return path


def convert_latent_dict(obj: Mapping[str, object], path: str, field: str = "images") -> str:
    """Converter for the reference's cache: ``obj`` is the dictionary a user obtained by loading THEIR OWN
    ``imageWordLineVae3*.pkl`` in their own environment (``{name: {"images": tensor[1,4,8,32], ...}}``,
    trainModifyCondition.py:300-325); only the ``field`` tensors are kept.  Plain ``{name: tensor}`` mappings pass too."""
    flat = {}
    for k, v in obj.items():
        flat[str(k)] = v[field] if isinstance(v, Mapping) else v
    return save_latent_cache(path, flat)


class LatentCache:
    """Read-only view over one or more containers; lookup order = argument order (word dictionary first, then the
    character dictionary: trainModifyCondition.py:452-456).  ``.safetensors`` files are memory-mapped and read lazily."""

    def __init__(self, *paths: str):
        if not paths:
            raise ValueError("LatentCache needs at least one container")
        self._stores = []
        for p in paths:
            if p.endswith(".npz"):
                z = np.load(p, allow_pickle=False)
                self._stores.append(("npz", z, set(z.files)))
            else:
                from safetensors import safe_open
                f = safe_open(p, framework="pt", device="cpu")
                self._stores.append(("st", f, set(f.keys())))

    def __contains__(self, name: str) -> bool:
        return any(name in keys for _, _, keys in self._stores)

    def __len__(self) -> int:
        return len(set().union(*[keys for _, _, keys in self._stores]))

    def keys(self) -> List[str]:
        seen, out = set(), []
        for _, _, keys in self._stores:
            for k in sorted(keys):
                if k not in seen:
                    seen.add(k)
                    out.append(k)
        return out

    def __getitem__(self, name: str) -> torch.Tensor:
        for kind, store, keys in self._stores:
            if name in keys:
                t = torch.from_numpy(store[name]) if kind == "npz" else store.get_tensor(name)
                return _as_chw(t)
        raise KeyError(name)


class CachedLatentDataset:
    """``rows``: gt rows ``(writer id, image name, transcription)`` (``driver.read_gt``); ``wr_dict``: writer id -> class index
    (``driver.writer_dict``); ``cache``: ``LatentCache``.  The image key is ``image + suffix`` with ``suffix='.png'`` as in
    ``train.py:378`` / the dictionary keys of the reference.  ``underscore`` selects the ``'_'`` alphabet of the
    ModifyCondition scripts.  Rows whose latent is missing are dropped up front when ``skip_missing`` (the reference would
    raise KeyError from ``__getitem__`` in the middle of an epoch)."""

    def __init__(self, rows: Sequence[Tuple[str, str, str]], wr_dict: Mapping[str, int], cache: LatentCache,
                 suffix: str = ".png", underscore: bool = False, phosc_of: Optional[Callable[[str], torch.Tensor]] = None,
                 skip_missing: bool = False, max_items: Optional[int] = None):
        self.cache, self.wr_dict, self.suffix, self.phosc_of = cache, wr_dict, suffix, phosc_of
        self._pad = label_padding_underscore if underscore else label_padding
        rows = list(rows[:max_items] if max_items is not None else rows)  # train.py:369 keeps the first 1000 lines
        if skip_missing:
            rows = [r for r in rows if (r[1] + suffix) in cache]
        self.rows = rows
        self._pre = None  # (latents [N,C,H,W], words [N,L], s_id [N][, phosc [N,P]]) once preloaded

    def preload(self, max_bytes: int = 2 << 30) -> bool:
        """Reads every row's latent / word ids / writer index once into dense host tensors (the reference keeps its whole
        dictionaries in memory too); ``batches`` then gathers rows by index instead of building 64 items per batch.  Skipped
        (returns False) when the latents would take more than ``max_bytes``."""
        if self._pre is not None:
            return True
        if not self.rows:
            return False
        first = self[0]
        if first["latent"].numel() * 4 * len(self.rows) > max_bytes:
            return False
        items = [first] + [self[i] for i in range(1, len(self.rows))]
        pre = [torch.stack([it["latent"] for it in items]), torch.stack([it["word"] for it in items]),
               torch.tensor([it["s_id"] for it in items], dtype=torch.int64)]
        if self.phosc_of is not None:
            pre.append(torch.stack([it["phosc"] for it in items]))
        self._pre = pre
        return True

    def __len__(self) -> int:
        return len(self.rows)

    def __getitem__(self, i: int):
        s_id, image, label = self.rows[i]
        name = image + self.suffix
        item = dict(image_name=name, latent=self.cache[name],
                    word=torch.tensor(self._pad(label, NUM_TOKENS), dtype=torch.int64),
                    s_id=int(self.wr_dict[s_id]), label=label)
        if self.phosc_of is not None:
            item["phosc"] = torch.as_tensor(self.phosc_of(label)).to(torch.int64)
        return item

    def batches(self, batch_size: int, shuffle: bool = True, seed: int = 0, epoch: int = 0, rank: int = 0, world: int = 1,
                drop_last: bool = True, pin: bool = True) -> Iterator[Dict[str, object]]:
        """Host-side loader: one permutation per (seed, epoch) shared by all ranks, cut into contiguous rank shards
        (``dist.shard_range``), then into batches.  ``drop_last`` keeps every step on one plan / one captured graph."""
        n = len(self.rows)
        order = np.random.RandomState((seed * 1000003 + epoch) & 0x7FFFFFFF).permutation(n) if shuffle else np.arange(n)
        start, count = shard_range(n, rank, world)
        mine = order[start:start + count]
        if world > 1:  # equal step counts on every rank, or the gradient all-reduce would hang
            per = n // world
            mine = mine[:per]
        for b0 in range(0, len(mine), batch_size):
            idx = mine[b0:b0 + batch_size]
            if len(idx) < batch_size and drop_last:
                break
            if self._pre is not None:
                sel = torch.from_numpy(np.ascontiguousarray(idx)).long()
                out = dict(image_names=[self.rows[int(i)][1] + self.suffix for i in idx], labels=[self.rows[int(i)][2] for i in idx],
                           latents=self._pre[0].index_select(0, sel), words=self._pre[1].index_select(0, sel),
                           s_id=self._pre[2].index_select(0, sel))
                if self.phosc_of is not None:
                    out["phosc"] = self._pre[3].index_select(0, sel)
            else:
                items = [self[int(i)] for i in idx]
                out = dict(image_names=[it["image_name"] for it in items], labels=[it["label"] for it in items],
                           latents=torch.stack([it["latent"] for it in items]), words=torch.stack([it["word"] for it in items]),
                           s_id=torch.tensor([it["s_id"] for it in items], dtype=torch.int64))
                if self.phosc_of is not None:
                    out["phosc"] = torch.stack([it["phosc"] for it in items])
            if pin and torch.cuda.is_available():
                for k in ("latents", "words", "s_id", "phosc"):
                    if k in out:
                        out[k] = out[k].pin_memory()
            yield out


def train_epoch(step, dataset: CachedLatentDataset, batch_size: int, device, epoch: int = 0, seed: int = 0,
                max_batches: Optional[int] = None, rank: int = 0, world: int = 1, shuffle: bool = True,
                on_batch: Optional[Callable[[int, torch.Tensor], None]] = None) -> Dict[str, object]:
    """One epoch of ``train.py:261-295`` over cached latents: for every batch ``step(latents, word ids, writer ids[, PHOSC])``
    (``training.TrainStep``: timesteps, noise, forward, loss, backward, gradient all-reduce, AdamW, EMA).
    ``max_batches=30`` reproduces the reference's ``if i == 30: break`` (``train.py:263-264``).  The loss stays on the
    device; it is read once at the end (the reference's per-batch ``loss.item()``, ``train.py:295``, is available through
    ``on_batch``).  Returns the number of batches / images and the mean loss."""
    import queue
    import threading
    if hasattr(dataset, "preload"):
        dataset.preload()
    nb, total = 0, None
    # host batching (container reads, stacking, pinning) runs one batch ahead on a worker thread; the id range check happens
    # there on the host tensors, so the step itself never synchronises with the device
    q: "queue.Queue" = queue.Queue(maxsize=2)
    eng = getattr(step, "eng", None)

    def produce():
        try:
            for i, b in enumerate(dataset.batches(batch_size, shuffle=shuffle, seed=seed, epoch=epoch, rank=rank, world=world)):
                if max_batches is not None and i == max_batches:
                    break
                if eng is not None:
                    eng.check_ids(b["words"], b["s_id"], b.get("phosc"))
                q.put(b)
            q.put(None)
        except BaseException as e:  # surfaced in the consumer
            q.put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    i = 0
    while True:
        b = q.get()
        if b is None:
            break
        if isinstance(b, BaseException):
            raise b
        lat = b["latents"].to(device, non_blocking=True)
        words = b["words"].to(device, non_blocking=True)
        s_id = b["s_id"].to(device, non_blocking=True)
        ph = b["phosc"].to(device, non_blocking=True) if "phosc" in b else None
        loss = step(lat, words, s_id, phoscLabels=ph, **({"check": False} if eng is not None else {}))
        total = loss.clone() if total is None else total + loss
        nb += 1
        if on_batch is not None:
            on_batch(i, loss)
        i += 1
    th.join()
    mean = float(total.item()) / nb if nb else float("nan")
    return dict(batches=nb, images=nb * batch_size, mean_loss=mean)
