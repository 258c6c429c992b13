"""CPU tests of the host side: parameter tree == reference state_dict, weight repack and gather tables
(checked against torch convolutions), C-ABI exports, DDPM host helpers."""
import copy
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests._common import DEEP, FULL, FWD_CASES, SMALL, golden_state_dict, load_golden, make_args
from worddiffusion_amd import Diffusion, UNetModel, UNetModelPhosc, label_padding
from worddiffusion_amd import _native as N
from worddiffusion_amd.engine import conv_gather_table, geglu_interleave


@pytest.mark.parametrize("tag", sorted(FWD_CASES))
def test_state_dict_layout_matches_reference(golden_dir, tag):
    cfg, variant, _ = FWD_CASES[tag]
    g = load_golden(golden_dir, tag)
    cls = UNetModel if variant == "base" else UNetModelPhosc
    m = cls(args=make_args(), **cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    m.load_state_dict(golden_state_dict(g), strict=True)
    # EMA copy as the reference makes it (train.py:409)
    m2 = copy.deepcopy(m).eval().requires_grad_(False)
    assert torch.equal(m2.state_dict()["out.2.weight"], sd["out.2.weight"])


def test_fresh_model_zero_init_like_reference():
    m = UNetModel(args=make_args(), **SMALL)
    for k in ("out.2.weight", "input_blocks.1.0.out_layers.3.weight", "input_blocks.1.1.proj_out.weight"):
        assert float(m.state_dict()[k].abs().max()) == 0.0  # zero_module, unet.py:152-158


def test_constructor_rejects_what_the_reference_cannot_run():
    for kw in (dict(resblock_updown=True), dict(use_spatial_transformer=False), dict(dims=3), dict(n_embed=8),
               dict(conv_resample=False)):
        with pytest.raises((NotImplementedError, AssertionError)):
            UNetModel(args=make_args(), **{**SMALL, **kw})
    with pytest.raises(NotImplementedError):
        UNetModel(args=make_args(attentionMaps=1), **SMALL)


def _gather_conv(x, w, b, mode):
    """Emulate wd_gemm's tap-gather on the CPU with the engine's tables and packing: x [B,C,h,w]."""
    B, C, h, wd = x.shape
    tab, ho, wo = conv_gather_table(h, wd, mode)
    tok = x.permute(0, 2, 3, 1).reshape(B, h * wd, C)
    tok = torch.cat([tok, torch.zeros(B, 1, C)], 1)  # index -1 -> zero row
    cols = [tok[:, torch.from_numpy(tab[t]).long()] for t in range(9)]  # [B, ho*wo, C] each
    a = torch.cat(cols, dim=2)
    wp = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    out = a @ wp.t() + b
    return out.reshape(B, ho, wo, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("h,w", [(8, 32), (4, 16), (5, 7), (1, 3)])
def test_gather_tables_equal_torch_convs(h, w):
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(2, 6, h, w, generator=g)
    wt = torch.randn(5, 6, 3, 3, generator=g)
    b = torch.randn(5, generator=g)
    assert torch.allclose(_gather_conv(x, wt, b, "same"), F.conv2d(x, wt, b, padding=1), atol=1e-5)
    assert torch.allclose(_gather_conv(x, wt, b, "down"), F.conv2d(x, wt, b, stride=2, padding=1), atol=1e-5)
    up = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding=1)
    assert torch.allclose(_gather_conv(x, wt, b, "up"), up, atol=1e-5)


@pytest.mark.parametrize("h,w", [(4, 16), (2, 8), (3, 5)])
def test_upsample_as_four_phase_convolutions(h, w):
    """upsample_phase_tables / upsample_phase_weights (four 2x2 convolutions of the source map, rows phase-major, read back through
    the raster permutation) == F.interpolate(nearest x2) + conv3x3 pad 1 (Upsample.forward, unet.py:488-499)."""
    from worddiffusion_amd.engine import upsample_phase_tables, upsample_phase_weights
    g = torch.Generator().manual_seed(h * 10 + w)
    x = torch.randn(2, 6, h, w, generator=g)
    wt = torch.randn(5, 6, 3, 3, generator=g)
    b = torch.randn(5, generator=g)
    tab, perm = upsample_phase_tables(h, w)
    wph = upsample_phase_weights(wt)                      # [phase][tap][n][c]
    hw = h * w
    tok = torch.cat([x.permute(0, 2, 3, 1).reshape(2, hw, 6), torch.zeros(2, 1, 6)], 1)   # index -1 -> zero row
    a = torch.cat([tok[:, torch.from_numpy(tab[t]).long()] for t in range(4)], dim=2)     # [B, 4 hw, 4 C], rows phase-major
    out = torch.empty(2, 4 * hw, 5)
    for ph in range(4):
        wp = wph[ph].permute(1, 0, 2).reshape(5, 4 * 6)    # [n][tap * C + c]
        out[:, ph * hw:(ph + 1) * hw] = a[:, ph * hw:(ph + 1) * hw] @ wp.t() + b
    raster = out[:, torch.from_numpy(perm).long()].reshape(2, 2 * h, 2 * w, 5).permute(0, 3, 1, 2)
    want = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding=1)
    assert torch.allclose(raster, want, atol=1e-5)
    assert sorted(perm.tolist()) == list(range(4 * hw))


def test_geglu_interleave_pairs_columns():
    w = torch.arange(2 * 64 * 3, dtype=torch.float32).reshape(128, 3)
    p = geglu_interleave(w, 32)
    for blk in range(2):
        assert torch.equal(p[64 * blk: 64 * blk + 32], w[32 * blk: 32 * blk + 32])
        assert torch.equal(p[64 * blk + 32: 64 * blk + 64], w[64 + 32 * blk: 64 + 32 * blk + 32])


def test_engine_recipes_cover_the_forward(golden_dir):
    """The repacked matrices reproduce the reference layers (ResBlock with skip: conv2 | skip along K)."""
    from worddiffusion_amd.engine import UNetEngine
    g = load_golden(golden_dir, "fwd_base_small")
    m = UNetModel(args=make_args(), **SMALL)
    m.load_state_dict(golden_state_dict(g))
    eng = UNetEngine.__new__(UNetEngine)
    eng.model, eng.variant = m, "base"
    rec = eng._recipes()
    name = "out0.0"  # ResBlock(128 -> 64) with a 1x1 skip
    rb = m.output_blocks[0][0]
    wcat = rec[name + ".c2.w"].host()
    assert wcat.shape == (64, 9 * 64 + 128)
    x = torch.randn(2, 128, 4, 8)
    h = torch.randn(2, 64, 4, 8)
    ref = F.conv2d(h, rb.out_layers[3].weight, rb.out_layers[3].bias, padding=1) + \
        F.conv2d(x, rb.skip_connection.weight, rb.skip_connection.bias)
    tab, _, _ = conv_gather_table(4, 8, "same")
    tok = torch.cat([h.permute(0, 2, 3, 1).reshape(2, 32, 64), torch.zeros(2, 1, 64)], 1)
    a = torch.cat([tok[:, torch.from_numpy(tab[t]).long()] for t in range(9)] +
                  [x.permute(0, 2, 3, 1).reshape(2, 32, 128)], dim=2)
    out = (a @ wcat.t() + rec[name + ".c2.b"].host()).reshape(2, 4, 8, 64).permute(0, 3, 1, 2)
    assert torch.allclose(out, ref, atol=1e-4)
    # every ResBlock has a FiLM slice; K/V slices exist for both cross-attentions of every block (base variant)
    assert eng.film_total == sum(mod.cout for _, mod in eng._walk() if hasattr(mod, "emb_layers"))
    assert eng.kv_total == 2 * 64 * 2 * 4
    assert rec["film.w"].host().shape == (eng.film_total, 256)
    # GEGLU projection: x rows and gate rows interleaved per half tile; data-gradient operands are the transposes
    tb = m.input_blocks[1][1].transformer_blocks[0]
    from worddiffusion_amd.engine import geglu_interleave, geglu_tile
    g_ = geglu_tile(tb.ff.net[2].in_features) % 1000 // 2
    assert torch.equal(rec["in1.1.tb0.ff1.w"].host(), geglu_interleave(tb.ff.net[0].proj.weight.detach(), g_))
    from worddiffusion_amd.backward import pack_dx_weight
    from worddiffusion_amd.train_engine import TrainEngine
    teng = TrainEngine.__new__(TrainEngine)
    teng.model, teng.variant = m, "base"
    trec = teng._recipes()
    assert torch.equal(trec["B:out0.0.c1.w"].host(), pack_dx_weight(rb.in_layers[2].weight.detach()))
    assert torch.equal(trec["B:out0.0.skip.w"].host(), pack_dx_weight(rb.skip_connection.weight.detach().flatten(1)))
    pad = torch.zeros(32, 64, 3, 3)
    pad[:4] = m.out[2].weight.detach()
    assert torch.equal(trec["B:out.w"].host(), pack_dx_weight(pad))
    assert torch.equal(trec["B:film.w"].host(), rec["film.w"].host().t())


def test_c_abi_exports_every_declared_symbol():
    assert os.path.exists(N.LIB_PATH), "build first: python -m worddiffusion_amd.build"
    lib = ctypes.CDLL(N.LIB_PATH)
    declared = N.header_symbols()
    assert len(declared) >= 25
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert set(N._SIGS) == set(declared)
    assert b"gfx950" in N.lib().wd_version()
    # the ctypes mirror of wd_gemm_args has the library's layout (N.lib() refuses to load a library where it has not)
    assert N.lib().wd_gemm_args_bytes() == ctypes.sizeof(N.WdGemmArgs)


def test_label_padding_and_schedule(golden_dir):
    g = load_golden(golden_dir, "primitives")
    for w, ref in zip(g["words"], g["label_padding"]):
        assert label_padding(str(w)) == [int(v) for v in ref]
    for T in (1000, 600, 51):
        d = Diffusion(noise_steps=T, img_size=(64, 256), args=make_args())
        assert torch.equal(d.beta, torch.from_numpy(g[f"beta{T}"]))
        assert torch.equal(d.alpha, torch.from_numpy(g[f"alpha{T}"]))
        assert torch.equal(d.alpha_hat, torch.from_numpy(g[f"alpha_hat{T}"]))
    d = Diffusion(args=make_args())
    assert d.sample is not None and Diffusion.sample is Diffusion.sampling
    t = d.sample_timesteps(64)
    assert t.min() >= 1 and t.max() < 1000


def test_no_cpu_fallback():
    m = UNetModelPhosc(args=make_args(), **SMALL).eval()
    x = torch.zeros(1, 4, 4, 8)
    with torch.no_grad(), pytest.raises(N.NativeError):
        m(x, None, timesteps=torch.tensor([3]), context=torch.zeros(1, 10, dtype=torch.long), y=torch.tensor([0]))


def test_driver_host_pieces(tmp_path):
    """gt reader / writer dictionary / PNG writer of the bulk sampling driver (full_sampling.py:132-153)."""
    import struct
    import zlib
    from worddiffusion_amd.driver import read_gt, write_png, writer_dict
    gt = tmp_path / "gt.txt"
    gt.write_text("000,a01-000u-00-00 A\n011,a01-000u-00-01 MOVE\n000,a01-000u-00-02 to\n\nbroken-line\n")
    rows = read_gt(str(gt))
    assert rows == [("000", "a01-000u-00-00", "A"), ("011", "a01-000u-00-01", "MOVE"), ("000", "a01-000u-00-02", "to")]
    assert writer_dict(rows) == {"000": 0, "011": 1}
    wj = tmp_path / "writers.json"
    wj.write_text('{"000": 7, "011": 3}')
    assert writer_dict(rows, str(wj)) == {"000": 7, "011": 3}
    img = (np.arange(5 * 7 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)
    p = tmp_path / "x.png"
    write_png(str(p), img)
    data = p.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    w, h, depth, color = struct.unpack(">IIBB", data[16:26])
    assert (w, h, depth, color) == (7, 5, 8, 2)
    i0 = data.index(b"IDAT")
    n = struct.unpack(">I", data[i0 - 4:i0])[0]
    raw = zlib.decompress(data[i0 + 4:i0 + 4 + n])
    rows_px = np.frombuffer(raw, dtype=np.uint8).reshape(5, 1 + 7 * 3)
    assert (rows_px[:, 0] == 0).all() and np.array_equal(rows_px[:, 1:].reshape(5, 7, 3), img)
    # the step-skipping predicate: T-1 and every multiple of 5
    from oracle import ddpm_oracle as D
    calls = [i for i in range(1, 600) if D.sampling3_calls_model(i, 600, epoch=12)]
    assert calls == [i for i in range(1, 600) if i % 5 == 0 or i == 599]
    assert all(Diffusion.sampling3_calls_model(i, 600, 12) == D.sampling3_calls_model(i, 600, 12) for i in range(1, 600))


def test_driver_reads_the_references_own_gt_lines(golden_dir):
    """``read_gt`` / ``writer_dict`` on real lines of the reference's gt/* lists (IAM, CVL, OOV, Norwegian formats) against the
    literal restatement of full_sampling.py:132-143 / train.py:370-388."""
    from oracle import driver_oracle as DO
    from worddiffusion_amd.driver import read_gt, writer_dict
    path = os.path.join(golden_dir, "gt_samples.txt")
    with open(path) as f:
        lines = [ln for ln in f.readlines() if ln.strip()]
    want = DO.parse_gt_lines(lines)
    rows = read_gt(path)
    assert rows == want and len(rows) == 23
    assert rows[0] == ("049", "a03-034-00-00", "Members") and rows[17] == ("202", "f07-069-03-07_202_owes_155_owes", "owes")
    assert rows[20][2] == "skr\u00e6ggs"
    assert writer_dict(rows) == DO.writer_dict_train(want)
    assert writer_dict(rows)["049"] == 0 and writer_dict(rows)["537"] == 1 and len(writer_dict(rows)) == 11


def test_underscore_alphabet_matches_reference_golden(golden_dir):
    """``label_padding`` of trainModifyCondition.py:166-180 ('_' alphabet, vocab 54) - outputs of the reference function."""
    from worddiffusion_amd.diffusion import (C_CLASSES_UNDERSCORE, VOCAB_SIZE_UNDERSCORE, label_padding_underscore)
    g = load_golden(golden_dir, "primitives_modcond")
    for w, ref in zip(g["words"], g["label_padding"]):
        assert label_padding_underscore(str(w)) == [int(v) for v in ref]
    assert VOCAB_SIZE_UNDERSCORE == int(g["vocab_size"]) == 54 and C_CLASSES_UNDERSCORE == str(g["c_classes"])
    with pytest.raises(KeyError):
        label_padding("a_b")  # train.py's 52-letter alphabet has no '_'
    with pytest.raises(KeyError):
        label_padding_underscore("a-b")
    d = Diffusion(args=make_args())
    assert d._text_features("to be", 2, underscore=True).tolist() == [[int(v) for v in g["label_padding"][0]]] * 2


def test_char_level_emb_flag_constructs(golden_dir):
    """args.charLevelEmb=1 (default of unet.py:1871) is accepted: same parameter tree; only context_dim=320 is legal."""
    m0 = UNetModel(args=make_args(charLevelEmb=0), **FULL)
    m1 = UNetModel(args=make_args(charLevelEmb=1), **FULL)
    assert list(m0.state_dict().keys()) == list(m1.state_dict().keys())
    g = load_golden(golden_dir, "fwd_base_full_charlevel")
    assert [str(k) for k in g["keys"]] == list(m1.state_dict().keys())
    with pytest.raises(NotImplementedError):
        UNetModel(args=make_args(charLevelEmb=1), **SMALL)  # unet.py:864 views (B, 10, 320)
    UNetModelPhosc(args=make_args(charLevelEmb=1), **SMALL)  # unetPhosc.py never reads the flag


def test_latent_cache_round_trip_and_dataset(tmp_path, golden_dir):
    """Cached-latent container (the tensor-only form of the reference's imageWordLineVae3*.pkl dictionaries,
    trainModifyCondition.py:300-325,452-458): converter, word -> character fallback order, dataset items, rank shards."""
    from worddiffusion_amd.driver import read_gt, writer_dict
    from worddiffusion_amd.latents import CachedLatentDataset, LatentCache, convert_latent_dict, save_latent_cache
    rows = read_gt(os.path.join(golden_dir, "gt_samples.txt"))[:14]  # the IAM lines
    wr = writer_dict(rows)
    rs = np.random.RandomState(0)
    # the reference's layout: {name: {"images": [1,4,8,32], ...}}; the last 4 images only exist in the second dictionary
    word_dict = {r[1] + ".png": {"images": torch.from_numpy(rs.standard_normal((1, 4, 8, 32)).astype(np.float32)), "x": 1}
                 for r in rows[:10]}
    char_dict = {r[1] + ".png": {"images": torch.from_numpy(rs.standard_normal((1, 4, 8, 32)).astype(np.float32))}
                 for r in rows[8:]}
    p1 = convert_latent_dict(word_dict, str(tmp_path / "word.safetensors"))
    p2 = convert_latent_dict(char_dict, str(tmp_path / "char.npz"))
    cache = LatentCache(p1, p2)
    assert len(cache) == 14 and rows[0][1] + ".png" in cache and "nope.png" not in cache
    k9 = rows[9][1] + ".png"
    assert torch.equal(cache[k9], word_dict[k9]["images"][0])          # the word dictionary wins (:452-456)
    k12 = rows[12][1] + ".png"
    assert torch.equal(cache[k12], char_dict[k12]["images"][0])        # fallback to the character dictionary
    with pytest.raises(KeyError):
        cache["nope.png"]
    ds = CachedLatentDataset(rows, wr, cache)
    it = ds[3]
    assert it["image_name"] == "a03-034-00-03.png" and it["label"] == "Cabinet" and it["s_id"] == 0
    assert it["word"].tolist() == label_padding("Cabinet") and it["latent"].shape == (4, 8, 32)
    seen = []
    for r in range(2):
        for b in ds.batches(3, shuffle=True, seed=5, epoch=1, rank=r, world=2, pin=False):
            assert b["latents"].shape == (3, 4, 8, 32) and b["words"].shape == (3, 10) and b["s_id"].dtype == torch.int64
            seen += b["image_names"]
    assert len(seen) == 12 and len(set(seen)) == 12  # 7 rows per rank, drop_last -> 2 batches of 3 each, disjoint
    a = [b["image_names"] for b in ds.batches(4, seed=5, epoch=0, pin=False)]
    assert a == [b["image_names"] for b in ds.batches(4, seed=5, epoch=0, pin=False)]
    assert a != [b["image_names"] for b in ds.batches(4, seed=5, epoch=1, pin=False)]
    # preloaded form: same batches
    ds2 = CachedLatentDataset(rows, wr, cache)
    assert ds2.preload()
    for b1, b2 in zip(ds.batches(4, seed=5, epoch=0, pin=False), ds2.batches(4, seed=5, epoch=0, pin=False)):
        assert b1["image_names"] == b2["image_names"] and b1["labels"] == b2["labels"]
        assert torch.equal(b1["latents"], b2["latents"]) and torch.equal(b1["words"], b2["words"]) and torch.equal(b1["s_id"], b2["s_id"])
    ds_u = CachedLatentDataset([("049", rows[0][1], "to be")], wr, cache, underscore=True)
    assert ds_u[0]["word"].tolist()[:5] == [46, 41, 53, 28, 31]
    short = CachedLatentDataset(rows + [("049", "missing-image", "x")], wr, cache, skip_missing=True)
    assert len(short) == 14
    save_latent_cache(str(tmp_path / "plain.safetensors"), {"a.png": torch.zeros(4, 8, 32)})
    assert LatentCache(str(tmp_path / "plain.safetensors"))["a.png"].shape == (4, 8, 32)


def test_phosc_descriptors_match_reference_golden(golden_dir, tmp_path):
    """PHOS / PHOC / PHOSC vectors against the outputs of the reference's own generators (tests/golden/phosc.npz, made by
    oracle/make_golden_phosc.py); the shape-count table travels as data inside the golden file."""
    from worddiffusion_amd.phosc import load_alphabet, phoc_vector, phos_vector, phosc_vector
    g = load_golden(golden_dir, "phosc")
    for version in ("eng", "gw", "nor"):
        letters = [str(x) for x in g[f"{version}:letters"]]
        table = g[f"{version}:table"]
        index = {k: i for i, k in enumerate(letters)}  # later rows win, as in create_alphabet_dictionary
        for w, ph, pc in zip(g[f"{version}:words"], g[f"{version}:phos"], g[f"{version}:phoc"]):
            w = str(w)
            assert np.array_equal(phos_vector(w, index, table), ph), (version, w)
            assert phoc_vector(w, version) == [int(v) for v in pc], (version, w)
            full = phosc_vector(w, index, table, version)
            assert full.dtype == np.int64 and np.array_equal(full, np.concatenate([ph, pc]).astype(np.int64))
        # csv round trip of the loader (same parse as the reference: first column = letter)
        p = tmp_path / f"{version}.csv"
        p.write_text("\n".join(",".join([k] + [str(int(v)) for v in row]) for k, row in zip(letters, table)))
        idx2, tab2 = load_alphabet(str(p))
        assert idx2 == index and np.array_equal(tab2, table)
    assert len(phosc_vector("Stop", {k: i for i, k in enumerate(str(x) for x in g["eng:letters"])}, g["eng:table"])) == 769
    with pytest.raises(KeyError):
        phos_vector("é", {"a": 0}, np.zeros((1, 11), dtype=int))


def test_vae_decoder_state_dict_layout(tmp_path):
    """The decoder container has the diffusers AutoencoderKL keys / shapes of the SD-v1.5 VAE (49,490,179 decoder parameters +
    the 1x1 post_quant_conv), accepts the pre-0.14 attention names, ignores the encoder half, and round-trips through a local
    diffusers-layout directory (safetensors)."""
    import json
    from safetensors.torch import save_file
    from worddiffusion_amd.vae import AutoencoderKL
    m = AutoencoderKL()
    sd = m.state_dict()
    assert sum(v.numel() for k, v in sd.items() if k.startswith("decoder.")) == 49_490_179
    assert sum(v.numel() for k, v in sd.items() if k.startswith("post_quant_conv.")) == 20
    expect = {"decoder.conv_in.weight": (512, 4, 3, 3), "decoder.mid_block.attentions.0.to_q.weight": (512, 512),
              "decoder.mid_block.attentions.0.to_out.0.bias": (512,), "decoder.up_blocks.0.upsamplers.0.conv.weight": (512, 512, 3, 3),
              "decoder.up_blocks.2.resnets.0.conv_shortcut.weight": (256, 512, 1, 1),
              "decoder.up_blocks.3.resnets.0.conv1.weight": (128, 256, 3, 3), "decoder.up_blocks.3.resnets.2.norm2.weight": (128,),
              "decoder.conv_norm_out.weight": (128,), "decoder.conv_out.weight": (3, 128, 3, 3), "post_quant_conv.weight": (4, 4, 1, 1)}
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    assert not any("upsamplers" in k for k in sd if k.startswith("decoder.up_blocks.3."))
    # old attention naming + encoder entries in a checkpoint
    old = {}
    for k, v in sd.items():
        for new_n, old_n in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if f"attentions.0.{new_n}." in k:
                k = k.replace(f"attentions.0.{new_n}.", f"attentions.0.{old_n}.")
                if v.dim() == 2:
                    v = v[:, :, None, None]  # some exports keep the projections as 1x1 convolutions
        old[k] = torch.full_like(v, 0.5)
    old["encoder.conv_in.weight"] = torch.zeros(128, 3, 3, 3)
    old["quant_conv.weight"] = torch.zeros(8, 8, 1, 1)
    m.load_state_dict(old)
    assert all(float(v.min()) == 0.5 == float(v.max()) for v in m.state_dict().values())
    d = tmp_path / "sd" / "vae"
    d.mkdir(parents=True)
    small = AutoencoderKL(block_out_channels=(64, 128), layers_per_block=1)
    save_file({k: v.contiguous() for k, v in small.state_dict().items()}, str(d / "diffusion_pytorch_model.safetensors"))
    (d / "config.json").write_text(json.dumps({"block_out_channels": [64, 128], "layers_per_block": 1, "latent_channels": 4,
                                               "_class_name": "AutoencoderKL", "sample_size": 512}))
    back = AutoencoderKL.from_pretrained(str(tmp_path / "sd"), subfolder="vae")
    assert back.config.block_out_channels == (64, 128)
    for k, v in small.state_dict().items():
        assert torch.equal(v, back.state_dict()[k])
    with pytest.raises(Exception):
        back.decode(torch.zeros(1, 4, 4, 8))  # no CPU fallback
