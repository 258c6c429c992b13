"""ctypes binding of libwdiff_hip.so (include/wdiff_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing or a symbol is
absent, importing/using it raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WDIFF_LIB") or os.path.join(_HERE, "libwdiff_hip.so")  # (WDIFF_LIB: A/B builds of the kernels)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "wdiff_hip.h")

WD_OK, WD_EINVAL, WD_ELAUNCH, WD_ESTATE = 0, -1, -2, -3
ACT_NONE, ACT_SILU, ACT_GEGLU = 0, 1, 2
NCLASS = 13
CLASS_NAMES = ("gemm", "gn_stats", "gn_apply", "layernorm", "attention", "other", "gemm_other_tiles", "gemm_splitk_reduce",
               "gemm_two_per_cu", "gemm_weights_to_registers", "feed_forward_fused", "weight_gradient", "gemm_small_maps_whole_k")

_vp = C.c_void_p
_i = C.c_int
_f = C.c_float
_u64 = C.c_uint64


class WdSrc(C.Structure):
    _fields_ = [("hi", _vp), ("lo", _vp), ("gather", _vp), ("ld", C.c_int32), ("c", C.c_int32),
                ("ntaps", C.c_int32), ("hw_src", C.c_int32)]


class WdGemmArgs(C.Structure):
    _fields_ = [("src", WdSrc * 2), ("nsrc", C.c_int32), ("npass", C.c_int32), ("w_hi", _vp), ("w_lo", _vp),
                ("m", C.c_int32), ("n", C.c_int32), ("ktot", C.c_int32), ("hw_out", C.c_int32),
                ("bias", _vp), ("rowvec", _vp), ("rowvec_ld", C.c_int32), ("resid", _vp), ("resid_ld", C.c_int32),
                ("resid_rows", _vp), ("act", C.c_int32), ("out_f32", _vp), ("out_ld", C.c_int32),
                ("out_hi", _vp), ("out_lo", _vp), ("out_pl_ld", C.c_int32), ("tile", C.c_int32),
                ("w_layout", C.c_int32), ("slab_rows", C.c_int32), ("ksplit", C.c_int32), ("ws", _vp),
                ("ws_floats", C.c_int64), ("stat_part", _vp), ("stat_cpg", C.c_int32), ("dbg", C.c_int32),
                ("tickets", _vp), ("ntickets", C.c_int32), ("gn_gamma", _vp), ("gn_beta", _vp), ("gn_eps", C.c_float),
                ("gn_silu", C.c_int32), ("gn_cpg", C.c_int32), ("a32", _vp), ("a32_ld", C.c_int32), ("a32_part", _vp),
                ("a32_nchunk", C.c_int32), ("a32_pcpg", C.c_int32), ("a32_cpg", C.c_int32), ("a32_gamma", _vp), ("a32_beta", _vp),
                ("a32_eps", C.c_float), ("a32_silu", C.c_int32), ("ln_gamma", _vp), ("ln_beta", _vp), ("ln_eps", C.c_float),
                ("w_ngroups", C.c_int32), ("w_group_stride", C.c_int64)]


class WdFfArgs(C.Structure):
    _fields_ = [("x_hi", _vp), ("x_lo", _vp), ("x_ld", C.c_int32), ("m", C.c_int32), ("c", C.c_int32), ("inner", C.c_int32),
                ("w1_hi", _vp), ("w1_lo", _vp), ("b1", _vp), ("w2_hi", _vp), ("w2_lo", _vp), ("b2", _vp), ("resid", _vp),
                ("resid_ld", C.c_int32), ("out_f32", _vp), ("out_ld", C.c_int32), ("out_hi", _vp), ("out_lo", _vp),
                ("out_pl_ld", C.c_int32), ("w3_hi", _vp), ("w3_lo", _vp), ("b3", _vp), ("resid3", _vp), ("resid3_ld", C.c_int32),
                ("stat_part", _vp), ("stat_cpg", C.c_int32), ("hw_out", C.c_int32), ("npass", C.c_int32)]


class WdDwArgs(C.Structure):
    _fields_ = [("d_hi", _vp), ("d_lo", _vp), ("x_hi", _vp), ("x_lo", _vp), ("gather", _vp), ("grad", _vp), ("ws", _vp),
                ("ws_floats", C.c_int64), ("d_ld", C.c_int32), ("x_ld", C.c_int32), ("grad_ld", C.c_int32), ("ntaps", C.c_int32),
                ("hw_out", C.c_int32), ("hw_src", C.c_int32), ("m", C.c_int32), ("n", C.c_int32), ("c", C.c_int32),
                ("npass", C.c_int32), ("accumulate", C.c_int32), ("nslice", C.c_int32), ("dbg", C.c_int32), ("reserved", C.c_int32),
                ("stamps", _vp), ("items", _vp), ("nitems", C.c_int32), ("reserved2", C.c_int32)]


class WdDwItem(C.Structure):
    _fields_ = [("d_hi", _vp), ("d_lo", _vp), ("x_hi", _vp), ("x_lo", _vp), ("grad", _vp), ("d_ld", C.c_int32), ("x_ld", C.c_int32),
                ("grad_ld", C.c_int32), ("accumulate", C.c_int32)]


_SIGS = {
    "wd_gemm": (_i, [C.POINTER(WdGemmArgs), _vp]),
    "wd_dw": (_i, [C.POINTER(WdDwArgs), _vp]),
    "wd_dw_supported": (_i, [_i, _i, _i, _i, _i]),
    "wd_dw_slices": (_i, [_i, _i, _i, _i]),
    "wd_dw_args_bytes": (_i, []),
    "wd_dw_group": (_i, [C.POINTER(WdDwArgs), _vp, _vp, _i, _vp]),
    "wd_dw_group_slices": (_i, [_i, _i, _i, _i, _i]),
    "wd_dw_item_bytes": (_i, []),
    "wd_gemm_auto_ksplit": (_i, [_i, _i, _i, C.c_int64]),
    "wd_gemm_pack_w": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "wd_ff_fused": (_i, [C.POINTER(WdFfArgs), _vp]),
    "wd_ff_supported": (_i, [_i, _i]),
    "wd_ff_args_bytes": (_i, []),
    "wd_gemm_args_bytes": (_i, []),
    "wd_gemm_experimental": (_i, []),
    "wd_gn_nchunk": (_i, [_i]),
    "wd_gn_stats": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "wd_gn_fold_chunks": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "wd_gn_conv3x3_few_supported": (_i, [_i, _i, _i]),
    "wd_gn_conv3x3_few": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _i, _vp, _vp]),
    "wd_gn_apply": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "wd_gn_apply2": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _i, _vp,
                          _vp, _vp, _vp]),
    "wd_layernorm": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    "wd_split": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "wd_attention": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "wd_attention_packed_elems": (C.c_int64, [_i, _i, _i, _i]),
    "wd_attention_pack_kv": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "wd_attention_packed": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "wd_timestep_embedding": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "wd_embed_tokens": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "wd_im2col3x3": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "wd_nchw_to_tokens": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "wd_tokens_to_nchw": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "wd_ddpm_step": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp]),
    "wd_advance_timestep": (_i, [_vp, _i, _vp, _i, _vp]),
    "wd_randn": (_i, [_vp, _i, _i, _u64, _u64, C.c_uint32, _vp]),
    "wd_noise_images": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "wd_copy2d": (_i, [_vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int64, _vp]),
    "wd_ema_update": (_i, [_vp, _vp, C.c_int64, C.c_double, _vp]),
    "wd_adamw_table_entry_bytes": (_i, []),
    "wd_adamw_chunk": (_i, []),
    "wd_adamw_multi": (_i, [_vp, _i, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, _i,
                           C.c_double, _vp]),
    "wd_mse_loss": (_i, [_vp, _vp, C.c_int64, _vp, _vp, _vp, _i, _vp]),
    "wd_transpose_planes": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "wd_repack_entry_bytes": (_i, []),
    "wd_repack_tile": (_i, []),
    "wd_repack_vchunk": (_i, []),
    "wd_repack_multi": (_i, [_vp, _i, C.c_int64, _vp]),
    "wd_attention_bwd_scratch_floats": (C.c_int64, [_i, _i, _i, _i]),
    "wd_attention_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _vp, _i, _vp, C.c_int64, _vp]),
    "wd_xattn_supported": (_i, [_i, _i, _i]),
    "wd_xattn_fold": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "wd_xattn_fused": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _i, _vp, _vp, _vp]),
    "wd_dout_prep_rows": (_i, []),
    "wd_dout_prep": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wd_dout_prep_geglu": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wd_colsum_finish": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _f, _vp]),
    "wd_colsum_entry_bytes": (_i, []),
    "wd_colsum_finish_multi": (_i, [_vp, _i, _i, _vp]),
    "wd_emb_combine": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "wd_select_rows": (_i, [_vp, _vp, _i, C.c_int64, _i, _vp, _vp]),
    "wd_xattn_pair": (_i, [_vp, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f,
                           _vp, _vp, _i, _vp]),
    "wd_add": (_i, [_vp, _vp, C.c_int64, _vp]),
    "wd_permute_dw": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "wd_colsum": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _f, _vp, C.c_int64, _vp]),
    "wd_gn_bwd_nchunk": (_i, [_i]),
    "wd_gn_bwd_stats": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _f, _i, _vp, _vp]),
    "wd_gn_bwd_apply": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _f, _i, _vp, _vp, _i, _i, _vp]),
    "wd_gn_bwd_fused_supported": (_i, [_i, _i, _i]),
    "wd_gn_bwd_fused": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _f, _i, _vp, _vp, _i, _i, _vp]),
    "wd_layernorm_bwd_nblk": (_i, [_i]),
    "wd_layernorm_bwd": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _f, _vp, _i, _i, _vp, _vp]),
    "wd_attention_bwd_small_nwg": (_i, [_i, _i, _i, _i]),
    "wd_attention_bwd_small": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp,
                                   C.POINTER(_i), _vp]),
    "wd_geglu_fwd": (_i, [_vp, _i, C.c_int64, _i, _vp, _vp, _i, _vp]),
    "wd_geglu_bwd": (_i, [_vp, _i, _vp, _i, C.c_int64, _i, _vp, _i, _vp]),
    "wd_silu_bwd": (_i, [_vp, _vp, C.c_int64, _vp, _vp]),
    "wd_pool2x2_sum": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "wd_embedding_bwd": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp]),
    "wd_graph_begin": (_i, [_vp]),
    "wd_graph_end": (_i, [_vp, C.POINTER(_vp)]),
    "wd_graph_launch": (_i, [_vp, _vp]),
    "wd_graph_destroy": (_i, [_vp]),
    "wd_prof_enable": (_i, [_i]),
    "wd_prof_collect": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "wd_prof_collect_flops": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "wd_version": (C.c_char_p, []),
    "wd_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
}


def header_symbols() -> list:
    """Every function the public header declares (used by the CPU test that the .so exports them all)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wd_[a-z0-9_]+)\s*\(", txt)))


class NativeError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load (once) and type the shared library.  Raises if it is absent: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} is missing: build it with `python -m worddiffusion_amd.build` "
                          "(hipcc --offload-arch=gfx950); worddiffusion_amd has no CPU/eager fallback")
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(l, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if l.wd_gemm_args_bytes() != C.sizeof(WdGemmArgs):  # a stale library beside newer Python (or the reverse)
        raise NativeError(f"{LIB_PATH}: wd_gemm_args is {l.wd_gemm_args_bytes()} bytes in the library, {C.sizeof(WdGemmArgs)} in "
                          "worddiffusion_amd/_native.py - rebuild with `python -m worddiffusion_amd.build`")
    if l.wd_ff_args_bytes() != C.sizeof(WdFfArgs):
        raise NativeError(f"{LIB_PATH}: wd_ff_args is {l.wd_ff_args_bytes()} bytes in the library, {C.sizeof(WdFfArgs)} in "
                          "worddiffusion_amd/_native.py - rebuild with `python -m worddiffusion_amd.build`")
    if l.wd_dw_item_bytes() != C.sizeof(WdDwItem):
        raise NativeError(f"{LIB_PATH}: wd_dw_item is {l.wd_dw_item_bytes()} bytes in the library, {C.sizeof(WdDwItem)} in "
                          "worddiffusion_amd/_native.py - rebuild with `python -m worddiffusion_amd.build`")
    if l.wd_dw_args_bytes() != C.sizeof(WdDwArgs):
        raise NativeError(f"{LIB_PATH}: wd_dw_args is {l.wd_dw_args_bytes()} bytes in the library, {C.sizeof(WdDwArgs)} in "
                          "worddiffusion_amd/_native.py - rebuild with `python -m worddiffusion_amd.build`")
    _lib = l
    return l


_ERR = {WD_EINVAL: "invalid argument", WD_ELAUNCH: "HIP launch/runtime error", WD_ESTATE: "bad call order"}


def check(rc: int, what: str) -> None:
    if rc != WD_OK:
        raise NativeError(f"{what} failed: {_ERR.get(rc, rc)}")
