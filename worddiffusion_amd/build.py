"""Build libwdiff_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m worddiffusion_amd.build [--force]
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwdiff_hip.so")
SOURCES = ["wd_gemm.hip", "wd_gemmw.hip", "wd_gemmq.hip", "wd_ff.hip", "wd_dw.hip", "wd_norm.hip", "wd_attn.hip", "wd_xattn.hip", "wd_misc.hip", "wd_train.hip", "wd_bwd.hip", "wd_pack.hip", "wd_runtime.hip"]
HEADERS = [os.path.join(CSRC, "wd_common.h"), os.path.join(CSRC, "wd_gemm_epi.h"), os.path.join(CSRC, "wd_gemm_priv.h"), os.path.join(os.path.dirname(HERE), "include", "wdiff_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-gpu-rdc"]
# per-file flags.  wd_attn.hip: MFMA accumulators in ordinary VGPRs - for its 256-thread kernels hipcc otherwise keeps them in AGPRs and
# moves every value the softmax touches through v_accvgpr_read / _write (432 such moves per key block in attn_mfma_kernel<5,3>);
# the 4x16-level attention launches run 7-8 % faster without them, the 8x32 ones 2-3 %.  (The 512-thread GEMM kernels are tuned
# against the default register allocation and keep it.)
FILE_FLAGS = {"wd_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
if os.environ.get("WDIFF_EXPERIMENTAL", "0") != "0":  # the opt-in GEMM variants that lost their A/B (csrc/wd_gemm.hip)
    FLAGS.append("-DWDIFF_EXPERIMENTAL")
if os.environ.get("WDIFF_STAMPS", "0") != "0":  # s_memtime stamps inside wd_dw_kernel / wd_gemmw_kernel (tools/*_bench.py --stamps)
    FLAGS += ["-DWD_DW_STAMPS", "-DWD_GEMMW_STAMPS", "-DWD_Q_STAMPS"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    h.update(" ".join(FLAGS + FILE_FLAGS.get(os.path.basename(paths[0]), [])).encode())
    return h.hexdigest()


def _stale(target: str, deps, manifest: dict) -> bool:
    """An object is up to date when it exists and was built from exactly these source bytes and flags (content hash kept in
    csrc/build/manifest.json) - not merely when its mtime is newer, which says nothing after a checkout or a copy."""
    return not os.path.exists(target) or manifest.get(os.path.basename(target)) != _digest(deps)


def build_native(force: bool = False, verbose: bool = True) -> str:
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    mpath = os.path.join(objdir, "manifest.json")
    try:
        with open(mpath) as f:
            manifest = json.load(f)
    except (OSError, ValueError):
        manifest = {}
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS, manifest):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
        for s, o in jobs:
            manifest[os.path.basename(o)] = _digest([s] + HEADERS)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    lib_key = hashlib.sha256("".join(manifest.get(os.path.basename(o), "") for o in objs).encode()).hexdigest()
    if force or jobs or not os.path.exists(LIB) or manifest.get("libwdiff_hip.so") != lib_key:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        manifest["libwdiff_hip.so"] = lib_key
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1)
    return LIB


def source_digest() -> str:
    """Content hash of every kernel source + header + the compiler flags: what the library in this tree was built from
    (recorded in csrc/build/manifest.json by build_native; bench.py / smoke() can report it)."""
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv))
