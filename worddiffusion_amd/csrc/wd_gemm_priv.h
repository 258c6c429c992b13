// Internal links between the translation units of the tap-gather GEMM (not part of the C ABI).
#pragma once
#include "wd_common.h"

// wd_gemmw.hip: the 64 x 320 "weights straight to registers" kernel (wd_gemm_args.w_layout == 3).  `a` has been validated and
// its ksplit resolved by wd_gemm().
int wd_gemmw_launch(const wd_gemm_args& a, hipStream_t st);
// wd_gemm.hip: the split-K combine launch for slabs of a launch with bm-row tiles (statistics layout follows bm).
int wd_gemm_launch_reduce(const wd_gemm_args& a, hipStream_t st, int bm);
// wd_gemmq.hip: 64 x 80 tiles, all of K inside the workgroup (wd_gemm_args.tile == 64080: the 3x3 layers over 64-position samples).
bool wd_gemmq_applies(const wd_gemm_args& a);
int wd_gemmq_launch(const wd_gemm_args& a, hipStream_t st);
