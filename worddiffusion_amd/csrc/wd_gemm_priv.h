// Internal links between the translation units of the tap-gather GEMM (not part of the C ABI).
#pragma once
#include "wd_common.h"

// wd_gemmw.hip: the 64 x 320 "weights straight to registers" kernel (wd_gemm_args.w_layout == 3).  `a` has been validated and
// its ksplit resolved by wd_gemm().
int wd_gemmw_launch(const wd_gemm_args& a, hipStream_t st);
// wd_gemm.hip: the split-K combine launch for slabs of a launch with bm-row tiles (statistics layout follows bm).
int wd_gemm_launch_reduce(const wd_gemm_args& a, hipStream_t st, int bm);
