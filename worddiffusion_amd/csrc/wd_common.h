// Shared device helpers and the launch wrapper of libwdiff_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wdiff_hip.h"

#define WD_CLS_GEMM 0
#define WD_CLS_GNSTATS 1
#define WD_CLS_GNAPPLY 2
#define WD_CLS_LN 3
#define WD_CLS_ATTN 4
#define WD_CLS_OTHER 5
#define WD_CLS_GEMM_OTHER 6   // wd_gemm tile shapes other than the dominant 128x160 kernel
#define WD_CLS_GEMM_REDUCE 7  // split-K combine pass
#define WD_CLS_GEMM_2CU 8     // wd_gemm4_kernel (two workgroups per CU)
#define WD_CLS_GEMM_WDIRECT 9 // wd_gemmw_kernel (weights straight to registers)
#define WD_CLS_FF 10          // wd_ff_kernel (fused GEGLU feed-forward)
#define WD_CLS_DW 11          // wd_dw_kernel (weight gradients from the row-major planes)
#define WD_CLS_GEMM_Q 12      // wd_gemmq_kernel (64 x 80 tiles, all of K in the workgroup: the 4 x 16 level)

// ---- profiling hooks (wd_runtime.hip) -------------------------------------------------------------
extern "C" int wd_prof_is_on();
void wd_prof_begin(int cls, hipStream_t s, double flops);
void wd_prof_end(hipStream_t s);

struct WdLaunchScope {
    hipStream_t s;
    bool on;
    WdLaunchScope(int cls, hipStream_t st, double flops = 0.0) : s(st), on(wd_prof_is_on() != 0) {
        if (on) wd_prof_begin(cls, s, flops);
    }
    ~WdLaunchScope() {
        if (on) wd_prof_end(s);
    }
};

static inline int wd_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? WD_OK : WD_ELAUNCH;
}

// ---- bf16 split ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wd_f2bf_bits(float x) {  // round-to-nearest-even, finite inputs
    uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float wd_bf_bits2f(uint32_t b) { return __uint_as_float(b << 16); }

// gfx950 converts fp32 -> bf16 in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even, two values per instruction)
typedef __attribute__((ext_vector_type(2))) float wd_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 wd_bf16x2;
__device__ __forceinline__ uint32_t wd_pack_bf16x2(float a, float b) {  // a in the low half
    const wd_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wd_bf16x2));
}

__device__ __forceinline__ void wd_split1(float x, uint32_t& hi, uint32_t& lo) {
    hi = wd_pack_bf16x2(x, 0.f) & 0xffffu;
    lo = wd_pack_bf16x2(x - __uint_as_float(hi << 16), 0.f) & 0xffffu;
}

// four floats -> 4 hi (8 bytes) + 4 lo (8 bytes)
__device__ __forceinline__ void wd_split4(const float4 v, uint2& hi, uint2& lo) {
    const uint32_t h01 = wd_pack_bf16x2(v.x, v.y), h23 = wd_pack_bf16x2(v.z, v.w);
    hi = make_uint2(h01, h23);
    lo = make_uint2(wd_pack_bf16x2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u)),
                    wd_pack_bf16x2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u)));
}

// x * sigmoid(x) on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: 1 ulp each) - the IEEE expf + division form costs
// ~25 VALU instructions per element, which made the SiLU the largest part of the GroupNorm-apply kernels' arithmetic
__device__ __forceinline__ float wd_silu(float x) { return __fdividef(x, 1.0f + __expf(-x)); }
// exact-erf GELU (unet.py:136 F.gelu default).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, the size of an fp32 ulp
// of 1) on the hardware rcp / exp units: about a third of the instructions of libm's erff, which made this the largest
// part of the GEGLU epilogue's arithmetic (40 gate values per thread and tile).
__device__ __forceinline__ float wd_erf(float x) {
    const float ax = fabsf(x);
    const float t = __fdividef(1.0f, 1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float wd_gelu_erf(float x) { return 0.5f * x * (1.0f + wd_erf(x * 0.70710678118654752440f)); }

// 64-lane reductions on the DPP path (quad permutes + row rotates inside each row of 16 lanes, then one v_readlane per
// row): ~10 VALU-rate instructions instead of six dependent ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float wd_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wd_row16_sum(float v) {  // sum over the 16 lanes of a DPP row, result in all of them
    v += wd_dpp<0xB1>(v);
    v += wd_dpp<0x4E>(v);
    v += wd_dpp<0x124>(v);
    v += wd_dpp<0x128>(v);
    return v;
}
__device__ __forceinline__ float wd_wave_sum(float v) {
    v += wd_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v += wd_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v += wd_dpp<0x124>(v);  // row_ror:4
    v += wd_dpp<0x128>(v);  // row_ror:8  -> every lane holds the sum of its row of 16
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wd_wave_max(float v) {
    v = fmaxf(v, wd_dpp<0xB1>(v));
    v = fmaxf(v, wd_dpp<0x4E>(v));
    v = fmaxf(v, wd_dpp<0x124>(v));
    v = fmaxf(v, wd_dpp<0x128>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
