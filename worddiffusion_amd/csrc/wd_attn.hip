// softmax(q k^T * scale) v, exact fp32, any head size / key count (scores for a tile of TQ queries live in LDS).
// Reference: CrossAttention.forward unet.py:185-279 / unetPhosc.py:176-198 (scale = d_head^-0.5),
// Word_Attention.forward unet.py:823-836 (one head of width 320, scale 1).
//
// One workgroup = one (sample, head, TQ-query tile):
//   1. thread j computes the TQ scores of key j (k row streamed once from global as float4, q broadcast from LDS);
//   2. one wave per query row does max / exp / sum in LDS (wave shuffles);
//   3. thread d accumulates out[:, d] = sum_j p[:, j] v[j, d]  (v rows read coalesced across threads).
// On the bench configuration (base UNet) the keys are the 10 text tokens, so this kernel is latency-bound and
// small; the PHOSC variant (779 keys, 256-token self-attention) is where an MFMA flash kernel will replace it.
#include <stdlib.h>

#include "wd_common.h"

namespace {

constexpr int TQ = 8;

__global__ void __launch_bounds__(256) attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                   int ldk, const float* __restrict__ v, int ldv, int heads, int nq,
                                                   int nk, int d, float scale, float* __restrict__ out_f32,
                                                   wd_bf16* __restrict__ out_hi, wd_bf16* __restrict__ out_lo,
                                                   int out_ld, int out_rows, int out_row0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_q = reinterpret_cast<float*>(smem);  // [TQ][d]
    float* s_p = s_q + TQ * d;                    // [TQ][nk]
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
    const int i0 = blockIdx.x * TQ;
    const int nqt = min(TQ, nq - i0);
    const int tid = threadIdx.x;
    const int hoff = h * d;

    for (int e = tid; e < TQ * d; e += 256) {
        const int i = e / d, dd = e - i * d;
        s_q[e] = (i < nqt) ? q[((long)b * nq + i0 + i) * ldq + hoff + dd] : 0.0f;
    }
    __syncthreads();

    const int d4 = d >> 2;
    for (int j = tid; j < nk; j += 256) {
        const float* kr = k + ((long)b * nk + j) * ldk + hoff;
        float acc[TQ];
#pragma unroll
        for (int i = 0; i < TQ; ++i) acc[i] = 0.0f;
        for (int c = 0; c < d4; ++c) {
            const float4 kv = *reinterpret_cast<const float4*>(kr + c * 4);
#pragma unroll
            for (int i = 0; i < TQ; ++i) {
                const float4 qv = *reinterpret_cast<const float4*>(s_q + i * d + c * 4);
                acc[i] = fmaf(qv.x, kv.x, acc[i]);
                acc[i] = fmaf(qv.y, kv.y, acc[i]);
                acc[i] = fmaf(qv.z, kv.z, acc[i]);
                acc[i] = fmaf(qv.w, kv.w, acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < TQ; ++i) s_p[i * nk + j] = acc[i] * scale;
    }
    __syncthreads();

    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int i = wave; i < TQ; i += 4) {
            float* p = s_p + i * nk;
            float mx = -3.4e38f;
            for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, p[j]);
            mx = wd_wave_max(mx);
            float sum = 0.0f;
            for (int j = lane; j < nk; j += 64) {
                const float e = expf(p[j] - mx);
                p[j] = e;
                sum += e;
            }
            sum = wd_wave_sum(sum);
            const float inv = 1.0f / sum;
            for (int j = lane; j < nk; j += 64) p[j] *= inv;
        }
    }
    __syncthreads();

    for (int dd = tid; dd < d; dd += 256) {
        float acc[TQ];
#pragma unroll
        for (int i = 0; i < TQ; ++i) acc[i] = 0.0f;
        const float* vc = v + ((long)b * nk) * ldv + hoff + dd;
        for (int j = 0; j < nk; ++j) {
            const float vv = vc[(long)j * ldv];
#pragma unroll
            for (int i = 0; i < TQ; ++i) acc[i] = fmaf(s_p[i * nk + j], vv, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < TQ; ++i) {
            if (i < nqt) {
                const long row = (long)b * out_rows + out_row0 + i0 + i;
                const long o = row * out_ld + hoff + dd;
                if (out_f32) out_f32[o] = acc[i];
                if (out_hi) {
                    uint32_t hb, lb;
                    wd_split1(acc[i], hb, lb);
                    out_hi[o] = (wd_bf16)hb;
                    if (out_lo) out_lo[o] = (wd_bf16)lb;
                }
            }
        }
    }
}

// ---- few keys (the base UNet attends to the 10 text tokens only, unet.py:337-341): one thread per (query token,
// head).  K, V of the sample and the q rows of TPW tokens are staged in LDS with coalesced 16-byte loads (token rows
// padded by 4 floats so that the per-(token, head) float4 reads are bank-conflict free); the nk scores stay in
// registers; the output overwrites the thread's own q slice in LDS and leaves through coalesced plane stores.
// HBM-bound: q is read once and the planes are written once, in full lines.
constexpr int NKS = 16;
template <int NK>  // NK > 0: exactly NK keys (branch-free, fully unrolled); NK == 0: up to NKS keys, predicated
__global__ void __launch_bounds__(256) attn_small_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                         int ldk, const float* __restrict__ v, int ldv, int heads, int nq,
                                                         int nk, int d, float scale, float* __restrict__ out_f32,
                                                         wd_bf16* __restrict__ out_hi, wd_bf16* __restrict__ out_lo,
                                                         int out_ld, int out_rows, int out_row0, int tpw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int inner = heads * d;
    const int pitch = inner + 4;
    float* s_k = reinterpret_cast<float*>(smem);  // [nk][inner]
    float* s_v = s_k + nk * inner;                // [nk][inner]
    float* s_q = s_v + nk * inner;                // [tpw][pitch]
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int i4 = inner >> 2;
    const int tok0 = blockIdx.x * tpw;
    const int ntok = min(tpw, nq - tok0);
    for (int e = tid; e < nk * i4; e += 256) {
        const int j = e / i4, c = (e - j * i4) * 4;
        *reinterpret_cast<float4*>(s_k + j * inner + c) =
            *reinterpret_cast<const float4*>(k + ((long)b * nk + j) * ldk + c);
        *reinterpret_cast<float4*>(s_v + j * inner + c) =
            *reinterpret_cast<const float4*>(v + ((long)b * nk + j) * ldv + c);
    }
    for (int e = tid; e < ntok * i4; e += 256) {
        const int t = e / i4, c = (e - t * i4) * 4;
        *reinterpret_cast<float4*>(s_q + t * pitch + c) =
            *reinterpret_cast<const float4*>(q + ((long)b * nq + tok0 + t) * ldq + c);
    }
    __syncthreads();
    const int h = tid % heads, tl = tid / heads;
    if (tl < ntok) {
        const int hoff = h * d;
        float* qr = s_q + tl * pitch + hoff;
        constexpr int NJ = NK > 0 ? NK : NKS;
        float sc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) sc[j] = 0.0f;
        const int d4 = d >> 2;
        for (int c = 0; c < d4; ++c) {
            const float4 qv = *reinterpret_cast<const float4*>(qr + c * 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (NK > 0 || j < nk) {
                    const float4 kv = *reinterpret_cast<const float4*>(s_k + j * inner + hoff + c * 4);
                    sc[j] = fmaf(qv.x, kv.x, sc[j]);
                    sc[j] = fmaf(qv.y, kv.y, sc[j]);
                    sc[j] = fmaf(qv.z, kv.z, sc[j]);
                    sc[j] = fmaf(qv.w, kv.w, sc[j]);
                }
            }
        }
        float mx = -3.4e38f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (NK > 0 || j < nk) {
                sc[j] *= scale;
                mx = fmaxf(mx, sc[j]);
            }
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (NK > 0 || j < nk) {
                sc[j] = expf(sc[j] - mx);
                sum += sc[j];
            }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int j = 0; j < NJ; ++j) sc[j] *= inv;
        for (int c = 0; c < d4; ++c) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (NK > 0 || j < nk) {
                    const float4 vv = *reinterpret_cast<const float4*>(s_v + j * inner + hoff + c * 4);
                    o.x = fmaf(sc[j], vv.x, o.x);
                    o.y = fmaf(sc[j], vv.y, o.y);
                    o.z = fmaf(sc[j], vv.z, o.z);
                    o.w = fmaf(sc[j], vv.w, o.w);
                }
            }
            *reinterpret_cast<float4*>(qr + c * 4) = o;  // this (token, head) slice is private to the thread
        }
    }
    __syncthreads();
    for (int e = tid; e < ntok * i4; e += 256) {
        const int t = e / i4, c = (e - t * i4) * 4;
        const float4 o = *reinterpret_cast<const float4*>(s_q + t * pitch + c);
        const long oo = ((long)b * out_rows + out_row0 + tok0 + t) * out_ld + c;
        if (out_f32) *reinterpret_cast<float4*>(out_f32 + oo) = o;
        if (out_hi) {
            uint2 hb, lb;
            wd_split4(o, hb, lb);
            *reinterpret_cast<uint2*>(out_hi + oo) = hb;
            if (out_lo) *reinterpret_cast<uint2*>(out_lo + oo) = lb;
        }
    }
}

}  // namespace

extern "C" int wd_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int batch,
                            int heads, int nq, int nk, int d, float scale, float* out_f32, wd_bf16* out_hi,
                            wd_bf16* out_lo, int out_ld, int out_rows, int out_row0, void* stream) {
    if (!q || !k || !v || (!out_f32 && !out_hi)) return WD_EINVAL;
    if (batch <= 0 || heads <= 0 || nq <= 0 || nk <= 0 || d <= 0 || d % 4) return WD_EINVAL;
    if (ldq % 4 || ldk % 4 || ldv % 4) return WD_EINVAL;
    const size_t smem = (size_t)TQ * (d + nk) * sizeof(float);
    if (smem > 160 * 1024) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {
        const int inner = heads * d;
        const int tpw = 256 / heads;  // tokens per workgroup
        const size_t small_smem = ((size_t)2 * nk * inner + (size_t)tpw * (inner + 4)) * sizeof(float);
        if (nk <= NKS && heads <= 256 && small_smem <= 150 * 1024 && out_ld % 4 == 0 && ldq % 4 == 0 &&
            !getenv("WDIFF_ATTN_GENERIC")) {
            static size_t small_set = 64 * 1024;
            if (small_smem > small_set) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_small_kernel<10>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_smem) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_small_kernel<0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_smem) != hipSuccess)
                    return WD_ELAUNCH;
                small_set = small_smem;
            }
            WdLaunchScope scope(WD_CLS_ATTN, st);
            if (nk == 10)  // MAX_CHARS = 10 text tokens (train.py:28): the shape of every base-UNet cross-attention
                hipLaunchKernelGGL(attn_small_kernel<10>, dim3((nq + tpw - 1) / tpw, batch), dim3(256), small_smem, st, q,
                                   ldq, k, ldk, v, ldv, heads, nq, nk, d, scale, out_f32, out_hi, out_lo, out_ld, out_rows,
                                   out_row0, tpw);
            else
                hipLaunchKernelGGL(attn_small_kernel<0>, dim3((nq + tpw - 1) / tpw, batch), dim3(256), small_smem, st, q,
                                   ldq, k, ldk, v, ldv, heads, nq, nk, d, scale, out_f32, out_hi, out_lo, out_ld, out_rows,
                                   out_row0, tpw);
            return wd_check_launch();
        }
    }
    static size_t max_set = 64 * 1024;
    if (smem > max_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return WD_ELAUNCH;
        max_set = smem;
    }
    WdLaunchScope scope(WD_CLS_ATTN, st);
    hipLaunchKernelGGL(attn_kernel, dim3((nq + TQ - 1) / TQ, batch * heads), dim3(256), smem, st, q, ldq, k, ldk, v,
                       ldv, heads, nq, nk, d, scale, out_f32, out_hi, out_lo, out_ld, out_rows, out_row0);
    return wd_check_launch();
}
