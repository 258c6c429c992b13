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
