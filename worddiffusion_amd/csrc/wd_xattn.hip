// Cross-attention over a handful of context tokens, folded and fused (base UNet: both attentions of every transformer
// block attend to the <= 10 character tokens, unet.py:164-279,337-345; PHOSC UNet: attn2 without a PHOSC vector).
//
// K and V depend only on the context, so for every sample b and head h the projections around the attention fold into two
// small per-sample matrices, computed once per sampling call (wd_xattn_fold):
//     scores[t][h, j] = LN(x_t) . Mq[b][h, j][:],     Mq[b][h, j][n] = scale * sum_c K[b, j][h, c] * Wq[h*d + c][n]
//     out[t][n]       = sum_{h, j} softmax_j(scores)[h, j] * Mo[b][h, j][n] + bias[n] + x_t[n],
//                                                     Mo[b][h, j][n] = sum_c V[b, j][h, c] * Wo[n][h*d + c]
// which is 8x fewer multiply-adds than  LN -> to_q GEMM -> attention -> to_out GEMM  and one launch instead of four.
// Everything is fp32 (VALU): LayerNorm, 2 x (64 tokens x 320 x 40) products per workgroup out of LDS, softmax, residual.
#include "wd_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int XHJ = 64;  // (head, key) pairs padded to two 32-wide MFMA tiles

// The split-bf16 planes of the folded matrices are stored FRAGMENT-MAJOR: the 16 bytes a lane of the consuming MFMA needs sit at
// block * 1 KB + lane * 16 B, so a wave's operand load is one contiguous kilobyte (eight full lines) instead of 16-byte pieces of 16
// to 64 different lines (row pitch 640 B / 128 B) - stamped before the change: the score phase spent ~10k of its 14k cycles waiting
// for those pieces.
//   Mq plane [64 hj][c], consumer v_mfma_f32_16x16x32_bf16 B operand: lane = 16 * (k chunk of 8) + (hj row in its tile of 16),
//   block = (hj tile, 32-deep k step).
__device__ __forceinline__ long xa_mq_index(int hj, int n, int c) {
    return ((long)((hj >> 4) * (c >> 5) + (n >> 5)) << 9) + ((((n >> 3) & 3) * 16 + (hj & 15)) << 3) + (n & 7);
}
//   Mo^T plane [c][64 hj], consumer v_mfma_f32_16x16x32_bf16 B operand: block = (column tile of 16, k-step of 32 (head, key)
//   pairs), lane = 16 * (chunk of 8 pairs) + column.
__device__ __forceinline__ long xa_mo16_index(int n, int hj) {
    return ((long)((n >> 4) * 2 + (hj >> 5)) << 9) + ((((hj >> 3) & 3) * 16 + (n & 15)) << 3) + (hj & 7);
}

__global__ void __launch_bounds__(256) xattn_fold_kernel(const float* __restrict__ k, int ldk, const float* __restrict__ v,
                                                         int ldv, int heads, int L, int d, float scale,
                                                         const float* __restrict__ wq, const float* __restrict__ wo, int c,
                                                         float* __restrict__ mq, float* __restrict__ mo,
                                                         wd_bf16* __restrict__ mq_pl, wd_bf16* __restrict__ mot_pl) {
    // grid (L, heads, batch); one (b, h, j) row of both matrices per workgroup
    extern __shared__ float s_kv[];  // [2][d]
    const int j = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int inner = heads * d;
    const float* kr = k + ((long)b * L + j) * ldk + h * d;
    const float* vr = v + ((long)b * L + j) * ldv + h * d;
    for (int i = threadIdx.x; i < d; i += 256) {
        s_kv[i] = kr[i];
        s_kv[d + i] = vr[i];
    }
    __syncthreads();
    const long orow = ((long)b * heads + h) * L + j;
    for (int n = threadIdx.x; n < c; n += 256) {
        float aq = 0.f, ao = 0.f;
        const float* wqc = wq + (long)h * d * c + n;       // Wq[h*d + cc][n], coalesced over n
        const float* woc = wo + (long)n * inner + h * d;   // Wo[n][h*d + cc]
        for (int cc = 0; cc < d; ++cc) {
            aq += s_kv[cc] * wqc[(long)cc * c];
            ao += s_kv[d + cc] * woc[cc];
        }
        mq[orow * c + n] = aq * scale;
        mo[orow * c + n] = ao;
        if (mq_pl) {
            // operands of the MFMA kernel: Mq [b][plane][64 hj x c], Mo transposed [b][plane][c x 64 hj], both fragment-major
            const int hj = h * L + j;
            const long pq = (long)XHJ * c;
            uint32_t hi, lo;
            wd_split1(aq * scale, hi, lo);
            mq_pl[((long)b * 2 + 0) * pq + xa_mq_index(hj, n, c)] = (wd_bf16)hi;
            mq_pl[((long)b * 2 + 1) * pq + xa_mq_index(hj, n, c)] = (wd_bf16)lo;
            wd_split1(ao, hi, lo);
            const long io = xa_mo16_index(n, hj);
            mot_pl[((long)b * 2 + 0) * pq + io] = (wd_bf16)hi;
            mot_pl[((long)b * 2 + 1) * pq + io] = (wd_bf16)lo;
        }
    }
}

constexpr int XT = 64;  // tokens per workgroup: 32 token pairs x 8 column groups

template <int NI, int NH>  // NI = c / 32 (float4 columns per thread in the output phase), NH = ceil(heads * L / 8)
__global__ void __launch_bounds__(256) xattn_fused_kernel(const float* __restrict__ x, int ld, int hw,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, const float* __restrict__ mq, const float* __restrict__ mo,
                                                          int heads, int L, const float* __restrict__ bias,
                                                          float* __restrict__ out, int out_ld, const float* __restrict__ gamma2,
                                                          const float* __restrict__ beta2, float eps2,
                                                          wd_bf16* __restrict__ n_hi, wd_bf16* __restrict__ n_lo, int n_ld, int dbg) {
    constexpr int C = NI * 32, C4 = C / 4, CP = C + 4, HJP = NH * 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_x = reinterpret_cast<float*>(smem);  // [XT][CP] normalised tokens
    float* s_m = s_x + XT * CP;                   // [HJP][CP] Mq, later Mo (rows >= heads*L are zero)
    float* s_p = s_m + HJP * CP;                  // [XT][HJP + 1] scores / probabilities
    const int b = blockIdx.y, t0 = blockIdx.x * XT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HJ = heads * L;
    const long row0 = (long)b * hw + t0;
    const int ntok = min(XT, hw - t0);

    // ---- LayerNorm of the 64 tokens (one wave per token, two-pass in registers) -> s_x
    constexpr int F = (C4 + 63) / 64;
    for (int t = wave; t < ((dbg & 1) ? 4 : XT); t += 4) {
        float4 v[F];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < F; ++i) {
            const int f = lane + 64 * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < C4 && t < ntok) v[i] = *reinterpret_cast<const float4*>(x + (row0 + t) * ld + f * 4);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wd_wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < F; ++i) {
            const int f = lane + 64 * i;
            if (f < C4) {
                const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = 1.0f / sqrtf(wd_wave_sum(q) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < F; ++i) {
            const int f = lane + 64 * i;
            if (f < C4) {
                const float4 ga = *reinterpret_cast<const float4*>(gamma + f * 4);
                const float4 be = *reinterpret_cast<const float4*>(beta + f * 4);
                float4 o;
                o.x = (v[i].x - mean) * rstd * ga.x + be.x; o.y = (v[i].y - mean) * rstd * ga.y + be.y;
                o.z = (v[i].z - mean) * rstd * ga.z + be.z; o.w = (v[i].w - mean) * rstd * ga.w + be.w;
                *reinterpret_cast<float4*>(s_x + t * CP + f * 4) = o;
            }
        }
    }
    const float* mqb = mq + (long)b * HJ * C;
    const float* mob = mo + (long)b * HJ * C;
    for (int e = tid; e < HJP * C4; e += 256) {
        const int r = e / C4, f = e - r * C4;
        float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < HJ) m4 = *reinterpret_cast<const float4*>(mqb + (long)r * C + f * 4);
        *reinterpret_cast<float4*>(s_m + r * CP + f * 4) = m4;
    }
    __syncthreads();

    // ---- scores: thread = (token pair, column group g); columns g + 8 i
    const int tp = tid >> 3, g = tid & 7;
    {
        float acc[2][NH];
#pragma unroll
        for (int i = 0; i < NH; ++i) acc[0][i] = acc[1][i] = 0.f;
        const float* xa = s_x + (2 * tp) * CP;
        const float* xb = xa + CP;
        for (int f = 0; f < ((dbg & 2) ? 1 : C4); ++f) {
            const float4 a = *reinterpret_cast<const float4*>(xa + f * 4), c4 = *reinterpret_cast<const float4*>(xb + f * 4);
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const float4 m4 = *reinterpret_cast<const float4*>(s_m + (g + 8 * i) * CP + f * 4);
                acc[0][i] += (a.x * m4.x + a.y * m4.y) + (a.z * m4.z + a.w * m4.w);
                acc[1][i] += (c4.x * m4.x + c4.y * m4.y) + (c4.z * m4.z + c4.w * m4.w);
            }
        }
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            s_p[(2 * tp) * (HJP + 1) + g + 8 * i] = acc[0][i];
            s_p[(2 * tp + 1) * (HJP + 1) + g + 8 * i] = acc[1][i];
        }
    }
    __syncthreads();
    // ---- softmax per (token, head) over the L keys; meanwhile Mo replaces Mq in LDS
    for (int idx = tid; idx < XT * heads; idx += 256) {
        const int t = idx / heads, h = idx - t * heads;
        float* pr = s_p + t * (HJP + 1) + h * L;
        float mx = -3.4e38f;
        for (int j = 0; j < L; ++j) mx = fmaxf(mx, pr[j]);
        float sum = 0.f;
        for (int j = 0; j < L; ++j) {
            const float e = expf(pr[j] - mx);
            pr[j] = e;
            sum += e;
        }
        const float inv = 1.f / sum;
        for (int j = 0; j < L; ++j) pr[j] *= inv;
    }
    for (int e = tid; e < HJ * C4; e += 256) {
        const int r = e / C4, f = e - r * C4;
        *reinterpret_cast<float4*>(s_m + r * CP + f * 4) = *reinterpret_cast<const float4*>(mob + (long)r * C + f * 4);
    }
    __syncthreads();

    // ---- output: thread = (token pair, float4 column group g); float4 columns g + 8 i
    float4 o[2][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) o[0][i] = o[1][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float* pa = s_p + (2 * tp) * (HJP + 1);
        const float* pb = pa + (HJP + 1);
        for (int hj = 0; hj < ((dbg & 4) ? 1 : HJ); ++hj) {
            const float wa = pa[hj], wb = pb[hj];
            const float* mr = s_m + hj * CP + g * 4;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float4 m4 = *reinterpret_cast<const float4*>(mr + i * 32);
                o[0][i].x += wa * m4.x; o[0][i].y += wa * m4.y; o[0][i].z += wa * m4.z; o[0][i].w += wa * m4.w;
                o[1][i].x += wb * m4.x; o[1][i].y += wb * m4.y; o[1][i].z += wb * m4.z; o[1][i].w += wb * m4.w;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = 2 * tp + u;
        const bool ok = t < ntok;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = (g + 8 * i) * 4;
            const float4 bi = *reinterpret_cast<const float4*>(bias + n);
            float4 xr = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) xr = *reinterpret_cast<const float4*>(x + (row0 + t) * ld + n);
            o[u][i].x += bi.x + xr.x; o[u][i].y += bi.y + xr.y; o[u][i].z += bi.z + xr.z; o[u][i].w += bi.w + xr.w;
            if (ok) *reinterpret_cast<float4*>(out + (row0 + t) * out_ld + n) = o[u][i];
            s += (o[u][i].x + o[u][i].y) + (o[u][i].z + o[u][i].w);
        }
        if (n_hi) {
            // the LayerNorm that follows (norm3 before the feed-forward), written as GEMM operand planes: a token's row is
            // spread over the 8 lanes of its column groups
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            const float mean = s / (float)C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float a0 = o[u][i].x - mean, a1 = o[u][i].y - mean, a2 = o[u][i].z - mean, a3 = o[u][i].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
            const float rstd = 1.0f / sqrtf(q / (float)C + eps2);
            if (ok) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int n = (g + 8 * i) * 4;
                    const float4 ga = *reinterpret_cast<const float4*>(gamma2 + n);
                    const float4 be = *reinterpret_cast<const float4*>(beta2 + n);
                    float4 y;
                    y.x = (o[u][i].x - mean) * rstd * ga.x + be.x; y.y = (o[u][i].y - mean) * rstd * ga.y + be.y;
                    y.z = (o[u][i].z - mean) * rstd * ga.z + be.z; y.w = (o[u][i].w - mean) * rstd * ga.w + be.w;
                    uint2 hi, lo;
                    wd_split4(y, hi, lo);
                    *reinterpret_cast<uint2*>(n_hi + (row0 + t) * n_ld + n) = hi;
                    if (n_lo) *reinterpret_cast<uint2*>(n_lo + (row0 + t) * n_ld + n) = lo;
                }
            }
        }
    }
}

template <int NI, int NH>
int launch_fused(const float* x, int ld, int batch, int hw, const float* gamma, const float* beta, float eps, const float* mq,
                 const float* mo, int heads, int L, const float* bias, float* out, int out_ld, const float* gamma2,
                 const float* beta2, float eps2, wd_bf16* n_hi, wd_bf16* n_lo, int n_ld, hipStream_t st) {
    constexpr int C = NI * 32, CP = C + 4, HJP = NH * 8;
    constexpr size_t smem = (size_t)(XT * CP + HJP * CP + XT * (HJP + 1)) * sizeof(float);
    static_assert(smem <= 160 * 1024, "tile does not fit LDS");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<NI, NH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return WD_ELAUNCH;
        attr = true;
    }
    WdLaunchScope scope(WD_CLS_ATTN, st);
    hipLaunchKernelGGL((xattn_fused_kernel<NI, NH>), dim3((hw + XT - 1) / XT, batch), dim3(256), smem, st, x, ld, hw, gamma, beta,
                       eps, mq, mo, heads, L, bias, out, out_ld, gamma2, beta2, eps2, n_hi, n_lo, n_ld,
                       getenv("WDIFF_XATTN_DBG") ? atoi(getenv("WDIFF_XATTN_DBG")) : 0);
    return wd_check_launch();
}


// ---- MFMA form of the fused kernel (split-bf16 operands: the normalised tokens and the probabilities from LDS, the folded
// matrices straight from L2), everything else as above.
typedef __attribute__((ext_vector_type(8))) __bf16 xa_bf16x8;

struct XaLayer {  // one folded cross-attention: its LayerNorm, the folded matrices of this batch, the to_out bias
    const float* gamma;
    const float* beta;
    const wd_bf16* mq_pl;
    const wd_bf16* mot_pl;
    const float* bias;
};

// ---- 16 tokens per workgroup of four waves (B = 64 at 8 x 32: 1024 workgroups, three resident per CU).  The kernel this replaces
// gave 32 tokens to two waves and ran one wave per SIMD, so every memory and LDS latency of its chain (rows -> LayerNorm -> scores
// -> softmax -> output -> epilogue, twice) was exposed: 128 workgroups (the 4 x 16 level) took 23-30 us, 512 took 32-40; this one
// takes 11 and 28.  Everything on v_mfma_f32_16x16x32_bf16:
//   scores: wave w owns the (head, key) tile of 16 number w (heads * L <= 40: three tiles, the fourth wave has none); its three
//           split products run in three independent accumulators;
//   output: wave w owns C / 64 column tiles of 16; two k-steps over the (head, key) pairs 0..31 and 32..63 (zero beyond heads * L:
//           lanes whose 8-pair chunk is all padding do not load).
// (A 16x16x16 step for the pairs 32..47 would halve the second step, but v_mfma_f32_16x16x16_bf16 is not safe here: the compiler
// pads s_nop 7 between it and a DS read of its result, the hardware needs more - rows 4 q + {0, 1} of the tile came back stale in
// the two-attention instantiation until 32 wait states were added by hand.)
typedef __attribute__((ext_vector_type(4))) float xa_f32x4;

typedef __attribute__((ext_vector_type(4))) unsigned xa_u32x4;
__device__ __forceinline__ xa_bf16x8 xa_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(xa_bf16x8, (xa_u32x4)__builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int NI, int NP, int LL>  // c = NI * 32; NP attentions chained; LL = keys per head when known at compile time (0: run time)
__global__ void __launch_bounds__(256, 3) xattn16_kernel(const float* __restrict__ x, int ld, int hw, const XaLayer la,
                                                      const XaLayer lb, float eps, int heads, int L, float* __restrict__ out,
                                                      int out_ld, const float* __restrict__ gamma2,
                                                      const float* __restrict__ beta2, float eps2, wd_bf16* __restrict__ n_hi,
                                                      wd_bf16* __restrict__ n_lo, int n_ld) {
    constexpr int XT = 16, NTH = 256;
    constexpr int C = NI * 32, XP = C + 8, OP = C + 4, SP = 65, PP = 72, NT = NI / 2, KS32 = C / 32;
    static_assert(NI % 2 == 0, "column tiles per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wd_bf16* sX = reinterpret_cast<wd_bf16*>(smem);                              // [2][XT][XP] normalised tokens (planes)
    float* sO = reinterpret_cast<float*>(smem);                                   // [XT][OP] output image (overlays sX)
    constexpr size_t XBYTES = (size_t)2 * XT * XP * 2 > (size_t)XT * OP * 4 ? (size_t)2 * XT * XP * 2 : (size_t)XT * OP * 4;
    float* sS = reinterpret_cast<float*>(smem + XBYTES);                          // [XT][SP] scores
    wd_bf16* sP = reinterpret_cast<wd_bf16*>(smem + XBYTES + (size_t)XT * SP * 4);  // [2][XT][PP] probabilities (planes)
    // parameter vectors used after the first barrier, staged once (read from global where they are used, each is an exposed L2
    // round trip of the chain): [0] la.bias [1] lb.gamma [2] lb.beta [3] lb.bias [4] gamma2 [5] beta2
    float* sV = reinterpret_cast<float*>(smem + XBYTES + (size_t)XT * SP * 4 + (size_t)2 * XT * PP * 2);  // [6][C]
    // grid (batch, token tiles): consecutive workgroup ids = consecutive samples, so the token tiles of one sample (which share its
    // folded matrices) are dealt to the same XCD and find them in its L2
    const int b = blockIdx.x, t0 = blockIdx.y * XT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the per-wave branches below are s_cbranch)
    const int HJ = heads * L;
    const long row0 = (long)b * hw + t0;
    const int ntok = min(XT, hw - t0);
    constexpr int FI = NI / 2;  // float4 per lane per row; a row is handled by a quarter wave, four rows per wave
    const int qd = lane >> 4, l15 = lane & 15;
    const int trow = wave * 4 + qd;
    const bool ok = trow < ntok;
    float4 xr[FI];
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) xr[i] = *reinterpret_cast<const float4*>(x + (row0 + trow) * ld + (l15 + 16 * i) * 4);
    }
    const bool row_live = wave * 16 + l15 < HJ;   // this lane's row of the wave's (head, key) tile is not padding
    const long pq = (long)XHJ * C;
    // The whole score operand of this wave's (head, key) tile: 80 registers, requested after the LayerNorm (whose temporaries would
    // push them to scratch; requested a phase earlier the launch took 31 us instead of 27.8 - the loads then queue behind the row
    // burst - and the step was no faster).  Raw buffer loads: a lane whose row / chunk is padding gets an out-of-range offset -
    // zeros, no memory traffic, no branch; straight-line for all four waves (conditional definitions of these registers across a
    // barrier sent them through scratch).
    xa_bf16x8 bh[KS32], bl[KS32];
    auto request_scores = [&](const XaLayer& ly) {
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<wd_bf16*>(ly.mq_pl + (long)b * 2 * pq), 0,
                                                                            (int)(2 * pq * 2), 0x00020000);
        const unsigned vo = row_live ? lane * 16u : 0x80000000u;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
            bh[ks] = xa_bload(rq, vo, (unsigned)((wave * KS32 + ks) << 10));
            bl[ks] = xa_bload(rq, vo, (unsigned)(((wave * KS32 + ks) << 10) + pq * 2));
        }
    };
    for (int e = tid; e < 2 * XT * PP / 2; e += NTH) reinterpret_cast<uint32_t*>(sP)[e] = 0u;  // padding columns stay zero
    for (int e = tid; e < 6 * (C / 4); e += NTH) {
        const int v = e / (C / 4), i = e - v * (C / 4);
        const float* src = v == 0 ? la.bias : v == 1 ? lb.gamma : v == 2 ? lb.beta : v == 3 ? lb.bias : v == 4 ? gamma2 : beta2;
        if (src) reinterpret_cast<float4*>(sV)[e] = reinterpret_cast<const float4*>(src)[i];
    }
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
    const XaLayer& ly = ps == 0 ? la : lb;
    const float* gamma = ps == 0 ? la.gamma : sV + C;      // (first LayerNorm: from global, requested together with the rows)
    const float* beta = ps == 0 ? la.beta : sV + 2 * C;
    const float* bias = ps == 0 ? sV : sV + 3 * C;
    const bool last = ps == NP - 1;
    // ---- LayerNorm -> split planes in LDS
    {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < FI; ++i) s += (xr[i].x + xr[i].y) + (xr[i].z + xr[i].w);
        const float mean = wd_row16_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const float a0 = xr[i].x - mean, a1 = xr[i].y - mean, a2 = xr[i].z - mean, a3 = xr[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        const float rstd = 1.0f / sqrtf(wd_row16_sum(q) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const float4 ga = *reinterpret_cast<const float4*>(gamma + (l15 + 16 * i) * 4);
            const float4 be = *reinterpret_cast<const float4*>(beta + (l15 + 16 * i) * 4);
            float4 o;
            o.x = (xr[i].x - mean) * rstd * ga.x + be.x; o.y = (xr[i].y - mean) * rstd * ga.y + be.y;
            o.z = (xr[i].z - mean) * rstd * ga.z + be.z; o.w = (xr[i].w - mean) * rstd * ga.w + be.w;
            uint2 hi, lo;
            wd_split4(o, hi, lo);
            *reinterpret_cast<uint2*>(sX + (long)trow * XP + (l15 + 16 * i) * 4) = hi;
            *reinterpret_cast<uint2*>(sX + (long)(XT + trow) * XP + (l15 + 16 * i) * 4) = lo;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    request_scores(ly);
    __syncthreads();

    // ---- scores S[token][hj] = Xn . Mq^T: operand lane (l15 = row of the tile, qd = 8-element k chunk)
    if (wave * 16 < HJ) {  // (the wave without a tile leaves the MFMA pipe to the others)
        xa_f32x4 a_lh = {0.f, 0.f, 0.f, 0.f}, a_hl = a_lh, a_hh = a_lh;
        const wd_bf16* ax = sX + (long)l15 * XP + qd * 8;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
            const xa_bf16x8 ah = *reinterpret_cast<const xa_bf16x8*>(ax + ks * 32);
            const xa_bf16x8 al = *reinterpret_cast<const xa_bf16x8*>(ax + (long)XT * XP + ks * 32);
            a_lh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[ks], a_lh, 0, 0, 0);
            a_hl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[ks], a_hl, 0, 0, 0);
            a_hh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[ks], a_hh, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sS[(4 * qd + r) * SP + wave * 16 + l15] = (a_lh[r] + a_hl[r]) + a_hh[r];
    }
    // the operand of the output product does not depend on the softmax: request it now, it lands during the softmax
    __builtin_amdgcn_sched_barrier(0);  // (not before the score products have consumed their operand: 80 + 80 registers)
    xa_bf16x8 mh[2][NT], ml[2][NT];
    {
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<wd_bf16*>(ly.mot_pl + (long)b * 2 * pq), 0,
                                                                            (int)(2 * pq * 2), 0x00020000);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned vo = ks * 32 + qd * 8 < HJ ? lane * 16u : 0x80000000u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {  // block (column tile, ks) = ((wave NT + t) 2 + ks) KB
                mh[ks][t] = xa_bload(rm, vo, (unsigned)((((wave * NT + t) * 2 + ks) << 10)));
                ml[ks][t] = xa_bload(rm, vo, (unsigned)((((wave * NT + t) * 2 + ks) << 10) + pq * 2));
            }
        }
    }
    __syncthreads();
    // ---- softmax per (token, head) -> probability planes
    for (int idx = tid; idx < XT * heads; idx += NTH) {
        const int t = idx / heads, h = idx - t * heads;
        const float* pr = sS + t * SP + h * L;
        wd_bf16* ph = sP + (long)t * PP + h * L;
        if (LL > 0) {
            constexpr int LN_ = LL > 0 ? LL : 1;
            float v[LN_];
#pragma unroll
            for (int j = 0; j < LN_; ++j) v[j] = pr[j];
            float mx = v[0];
#pragma unroll
            for (int j = 1; j < LN_; ++j) mx = fmaxf(mx, v[j]);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < LN_; ++j) {
                v[j] = __expf(v[j] - mx);
                sum += v[j];
            }
            const float inv = __fdividef(1.f, sum);
#pragma unroll
            for (int j = 0; j < LN_; ++j) {
                uint32_t hi, lo;
                wd_split1(v[j] * inv, hi, lo);
                ph[j] = (wd_bf16)hi;
                ph[(long)XT * PP + j] = (wd_bf16)lo;
            }
        } else {
            float mx = -3.4e38f;
            for (int j = 0; j < L; ++j) mx = fmaxf(mx, pr[j]);
            float sum = 0.f;
            for (int j = 0; j < L; ++j) sum += __expf(pr[j] - mx);
            const float inv = __fdividef(1.f, sum);
            for (int j = 0; j < L; ++j) {
                uint32_t hi, lo;
                wd_split1(__expf(pr[j] - mx) * inv, hi, lo);
                ph[j] = (wd_bf16)hi;
                ph[(long)XT * PP + j] = (wd_bf16)lo;
            }
        }
    }
    __syncthreads();
    // ---- output image O[token][n] = P . Mo: NT column tiles of 16 per wave, independent accumulators
    {
        xa_f32x4 o[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) o[t] = xa_f32x4{0.f, 0.f, 0.f, 0.f};
        const wd_bf16* ap = sP + (long)l15 * PP + qd * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const xa_bf16x8 ah = *reinterpret_cast<const xa_bf16x8*>(ap + ks * 32);
            const xa_bf16x8 al = *reinterpret_cast<const xa_bf16x8*>(ap + (long)XT * PP + ks * 32);
#pragma unroll
            for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, mh[ks][t], o[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ml[ks][t], o[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, mh[ks][t], o[t], 0, 0, 0);
        }
        // (sX is dead: every wave passed the barrier after the score phase)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sO[(4 * qd + r) * OP + (wave * NT + t) * 16 + l15] = o[t][r];
    }
    __syncthreads();
    // ---- epilogue: + bias + residual (the row is still in registers), store, optional following LayerNorm -> planes
    {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int n = (l15 + 16 * i) * 4;
            const float4 bi = *reinterpret_cast<const float4*>(bias + n);
            const float4 at = *reinterpret_cast<const float4*>(sO + trow * OP + n);
            xr[i] = make_float4(at.x + bi.x + xr[i].x, at.y + bi.y + xr[i].y, at.z + bi.z + xr[i].z, at.w + bi.w + xr[i].w);
            if (ok && last) *reinterpret_cast<float4*>(out + (row0 + trow) * out_ld + n) = xr[i];
            s += (xr[i].x + xr[i].y) + (xr[i].z + xr[i].w);
        }
        if (n_hi && last) {
            const float mean = wd_row16_sum(s) / (float)C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const float a0 = xr[i].x - mean, a1 = xr[i].y - mean, a2 = xr[i].z - mean, a3 = xr[i].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
            const float rstd = 1.0f / sqrtf(wd_row16_sum(q) / (float)C + eps2);
            if (ok) {
#pragma unroll
                for (int i = 0; i < FI; ++i) {
                    const int n = (l15 + 16 * i) * 4;
                    const float4 g2 = *reinterpret_cast<const float4*>(sV + 4 * C + n);
                    const float4 b2 = *reinterpret_cast<const float4*>(sV + 5 * C + n);
                    float4 y;
                    y.x = (xr[i].x - mean) * rstd * g2.x + b2.x; y.y = (xr[i].y - mean) * rstd * g2.y + b2.y;
                    y.z = (xr[i].z - mean) * rstd * g2.z + b2.z; y.w = (xr[i].w - mean) * rstd * g2.w + b2.w;
                    uint2 hi, lo;
                    wd_split4(y, hi, lo);
                    *reinterpret_cast<uint2*>(n_hi + (row0 + trow) * n_ld + n) = hi;
                    if (n_lo) *reinterpret_cast<uint2*>(n_lo + (row0 + trow) * n_ld + n) = lo;
                }
            }
        }
    }
    if (!last) __syncthreads();  // the output image is consumed before the next attention overwrites the token planes
    }
}

template <int NI, int NP>
int launch16(const float* x, int ld, int batch, int hw, const XaLayer& la, const XaLayer& lb, float eps, int heads, int L,
             float* out, int out_ld, const float* gamma2, const float* beta2, float eps2, wd_bf16* n_hi, wd_bf16* n_lo, int n_ld,
             hipStream_t st) {
    constexpr int C = NI * 32, XTM = 16;
    constexpr size_t xb = (size_t)2 * XTM * (C + 8) * 2 > (size_t)XTM * (C + 4) * 4 ? (size_t)2 * XTM * (C + 8) * 2 : (size_t)XTM * (C + 4) * 4;
    constexpr size_t smem = xb + (size_t)XTM * 65 * 4 + (size_t)2 * XTM * 72 * 2 + (size_t)6 * C * 4;
    static_assert(smem <= 48 * 1024, "static LDS limit");
    WdLaunchScope scope(WD_CLS_ATTN, st);
    const dim3 grid(batch, (hw + XTM - 1) / XTM);
    // (three workgroups per CU: 145 / 152 registers.  Four - the whole grid of the 8 x 32 level resident at once - spill 10 / 30
    // registers and ran 33.5 us instead of 27.8.)
    if (L == 10)
        hipLaunchKernelGGL((xattn16_kernel<NI, NP, 10>), grid, dim3(256), smem, st, x, ld, hw, la, lb, eps, heads, L, out, out_ld,
                           gamma2, beta2, eps2, n_hi, n_lo, n_ld);
    else
        hipLaunchKernelGGL((xattn16_kernel<NI, NP, 0>), grid, dim3(256), smem, st, x, ld, hw, la, lb, eps, heads, L, out, out_ld,
                           gamma2, beta2, eps2, n_hi, n_lo, n_ld);
    return wd_check_launch();
}

}  // namespace

extern "C" int wd_xattn_supported(int c, int heads, int L) {
    return (c == 64 || c == 320) && heads > 0 && L > 0 && heads * L <= 40 && c % heads == 0;
}

extern "C" int wd_xattn_fold(const float* k, int ldk, const float* v, int ldv, int batch, int heads, int L, int d, float scale,
                             const float* wq, const float* wo, int c, float* mq, float* mo, wd_bf16* mq_pl, wd_bf16* mot_pl,
                             void* stream) {
    if (!k || !v || !wq || !wo || !mq || !mo || batch <= 0 || heads <= 0 || L <= 0 || d <= 0 || c <= 0) return WD_EINVAL;
    if ((mq_pl != nullptr) != (mot_pl != nullptr) || (mq_pl && heads * L > XHJ)) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(xattn_fold_kernel, dim3(L, heads, batch), dim3(256), 2 * d * sizeof(float), st, k, ldk, v, ldv, heads, L, d,
                       scale, wq, wo, c, mq, mo, mq_pl, mot_pl);
    return wd_check_launch();
}

extern "C" int wd_xattn_fused(const float* x, int ld, int batch, int hw, int c, const float* gamma, const float* beta, float eps,
                              const float* mq, const float* mo, int heads, int L, const float* bias, float* out, int out_ld,
                              const float* gamma2, const float* beta2, float eps2, wd_bf16* n_hi, wd_bf16* n_lo, int n_ld,
                              const wd_bf16* mq_pl, const wd_bf16* mot_pl, void* stream) {
    if (!x || !gamma || !beta || !mq || !mo || !bias || !out || batch <= 0 || hw <= 0) return WD_EINVAL;
    if (!wd_xattn_supported(c, heads, L) || ld % 4 || out_ld % 4 || (n_hi && (n_ld % 4 || !gamma2 || !beta2))) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mq_pl && mot_pl && !getenv("WDIFF_XATTN_VALU")) {
        const XaLayer la = {gamma, beta, mq_pl, mot_pl, bias};
        if (c == 320)
            return launch16<10, 1>(x, ld, batch, hw, la, la, eps, heads, L, out, out_ld, gamma2, beta2, eps2, n_hi, n_lo, n_ld, st);
        return launch16<2, 1>(x, ld, batch, hw, la, la, eps, heads, L, out, out_ld, gamma2, beta2, eps2, n_hi, n_lo, n_ld, st);
    }
    const int nh = (heads * L + 7) / 8;
#define WD_XA(NI_, NH_)                                                                                                     \
    return launch_fused<NI_, NH_>(x, ld, batch, hw, gamma, beta, eps, mq, mo, heads, L, bias, out, out_ld, gamma2, beta2, eps2, \
                                  n_hi, n_lo, n_ld, st)
    if (c == 320) {
        switch (nh) {
            case 1: WD_XA(10, 1);
            case 2: WD_XA(10, 2);
            case 3: WD_XA(10, 3);
            case 4: WD_XA(10, 4);
            case 5: WD_XA(10, 5);
        }
    } else {
        switch (nh) {
            case 1: WD_XA(2, 1);
            case 2: WD_XA(2, 2);
            case 3: WD_XA(2, 3);
            case 4: WD_XA(2, 4);
            case 5: WD_XA(2, 5);
        }
    }
#undef WD_XA
    return WD_EINVAL;
}

extern "C" int wd_xattn_pair(const float* x, int ld, int batch, int hw, int c, float eps, int heads, int L, const float* gamma_a,
                             const float* beta_a, const wd_bf16* mq_pl_a, const wd_bf16* mot_pl_a, const float* bias_a,
                             const float* gamma_b, const float* beta_b, const wd_bf16* mq_pl_b, const wd_bf16* mot_pl_b,
                             const float* bias_b, float* out, int out_ld, const float* gamma2, const float* beta2, float eps2,
                             wd_bf16* n_hi, wd_bf16* n_lo, int n_ld, void* stream) {
    if (!x || !out || !gamma_a || !beta_a || !mq_pl_a || !mot_pl_a || !bias_a || !gamma_b || !beta_b || !mq_pl_b || !mot_pl_b ||
        !bias_b || batch <= 0 || hw <= 0)
        return WD_EINVAL;
    if (!wd_xattn_supported(c, heads, L) || ld % 4 || out_ld % 4 || (n_hi && (n_ld % 4 || !gamma2 || !beta2))) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const XaLayer la = {gamma_a, beta_a, mq_pl_a, mot_pl_a, bias_a}, lb = {gamma_b, beta_b, mq_pl_b, mot_pl_b, bias_b};
    if (c == 320)
        return launch16<10, 2>(x, ld, batch, hw, la, lb, eps, heads, L, out, out_ld, gamma2, beta2, eps2, n_hi, n_lo, n_ld, st);
    return launch16<2, 2>(x, ld, batch, hw, la, lb, eps, heads, L, out, out_ld, gamma2, beta2, eps2, n_hi, n_lo, n_ld, st);
}
