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


// ---- MFMA attention (spatial self-attention of the PHOSC UNet, its 779-token cross-attention; unetPhosc.py:157-198) -------
// One workgroup = 128 queries of one (sample, head); each of its 4 waves owns 32 queries.  Per 32-key block:
//   S^T[key][query] = K . Q^T on v_mfma_f32_32x32x16_bf16: a lane then holds ONE query (column lane & 31) and 16 keys in
//   its accumulator registers, so the softmax statistics are per-lane scalars (one cross-half exchange);
//   O^T[dv][query] += V^T . P^T with the accumulator registers themselves as the B operand: the MFMA reduction index is
//   only a label, so V^T is stored in LDS with the keys permuted to the order the S^T accumulator delivers them.
// Operands are split-bf16 (hi + lo, three MFMAs per product) like the GEMM: results stay at fp32-class accuracy.
typedef __attribute__((ext_vector_type(8))) __bf16 fa_bf16x8;
typedef __attribute__((ext_vector_type(16))) float fa_f32x16;
constexpr int FA_KB = 64;    // keys per LDS block (two 32-key MFMA sub-blocks)
constexpr float FA_LAZY = 8.0f;  // log2 headroom of the lazily updated softmax maximum (attn_mfma_kernel)
constexpr int FA_QB = 128;   // queries per workgroup

__device__ __forceinline__ void fa_split8(const float* f, fa_bf16x8& hi, fa_bf16x8& lo) {
    uint2 h0, l0, h1, l1;
    wd_split4(make_float4(f[0], f[1], f[2], f[3]), h0, l0);
    wd_split4(make_float4(f[4], f[5], f[6], f[7]), h1, l1);
    const uint4 hv = make_uint4(h0.x, h0.y, h1.x, h1.y), lv = make_uint4(l0.x, l0.y, l1.x, l1.y);
    hi = *reinterpret_cast<const fa_bf16x8*>(&hv);
    lo = *reinterpret_cast<const fa_bf16x8*>(&lv);
}

// One block of FA_KB keys of one (sample, head): fp32 K / V rows -> split-bf16 LDS image (K planes by key, V^T planes with the keys
// of each 32-block in the order the S^T accumulator delivers them).
template <int D, int KP, int VP, int DVR>
__device__ __forceinline__ void fa_convert_block(wd_bf16* sK, wd_bf16* sV, const float* __restrict__ k, int ldk,
                                                 const float* __restrict__ v, int ldv, const int b, const int h, const int nk,
                                                 const int kb, const int tid) {
    constexpr int D4 = D / 4;
    for (int e = tid; e < FA_KB * D4; e += 256) {
        const int key = e / D4, c4 = e - key * D4;
        const int gk = kb + key;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (gk < nk) {
            kv = *reinterpret_cast<const float4*>(k + ((long)b * nk + gk) * ldk + h * D + c4 * 4);
            vv = *reinterpret_cast<const float4*>(v + ((long)b * nk + gk) * ldv + h * D + c4 * 4);
        }
        uint2 hi, lo;
        wd_split4(kv, hi, lo);
        *reinterpret_cast<uint2*>(sK + (long)key * KP + c4 * 4) = hi;
        *reinterpret_cast<uint2*>(sK + (long)(FA_KB + key) * KP + c4 * 4) = lo;
        // V^T with the keys of each 32-block in accumulator order: position = key with bits 2 and 3 swapped
        const int kl = key & 31;
        const int pos = (key & 32) | (kl & 0x13) | ((kl & 4) << 1) | ((kl & 8) >> 1);
        wd_split4(vv, hi, lo);
        wd_bf16* vh = sV + (long)(c4 * 4) * VP + pos;
        wd_bf16* vl = sV + (long)(DVR + c4 * 4) * VP + pos;
        vh[0] = (wd_bf16)(hi.x & 0xffff); vh[VP] = (wd_bf16)(hi.x >> 16);
        vh[2 * VP] = (wd_bf16)(hi.y & 0xffff); vh[3 * VP] = (wd_bf16)(hi.y >> 16);
        vl[0] = (wd_bf16)(lo.x & 0xffff); vl[VP] = (wd_bf16)(lo.x >> 16);
        vl[2 * VP] = (wd_bf16)(lo.y & 0xffff); vl[3 * VP] = (wd_bf16)(lo.y >> 16);
    }
}

// KSPLIT: 64 queries per workgroup instead of 128 - the wave pairs {0, 1} and {2, 3} own the SAME 64 queries and take the even /
// odd 32-key sub-blocks of every LDS block; their (max, sum, O) states are merged once at the end (the flash-decoding merge).  For
// maps of <= 64 positions (4x16 level) the 128-query form left half of every MFMA tile empty and 256 workgroups to stream 779 keys
// each: 74.6 us for a quarter of the work the 8x32 level does in 102.
// PACKED: k points at the LDS images of the key blocks as wd_attention_pack_kv stored them ([sample][head][block][image], image =
// exactly sK | sV below): a context that does not change over the 999 steps of a sampling call (the PHOSC cross-attention: 779
// keys) is converted once, and a block is then 16-byte copies with the next block's in flight during this block's products -
// instead of fp32 loads, the bf16 split and 2-byte transposing LDS writes in every workgroup of every step.
template <int KS, int DVT, bool KSPLIT, bool PACKED = false>
__global__ void __launch_bounds__(256) attn_mfma_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                        int ldk, const float* __restrict__ v, int ldv, int heads, int nq,
                                                        int nk, float scale, float* __restrict__ out_f32,
                                                        wd_bf16* __restrict__ out_hi, wd_bf16* __restrict__ out_lo, int out_ld,
                                                        int out_rows, int out_row0) {
    constexpr int D = KS * 16;
    constexpr int KP = D + 8;        // sK row pitch (bf16 elements): 16-byte fragment reads, rows spread over the banks
    constexpr int VP = FA_KB + 8;    // sVt row pitch
    constexpr int DVR = DVT * 32;    // V^T rows (dv padded to whole 32-row MFMA tiles; the padding rows stay zero)
    constexpr int OP = D + 4;        // fp32 output staging pitch
    constexpr int D4 = D / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wd_bf16* sK = reinterpret_cast<wd_bf16*>(smem);   // [2 planes][FA_KB][KP]
    wd_bf16* sV = sK + 2 * FA_KB * KP;                // [2 planes][DVR][VP]
    // grid (heads, batch, query tiles): the query tiles of one (sample, head) share K/V - consecutive workgroup ids are
    // different (sample, head) pairs, so those tiles land on the same XCD (ids 8 apart... see launch) and hit its L2
    constexpr int QB = KSPLIT ? FA_QB / 2 : FA_QB;
    const int b = blockIdx.y, h = blockIdx.x, q0 = blockIdx.z * QB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = KSPLIT ? (wave & 1) : wave;   // which 32 queries of the tile
    const int kgrp = KSPLIT ? (wave >> 1) : 0;   // which 32-key sub-block of every 64-key block
    const int l31 = lane & 31, lh = lane >> 5;

    // Q fragments (B operand of S^T = K . Q^T), scaled by softmax_scale * log2(e) so that the exponentials are exp2
    fa_bf16x8 qh[KS], ql[KS];
    {
        const int qi = q0 + qw * 32 + l31;
        const float sc = scale * 1.44269504088896340736f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float f[8];
            if (qi < nq) {
                const float* p = q + ((long)b * nq + qi) * ldq + h * D + ks * 16 + lh * 8;
                const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
                f[0] = a.x * sc; f[1] = a.y * sc; f[2] = a.z * sc; f[3] = a.w * sc;
                f[4] = c.x * sc; f[5] = c.y * sc; f[6] = c.z * sc; f[7] = c.w * sc;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = 0.f;
            }
            fa_split8(f, qh[ks], ql[ks]);
        }
    }
    constexpr int IMG = 2 * FA_KB * KP + 2 * DVR * VP;  // bf16 elements of one block's image
    constexpr int NP16 = IMG / 8, NPF = PACKED ? (NP16 + 255) / 256 : 1;
    static_assert(IMG % 8 == 0, "16-byte pieces");
    const uint4* gimg = nullptr;
    uint4 pre[NPF];
    if constexpr (PACKED) {
        const int nkb = (nk + FA_KB - 1) / FA_KB;
        gimg = reinterpret_cast<const uint4*>(reinterpret_cast<const wd_bf16*>(k) + ((long)(b * heads + h) * nkb) * IMG);
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + i * 256;
            pre[i] = idx < NP16 ? gimg[idx] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if constexpr (DVR > D && !PACKED) {  // zero the dv padding rows of both planes once (the packed image carries its zeros)
        constexpr int PADN = (DVR - D) * VP;
        for (int e = tid; e < 2 * PADN; e += 256) {
            const int pl = e / PADN, r = e - pl * PADN;
            sV[(pl * DVR + D) * VP + r] = 0;
        }
    }
    fa_f32x16 o[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -1e30f, lsum = 0.f;

    for (int kb = 0; kb < nk; kb += FA_KB) {
        __syncthreads();  // the previous block has been consumed
        if constexpr (PACKED) {
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int idx = tid + i * 256;
                if (idx < NP16) reinterpret_cast<uint4*>(smem)[idx] = pre[i];
            }
            if (kb + FA_KB < nk) {  // the next block's image: in flight during this block's products
                const uint4* nx = gimg + (long)(kb / FA_KB + 1) * NP16;
#pragma unroll
                for (int i = 0; i < NPF; ++i) {
                    const int idx = tid + i * 256;
                    if (idx < NP16) pre[i] = nx[idx];
                }
            }
        } else {
            fa_convert_block<D, KP, VP, DVR>(sK, sV, k, ldk, v, ldv, b, h, nk, kb, tid);
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < FA_KB / 32; ++sub) {
            const int k0 = kb + sub * 32;
            if (k0 >= nk) break;  // uniform over the workgroup
            if (KSPLIT && sub != kgrp) continue;  // (wave-uniform) the other wave pair's sub-block
            fa_f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const wd_bf16* kp = sK + (long)(sub * 32 + l31) * KP + ks * 16 + lh * 8;
                const fa_bf16x8 ah = *reinterpret_cast<const fa_bf16x8*>(kp);
                const fa_bf16x8 al = *reinterpret_cast<const fa_bf16x8*>(kp + FA_KB * KP);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[ks], s, 0, 0, 0);
            }
            if (k0 + 32 > nk) {  // (uniform) only the last sub-block has keys to mask
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= nk) s[r] = -1e30f;
                }
            }
            float bm = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) bm = fmaxf(bm, s[r]);
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            // Lazy rescale: the running maximum only moves when some query of the wave outgrew it by more than 2^FA_LAZY (the
            // exponentials then stay below 2^FA_LAZY, harmless in fp32 and in the split-bf16 product; the final division by lsum
            // uses the same reference, so the softmax is unchanged) - after the first blocks that is almost never, and the 48
            // multiplications of the output accumulators per sub-block go away with it.
            if (__any(bm > m + FA_LAZY)) {
                const float m_new = fmaxf(m, bm);
                const float alpha = __builtin_amdgcn_exp2f(m - m_new);
                m = m_new;
                lsum *= alpha;
#pragma unroll
                for (int t = 0; t < DVT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
            }
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(s[r] - m);   // (raw v_exp_f32: arguments <= FA_LAZY, large negatives flush to 0)
                lsum += p[r];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa_bf16x8 ph, pl;
                fa_split8(p + 8 * i, ph, pl);
#pragma unroll
                for (int t = 0; t < DVT; ++t) {
                    const wd_bf16* vp = sV + (long)(t * 32 + l31) * VP + sub * 32 + i * 16 + lh * 8;
                    const fa_bf16x8 ah = *reinterpret_cast<const fa_bf16x8*>(vp);
                    const fa_bf16x8 al = *reinterpret_cast<const fa_bf16x8*>(vp + DVR * VP);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ph, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pl, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph, o[t], 0, 0, 0);
                }
            }
        }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    __syncthreads();
    if constexpr (KSPLIT) {
        // merge the two key groups' states of every query (lane-for-lane: both pairs hold the same query in the same lane)
        float* sM = reinterpret_cast<float*>(smem);  // [2 query waves][DVT * 16 + 2][64 lanes]
        constexpr int NV = DVT * 16 + 2;
        float* mine = sM + ((long)qw * NV) * 64 + lane;
        if (kgrp == 1) {
#pragma unroll
            for (int t = 0; t < DVT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(t * 16 + r) * 64] = o[t][r];
            mine[(DVT * 16) * 64] = m;
            mine[(DVT * 16 + 1) * 64] = lsum;
        }
        __syncthreads();
        if (kgrp == 0) {
            const float m1 = mine[(DVT * 16) * 64], l1 = mine[(DVT * 16 + 1) * 64];
            const float mt = fmaxf(m, m1);
            const float a0 = exp2f(m - mt), a1 = exp2f(m1 - mt);
            lsum = lsum * a0 + l1 * a1;
#pragma unroll
            for (int t = 0; t < DVT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] = o[t][r] * a0 + mine[(t * 16 + r) * 64] * a1;
        }
        __syncthreads();  // the merge image is consumed before the output staging overlays it
    }
    const float inv = 1.0f / lsum;
    float* sO = reinterpret_cast<float*>(smem);  // [QB][OP]
    if (kgrp == 0) {
#pragma unroll
        for (int t = 0; t < DVT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dv = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (dv < D) sO[(qw * 32 + l31) * OP + dv] = o[t][r] * inv;
            }
    }
    __syncthreads();
    for (int e = tid; e < QB * D4; e += 256) {
        const int qlr = e / D4, c4 = e - qlr * D4;
        const int qi = q0 + qlr;
        if (qi >= nq) continue;
        const float4 val = *reinterpret_cast<const float4*>(sO + qlr * OP + c4 * 4);
        const long orow = (long)b * out_rows + out_row0 + qi;
        const long off = orow * out_ld + h * D + c4 * 4;
        if (out_f32) *reinterpret_cast<float4*>(out_f32 + off) = val;
        if (out_hi) {
            uint2 hi, lo;
            wd_split4(val, hi, lo);
            *reinterpret_cast<uint2*>(out_hi + off) = hi;
            if (out_lo) *reinterpret_cast<uint2*>(out_lo + off) = lo;
        }
    }
}

template <int KS, int DVT>
__global__ void __launch_bounds__(256) attn_pack_kernel(const float* __restrict__ k, int ldk, const float* __restrict__ v, int ldv,
                                                        int heads, int nk, wd_bf16* __restrict__ img) {
    constexpr int D = KS * 16, KP = D + 8, VP = FA_KB + 8, DVR = DVT * 32;
    constexpr int IMG = 2 * FA_KB * KP + 2 * DVR * VP, NP16 = IMG / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wd_bf16* sK = reinterpret_cast<wd_bf16*>(smem);
    wd_bf16* sV = sK + 2 * FA_KB * KP;
    const int kbi = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    for (int i = tid; i < NP16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);  // every padding element
    __syncthreads();
    fa_convert_block<D, KP, VP, DVR>(sK, sV, k, ldk, v, ldv, b, h, nk, kbi * FA_KB, tid);
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(img + ((long)(b * heads + h) * gridDim.x + kbi) * IMG);
    for (int i = tid; i < NP16; i += 256) dst[i] = reinterpret_cast<const uint4*>(smem)[i];
}

template <int KS, int DVT>
static constexpr long fa_image_elems() {
    return 2L * FA_KB * (KS * 16 + 8) + 2L * DVT * 32 * (FA_KB + 8);
}

template <int KS, int DVT, bool KSPLIT, bool PACKED = false>
static int launch_attn_mfma(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int batch, int heads,
                            int nq, int nk, float scale, float* out_f32, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld,
                            int out_rows, int out_row0, hipStream_t st) {
    constexpr int D = KS * 16;
    constexpr int QB = KSPLIT ? FA_QB / 2 : FA_QB;
    constexpr size_t tiles = (size_t)(2 * FA_KB * (D + 8) + 2 * DVT * 32 * (FA_KB + 8)) * sizeof(wd_bf16);
    constexpr size_t stage = (size_t)QB * (D + 4) * sizeof(float);
    constexpr size_t merge = KSPLIT ? (size_t)2 * (DVT * 16 + 2) * 64 * sizeof(float) : 0;
    constexpr size_t smem = (tiles > stage ? tiles : stage) > merge ? (tiles > stage ? tiles : stage) : merge;
    static_assert(smem <= 64 * 1024, "attention tile does not fit the default dynamic LDS limit");
    WdLaunchScope scope(WD_CLS_ATTN, st);
    hipLaunchKernelGGL((attn_mfma_kernel<KS, DVT, KSPLIT, PACKED>), dim3(heads, batch, (nq + QB - 1) / QB), dim3(256), smem, st, q, ldq, k,
                       ldk, v, ldv, heads, nq, nk, scale, out_f32, out_hi, out_lo, out_ld, out_rows, out_row0);
    return wd_check_launch();
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
    if (nk > NKS && d % 16 == 0 && d <= 96 && out_ld % 4 == 0 && !getenv("WDIFF_ATTN_GENERIC")) {
        // spatial self-attention / long-context cross-attention: MFMA kernel (split-bf16, fp32 softmax)
        // 64-query tiles with the keys split over the wave pairs: maps of <= 64 positions (WDIFF_ATTN_KSPLIT=1 always, 0 never)
        static const int ksplit_env = getenv("WDIFF_ATTN_KSPLIT") ? atoi(getenv("WDIFF_ATTN_KSPLIT")) : 2;
        const bool ksplit = ksplit_env == 1 || (ksplit_env == 2 && nq <= 64);
#define WD_FA(KS_, DVT_)                                                                                                          \
    return ksplit ? launch_attn_mfma<KS_, DVT_, true>(q, ldq, k, ldk, v, ldv, batch, heads, nq, nk, scale, out_f32, out_hi, out_lo, \
                                                      out_ld, out_rows, out_row0, st)                                               \
                  : launch_attn_mfma<KS_, DVT_, false>(q, ldq, k, ldk, v, ldv, batch, heads, nq, nk, scale, out_f32, out_hi,        \
                                                       out_lo, out_ld, out_rows, out_row0, st)
        switch (d / 16) {
            case 1: WD_FA(1, 1);
            case 2: WD_FA(2, 1);
            case 3: WD_FA(3, 2);
            case 4: WD_FA(4, 2);
            case 5: WD_FA(5, 3);
            case 6: WD_FA(6, 3);
        }
#undef WD_FA
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

// ---- key / value images for a context that stays fixed over a sampling call (see attn_mfma_kernel PACKED)
#define WD_FA_CASES(X) \
    switch (d / 16) {  \
        case 1: X(1, 1); \
        case 2: X(2, 1); \
        case 3: X(3, 2); \
        case 4: X(4, 2); \
        case 5: X(5, 3); \
        case 6: X(6, 3); \
    }

extern "C" int64_t wd_attention_packed_elems(int batch, int heads, int nk, int d) {
    if (batch <= 0 || heads <= 0 || nk <= NKS || d <= 0 || d % 16 || d > 96) return 0;
    const long nkb = (nk + FA_KB - 1) / FA_KB;
#define WD_X(KS_, DVT_) return (int64_t)batch * heads * nkb * fa_image_elems<KS_, DVT_>()
    WD_FA_CASES(WD_X)
#undef WD_X
    return 0;
}

extern "C" int wd_attention_pack_kv(const float* k, int ldk, const float* v, int ldv, int batch, int heads, int nk, int d,
                                    wd_bf16* img, void* stream) {
    if (!k || !v || !img || ldk % 4 || ldv % 4 || wd_attention_packed_elems(batch, heads, nk, d) == 0) return WD_EINVAL;
    if (reinterpret_cast<uintptr_t>(img) & 15) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nkb = (nk + FA_KB - 1) / FA_KB;
    WdLaunchScope scope(WD_CLS_ATTN, st);
#define WD_X(KS_, DVT_)                                                                                                    \
    {                                                                                                                      \
        hipLaunchKernelGGL((attn_pack_kernel<KS_, DVT_>), dim3(nkb, heads, batch), dim3(256),                               \
                           ((size_t)fa_image_elems<KS_, DVT_>() * sizeof(wd_bf16)), st, k, ldk, v, ldv, heads, nk, img);    \
        return wd_check_launch();                                                                                          \
    }
    WD_FA_CASES(WD_X)
#undef WD_X
    return WD_EINVAL;
}

extern "C" int wd_attention_packed(const float* q, int ldq, const wd_bf16* img, int batch, int heads, int nq, int nk, int d,
                                   float scale, float* out_f32, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, int out_rows,
                                   int out_row0, void* stream) {
    if (!q || !img || (!out_f32 && !out_hi) || nq <= 0 || ldq % 4 || out_ld % 4) return WD_EINVAL;
    if (wd_attention_packed_elems(batch, heads, nk, d) == 0 || (reinterpret_cast<uintptr_t>(img) & 15)) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float* kimg = reinterpret_cast<const float*>(img);
    static const int ksplit_env = getenv("WDIFF_ATTN_KSPLIT") ? atoi(getenv("WDIFF_ATTN_KSPLIT")) : 2;
    const bool ksplit = ksplit_env == 1 || (ksplit_env == 2 && nq <= 64);
#define WD_X(KS_, DVT_)                                                                                                            \
    return ksplit ? launch_attn_mfma<KS_, DVT_, true, true>(q, ldq, kimg, 0, nullptr, 0, batch, heads, nq, nk, scale, out_f32,       \
                                                            out_hi, out_lo, out_ld, out_rows, out_row0, st)                          \
                  : launch_attn_mfma<KS_, DVT_, false, true>(q, ldq, kimg, 0, nullptr, 0, batch, heads, nq, nk, scale, out_f32,      \
                                                             out_hi, out_lo, out_ld, out_rows, out_row0, st)
    WD_FA_CASES(WD_X)
#undef WD_X
    return WD_EINVAL;
}
#undef WD_FA_CASES
