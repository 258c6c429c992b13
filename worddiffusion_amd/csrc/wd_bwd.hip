// Backward building blocks of the training step (SURVEY.md row a-T; reference: loss.backward() of train.py:290-292 through
// unet.py's ResBlock / SpatialTransformer / CrossAttention / GEGLU modules).  All contractions of the backward pass run
// on wd_gemm (data gradients: mirrored gather tables + transposed weights; weight gradients: a GEMM over the token
// dimension whose operands are the TRANSPOSED planes produced here).  This file holds the non-GEMM pieces:
//   wd_transpose_planes   planes / fp32 [rows][C] (+ 3x3 tap gather) -> planes [taps*C][M]   (operands of the dW GEMMs)
//   wd_colsum             deterministic (segmented) column sums: bias, FiLM and affine-parameter gradients
//   wd_gn_bwd_stats/apply GroupNorm(+SiLU) backward
//   wd_layernorm_bwd      LayerNorm backward
//   wd_attention_bwd_small  softmax attention backward for <= 16 keys (the 10 text tokens)
//   wd_geglu_fwd/bwd, wd_silu_bwd, wd_pool2x2_sum (nearest-x2 upsample backward), wd_embedding_bwd
// Reductions are two-stage with fixed order (no float atomics): gradients are bitwise reproducible.
#include "wd_common.h"

namespace {

__device__ __forceinline__ float silu_grad(float y) {  // d silu(y) / dy  (hardware exp / rcp, as wd_silu)
    const float s = __fdividef(1.0f, 1.0f + __expf(-y));
    return s * (1.0f + y * (1.0f - s));
}
__device__ __forceinline__ float gelu_grad(float g) {  // d gelu_erf(g) / dg
    const float cdf = 0.5f * (1.0f + wd_erf(g * 0.70710678118654752440f));
    return cdf + g * 0.39894228040143267794f * __expf(-0.5f * g * g);
}

// ---- transpose: out[(t*C + c)][m] = in[src(m, t)][c]   (64 x 64 tiles through LDS; src = gather table or identity)
// plane input: 8-byte loads (4 channels) and 8-byte stores (4 tokens) per thread and plane
template <bool F32IN>
__global__ void __launch_bounds__(256) transpose_planes_kernel(const void* __restrict__ in_hi, const void* __restrict__ in_lo,
                                                               int ld, int c, const int32_t* __restrict__ gather, int ntaps,
                                                               int hw_out, int hw_src, int m, int mpad, int tap_minor,
                                                               wd_bf16* __restrict__ out_hi, wd_bf16* __restrict__ out_lo) {
    __shared__ wd_bf16 th[64][68], tl[64][68];  // [token][channel]
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
    const int tid = threadIdx.x;
    {
        const int q = tid & 15, r4 = tid >> 4;  // q: channel quad, r4: 16 token rows per pass
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = r4 + 16 * ps;
            const int mm = m0 + r, cc = c0 + q * 4;
            uint2 h = make_uint2(0u, 0u), l = make_uint2(0u, 0u);
            if (mm < m && cc < c) {
                long src = mm;
                if (gather) {
                    const int b = mm / hw_out, p = mm - b * hw_out;
                    const int g = gather[tap * hw_out + p];
                    src = g >= 0 ? (long)b * hw_src + g : -1;
                }
                if (src >= 0) {
                    if (F32IN) {
                        const float* xp = reinterpret_cast<const float*>(in_hi) + src * ld + cc;
                        if (cc + 3 < c && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(xp) & 15) == 0)) {
                            wd_split4(*reinterpret_cast<const float4*>(xp), h, l);
                        } else {
                            float4 v = make_float4(xp[0], cc + 1 < c ? xp[1] : 0.f, cc + 2 < c ? xp[2] : 0.f, cc + 3 < c ? xp[3] : 0.f);
                            wd_split4(v, h, l);
                        }
                    } else {
                        const wd_bf16* hp = reinterpret_cast<const wd_bf16*>(in_hi) + src * ld + cc;
                        const wd_bf16* lp = in_lo ? reinterpret_cast<const wd_bf16*>(in_lo) + src * ld + cc : nullptr;
                        if (cc + 3 < c && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(hp) & 7) == 0) &&
                            (!lp || (reinterpret_cast<uintptr_t>(lp) & 7) == 0)) {
                            h = *reinterpret_cast<const uint2*>(hp);
                            if (lp) l = *reinterpret_cast<const uint2*>(lp);
                        } else {
                            uint32_t e[4] = {0, 0, 0, 0}, f[4] = {0, 0, 0, 0};
                            for (int u = 0; u < 4; ++u)
                                if (cc + u < c) {
                                    e[u] = hp[u];
                                    if (lp) f[u] = lp[u];
                                }
                            h = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
                            l = make_uint2(f[0] | (f[1] << 16), f[2] | (f[3] << 16));
                        }
                    }
                }
            }
            *reinterpret_cast<uint2*>(&th[r][q * 4]) = h;
            *reinterpret_cast<uint2*>(&tl[r][q * 4]) = l;
        }
    }
    __syncthreads();
    {
        const int mq = tid & 15, c4 = tid >> 4;  // mq: token quad, c4: 16 channel rows per pass
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = c4 + 16 * ps;          // channel inside the tile
            const int cc = c0 + r, mm = m0 + mq * 4;
            if (cc < c && mm < mpad) {
                const long o = (tap_minor ? (long)cc * ntaps + tap : (long)tap * c + cc) * mpad + mm;
                const uint32_t a0 = th[mq * 4][r], a1 = th[mq * 4 + 1][r], a2 = th[mq * 4 + 2][r], a3 = th[mq * 4 + 3][r];
                *reinterpret_cast<uint2*>(out_hi + o) = make_uint2(a0 | (a1 << 16), a2 | (a3 << 16));
                if (out_lo) {
                    const uint32_t b0 = tl[mq * 4][r], b1 = tl[mq * 4 + 1][r], b2 = tl[mq * 4 + 2][r], b3 = tl[mq * 4 + 3][r];
                    *reinterpret_cast<uint2*>(out_lo + o) = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
                }
            }
        }
    }
}

// ---- one pass over d(output) [M][n] fp32 for everything its layer's backward needs: row-major split planes (operand of the
// data-gradient GEMM), transposed split planes (operand of the weight-gradient GEMM) and per-64-row column sums (bias /
// FiLM gradients after wd_colsum_finish).  Any of the three outputs may be NULL.
// GEGLU: d is not read but computed - d(output) of the GEGLU projection from the saved pre-activation u = [x | gate] ([m][ld], ld >=
// 2 inner) and the gradient dh [m][dh_ld] of a . gelu(gate): columns [0, inner) = dh gelu(gate), [inner, 2 inner) = dh x gelu'(gate)
// (wd_geglu_bwd's arithmetic) - so the 2 inner wide fp32 gradient is never written and read back (168 + 168 MB per 8x32 block).
template <bool GEGLU>
__global__ void __launch_bounds__(256) dout_prep_kernel(const float* __restrict__ d, int ld, int m, int n, int npad, int mpad,
                                                        wd_bf16* __restrict__ pl_hi, wd_bf16* __restrict__ pl_lo,
                                                        wd_bf16* __restrict__ t_hi, wd_bf16* __restrict__ t_lo,
                                                        float* __restrict__ colpart, const float* __restrict__ dh, int dh_ld, int inner) {
    __shared__ wd_bf16 th[64][68], tl[64][68];
    __shared__ float cs[16][65];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int q = tid & 15, r4 = tid >> 4;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = r4 + 16 * ps;
        const int mm = m0 + r, cc = c0 + q * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (GEGLU) {
            if (mm < m && cc < n) {  // (n = 2 inner, inner % 64 == 0: a 64-column block lies in one half)
                const int j = cc < inner ? cc : cc - inner;
                const float4 g = *reinterpret_cast<const float4*>(d + (long)mm * ld + inner + j);
                const float4 dd = *reinterpret_cast<const float4*>(dh + (long)mm * dh_ld + j);
                if (cc < inner) {
                    v = make_float4(dd.x * wd_gelu_erf(g.x), dd.y * wd_gelu_erf(g.y), dd.z * wd_gelu_erf(g.z), dd.w * wd_gelu_erf(g.w));
                } else {
                    const float4 xa = *reinterpret_cast<const float4*>(d + (long)mm * ld + j);
                    v = make_float4(dd.x * xa.x * gelu_grad(g.x), dd.y * xa.y * gelu_grad(g.y), dd.z * xa.z * gelu_grad(g.z),
                                    dd.w * xa.w * gelu_grad(g.w));
                }
            }
        } else if (mm < m && cc < n) {
            const float* xp = d + (long)mm * ld + cc;
            if (cc + 3 < n) v = *reinterpret_cast<const float4*>(xp);
            else v = make_float4(xp[0], cc + 1 < n ? xp[1] : 0.f, cc + 2 < n ? xp[2] : 0.f, 0.f);
        }
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        uint2 h, l;
        wd_split4(v, h, l);
        *reinterpret_cast<uint2*>(&th[r][q * 4]) = h;
        *reinterpret_cast<uint2*>(&tl[r][q * 4]) = l;
        if (pl_hi && mm < m && cc < npad) {  // (columns n..npad-1 of the row-major planes are written as zeros)
            *reinterpret_cast<uint2*>(pl_hi + (long)mm * npad + cc) = h;
            if (pl_lo) *reinterpret_cast<uint2*>(pl_lo + (long)mm * npad + cc) = l;
        }
    }
    if (colpart) {
        cs[r4][q * 4 + 0] = sum.x; cs[r4][q * 4 + 1] = sum.y; cs[r4][q * 4 + 2] = sum.z; cs[r4][q * 4 + 3] = sum.w;
    }
    __syncthreads();
    if (colpart && tid < 64 && c0 + tid < n) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += cs[k][tid];
        colpart[(long)blockIdx.x * n + c0 + tid] = a;
    }
    if (t_hi) {
        const int mq = tid & 15, c4 = tid >> 4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = c4 + 16 * ps;
            const int cc = c0 + r, mm = m0 + mq * 4;
            if (cc < n && mm < mpad) {
                const long o = (long)cc * mpad + mm;
                const uint32_t a0 = th[mq * 4][r], a1 = th[mq * 4 + 1][r], a2 = th[mq * 4 + 2][r], a3 = th[mq * 4 + 3][r];
                *reinterpret_cast<uint2*>(t_hi + o) = make_uint2(a0 | (a1 << 16), a2 | (a3 << 16));
                if (t_lo) {
                    const uint32_t b0 = tl[mq * 4][r], b1 = tl[mq * 4 + 1][r], b2 = tl[mq * 4 + 2][r], b3 = tl[mq * 4 + 3][r];
                    *reinterpret_cast<uint2*>(t_lo + o) = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
                }
            }
        }
    }
}

// ---- column sums of x[rows][ld] over row segments of `seg` rows: out[s][c] (+= when accumulate), two stages, fixed order
constexpr int CS_ROWS = 128;  // rows per stage-1 workgroup: 4 row lanes x 32 rows, 64 column quads per workgroup
__global__ void __launch_bounds__(256) colsum_stage1(const float* __restrict__ x, int ld, int rows, int c, int seg,
                                                     float* __restrict__ part, int vec) {
    // grid (ceil(seg / CS_ROWS), nseg, ceil(c / 256)); thread = (column quad q, row lane rl)
    __shared__ float4 red[4][64];
    const int q = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.z * 256 + q * 4;
    const int s = blockIdx.y;
    const int r0 = s * seg + blockIdx.x * CS_ROWS, r1 = min(min(r0 + CS_ROWS, (s + 1) * seg), rows);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < c) {
        if (vec && col + 3 < c) {
            for (int r = r0 + rl; r < r1; r += 4) {
                const float4 v = *reinterpret_cast<const float4*>(x + (long)r * ld + col);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        } else {
            for (int r = r0 + rl; r < r1; r += 4) {
                const float* p = x + (long)r * ld + col;
                acc.x += p[0];
                if (col + 1 < c) acc.y += p[1];
                if (col + 2 < c) acc.z += p[2];
                if (col + 3 < c) acc.w += p[3];
            }
        }
    }
    red[rl][q] = acc;
    __syncthreads();
    if (rl == 0 && col < c) {
        const float4 a0 = red[0][q], a1 = red[1][q], a2 = red[2][q], a3 = red[3][q];
        float* o = part + ((long)s * gridDim.x + blockIdx.x) * c + col;
        o[0] = (a0.x + a1.x) + (a2.x + a3.x);
        if (col + 1 < c) o[1] = (a0.y + a1.y) + (a2.y + a3.y);
        if (col + 2 < c) o[2] = (a0.z + a1.z) + (a2.z + a3.z);
        if (col + 3 < c) o[3] = (a0.w + a1.w) + (a2.w + a3.w);
    }
}
__global__ void __launch_bounds__(256) colsum_stage2(const float* __restrict__ part, int nblk, int c, int nseg,
                                                     float* __restrict__ out, int out_ld, int accumulate, float scale) {
    // grid (ceil(c / 16), nseg); thread = (row lane rl of 16, column cl of 16): 64-byte coalesced reads, fixed-order sums
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl, s = blockIdx.y;
    double acc = 0.0;
    if (col < c)
        for (int k = rl; k < nblk; k += 16) acc += (double)part[((long)s * nblk + k) * c + col];
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && col < c) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        const float v = (float)t * scale;
        float* o = out + (long)s * out_ld + col;
        *o = accumulate ? *o + v : v;
    }
}

// The same finish for a whole list of (partial sums -> parameter gradient) pairs in one launch: grid (column tiles of the widest
// entry, entries).  The training backward defers its bias-gradient finishes to one of these at its end (they are launch-bound:
// ~45 launches of a few microseconds each otherwise).
struct ColRef {
    const float* part;
    float* out;
    int32_t nblk, c, ld, accumulate;
    float scale;
    int32_t pad;
};
__global__ void __launch_bounds__(256) colsum_multi_kernel(const ColRef* __restrict__ tab) {
    __shared__ double red[16][17];
    const ColRef e = tab[blockIdx.y];
    if ((int)blockIdx.x * 16 >= e.c) return;
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    double acc = 0.0;
    if (col < e.c) {
        const float* p = e.part + col;
        int k = rl;
        for (; k + 48 < e.nblk; k += 64) {  // four rows in flight per thread; the order of the adds is fixed
            const float v0 = p[(long)k * e.ld], v1 = p[(long)(k + 16) * e.ld], v2 = p[(long)(k + 32) * e.ld],
                        v3 = p[(long)(k + 48) * e.ld];
            acc += (double)v0;
            acc += (double)v1;
            acc += (double)v2;
            acc += (double)v3;
        }
        for (; k < e.nblk; k += 16) acc += (double)p[(long)k * e.ld];
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && col < e.c) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        const float v = (float)t * e.scale;
        float* o = e.out + col;
        *o = e.accumulate ? *o + v : v;
    }
}

// ---- GroupNorm (+SiLU) backward, pass 1: per (sample, chunk, channel) sums of dy and dy * xhat,
// sums[b][chunk][2][c] (planar: row-summing it gives [d beta | d gamma])
constexpr int GB_TOK = 32;
__global__ void gn_bwd_stats_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dz, int dz_ld,
                                    int dz_off, int hw, int c, int cpg, const double* __restrict__ part, int nchunk_f,
                                    int part_cpg, const float* __restrict__ gamma, const float* __restrict__ beta, int c_off,
                                    float eps, int silu, int nchunk, float* __restrict__ sums) {
    // grid (nchunk, batch), block (64 * ceil(c/4/64), 2); sums: [b][chunk][2][c] (planar)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_mean = reinterpret_cast<float*>(smem);
    float* s_rstd = s_mean + 32;
    float* s_acc = s_rstd + 32;  // [2 (y)][c][2]
    const int b = blockIdx.y, j = blockIdx.x;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int ng = c / cpg;
    if (tid < ng) {
        const int ratio = cpg / part_cpg, ngs = c / part_cpg;
        double ds = 0.0, dq = 0.0;
        for (int k = 0; k < nchunk_f; ++k) {
            const double* p = part + (((long)b * nchunk_f + k) * ngs + tid * ratio) * 2;
            for (int q = 0; q < ratio; ++q) {
                ds += p[2 * q];
                dq += p[2 * q + 1];
            }
        }
        const double n = (double)hw * cpg, mean = ds / n;
        double var = dq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cx = threadIdx.x * 4;
    if (cx < c) {
        const int t0 = j * GB_TOK, t1 = min(hw, t0 + GB_TOK);
        float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c_off + cx);
        const float4 be = *reinterpret_cast<const float4*>(beta + c_off + cx);
        const float gam[4] = {ga.x, ga.y, ga.z, ga.w}, bet[4] = {be.x, be.y, be.z, be.w};
        for (int t = t0 + threadIdx.y; t < t1; t += 2) {
            const long row = (long)b * hw + t;
            const float4 xv = *reinterpret_cast<const float4*>(x + row * ld + cx);
            const float4 dv = *reinterpret_cast<const float4*>(dz + row * dz_ld + dz_off + cx);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds4[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = (cx + k) / cpg;
                const float xh = (xs[k] - s_mean[g]) * s_rstd[g];
                float dy = ds4[k];
                if (silu) dy *= silu_grad(gam[k] * xh + bet[k]);
                s1[k] += dy;
                s2[k] += dy * xh;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s_acc[(threadIdx.y * c + cx + k) * 2] = s1[k];
            s_acc[(threadIdx.y * c + cx + k) * 2 + 1] = s2[k];
        }
    }
    __syncthreads();
    for (int i = tid; i < c; i += blockDim.x * blockDim.y) {
        float* o = sums + ((long)b * nchunk + j) * 2 * c + i;  // [b][chunk][{sum dy, sum dy*xhat}][c]
        o[0] = s_acc[i * 2] + s_acc[(c + i) * 2];
        o[c] = s_acc[i * 2 + 1] + s_acc[(c + i) * 2 + 1];
    }
}

// pass 2: dx = rstd * (gamma dy - m1_g - xhat m2_g),  m1_g = mean_g(gamma dy), m2_g = mean_g(gamma dy xhat)
constexpr int GA_TOK = 16;
__global__ void gn_bwd_apply_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dz, int dz_ld, int dz_off,
                                    int hw, int c, int cpg, const double* __restrict__ part, int nchunk_f, int part_cpg,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int c_off, float eps,
                                    int silu, const float* __restrict__ sums, int nchunk, float* __restrict__ dx, int dx_ld,
                                    int accumulate) {
    __shared__ float s_mean[32], s_rstd[32], s_m1[32], s_m2[32];
    const int b = blockIdx.y;
    const int ng = c / cpg;
    if (threadIdx.x < ng) {
        const int g = threadIdx.x;
        const int ratio = cpg / part_cpg, ngs = c / part_cpg;
        double ds = 0.0, dq = 0.0;
        for (int k = 0; k < nchunk_f; ++k) {
            const double* p = part + (((long)b * nchunk_f + k) * ngs + g * ratio) * 2;
            for (int q = 0; q < ratio; ++q) {
                ds += p[2 * q];
                dq += p[2 * q + 1];
            }
        }
        const double n = (double)hw * cpg, mean = ds / n;
        double var = dq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[g] = (float)mean;
        s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
        double a1 = 0.0, a2 = 0.0;
        for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) {
            double t1 = 0.0, t2 = 0.0;
            for (int k = 0; k < nchunk; ++k) {
                const float* s = sums + ((long)b * nchunk + k) * 2 * c + cc;
                t1 += (double)s[0];
                t2 += (double)s[c];
            }
            a1 += (double)gamma[c_off + cc] * t1;
            a2 += (double)gamma[c_off + cc] * t2;
        }
        s_m1[g] = (float)(a1 / n);
        s_m2[g] = (float)(a2 / n);
    }
    __syncthreads();
    const int c4 = c >> 2;
    const int t0 = blockIdx.x * GA_TOK, nt = min(GA_TOK, hw - t0);
    if (c4 <= (int)blockDim.x) {
        // thread = (token lane, channel quad): the quad is fixed, so the group lookups (integer divisions by a run-time cpg) and
        // the per-channel constants are set up once per thread instead of once per element
        const int rpp = blockDim.x / c4;
        const int tl = threadIdx.x / c4, cx = (threadIdx.x - tl * c4) * 4;
        if (tl >= rpp) return;
        float mean[4], rstd[4], m1[4], m2[4], gam[4], bet[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = (cx + k) / cpg;
            mean[k] = s_mean[g]; rstd[k] = s_rstd[g]; m1[k] = s_m1[g]; m2[k] = s_m2[g];
            gam[k] = gamma[c_off + cx + k]; bet[k] = beta[c_off + cx + k];
        }
        for (int t = tl; t < nt; t += rpp) {
            const long row = (long)b * hw + t0 + t;
            const float4 xv = *reinterpret_cast<const float4*>(x + row * ld + cx);
            const float4 dv = *reinterpret_cast<const float4*>(dz + row * dz_ld + dz_off + cx);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds4[4] = {dv.x, dv.y, dv.z, dv.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xs[k] - mean[k]) * rstd[k];
                float dy = ds4[k];
                if (silu) dy *= silu_grad(gam[k] * xh + bet[k]);
                o[k] = rstd[k] * (gam[k] * dy - m1[k] - xh * m2[k]);
            }
            float4* op = reinterpret_cast<float4*>(dx + row * dx_ld + cx);
            float4 r = make_float4(o[0], o[1], o[2], o[3]);
            if (accumulate) {
                const float4 old = *op;
                r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w;
            }
            *op = r;
        }
        return;
    }
    for (int i = threadIdx.x; i < nt * c4; i += blockDim.x) {
        const int t = i / c4, cx = (i - t * c4) * 4;
        const long row = (long)b * hw + t0 + t;
        const float4 xv = *reinterpret_cast<const float4*>(x + row * ld + cx);
        const float4 dv = *reinterpret_cast<const float4*>(dz + row * dz_ld + dz_off + cx);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c_off + cx);
        const float4 be = *reinterpret_cast<const float4*>(beta + c_off + cx);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds4[4] = {dv.x, dv.y, dv.z, dv.w};
        const float gam[4] = {ga.x, ga.y, ga.z, ga.w}, bet[4] = {be.x, be.y, be.z, be.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = (cx + k) / cpg;
            const float xh = (xs[k] - s_mean[g]) * s_rstd[g];
            float dy = ds4[k];
            if (silu) dy *= silu_grad(gam[k] * xh + bet[k]);
            o[k] = s_rstd[g] * (gam[k] * dy - s_m1[g] - xh * s_m2[g]);
        }
        float4* op = reinterpret_cast<float4*>(dx + row * dx_ld + cx);
        float4 r = make_float4(o[0], o[1], o[2], o[3]);
        if (accumulate) {
            const float4 old = *op;
            r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w;
        }
        *op = r;
    }
}

// ---- LayerNorm backward: one wave per row; a block walks LB_ROWS rows and emits per-column partial sums of
// dy * xhat (-> d gamma) and dy (-> d beta): colpart[blk][c][2]
constexpr int LB_MAX4 = 8;
constexpr int LB_ROWS = 16;
// MAX4 = float4 per lane and row (c <= 256 MAX4): the per-lane row image and the column sums live in registers, so the small
// instantiation (c <= 512) runs at full occupancy where the general one is limited to two waves per SIMD
template <int MAX4>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dy,
                                                            int dy_ld, int rows, int c, const float* __restrict__ gamma,
                                                            float eps, float* __restrict__ dx, int dx_ld, int accumulate,
                                                            float* __restrict__ colpart) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_col = reinterpret_cast<float*>(smem);  // [4 waves][c][2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = c >> 2;
    float4 ag[MAX4], ab[MAX4];
#pragma unroll
    for (int i = 0; i < MAX4; ++i) ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int r0 = blockIdx.x * LB_ROWS;
    for (int rr = wave; rr < LB_ROWS; rr += 4) {
        const int row = r0 + rr;
        if (row >= rows) break;
        float4 v[MAX4], d[MAX4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAX4; ++i) {
            const int f = lane + 64 * i;
            if (f < c4) {
                v[i] = *reinterpret_cast<const float4*>(x + (long)row * ld + f * 4);
                d[i] = *reinterpret_cast<const float4*>(dy + (long)row * dy_ld + f * 4);
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        const float mean = wd_wave_sum(s) / (float)c;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAX4; ++i) {
            const int f = lane + 64 * i;
            if (f < c4) {
                const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = 1.0f / sqrtf(wd_wave_sum(q) / (float)c + eps);
        float t1 = 0.f, t2 = 0.f;  // sum(dxhat), sum(dxhat * xhat)
#pragma unroll
        for (int i = 0; i < MAX4; ++i) {
            const int f = lane + 64 * i;
            if (f < c4) {
                const float4 ga = *reinterpret_cast<const float4*>(gamma + f * 4);
                float4 xh;
                xh.x = (v[i].x - mean) * rstd; xh.y = (v[i].y - mean) * rstd;
                xh.z = (v[i].z - mean) * rstd; xh.w = (v[i].w - mean) * rstd;
                ag[i].x += d[i].x * xh.x; ag[i].y += d[i].y * xh.y; ag[i].z += d[i].z * xh.z; ag[i].w += d[i].w * xh.w;
                ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
                d[i].x *= ga.x; d[i].y *= ga.y; d[i].z *= ga.z; d[i].w *= ga.w;  // dxhat
                t1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
                t2 += (d[i].x * xh.x + d[i].y * xh.y) + (d[i].z * xh.z + d[i].w * xh.w);
                v[i] = xh;
            }
        }
        t1 = wd_wave_sum(t1) / (float)c;
        t2 = wd_wave_sum(t2) / (float)c;
#pragma unroll
        for (int i = 0; i < MAX4; ++i) {
            const int f = lane + 64 * i;
            if (f < c4) {
                float4 o;
                o.x = rstd * (d[i].x - t1 - v[i].x * t2); o.y = rstd * (d[i].y - t1 - v[i].y * t2);
                o.z = rstd * (d[i].z - t1 - v[i].z * t2); o.w = rstd * (d[i].w - t1 - v[i].w * t2);
                float4* op = reinterpret_cast<float4*>(dx + (long)row * dx_ld + f * 4);
                if (accumulate) {
                    const float4 old = *op;
                    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                *op = o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAX4; ++i) {
        const int f = lane + 64 * i;
        if (f < c4) {
            *reinterpret_cast<float4*>(s_col + (wave * 2) * c + f * 4) = ag[i];      // [wave][{d gamma, d beta}][c]
            *reinterpret_cast<float4*>(s_col + (wave * 2 + 1) * c + f * 4) = ab[i];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c * 2; i += 256) {
        const float v = (s_col[i] + s_col[c * 2 + i]) + (s_col[2 * c * 2 + i] + s_col[3 * c * 2 + i]);
        colpart[(long)blockIdx.x * c * 2 + i] = v;
    }
}

// ---- attention backward, few keys: thread = (token, head); dq written per token; dK / dV reduced over the
// workgroup's tokens through LDS in fixed order and written as per-workgroup partials [b][nwg][nk][2][inner]
constexpr int NKB = 16;
__global__ void __launch_bounds__(256) attn_bwd_small_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                             int ldk, const float* __restrict__ v, int ldv,
                                                             const float* __restrict__ dout, int ldo, int heads, int nq,
                                                             int nk, int d, float scale, float* __restrict__ dq, int lddq,
                                                             float* __restrict__ dkv_part, int tpw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int inner = heads * d;
    float* s_k = reinterpret_cast<float*>(smem);   // [nk][inner]
    float* s_v = s_k + nk * inner;                 // [nk][inner]
    float* s_ds = s_v + nk * inner;                // [tpw][heads][NKB]: d(score) per (token, head, key)
    float* s_p = s_ds + tpw * heads * NKB;         // [tpw][heads][NKB]: softmax probabilities
    const int b = blockIdx.y, tid = threadIdx.x;
    const int i4 = inner >> 2;
    const int tok0 = blockIdx.x * tpw, ntok = min(tpw, nq - tok0);
    for (int e = tid; e < nk * i4; e += 256) {
        const int j = e / i4, c = (e - j * i4) * 4;
        *reinterpret_cast<float4*>(s_k + j * inner + c) = *reinterpret_cast<const float4*>(k + ((long)b * nk + j) * ldk + c);
        *reinterpret_cast<float4*>(s_v + j * inner + c) = *reinterpret_cast<const float4*>(v + ((long)b * nk + j) * ldv + c);
    }
    __syncthreads();
    const int h = tid % heads, tl = tid / heads;
    const bool act = tl < ntok;
    const int hoff = h * d, d4 = d >> 2;
    const long row = (long)b * nq + tok0 + tl;
    float p[NKB], ds[NKB];
#pragma unroll
    for (int j = 0; j < NKB; ++j) p[j] = ds[j] = 0.f;
    if (act) {
        const float* qr = q + row * ldq + hoff;
        const float* dor = dout + row * ldo + hoff;
        float dp[NKB];
#pragma unroll
        for (int j = 0; j < NKB; ++j) dp[j] = 0.f;
        for (int c = 0; c < d4; ++c) {
            const float4 qv = *reinterpret_cast<const float4*>(qr + c * 4);
            const float4 gv = *reinterpret_cast<const float4*>(dor + c * 4);
#pragma unroll
            for (int j = 0; j < NKB; ++j)
                if (j < nk) {
                    const float4 kv = *reinterpret_cast<const float4*>(s_k + j * inner + hoff + c * 4);
                    const float4 vv = *reinterpret_cast<const float4*>(s_v + j * inner + hoff + c * 4);
                    p[j] += qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w;
                    dp[j] += gv.x * vv.x + gv.y * vv.y + gv.z * vv.z + gv.w * vv.w;
                }
        }
        float mx = -3.4e38f;
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) {
                p[j] *= scale;
                mx = fmaxf(mx, p[j]);
            }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) {
                p[j] = expf(p[j] - mx);
                sum += p[j];
            }
        const float inv = 1.f / sum;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) {
                p[j] *= inv;
                dot += p[j] * dp[j];
            }
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) ds[j] = p[j] * (dp[j] - dot) * scale;  // d(score before scaling) incl. the scale factor
        // dq = sum_j ds_j k_j
        for (int c = 0; c < d4; ++c) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < NKB; ++j)
                if (j < nk) {
                    const float4 kv = *reinterpret_cast<const float4*>(s_k + j * inner + hoff + c * 4);
                    o.x += ds[j] * kv.x; o.y += ds[j] * kv.y; o.z += ds[j] * kv.z; o.w += ds[j] * kv.w;
                }
            *reinterpret_cast<float4*>(dq + row * lddq + hoff + c * 4) = o;
        }
    }
    // dK_j = sum_tokens ds_j q ; dV_j = sum_tokens p_j dO: the per-(token, head) coefficients go through LDS, then one thread
    // per output column accumulates all keys over the workgroup's tokens in a fixed order
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
        if (tl < tpw) {
            s_ds[(tl * heads + h) * NKB + j] = act ? ds[j] : 0.f;
            s_p[(tl * heads + h) * NKB + j] = act ? p[j] : 0.f;
        }
    }
    __syncthreads();
    float* outp = dkv_part + (((long)b * gridDim.x + blockIdx.x) * 2) * nk * inner;
    const long row0 = (long)b * nq + tok0;
    for (int col = tid; col < inner; col += 256) {
        const int hh = col / d;
        float ak[NKB], av[NKB];
#pragma unroll
        for (int j = 0; j < NKB; ++j) ak[j] = av[j] = 0.f;
        for (int t0 = 0; t0 < ntok; t0 += 8) {  // eight tokens per round: their loads are in flight together
            float qv[8], gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool okt = t0 + u < ntok;
                qv[u] = okt ? q[(row0 + t0 + u) * ldq + col] : 0.f;
                gv[u] = okt ? dout[(row0 + t0 + u) * ldo + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u < ntok ? t0 + u : 0;  // (padding tokens multiply by 0)
                const float* cds = s_ds + (t * heads + hh) * NKB;
                const float* cp = s_p + (t * heads + hh) * NKB;
#pragma unroll
                for (int j = 0; j < NKB; ++j)
                    if (j < nk) {
                        ak[j] += cds[j] * qv[u];
                        av[j] += cp[j] * gv[u];
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) {
                outp[((long)j * 2 + 0) * inner + col] = ak[j];
                outp[((long)j * 2 + 1) * inner + col] = av[j];
            }
    }
}


// ---- attention backward, any number of keys (spatial self-attention / the 779-token PHOSC context; unetPhosc.py:157-198).
// Pass 1, one wave per query row: recompute the softmax row, write P and dS (scaled) to scratch, and dq.
// Pass 2, one wave per key: dK_j = sum_i dS_ij q_i, dV_j = sum_i P_ij dO_i in query order (deterministic, no partials).
constexpr int AG_NJ = 16;  // keys per lane: up to 64 * 16 = 1024 keys
__global__ void __launch_bounds__(256) attn_bwd_rows_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                            int ldk, const float* __restrict__ v, int ldv,
                                                            const float* __restrict__ dout, int ldo, int heads, int nq, int nk,
                                                            int d, float scale, float* __restrict__ dq, int lddq,
                                                            float* __restrict__ pbuf, float* __restrict__ dsbuf,
                                                            long total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s_q = reinterpret_cast<float*>(smem) + wave * (2 * d + nk);  // [d] q row, [d] dO row, [nk] dS row
    float* s_g = s_q + d;
    float* s_ds = s_g + d;
    const long rowid = (long)blockIdx.x * 4 + wave;                     // (b * heads + h) * nq + i
    if (rowid >= total) return;
    const int i = (int)(rowid % nq);
    const int h = (int)((rowid / nq) % heads);
    const int b = (int)(rowid / ((long)nq * heads));
    const float* qr = q + ((long)b * nq + i) * ldq + h * d;
    const float* gr = dout + ((long)b * nq + i) * ldo + h * d;
    for (int c = lane; c < d; c += 64) {
        s_q[c] = qr[c];
        s_g[c] = gr[c];
    }
    __builtin_amdgcn_wave_barrier();
    float sc[AG_NJ], dp[AG_NJ];
    float mx = -3.4e38f;
#pragma unroll
    for (int t = 0; t < AG_NJ; ++t) {
        const int j = lane + 64 * t;
        sc[t] = -3.4e38f;
        dp[t] = 0.f;
        if (j < nk) {
            const float* kr = k + ((long)b * nk + j) * ldk + h * d;
            const float* vr = v + ((long)b * nk + j) * ldv + h * d;
            float a = 0.f, g = 0.f;
            for (int c = 0; c < d; c += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(kr + c), vv = *reinterpret_cast<const float4*>(vr + c);
                a += s_q[c] * kv.x + s_q[c + 1] * kv.y + s_q[c + 2] * kv.z + s_q[c + 3] * kv.w;
                g += s_g[c] * vv.x + s_g[c + 1] * vv.y + s_g[c + 2] * vv.z + s_g[c + 3] * vv.w;
            }
            sc[t] = a * scale;
            dp[t] = g;
            mx = fmaxf(mx, sc[t]);
        }
    }
    mx = wd_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < AG_NJ; ++t) {
        const int j = lane + 64 * t;
        sc[t] = j < nk ? expf(sc[t] - mx) : 0.f;
        sum += sc[t];
    }
    const float inv = 1.f / wd_wave_sum(sum);
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < AG_NJ; ++t) {
        sc[t] *= inv;
        dot += sc[t] * dp[t];
    }
    dot = wd_wave_sum(dot);
    float* prow = pbuf + rowid * nk;
    float* dsrow = dsbuf + rowid * nk;
#pragma unroll
    for (int t = 0; t < AG_NJ; ++t) {
        const int j = lane + 64 * t;
        if (j < nk) {
            const float ds = sc[t] * (dp[t] - dot) * scale;
            prow[j] = sc[t];
            dsrow[j] = ds;
            s_ds[j] = ds;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < d; c += 64) {
        float acc = 0.f;
        const float* kc = k + (long)b * nk * ldk + h * d + c;
        for (int j = 0; j < nk; ++j) acc += s_ds[j] * kc[(long)j * ldk];
        dq[((long)b * nq + i) * lddq + h * d + c] = acc;
    }
}

__global__ void __launch_bounds__(256) attn_bwd_cols_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ dout,
                                                            int ldo, int heads, int nq, int nk, int d,
                                                            const float* __restrict__ pbuf, const float* __restrict__ dsbuf,
                                                            float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv,
                                                            long total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long colid = (long)blockIdx.x * 4 + wave;  // (b * heads + h) * nk + j
    if (colid >= total) return;
    const int j = (int)(colid % nk);
    const int h = (int)((colid / nk) % heads);
    const int b = (int)(colid / ((long)nk * heads));
    const float* pcol = pbuf + ((long)(b * heads + h) * nq) * nk + j;
    const float* dscol = dsbuf + ((long)(b * heads + h) * nq) * nk + j;
    for (int c0 = 0; c0 < d; c0 += 64) {
        const int c = c0 + lane;
        float ak = 0.f, av = 0.f;
        if (c < d) {
            const float* qc = q + (long)b * nq * ldq + h * d + c;
            const float* gc = dout + (long)b * nq * ldo + h * d + c;
            for (int i = 0; i < nq; ++i) {
                ak += dscol[(long)i * nk] * qc[(long)i * ldq];
                av += pcol[(long)i * nk] * gc[(long)i * ldo];
            }
            dk[((long)b * nk + j) * lddk + h * d + c] = ak;
            dv[((long)b * nk + j) * lddv + h * d + c] = av;
        }
    }
}

__global__ void geglu_fwd_kernel(const float* __restrict__ u, int ld, long rows, int inner, wd_bf16* __restrict__ out_hi,
                                 wd_bf16* __restrict__ out_lo, int out_ld) {
    const int i4 = inner >> 2;
    const long total = rows * i4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / i4;
        const int c = (int)(i - r * i4) * 4;
        const float4 a = *reinterpret_cast<const float4*>(u + r * ld + c);
        const float4 g = *reinterpret_cast<const float4*>(u + r * ld + inner + c);
        float4 o;
        o.x = a.x * wd_gelu_erf(g.x); o.y = a.y * wd_gelu_erf(g.y); o.z = a.z * wd_gelu_erf(g.z); o.w = a.w * wd_gelu_erf(g.w);
        uint2 h, l;
        wd_split4(o, h, l);
        *reinterpret_cast<uint2*>(out_hi + r * out_ld + c) = h;
        if (out_lo) *reinterpret_cast<uint2*>(out_lo + r * out_ld + c) = l;
    }
}
__global__ void geglu_bwd_kernel(const float* __restrict__ u, int ld, const float* __restrict__ dh, int dh_ld, long rows,
                                 int inner, float* __restrict__ du, int du_ld) {
    const long total = rows * inner;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / inner;
        const int c = (int)(i - r * inner);
        const float a = u[r * ld + c], g = u[r * ld + inner + c], d = dh[r * dh_ld + c];
        du[r * du_ld + c] = d * wd_gelu_erf(g);
        du[r * du_ld + inner + c] = d * a * gelu_grad(g);
    }
}
// dpre = dact * silu'(pre)   (time-embedding MLP: the saved tensors are SiLU outputs' pre-activations)
__global__ void silu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dact, long n,
                                float* __restrict__ dpre) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dpre[i] = dact[i] * silu_grad(pre[i]);
}
// nearest-x2 upsample backward: out[b][y][x][c] = sum of the 2x2 block of in[b][2y..][2x..][c]
__global__ void pool2x2_sum_kernel(const float* __restrict__ in, int batch, int h, int w, int c, float* __restrict__ out) {
    const int c4 = c >> 2;
    const long total = (long)batch * h * w * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cx = (int)(i % c4) * 4;
        const long t = i / c4;
        const int x = (int)(t % w), y = (int)((t / w) % h), b = (int)(t / ((long)w * h));
        const long base = (((long)b * 2 * h + 2 * y) * 2 * w + 2 * x) * c + cx;
        const float4 a0 = *reinterpret_cast<const float4*>(in + base);
        const float4 a1 = *reinterpret_cast<const float4*>(in + base + c);
        const float4 a2 = *reinterpret_cast<const float4*>(in + base + (long)2 * w * c);
        const float4 a3 = *reinterpret_cast<const float4*>(in + base + (long)2 * w * c + c);
        *reinterpret_cast<float4*>(out + t * c + cx) = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                                                                  (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    }
}
// embedding backward: dtable[v][c] (+)= sum over rows r with ids[r] == v of d[r][c]  (one thread per table element)
__global__ void embedding_bwd_kernel(const void* __restrict__ ids, int i64, int rows, const float* __restrict__ d, int ld,
                                     int vocab, int c, float* __restrict__ dtable, int accumulate) {
    const long total = (long)vocab * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int vv = (int)(i / c), col = (int)(i - (long)vv * c);
        float acc = 0.f;
        for (int r = 0; r < rows; ++r) {
            const long id = i64 ? (long)reinterpret_cast<const int64_t*>(ids)[r] : (long)reinterpret_cast<const int32_t*>(ids)[r];
            if (id == vv) acc += d[(long)r * ld + col];
        }
        dtable[i] = accumulate ? dtable[i] + acc : acc;
    }
}

// dst[i] += src[i]
__global__ void add_kernel(float* __restrict__ dst, const float* __restrict__ src, long n4, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(dst)[i];
        const float4 b = reinterpret_cast<const float4*>(src)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        reinterpret_cast<float4*>(dst)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] += src[n4 * 4 + threadIdx.x];
}
// packed weight gradient [n][tap * c + ch] (row pitch ld) -> OIHW [n][ch][tap]
__global__ void permute_dw_kernel(const float* __restrict__ packed, int ld, int n, int c, int ntaps, float* __restrict__ out) {
    const long total = (long)n * c * ntaps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % ntaps), ch = (int)((i / ntaps) % c), r = (int)(i / ((long)ntaps * c));
        out[i] = packed[(long)r * ld + t * c + ch];
    }
}

inline int grid_for(long total, int block = 256, int cap = 4096) {
    long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

}  // namespace

extern "C" int wd_transpose_planes(const void* in_hi, const void* in_lo, int in_is_f32, int ld, int c, const int32_t* gather,
                                   int ntaps, int hw_out, int hw_src, int m, int mpad, int tap_minor, wd_bf16* out_hi,
                                   wd_bf16* out_lo, void* stream) {
    if (!in_hi || !out_hi || c <= 0 || m <= 0 || mpad < m || ntaps < 1 || (gather && (hw_out <= 0 || hw_src <= 0)))
        return WD_EINVAL;
    if (!gather && ntaps != 1) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    const dim3 grid((mpad + 63) / 64, (c + 63) / 64, ntaps);
    if (in_is_f32)
        hipLaunchKernelGGL(transpose_planes_kernel<true>, grid, dim3(256), 0, st, in_hi, in_lo, ld, c, gather, ntaps, hw_out,
                           hw_src, m, mpad, tap_minor, out_hi, out_lo);
    else
        hipLaunchKernelGGL(transpose_planes_kernel<false>, grid, dim3(256), 0, st, in_hi, in_lo, ld, c, gather, ntaps, hw_out,
                           hw_src, m, mpad, tap_minor, out_hi, out_lo);
    return wd_check_launch();
}

extern "C" int wd_colsum(const float* x, int ld, int rows, int c, int seg, float* out, int out_ld, int accumulate, float scale,
                         float* scratch, int64_t scratch_floats, void* stream) {
    if (!x || !out || !scratch || rows <= 0 || c <= 0 || seg <= 0) return WD_EINVAL;
    const int nseg = (rows + seg - 1) / seg, nblk = (seg + CS_ROWS - 1) / CS_ROWS;
    if ((int64_t)nseg * nblk * c > scratch_floats) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    const int vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    hipLaunchKernelGGL(colsum_stage1, dim3(nblk, nseg, (c + 255) / 256), dim3(256), 0, st, x, ld, rows, c, seg, scratch, vec);
    hipLaunchKernelGGL(colsum_stage2, dim3((c + 15) / 16, nseg), dim3(256), 0, st, scratch, nblk, c, nseg, out, out_ld,
                       accumulate, scale);
    return wd_check_launch();
}

// ---- GroupNorm (+SiLU) backward in ONE pass over the tensors: a workgroup owns (sample, GF_CB channels = whole groups), keeps
// dy (d z through the SiLU) and xhat of its hw x GF_CB tile in LDS (320 B per token: 80 KB at 8 x 32), sums them per channel in a
// fixed order, derives the group means and writes dx - x and dz are read once (42 MB + 21 MB written for a 320-channel 8 x 32 map
// at batch 64 instead of 84 + 21 in the two-pass form), one launch instead of two.  sums[b][0][2][c] as pass 1 of the two-pass form
// writes them with one chunk.
constexpr int GF_CB = 40;      // channels per workgroup (4 groups of 10, 2 of 20, 1 of 40)
constexpr int GF_NT = 256;
constexpr int GF_Q = GF_CB / 4;            // channel quads
constexpr int GF_RPP = GF_NT / GF_Q;       // token lanes (25; 250 threads work)
__global__ void __launch_bounds__(GF_NT) gn_bwd_fused_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dz, int dz_ld,
                                                            int dz_off, int hw, int c, int cpg, const double* __restrict__ part,
                                                            int nchunk_f, int part_cpg, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int c_off, float eps, int silu,
                                                            float* __restrict__ sums, float* __restrict__ dx, int dx_ld, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_dy = reinterpret_cast<float*>(smem);           // [hw][GF_CB]
    float* s_xh = s_dy + (long)hw * GF_CB;                    // [hw][GF_CB]
    float* s_red = s_xh + (long)hw * GF_CB;                   // [GF_RPP][2][GF_CB]
    __shared__ float s_mean[8], s_rstd[8], s_m1[8], s_m2[8];
    __shared__ float s_s1[GF_CB], s_s2[GF_CB];
    const int b = blockIdx.y, c0 = blockIdx.x * GF_CB;
    const int gpb = GF_CB / cpg, g0 = c0 / cpg;               // groups of this workgroup
    const int tid = threadIdx.x;
    if (tid < gpb) {
        const int g = g0 + tid;
        const int ratio = cpg / part_cpg, ngs = c / part_cpg;
        double ds = 0.0, dq = 0.0;
        for (int k = 0; k < nchunk_f; ++k) {
            const double* p = part + (((long)b * nchunk_f + k) * ngs + g * ratio) * 2;
            for (int q = 0; q < ratio; ++q) {
                ds += p[2 * q];
                dq += p[2 * q + 1];
            }
        }
        const double n = (double)hw * cpg, mean = ds / n;
        double var = dq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int tl = tid / GF_Q, cx = (tid - tl * GF_Q) * 4;   // token lane, first channel of the quad (inside the block)
    const bool work = tl < GF_RPP;
    float mean[4], rstd[4], gam[4], bet[4];
    int gl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gl[k] = (cx + k) / cpg;
        mean[k] = s_mean[gl[k]];
        rstd[k] = s_rstd[gl[k]];
        gam[k] = gamma[c_off + c0 + cx + k];
        bet[k] = beta[c_off + c0 + cx + k];
    }
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    if (work) {
        for (int t = tl; t < hw; t += GF_RPP) {
            const long row = (long)b * hw + t;
            const float4 xv = *reinterpret_cast<const float4*>(x + row * ld + c0 + cx);
            const float4 dv = *reinterpret_cast<const float4*>(dz + row * dz_ld + dz_off + c0 + cx);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds4[4] = {dv.x, dv.y, dv.z, dv.w};
            float dy[4], xh[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xh[k] = (xs[k] - mean[k]) * rstd[k];
                dy[k] = ds4[k];
                if (silu) dy[k] *= silu_grad(gam[k] * xh[k] + bet[k]);
                a1[k] += dy[k];
                a2[k] += dy[k] * xh[k];
            }
            *reinterpret_cast<float4*>(s_dy + (long)t * GF_CB + cx) = make_float4(dy[0], dy[1], dy[2], dy[3]);
            *reinterpret_cast<float4*>(s_xh + (long)t * GF_CB + cx) = make_float4(xh[0], xh[1], xh[2], xh[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s_red[(tl * 2 + 0) * GF_CB + cx + k] = a1[k];
            s_red[(tl * 2 + 1) * GF_CB + cx + k] = a2[k];
        }
    }
    __syncthreads();
    if (tid < 2 * GF_CB) {  // per-channel sums over the token lanes, fixed order
        const int which = tid / GF_CB, cc = tid - which * GF_CB;
        double t = 0.0;
        for (int k = 0; k < GF_RPP; ++k) t += (double)s_red[(k * 2 + which) * GF_CB + cc];
        (which ? s_s2 : s_s1)[cc] = (float)t;
        sums[((long)b * 2 + which) * c + c0 + cc] = (float)t;  // [b][1 chunk][{sum dy, sum dy*xhat}][c]
    }
    __syncthreads();
    if (tid < gpb) {
        double m1 = 0.0, m2 = 0.0;
        for (int cc = tid * cpg; cc < (tid + 1) * cpg; ++cc) {
            m1 += (double)gamma[c_off + c0 + cc] * (double)s_s1[cc];
            m2 += (double)gamma[c_off + c0 + cc] * (double)s_s2[cc];
        }
        const double n = (double)hw * cpg;
        s_m1[tid] = (float)(m1 / n);
        s_m2[tid] = (float)(m2 / n);
    }
    __syncthreads();
    if (!work) return;
    float m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m1[k] = s_m1[gl[k]];
        m2[k] = s_m2[gl[k]];
    }
    for (int t = tl; t < hw; t += GF_RPP) {
        const float4 dv = *reinterpret_cast<const float4*>(s_dy + (long)t * GF_CB + cx);
        const float4 hv = *reinterpret_cast<const float4*>(s_xh + (long)t * GF_CB + cx);
        const float dy[4] = {dv.x, dv.y, dv.z, dv.w}, xh[4] = {hv.x, hv.y, hv.z, hv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = rstd[k] * (gam[k] * dy[k] - m1[k] - xh[k] * m2[k]);
        float4* op = reinterpret_cast<float4*>(dx + ((long)b * hw + t) * dx_ld + c0 + cx);
        float4 r = make_float4(o[0], o[1], o[2], o[3]);
        if (accumulate) {
            const float4 old = *op;
            r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w;
        }
        *op = r;
    }
}

extern "C" int wd_gn_bwd_fused_supported(int hw, int c, int cpg) {
    return hw > 0 && c > 0 && cpg > 0 && c % GF_CB == 0 && GF_CB % cpg == 0 && GF_CB / cpg <= 8 && c % cpg == 0 && c / cpg <= 32 &&
           (long)hw * GF_CB * 8 + GF_RPP * 2 * GF_CB * 4 <= 150 * 1024;
}

extern "C" int wd_gn_bwd_fused(const float* x, int ld, const float* dz, int dz_ld, int dz_off, int batch, int hw, int c, int cpg,
                               const double* part, int nchunk_f, int part_cpg, const float* gamma, const float* beta, int c_off,
                               float eps, int silu, float* sums, float* dx, int dx_ld, int accumulate, void* stream) {
    if (!x || !dz || !part || !gamma || !beta || !sums || !dx || batch <= 0) return WD_EINVAL;
    if (!wd_gn_bwd_fused_supported(hw, c, cpg)) return WD_EINVAL;
    if (ld % 4 || dz_ld % 4 || dz_off % 4 || dx_ld % 4 || c_off % 4 || cpg % part_cpg) return WD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(dx)) & 15) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int smem = hw * GF_CB * 8 + GF_RPP * 2 * GF_CB * 4;
    static int attr_max = 0;
    if (smem > attr_max) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_bwd_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
            hipSuccess)
            return WD_ELAUNCH;
        attr_max = smem;
    }
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(gn_bwd_fused_kernel, dim3(c / GF_CB, batch), dim3(GF_NT), smem, st, x, ld, dz, dz_ld, dz_off, hw, c, cpg, part,
                       nchunk_f, part_cpg, gamma, beta, c_off, eps, silu, sums, dx, dx_ld, accumulate);
    return wd_check_launch();
}

extern "C" int wd_gn_bwd_nchunk(int hw) { return (hw + GB_TOK - 1) / GB_TOK; }

extern "C" int wd_gn_bwd_stats(const float* x, int ld, const float* dz, int dz_ld, int dz_off, int batch, int hw, int c,
                               int cpg, const double* part, int nchunk_f, int part_cpg, const float* gamma,
                               const float* beta, int c_off, float eps, int silu, float* sums, void* stream) {
    if (!x || !dz || !part || !gamma || !beta || !sums || batch <= 0 || hw <= 0 || c <= 0 || cpg <= 0) return WD_EINVAL;
    if (c % 4 || ld % 4 || dz_ld % 4 || dz_off % 4 || c_off % 4 || c % cpg || c / cpg > 32 || cpg % part_cpg) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nchunk = wd_gn_bwd_nchunk(hw);
    const int bx = 64 * ((c / 4 + 63) / 64);
    if (bx * 2 > 1024) return WD_EINVAL;
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(nchunk, batch), dim3(bx, 2), (64 + 4 * c) * sizeof(float), st, x, ld, dz,
                       dz_ld, dz_off, hw, c, cpg, part, nchunk_f, part_cpg, gamma, beta, c_off, eps, silu, nchunk, sums);
    return wd_check_launch();
}

extern "C" int wd_gn_bwd_apply(const float* x, int ld, const float* dz, int dz_ld, int dz_off, int batch, int hw, int c,
                               int cpg, const double* part, int nchunk_f, int part_cpg, const float* gamma,
                               const float* beta, int c_off, float eps, int silu, const float* sums, float* dx, int dx_ld,
                               int accumulate, void* stream) {
    if (!x || !dz || !part || !gamma || !beta || !sums || !dx || batch <= 0 || hw <= 0) return WD_EINVAL;
    if (c % 4 || ld % 4 || dz_ld % 4 || dz_off % 4 || dx_ld % 4 || c_off % 4 || c % cpg || c / cpg > 32 || cpg % part_cpg)
        return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((hw + GA_TOK - 1) / GA_TOK, batch), dim3(256), 0, st, x, ld, dz, dz_ld,
                       dz_off, hw, c, cpg, part, nchunk_f, part_cpg, gamma, beta, c_off, eps, silu, sums, wd_gn_bwd_nchunk(hw),
                       dx, dx_ld, accumulate);
    return wd_check_launch();
}

extern "C" int wd_layernorm_bwd_nblk(int rows) { return (rows + LB_ROWS - 1) / LB_ROWS; }

extern "C" int wd_layernorm_bwd(const float* x, int ld, const float* dy, int dy_ld, int rows, int c, const float* gamma,
                                float eps, float* dx, int dx_ld, int accumulate, float* colpart, void* stream) {
    if (!x || !dy || !gamma || !dx || !colpart || rows <= 0 || c <= 0) return WD_EINVAL;
    if (c % 4 || ld % 4 || dy_ld % 4 || dx_ld % 4 || c > 64 * 4 * LB_MAX4) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
if (c <= 512)
            hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(wd_layernorm_bwd_nblk(rows)), dim3(256), 4 * c * 2 * sizeof(float), st, x,
                       ld, dy, dy_ld, rows, c, gamma, eps, dx, dx_ld, accumulate, colpart);
    else
            hipLaunchKernelGGL(layernorm_bwd_kernel<LB_MAX4>, dim3(wd_layernorm_bwd_nblk(rows)), dim3(256), 4 * c * 2 * sizeof(float), st, x,
                       ld, dy, dy_ld, rows, c, gamma, eps, dx, dx_ld, accumulate, colpart);
    return wd_check_launch();
}

// Same contract as attn_bwd_small_kernel with more parallelism per token: a (token, head) pair is shared by FOUR lanes that
// each own a quarter of the head's channels (partial dot products combined with two DPP quad exchanges), so a workgroup
// covers 256 / (4 heads) tokens (16 at 4 heads) and a sample is spread over 4x as many workgroups - the 64-token form ran
// one workgroup per CU, one wave per SIMD, and was latency-bound end to end.  The column phase (dK, dV) uses every thread
// of the block (blockDim = inner rounded up to a wave when inner > 256) and reads its coefficients as float4.
// Requires d % 16 == 0 and 256 % (4 heads) == 0.
static __global__ void __launch_bounds__(512) attn_bwd_q4_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                          const float* __restrict__ v, int ldv, const float* __restrict__ dout,
                                                          int ldo, int heads, int nq, int nk, int d, float scale,
                                                          float* __restrict__ dq, int lddq, float* __restrict__ dkv_part,
                                                          int tpw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int inner = heads * d;
    float* s_k = reinterpret_cast<float*>(smem);   // [nk][inner]
    float* s_v = s_k + nk * inner;                 // [nk][inner]
    float* s_ds = s_v + nk * inner;                // [tpw][heads][NKB]
    float* s_p = s_ds + tpw * heads * NKB;         // [tpw][heads][NKB]
    const int b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
    const int i4 = inner >> 2;
    const int tok0 = blockIdx.x * tpw, ntok = min(tpw, nq - tok0);
    for (int e = tid; e < nk * i4; e += nthr) {
        const int j = e / i4, c = (e - j * i4) * 4;
        *reinterpret_cast<float4*>(s_k + j * inner + c) = *reinterpret_cast<const float4*>(k + ((long)b * nk + j) * ldk + c);
        *reinterpret_cast<float4*>(s_v + j * inner + c) = *reinterpret_cast<const float4*>(v + ((long)b * nk + j) * ldv + c);
    }
    __syncthreads();
    if (tid < 256) {
        const int qt = tid & 3, h = (tid >> 2) % heads, tl = tid / (4 * heads);
        const bool act = tl < ntok;
        const int dq4 = d >> 4;                    // float4 per quarter
        const int coff = h * d + qt * (d >> 2);    // first channel of this lane
        const long row = (long)b * nq + tok0 + (act ? tl : 0);
        float p[NKB], dp[NKB];
#pragma unroll
        for (int j = 0; j < NKB; ++j) p[j] = dp[j] = 0.f;
        const float* qr = q + row * ldq + coff;
        const float* dor = dout + row * ldo + coff;
#pragma unroll 1
        for (int c = 0; c < dq4; ++c) {
            const float4 qv = *reinterpret_cast<const float4*>(qr + c * 4);
            const float4 gv = *reinterpret_cast<const float4*>(dor + c * 4);
#pragma unroll
            for (int j = 0; j < NKB; ++j)
                if (j < nk) {
                    const float4 kv = *reinterpret_cast<const float4*>(s_k + j * inner + coff + c * 4);
                    const float4 vv = *reinterpret_cast<const float4*>(s_v + j * inner + coff + c * 4);
                    p[j] += qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w;
                    dp[j] += gv.x * vv.x + gv.y * vv.y + gv.z * vv.z + gv.w * vv.w;
                }
        }
        // the four quarter lanes of a (token, head) are one DPP quad: after two exchanges every lane holds the full sums
        // (added in the same order in all four lanes, so they agree bit for bit)
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) {
                float a0 = p[j], a1 = dp[j];
                a0 += wd_dpp<0xB1>(a0);
                a1 += wd_dpp<0xB1>(a1);
                a0 += wd_dpp<0x4E>(a0);
                a1 += wd_dpp<0x4E>(a1);
                p[j] = a0;
                dp[j] = a1;
            }
        float mx = -3.4e38f;
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) {
                p[j] *= scale;
                mx = fmaxf(mx, p[j]);
            }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) {
                p[j] = expf(p[j] - mx);
                sum += p[j];
            }
        const float inv = 1.f / sum;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) {
                p[j] *= inv;
                dot += p[j] * dp[j];
            }
        float ds[NKB];
#pragma unroll
        for (int j = 0; j < NKB; ++j) ds[j] = j < nk ? p[j] * (dp[j] - dot) * scale : 0.f;
        if (act) {
#pragma unroll 1
            for (int c = 0; c < dq4; ++c) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < NKB; ++j)
                    if (j < nk) {
                        const float4 kv = *reinterpret_cast<const float4*>(s_k + j * inner + coff + c * 4);
                        o.x += ds[j] * kv.x; o.y += ds[j] * kv.y; o.z += ds[j] * kv.z; o.w += ds[j] * kv.w;
                    }
                *reinterpret_cast<float4*>(dq + row * lddq + coff + c * 4) = o;
            }
        }
        if (qt == 0 && tl < tpw) {
#pragma unroll
            for (int j4 = 0; j4 < NKB / 4; ++j4) {
                const int j = j4 * 4;
                const float z = act ? 1.f : 0.f;
                *reinterpret_cast<float4*>(s_ds + (tl * heads + h) * NKB + j) =
                    make_float4(z * ds[j], z * ds[j + 1], z * ds[j + 2], z * ds[j + 3]);
                *reinterpret_cast<float4*>(s_p + (tl * heads + h) * NKB + j) =
                    make_float4(j < nk ? z * p[j] : 0.f, j + 1 < nk ? z * p[j + 1] : 0.f, j + 2 < nk ? z * p[j + 2] : 0.f,
                                j + 3 < nk ? z * p[j + 3] : 0.f);
            }
        }
    }
    __syncthreads();
    float* outp = dkv_part + (((long)b * gridDim.x + blockIdx.x) * 2) * nk * inner;
    const long row0 = (long)b * nq + tok0;
    for (int col = tid; col < inner; col += nthr) {
        const int hh = col / d;
        float ak[NKB], av[NKB];
#pragma unroll
        for (int j = 0; j < NKB; ++j) ak[j] = av[j] = 0.f;
        // two tokens per round: with more the compiler hoists every coefficient read of the round (255 VGPRs, one workgroup
        // per CU); 82 VGPRs keep four workgroups resident and their loads overlap instead
#pragma unroll 1
        for (int t0 = 0; t0 < ntok; t0 += 2) {
            float qv[2], gv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool okt = t0 + u < ntok;
                qv[u] = okt ? q[(row0 + t0 + u) * ldq + col] : 0.f;
                gv[u] = okt ? dout[(row0 + t0 + u) * ldo + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = t0 + u < ntok ? t0 + u : 0;  // (padding tokens multiply by 0)
                const float* cds = s_ds + (t * heads + hh) * NKB;
                const float* cp = s_p + (t * heads + hh) * NKB;
#pragma unroll
                for (int j4 = 0; j4 < NKB / 4; ++j4) {
                    if (j4 * 4 < nk) {
                        const float4 a4 = *reinterpret_cast<const float4*>(cds + j4 * 4);
                        const float4 p4 = *reinterpret_cast<const float4*>(cp + j4 * 4);
                        ak[j4 * 4 + 0] += a4.x * qv[u]; ak[j4 * 4 + 1] += a4.y * qv[u];
                        ak[j4 * 4 + 2] += a4.z * qv[u]; ak[j4 * 4 + 3] += a4.w * qv[u];
                        av[j4 * 4 + 0] += p4.x * gv[u]; av[j4 * 4 + 1] += p4.y * gv[u];
                        av[j4 * 4 + 2] += p4.z * gv[u]; av[j4 * 4 + 3] += p4.w * gv[u];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NKB; ++j)
            if (j < nk) {
                outp[((long)j * 2 + 0) * inner + col] = ak[j];
                outp[((long)j * 2 + 1) * inner + col] = av[j];
            }
    }
}

static bool attn_bwd_q4_ok(int heads, int d) { return d % 16 == 0 && heads <= 64 && 256 % (4 * heads) == 0; }

static int attn_bwd_tpw(int heads, int nq, int nk, int d) {
    const int inner = heads * d;
    int tpw = attn_bwd_q4_ok(heads, d) ? 256 / (4 * heads) : 256 / heads;
    if (nq < tpw) tpw = nq;
    const long floats = (long)2 * nk * inner + (long)2 * tpw * heads * NKB;
    return floats * 4 <= 150 * 1024 ? tpw : 0;
}

extern "C" int wd_attention_bwd_small_nwg(int heads, int nq, int nk, int d) {
    if (heads <= 0 || heads > 256 || nq <= 0 || nk <= 0 || d <= 0) return 0;
    const int tpw = attn_bwd_tpw(heads, nq, nk, d);
    return tpw > 0 ? (nq + tpw - 1) / tpw : 0;
}

extern "C" int wd_attention_bwd_small(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                      const float* dout, int ldo, int batch, int heads, int nq, int nk, int d, float scale,
                                      float* dq, int lddq, float* dkv_part, int* nwg_out, void* stream) {
    if (!q || !k || !v || !dout || !dq || !dkv_part) return WD_EINVAL;
    if (batch <= 0 || heads <= 0 || heads > 256 || nq <= 0 || nk <= 0 || nk > NKB || d <= 0 || d % 4) return WD_EINVAL;
    if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4 || lddq % 4) return WD_EINVAL;
    const int inner = heads * d, tpw = attn_bwd_tpw(heads, nq, nk, d);
    if (tpw <= 0) return WD_EINVAL;
    const size_t smem = ((size_t)2 * nk * inner + (size_t)2 * tpw * heads * NKB) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool q4 = attn_bwd_q4_ok(heads, d);
    static size_t set = 64 * 1024, set4 = 64 * 1024;
    if (smem > (q4 ? set4 : set)) {
        if (hipFuncSetAttribute(q4 ? reinterpret_cast<const void*>(&attn_bwd_q4_kernel)
                                   : reinterpret_cast<const void*>(&attn_bwd_small_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return WD_ELAUNCH;
        (q4 ? set4 : set) = smem;
    }
    const int nwg = (nq + tpw - 1) / tpw;
    if (nwg_out) *nwg_out = nwg;
    WdLaunchScope scope(WD_CLS_ATTN, st);
    if (q4) {
        int nthr = inner > 256 ? (inner + 63) / 64 * 64 : 256;
        if (nthr > 512) nthr = 512;
        hipLaunchKernelGGL(attn_bwd_q4_kernel, dim3(nwg, batch), dim3(nthr), smem, st, q, ldq, k, ldk, v, ldv, dout, ldo, heads, nq,
                           nk, d, scale, dq, lddq, dkv_part, tpw);
    } else {
        hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(nwg, batch), dim3(256), smem, st, q, ldq, k, ldk, v, ldv, dout, ldo, heads,
                           nq, nk, d, scale, dq, lddq, dkv_part, tpw);
    }
    return wd_check_launch();
}

extern "C" int wd_geglu_fwd(const float* u, int ld, int64_t rows, int inner, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld,
                            void* stream) {
    if (!u || !out_hi || rows <= 0 || inner <= 0 || inner % 4 || ld % 4 || out_ld % 4) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(rows * (inner / 4))), dim3(256), 0, st, u, ld, (long)rows, inner,
                       out_hi, out_lo, out_ld);
    return wd_check_launch();
}
extern "C" int wd_geglu_bwd(const float* u, int ld, const float* dh, int dh_ld, int64_t rows, int inner, float* du, int du_ld,
                            void* stream) {
    if (!u || !dh || !du || rows <= 0 || inner <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(rows * inner)), dim3(256), 0, st, u, ld, dh, dh_ld, (long)rows, inner,
                       du, du_ld);
    return wd_check_launch();
}
extern "C" int wd_silu_bwd(const float* pre, const float* dact, int64_t n, float* dpre, void* stream) {
    if (!pre || !dact || !dpre || n <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, pre, dact, (long)n, dpre);
    return wd_check_launch();
}
extern "C" int wd_pool2x2_sum(const float* in, int batch, int h, int w, int c, float* out, void* stream) {
    if (!in || !out || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 4) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(pool2x2_sum_kernel, dim3(grid_for((long)batch * h * w * (c / 4))), dim3(256), 0, st, in, batch, h, w, c,
                       out);
    return wd_check_launch();
}
extern "C" int wd_embedding_bwd(const void* ids, int ids_are_i64, int rows, const float* d, int ld, int vocab, int c,
                                float* dtable, int accumulate, void* stream) {
    if (!ids || !d || !dtable || rows <= 0 || vocab <= 0 || c <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for((long)vocab * c)), dim3(256), 0, st, ids, ids_are_i64, rows, d, ld,
                       vocab, c, dtable, accumulate);
    return wd_check_launch();
}

extern "C" int wd_add(float* dst, const float* src, int64_t n, void* stream) {
    if (!dst || !src || n <= 0 || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, st, dst, src, (long)(n / 4), (long)n);
    return wd_check_launch();
}

extern "C" int wd_permute_dw(const float* packed, int ld, int n, int c, int ntaps, float* out, void* stream) {
    if (!packed || !out || n <= 0 || c <= 0 || ntaps <= 0 || ld < c * ntaps) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(permute_dw_kernel, dim3(grid_for((long)n * c * ntaps)), dim3(256), 0, st, packed, ld, n, c, ntaps, out);
    return wd_check_launch();
}

extern "C" int64_t wd_attention_bwd_scratch_floats(int batch, int heads, int nq, int nk) {
    return (int64_t)2 * batch * heads * nq * nk;
}

extern "C" int wd_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* dout,
                                int ldo, int batch, int heads, int nq, int nk, int d, float scale, float* dq, int lddq,
                                float* dk, int lddk, float* dv, int lddv, float* scratch, int64_t scratch_floats,
                                void* stream) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || !scratch) return WD_EINVAL;
    if (batch <= 0 || heads <= 0 || nq <= 0 || nk <= 0 || nk > 64 * AG_NJ || d <= 0 || d % 4) return WD_EINVAL;
    if (ldk % 4 || ldv % 4) return WD_EINVAL;
    if (scratch_floats < wd_attention_bwd_scratch_floats(batch, heads, nq, nk)) return WD_EINVAL;
    const size_t smem = (size_t)4 * (2 * d + nk) * sizeof(float);
    if (smem > 64 * 1024) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* pbuf = scratch;
    float* dsbuf = scratch + (int64_t)batch * heads * nq * nk;
    WdLaunchScope scope(WD_CLS_ATTN, st);
    const long rows = (long)batch * heads * nq, cols = (long)batch * heads * nk;
    hipLaunchKernelGGL(attn_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), smem, st, q, ldq, k, ldk, v, ldv, dout,
                       ldo, heads, nq, nk, d, scale, dq, lddq, pbuf, dsbuf, rows);
    hipLaunchKernelGGL(attn_bwd_cols_kernel, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, st, q, ldq, dout, ldo, heads, nq, nk,
                       d, pbuf, dsbuf, dk, lddk, dv, lddv, cols);
    return wd_check_launch();
}

extern "C" int wd_dout_prep_rows(void) { return 64; }

extern "C" int wd_dout_prep(const float* d, int ld, int m, int n, int npad, int mpad, wd_bf16* pl_hi, wd_bf16* pl_lo,
                            wd_bf16* t_hi, wd_bf16* t_lo, float* colpart, void* stream) {
    if (!d || m <= 0 || n <= 0 || npad < n || mpad < m || (mpad & 3) || (npad & 3) || (ld & 3)) return WD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d) & 15) || (!pl_hi && !t_hi && !colpart)) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    const int ncols = pl_hi ? npad : n;
    hipLaunchKernelGGL(dout_prep_kernel<false>, dim3((mpad + 63) / 64, (ncols + 63) / 64), dim3(256), 0, st, d, ld, m, n, npad, mpad,
                       pl_hi, pl_lo, t_hi, t_lo, colpart, nullptr, 0, 0);
    return wd_check_launch();
}

extern "C" int wd_dout_prep_geglu(const float* u, int u_ld, const float* dh, int dh_ld, int m, int inner, int mpad, wd_bf16* pl_hi,
                                  wd_bf16* pl_lo, wd_bf16* t_hi, wd_bf16* t_lo, float* colpart, void* stream) {
    if (!u || !dh || m <= 0 || inner <= 0 || inner % 64 || mpad < m || (mpad & 3) || u_ld < 2 * inner || (u_ld & 3) || dh_ld < inner ||
        (dh_ld & 3))
        return WD_EINVAL;
    if (((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(dh)) & 15) || (!pl_hi && !t_hi && !colpart)) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    const int n = 2 * inner;
    hipLaunchKernelGGL(dout_prep_kernel<true>, dim3((mpad + 63) / 64, n / 64), dim3(256), 0, st, u, u_ld, m, n, n, mpad, pl_hi, pl_lo,
                       t_hi, t_lo, colpart, dh, dh_ld, inner);
    return wd_check_launch();
}

extern "C" int wd_colsum_finish(const float* part, int nblk, int c, int nseg, float* out, int out_ld, int accumulate,
                                float scale, void* stream) {
    if (!part || !out || nblk <= 0 || c <= 0 || nseg <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(colsum_stage2, dim3((c + 15) / 16, nseg), dim3(256), 0, st, part, nblk, c, nseg, out, out_ld, accumulate,
                       scale);
    return wd_check_launch();
}

extern "C" int wd_colsum_entry_bytes() { return (int)sizeof(ColRef); }

extern "C" int wd_colsum_finish_multi(const void* table, int n, int max_c, void* stream) {
    if (!table || n <= 0 || max_c <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((max_c + 15) / 16, n), dim3(256), 0, st, reinterpret_cast<const ColRef*>(table));
    return wd_check_launch();
}
