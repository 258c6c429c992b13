// Epilogue shared by every tap-gather GEMM kernel of libwdiff_hip.so (wd_gemm.hip, wd_gemmw.hip): the fp32 LDS image of an
// output tile -> bias / FiLM row vector / residual / activation / operand planes / fused GroupNorm statistics, or the split-K
// partial store.  Replaces the elementwise tails of reference unet.py:660-669 (FiLM add, residual) and :128-136 (GEGLU).
#pragma once
#include "wd_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

// ---- epilogue shared by all kernels.  The accumulators of all waves go through an fp32 LDS image of the output
// tile (which also sums the two k-halves of the 8-wave variants); wd_epilogue_from_image then reads float4s row-major
// and does bias / FiLM / residual / activation with 16-byte global loads and stores, and - when the consumer is a
// GroupNorm - reduces per-(sample, group) sum / sum of squares of the finished values in a fixed order, so that the
// statistics pass over the feature map (wd_gn_stats) disappears.
constexpr int WD_STAT_SCRATCH = 40 * 1024;  // LDS behind the epilogue image for the per-thread column sums
constexpr int WD_STAT_MAXNS = 2;          // samples per 128-row panel the fused statistics support (hw_out >= 64)

// WS: the tile's values are not in the LDS image but still spread over the split-K slabs of a.ws - the combine pass sums them
// here, row by row in the thread's own (row lane, column quad) pattern, instead of staging the sums through LDS first (the
// statistics scratch behind the image position is used as usual).  Vector path only; the caller checks.
template <int BM, int BN, int NT, bool WS = false>
__device__ __forceinline__ void wd_epilogue_from_image(const wd_gemm_args& a, float* ep, const int m0, const int n0,
                                                       const int tid, const int nslab = 1) {
    constexpr int LDE = BN + 4;  // row pitch in floats (16-byte aligned rows, bank-shifted)
    const bool geglu = a.act == WD_ACT_GEGLU;
    const int ocols = geglu ? BN / 2 : BN;  // output columns of this tile
    const int oc4 = ocols / 4;
    const int nout = geglu ? a.n / 2 : a.n;
    const int no0 = geglu ? n0 / 2 : n0;
    const bool stats = a.stat_part != nullptr;
    // 16-byte path needs aligned rows everywhere; otherwise (odd leading dimensions) one element at a time
    const bool vec = (((a.out_ld | a.rowvec_ld | a.resid_ld | a.out_pl_ld) & 3) == 0) &&
                     (((reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.rowvec) |
                        reinterpret_cast<uintptr_t>(a.resid) | reinterpret_cast<uintptr_t>(a.out_f32)) & 15) == 0) &&
                     (((reinterpret_cast<uintptr_t>(a.out_hi) | reinterpret_cast<uintptr_t>(a.out_lo)) & 7) == 0);
    if (!vec) {
        for (int i = tid; i < BM * ocols; i += NT) {
            const int row = i / ocols, c = i - row * ocols;
            const int m = m0 + row, no = no0 + c;
            if (m >= a.m || no >= nout) continue;
            float v;
            if (geglu) {
                const float x = ep[row * LDE + c] + (a.bias ? a.bias[n0 + c] : 0.f);
                const float g = ep[row * LDE + c + BN / 2] + (a.bias ? a.bias[n0 + c + BN / 2] : 0.f);
                v = x * wd_gelu_erf(g);
            } else {
                v = ep[row * LDE + c] + (a.bias ? a.bias[no] : 0.f);
            }
            if (a.rowvec) v += a.rowvec[(long)(m / a.hw_out) * a.rowvec_ld + no];
            if (a.resid) v += a.resid[(a.resid_rows ? (long)a.resid_rows[m] : (long)m) * a.resid_ld + no];
            if (a.act == WD_ACT_SILU) v = wd_silu(v);
            if (stats || a.ln_gamma) ep[row * LDE + c] = v;
            if (a.out_f32) a.out_f32[(long)m * a.out_ld + no] = v;
            if (a.out_hi && !a.ln_gamma) {
                uint32_t hb, lb;
                wd_split1(v, hb, lb);
                a.out_hi[(long)m * a.out_pl_ld + no] = (wd_bf16)hb;
                if (a.out_lo) a.out_lo[(long)m * a.out_pl_ld + no] = (wd_bf16)lb;
            }
        }
    } else {
        // thread -> (row lane, fixed column quad): coalesced rows, and per-thread column sums for the statistics
        const int nrl = NT / oc4;                 // row lanes
        const int rl = tid / oc4, c = (tid - rl * oc4) * 4;
        const int no = no0 + c;
        const int rps = BM > a.hw_out ? a.hw_out : BM;  // rows of one sample inside this panel
        // (with statistics rps is BM or a divisor of it - a power of two - so row / rps is a shift; the FiLM row of m needs m /
        // hw_out: a shift too for power-of-two images, a division otherwise.  A run-time integer division is ~25 instructions,
        // more than the rest of a row's work.)
        const int rps_sh = 31 - __clz(rps);
        const int hw_sh = (a.hw_out & (a.hw_out - 1)) == 0 ? 31 - __clz(a.hw_out) : -1;
        float* scr = ep + BM * LDE;               // statistics scratch [sample][row lane][BN][2]
        float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = ssum;
        int cur_s = 0;
        if (rl < nrl && no < nout) {
            float4 bx = make_float4(0, 0, 0, 0), bg = bx;
            if (a.bias) {
                bx = *reinterpret_cast<const float4*>(a.bias + (geglu ? n0 + c : no));
                if (geglu) bg = *reinterpret_cast<const float4*>(a.bias + n0 + c + BN / 2);
            }
            for (int row = rl; row < BM; row += nrl) {
                const int m = m0 + row;
                if (m >= a.m) break;
                float4 v;
                if constexpr (WS) {
                    const float* p = a.ws + (long)m * a.n + n0 + c;
                    const long total = (long)a.m * a.n;
                    v = *reinterpret_cast<const float4*>(p);
                    for (int sp = 1; sp < nslab; ++sp) {
                        const float4 q = *reinterpret_cast<const float4*>(p + (long)sp * total);
                        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                    }
                } else {
                    v = *reinterpret_cast<const float4*>(ep + row * LDE + c);
                }
                if (geglu) {  // columns [0, BN/2) of the tile are x, [BN/2, BN) their gates (weights packed that way)
                    const float4 g = *reinterpret_cast<const float4*>(ep + row * LDE + c + BN / 2);
                    v.x = (v.x + bx.x) * wd_gelu_erf(g.x + bg.x);
                    v.y = (v.y + bx.y) * wd_gelu_erf(g.y + bg.y);
                    v.z = (v.z + bx.z) * wd_gelu_erf(g.z + bg.z);
                    v.w = (v.w + bx.w) * wd_gelu_erf(g.w + bg.w);
                } else {
                    v.x += bx.x; v.y += bx.y; v.z += bx.z; v.w += bx.w;
                }
                if (a.rowvec) {
                    const int bi = hw_sh >= 0 ? m >> hw_sh : m / a.hw_out;
                    const float4 q = *reinterpret_cast<const float4*>(a.rowvec + (long)bi * a.rowvec_ld + no);
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                if (a.resid) {
                    const long rr = a.resid_rows ? (long)a.resid_rows[m] : (long)m;
                    const float4 q = *reinterpret_cast<const float4*>(a.resid + rr * a.resid_ld + no);
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                if (a.act == WD_ACT_SILU) {
                    v.x = wd_silu(v.x); v.y = wd_silu(v.y); v.z = wd_silu(v.z); v.w = wd_silu(v.w);
                }
                if (stats) {
                    const int sidx2 = row >> rps_sh;
                    if (sidx2 != cur_s) {  // rows ascend: flush the finished sample's column sums
                        float* o = scr + ((cur_s * nrl + rl) * BN + c) * 2;
                        *reinterpret_cast<float4*>(o) = make_float4(ssum.x, ssq.x, ssum.y, ssq.y);
                        *reinterpret_cast<float4*>(o + 4) = make_float4(ssum.z, ssq.z, ssum.w, ssq.w);
                        ssum = make_float4(0.f, 0.f, 0.f, 0.f);
                        ssq = ssum;
                        cur_s = sidx2;
                    }
                    ssum.x += v.x; ssum.y += v.y; ssum.z += v.z; ssum.w += v.w;
                    ssq.x += v.x * v.x; ssq.y += v.y * v.y; ssq.z += v.z * v.z; ssq.w += v.w * v.w;
                }
                if (a.ln_gamma) {  // the row LayerNorm below reads the finished values back from the image
                    if constexpr (!WS) *reinterpret_cast<float4*>(ep + row * LDE + c) = v;
                }
                if (no + 3 < nout) {
                    if (a.out_f32) *reinterpret_cast<float4*>(a.out_f32 + (long)m * a.out_ld + no) = v;
                    if (a.out_hi && !a.ln_gamma) {
                        uint2 hh, ll;
                        wd_split4(v, hh, ll);
                        *reinterpret_cast<uint2*>(a.out_hi + (long)m * a.out_pl_ld + no) = hh;
                        if (a.out_lo) *reinterpret_cast<uint2*>(a.out_lo + (long)m * a.out_pl_ld + no) = ll;
                    }
                } else {  // ragged right edge (n not a multiple of 4 columns inside this float4)
                    const float e[4] = {v.x, v.y, v.z, v.w};
                    for (int j = 0; j < 4 && no + j < nout; ++j) {
                        if (a.out_f32) a.out_f32[(long)m * a.out_ld + no + j] = e[j];
                        if (a.out_hi) {
                            uint32_t hb, lb;
                            wd_split1(e[j], hb, lb);
                            a.out_hi[(long)m * a.out_pl_ld + no + j] = (wd_bf16)hb;
                            if (a.out_lo) a.out_lo[(long)m * a.out_pl_ld + no + j] = (wd_bf16)lb;
                        }
                    }
                }
            }
        }
        if (stats) {
            // flush the running sample and zero-fill the (sample, row lane) slots this thread never reached
            const int ns = BM > a.hw_out ? BM / a.hw_out : 1;
            if (rl < nrl && c < BN) {
                for (int s2 = cur_s; s2 < ns; ++s2) {
                    float* o = scr + ((s2 * nrl + rl) * BN + c) * 2;
                    const bool cur = (s2 == cur_s);
                    *reinterpret_cast<float4*>(o) = cur ? make_float4(ssum.x, ssq.x, ssum.y, ssq.y) : make_float4(0, 0, 0, 0);
                    *reinterpret_cast<float4*>(o + 4) = cur ? make_float4(ssum.z, ssq.z, ssum.w, ssq.w) : make_float4(0, 0, 0, 0);
                }
            }
        }
    }
    if (a.ln_gamma) {
        // ---- LayerNorm of every finished row -> the operand planes of the next GEMM (nn.LayerNorm of the transformer block that
        // consumes this tensor, unetPhosc.py:241-246): the tile holds whole rows (BN == n, the host checked), one wave per row,
        // two-pass statistics in registers as wd_layernorm does.  Saves the wd_layernorm launch and its pass over the tensor.
        __syncthreads();
        constexpr int NW = NT / 64;
        const int lane = tid & 63, wv = tid >> 6;
        const bool has0 = lane * 4 < BN, has1 = (64 + lane) * 4 < BN;
        // (a lane's columns are the same for every row: the affine vectors are loaded once, not once per row)
        float4 gq[2], bq[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            gq[h] = bq[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h ? has1 : has0) {
                gq[h] = *reinterpret_cast<const float4*>(a.ln_gamma + (h * 64 + lane) * 4);
                bq[h] = *reinterpret_cast<const float4*>(a.ln_beta + (h * 64 + lane) * 4);
            }
        }
        for (int row = wv; row < BM; row += NW) {
            const int m = m0 + row;
            if (m >= a.m) break;
            const float* rp = ep + row * LDE;
            float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
            if (has0) x0 = *reinterpret_cast<const float4*>(rp + lane * 4);
            if (has1) x1 = *reinterpret_cast<const float4*>(rp + (64 + lane) * 4);
            const float mean = wd_wave_sum((x0.x + x0.y) + (x0.z + x0.w) + (x1.x + x1.y) + (x1.z + x1.w)) * (1.0f / BN);
            float d = 0.f;
            if (has0) d += (x0.x - mean) * (x0.x - mean) + (x0.y - mean) * (x0.y - mean) + (x0.z - mean) * (x0.z - mean) + (x0.w - mean) * (x0.w - mean);
            if (has1) d += (x1.x - mean) * (x1.x - mean) + (x1.y - mean) * (x1.y - mean) + (x1.z - mean) * (x1.z - mean) + (x1.w - mean) * (x1.w - mean);
            const float rstd = rsqrtf(wd_wave_sum(d) * (1.0f / BN) + a.ln_eps);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!(h ? has1 : has0)) continue;
                const int c = (h * 64 + lane) * 4;
                const float4 x = h ? x1 : x0;
                const float4 g = gq[h], b = bq[h];
                float4 y;
                y.x = (x.x - mean) * rstd * g.x + b.x; y.y = (x.y - mean) * rstd * g.y + b.y;
                y.z = (x.z - mean) * rstd * g.z + b.z; y.w = (x.w - mean) * rstd * g.w + b.w;
                uint2 hh, ll;
                wd_split4(y, hh, ll);
                *reinterpret_cast<uint2*>(a.out_hi + (long)m * a.out_pl_ld + c) = hh;
                if (a.out_lo) *reinterpret_cast<uint2*>(a.out_lo + (long)m * a.out_pl_ld + c) = ll;
            }
        }
    }
    if (!stats) return;
    // ---- GroupNorm partial statistics of the finished tile: (sample in panel, group) = fixed-order sum over the
    // row lanes and the group's columns of the per-thread column sums above (host guarantees the vector path, BN % cpg
    // == 0, n % cpg == 0, and that BM and hw_out divide one another).
    __syncthreads();
    {
        const int cpg = a.stat_cpg;
        const int ns = BM > a.hw_out ? BM / a.hw_out : 1;
        const int rps = BM > a.hw_out ? a.hw_out : BM;
        const int ngt = BN / cpg;
        const int nrl = NT / oc4;
        const float* scr = ep + BM * LDE;
        const int nchunk = a.hw_out > BM ? a.hw_out / BM : 1;
        const int ngs = a.n / cpg;  // groups per sample in the statistics array of this tensor
        // level 1: (sample, group, row lane) -> sum over the group's columns, written back over the lane's first slot
        float* scw = ep + BM * LDE;
        for (int it = tid; it < ns * ngt * nrl; it += NT) {
            const int l = it % nrl, g = (it / nrl) % ngt, sidx2 = it / (nrl * ngt);
            const float* p = scr + ((sidx2 * nrl + l) * BN + g * cpg) * 2;
            float su = 0.f, sq = 0.f;
            for (int cc = 0; cc < cpg; ++cc) {
                su += p[2 * cc];
                sq += p[2 * cc + 1];
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): every read of this group's span precedes the write-back
            scw[((sidx2 * nrl + l) * BN + g * cpg) * 2] = su;
            scw[((sidx2 * nrl + l) * BN + g * cpg) * 2 + 1] = sq;
        }
        __syncthreads();
        // level 2: (sample, group) -> sum over the row lanes in fixed order
        for (int it = tid; it < ns * ngt; it += NT) {
            const int g = it % ngt, sidx2 = it / ngt;
            const int mrow = m0 + sidx2 * rps;
            if (mrow >= a.m || n0 + g * cpg >= a.n) continue;
            double su = 0.0, sq = 0.0;
            for (int l = 0; l < nrl; ++l) {
                const float* p = scr + ((sidx2 * nrl + l) * BN + g * cpg) * 2;
                su += (double)p[0];
                sq += (double)p[1];
            }
            const int b = mrow / a.hw_out;
            const int chunk = (mrow - b * a.hw_out) / BM;
            double* o = a.stat_part + (((long)b * nchunk + chunk) * ngs + n0 / cpg + g) * 2;
            o[0] = su;
            o[1] = sq;
        }
    }
}

// after the fp32 image of the tile is complete: split-K partial store, or the fused epilogue
template <int BM, int BN, int NT>
__device__ __forceinline__ void wd_epilogue_tail(const wd_gemm_args& a, float* ep, const int m0, const int n0, const int tid,
                                                 const int sidx) {
    constexpr int LDE = BN + 4;
    if (a.ksplit > 1) {  // raw partial sums of this K slice -> ws[sidx][m][n]; wd_gemm_reduce applies the epilogue
        float* ws = a.ws + (long)sidx * a.m * a.n;
        const bool v4 = (a.n & 3) == 0;
        for (int i = tid; i < BM * (BN / 4); i += NT) {
            const int row = i / (BN / 4), c = (i - row * (BN / 4)) * 4;
            const int m = m0 + row, n = n0 + c;
            if (m >= a.m || n >= a.n) continue;
            const float4 v = *reinterpret_cast<const float4*>(ep + row * LDE + c);
            if (v4) {
                *reinterpret_cast<float4*>(ws + (long)m * a.n + n) = v;
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
                for (int j = 0; j < 4 && n + j < a.n; ++j) ws[(long)m * a.n + n + j] = e[j];
            }
        }
        if (!a.tickets) return;  // a separate wd_gemm_reduce launch combines the slabs
        // ---- in-launch combine (cdna_hip_programming.md section 5 "In-launch split-K reduction", Guideline 16): every slice
        // publishes its slab (all stores drained, barrier, ONE agent-scope release, then a relaxed agent-scope ticket add); the
        // workgroup that draws the last ticket acquires once and sums ALL slabs from the workspace in ascending slice order -
        // the result does not depend on which slice arrived last - inside the ordinary epilogue (bias, FiLM, residual,
        // statistics, planes).  No spinning anywhere: a workgroup that is not last simply ends.  The ticket word is zero
        // before the launch (zero-initialised by the owner, reset here by the last arriver).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* s_flag = reinterpret_cast<int*>(ep + BM * LDE);  // first word of the statistics scratch (idle until the epilogue)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int* cnt = a.tickets + (m0 / BM) * ((a.n + BN - 1) / BN) + n0 / BN;
            const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == a.ksplit - 1;
            if (last) {
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *s_flag = last;
        }
        __syncthreads();
        const int is_last = *s_flag;
        __syncthreads();  // (the epilogue reuses the scratch the flag sits in)
        if (!is_last) return;
        wd_gemm_args b = a;
        b.ksplit = 1;
        wd_epilogue_from_image<BM, BN, NT, true>(b, ep, m0, n0, tid, a.ksplit);
        return;
    }
    wd_epilogue_from_image<BM, BN, NT>(a, ep, m0, n0, tid);
}

// Combine + the consumer's GroupNorm for one 64-row x 40-column tile whose rows are ONE sample (hw_out == 64, wd_gemm_args::gn_*):
// the slabs summed in ascending order, bias / FiLM row / residual, the fp32 result, the (sample, group) statistics - written to
// stat_part exactly as the ordinary combine writes them (same summation order) - and SiLU?(GroupNorm(result)) as operand planes.
// A thread's (at most three) rows stay in its registers between the statistics and the normalisation, so the wd_gn_apply
// launch and its pass over the tensor disappear (ten of them on the 4x16 level of the base UNet).
// WS = true: the values are summed from the split-K slabs of a.ws (the combine launch, 64 x 40 tiles of 256 threads); WS = false:
// they are the finished fp32 image `img` (row pitch BN + 4) of a kernel that owns all of K (wd_gemmk_kernel, 64 x 80, 512 threads).
// `ep`: scratch of NRL * BN * 2 floats + BN / stat_cpg doubles, not overlapping img.
template <int BN, int NT, bool WS>
__device__ __forceinline__ void wd_gn_tile(const wd_gemm_args& a, const float* img, float* ep, const int m0, const int n0, const int tid) {
    constexpr int BM = 64, Q = BN / 4, NRL = NT / Q, KEEP = (BM + NRL - 1) / NRL;
    const int rl = tid / Q, c = (tid - rl * Q) * 4;
    const bool live = rl < NRL;
    const int no = n0 + c;
    const long total = (long)a.m * a.n;
    float4 v[KEEP];
    bool ok[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const int row = rl + k * NRL;
        ok[k] = live && row < BM && m0 + row < a.m;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok[k]) v[k] = WS ? *reinterpret_cast<const float4*>(a.ws + (long)(m0 + row) * a.n + no)
                             : *reinterpret_cast<const float4*>(img + row * (BN + 4) + c);
    }
    for (int sp = 1; WS && sp < a.ksplit; ++sp) {
        float4 q[KEEP];
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[k]) q[k] = *reinterpret_cast<const float4*>(a.ws + (long)sp * total + (long)(m0 + rl + k * NRL) * a.n + no);
        }
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            v[k].x += q[k].x; v[k].y += q[k].y; v[k].z += q[k].z; v[k].w += q[k].w;
        }
    }
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f), rv = bx;
    if (live && a.bias) bx = *reinterpret_cast<const float4*>(a.bias + no);
    if (live && a.rowvec) rv = *reinterpret_cast<const float4*>(a.rowvec + (long)(m0 / a.hw_out) * a.rowvec_ld + no);
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = ssum;
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        if (!ok[k]) continue;
        const long m = m0 + rl + k * NRL;
        v[k].x += bx.x; v[k].y += bx.y; v[k].z += bx.z; v[k].w += bx.w;
        if (a.rowvec) {
            v[k].x += rv.x; v[k].y += rv.y; v[k].z += rv.z; v[k].w += rv.w;
        }
        if (a.resid) {
            const float4 q = *reinterpret_cast<const float4*>(a.resid + m * a.resid_ld + no);
            v[k].x += q.x; v[k].y += q.y; v[k].z += q.z; v[k].w += q.w;
        }
        ssum.x += v[k].x; ssum.y += v[k].y; ssum.z += v[k].z; ssum.w += v[k].w;
        ssq.x += v[k].x * v[k].x; ssq.y += v[k].y * v[k].y; ssq.z += v[k].z * v[k].z; ssq.w += v[k].w * v[k].w;
        if (a.out_f32) *reinterpret_cast<float4*>(a.out_f32 + m * a.out_ld + no) = v[k];
    }
    // per-thread column sums -> (group, row lane) -> (group): the order of the ordinary combine
    float* scr = ep;                                                  // [NRL][BN][2]
    double* gs = reinterpret_cast<double*>(ep + NRL * BN * 2);        // [BN / stat_cpg][2]
    if (live) {
        float* o = scr + (rl * BN + c) * 2;
        *reinterpret_cast<float4*>(o) = make_float4(ssum.x, ssq.x, ssum.y, ssq.y);
        *reinterpret_cast<float4*>(o + 4) = make_float4(ssum.z, ssq.z, ssum.w, ssq.w);
    }
    __syncthreads();
    const int cps = a.stat_cpg, ngt = BN / cps;
    for (int it = tid; it < ngt * NRL; it += NT) {
        const int l = it % NRL, g = it / NRL;
        const float* p = scr + (l * BN + g * cps) * 2;
        float su = 0.f, sq = 0.f;
        for (int cc = 0; cc < cps; ++cc) {
            su += p[2 * cc];
            sq += p[2 * cc + 1];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): every read of this group's span precedes the write-back
        scr[(l * BN + g * cps) * 2] = su;
        scr[(l * BN + g * cps) * 2 + 1] = sq;
    }
    __syncthreads();
    if (tid < ngt && n0 + tid * cps < a.n) {
        double su = 0.0, sq = 0.0;
        for (int l = 0; l < NRL; ++l) {
            const float* p = scr + (l * BN + tid * cps) * 2;
            su += (double)p[0];
            sq += (double)p[1];
        }
        gs[2 * tid] = su;
        gs[2 * tid + 1] = sq;
        double* o = a.stat_part + ((long)(m0 / a.hw_out) * (a.n / cps) + n0 / cps + tid) * 2;  // (one chunk per sample: hw_out == BM)
        o[0] = su;
        o[1] = sq;
    }
    __syncthreads();
    if (!live) return;
    // the consumer's groups: gn_cpg channels = gn_cpg / stat_cpg statistics groups (wd_gn_apply's arithmetic)
    const int gcp = a.gn_cpg, ratio = gcp / cps;
    float sc[4], sh[4];
    const float4 ga = *reinterpret_cast<const float4*>(a.gn_gamma + no);
    const float4 be = *reinterpret_cast<const float4*>(a.gn_beta + no);
    const float gaa[4] = {ga.x, ga.y, ga.z, ga.w}, bea[4] = {be.x, be.y, be.z, be.w};
    int gprev = -1;
    float mean = 0.f, rstd = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = (c + j) / gcp;
        if (g != gprev) {
            double ds = 0.0, dq = 0.0;
            for (int k = 0; k < ratio; ++k) {
                ds += gs[2 * (g * ratio + k)];
                dq += gs[2 * (g * ratio + k) + 1];
            }
            const double n = (double)a.hw_out * gcp;
            const double mu = ds / n;
            double var = dq / n - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            rstd = (float)(1.0 / sqrt(var + (double)a.gn_eps));
            gprev = g;
        }
        sc[j] = rstd * gaa[j];
        sh[j] = bea[j] - mean * sc[j];
    }
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        if (!ok[k]) continue;
        const long m = m0 + rl + k * NRL;
        float4 y;
        y.x = v[k].x * sc[0] + sh[0]; y.y = v[k].y * sc[1] + sh[1]; y.z = v[k].z * sc[2] + sh[2]; y.w = v[k].w * sc[3] + sh[3];
        if (a.gn_silu) {
            y.x = wd_silu(y.x); y.y = wd_silu(y.y); y.z = wd_silu(y.z); y.w = wd_silu(y.w);
        }
        uint2 hh, ll;
        wd_split4(y, hh, ll);
        *reinterpret_cast<uint2*>(a.out_hi + m * a.out_pl_ld + no) = hh;
        if (a.out_lo) *reinterpret_cast<uint2*>(a.out_lo + m * a.out_pl_ld + no) = ll;
    }
}


}  // namespace
