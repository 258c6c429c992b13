// Tap-gather GEMM, "weights straight to registers" form (wd_gemm_args.w_layout == 3) for the 320-column layers of the UNet:
// every 3x3 convolution, the 1x1 projections and the feed-forward output projection (reference unet.py:595,621,632,540,488,
// 364,375,145).
//
// What bounded the 128 x 160 LDS-staged kernel (wd_gemm2_kernel) was not the MFMA pipe but the L2 -> LDS path: 73.7 KB of
// operand rows per 64-deep stage and CU through LDS-DMA (measured ceiling 66-73 GB/s per CU) plus 144 KB of fragment reads
// out of LDS, and more than half of those bytes were WEIGHTS.  A weight element is needed by exactly one wave of a workgroup
// if the waves split the tile by COLUMNS, so it does not have to pass through LDS at all:
//
//   * tile 64 rows x 320 columns (all of N), 8 waves = 2 K-halves x 4 column groups of 80 (4 x 5 tiles of 16 x 16,
//     v_mfma_f32_16x16x32_bf16, the same 80 accumulator registers as before);
//   * W is stored FRAGMENT-MAJOR (wd_gemm_pack_w): the 16 bytes lane l of a B fragment holds - W[16 ct + (l & 15)][32 ks +
//     8 (l >> 4) ...] - sit at ((ks * N/16 + ct) * 64 + l) * 16, so a fragment is ONE contiguous kilobyte and a wave's five
//     column tiles of a k-step are five consecutive kilobytes: buffer_load_dwordx4 straight into the MFMA operand registers,
//     issued one stage ahead (two register sets);
//   * only the A rows (64 x 64 per stage, 16 KB with both planes instead of 73.7 KB) go through LDS, register-staged
//     (global -> VGPR one stage ahead, ds_write after the barrier), so every load of the loop is an ordinary load that the
//     compiler's own vmcnt bookkeeping counts - no LDS-DMA beside register loads.
// Per stage and CU: 16 KB of LDS writes + 64 KB of fragment reads (was 74 + 144), 80 KB of weights through the vector
// memory path.  The epilogue (bias, FiLM, residual, planes, fused GroupNorm statistics, split-K slabs) is the shared one.
#include "wd_gemm_epi.h"
#include "wd_gemm_priv.h"

namespace {

constexpr int WNT = 512;

typedef __attribute__((ext_vector_type(4))) unsigned w_u32x4;

__device__ __forceinline__ int w_lds_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// RH = 1: tile 64 x 320, waves = 2 K-halves x 4 column groups of 80.
// RH = 2: tile 128 x 160, waves = 2 K-halves x 2 row halves x 2 column groups of 80: the two row halves of a column group ask for
//         the same W kilobytes within a few hundred cycles of one another (the second request is served by the CU's L1), so a
//         64-deep stage pulls 32 KB of A + 40 KB of W from L2 instead of 16 + 80.
// A32: src[0] is an fp32 map and the consumer's GroupNorm (+ SiLU) is applied while its rows are staged (wd_gemm_args.a32*): the
//      thread that moves eight channels of a row loads them as two float4 instead of two bf16x8 planes - the same 32 bytes -
//      and normalises, activates and splits them between the load (issued a stage earlier) and the ds_write; the
//      wd_gn_apply launch, its pass over the tensor and the planes it wrote disappear.  A tile's rows are one sample
//      (hw_out % BM == 0): its 2 c scale / shift values are derived from the statistics partials once per workgroup.
template <int NPASS, int RH, bool A32>
__global__ void __launch_bounds__(WNT, 1) wd_gemmw_kernel(const wd_gemm_args a, const int nbn, const int nbm) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int BM = 64 * RH, BN = 320 / RH;
    constexpr int A_PL = BM * 128;  // one plane of one stage: BM rows x 64 bf16
    constexpr int STAGE = NPL * A_PL;
    constexpr uint32_t WD_OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_tab = reinterpret_cast<int*>(smem + 2 * STAGE);  // [ntaps0][BM] source row of src[0] per tap, -1 = zero row

    const int ntile = nbn * nbm;
    const int nwg = ntile * a.ksplit;
    int wg;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int sidx = wg % a.ksplit;
    wg /= a.ksplit;
    const int bn_i = wg % nbn, bm_i = wg / nbn;
    const int m0 = bm_i * BM, n0 = bn_i * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2;
    const int rh = RH == 2 ? (wave >> 1) & 1 : 0;
    const int cg = RH == 2 ? wave & 1 : wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;

    // ---- source-row table of src[0] for this row panel
    {
        const int nt0 = a.src[0].ntaps;
        const int32_t* g0 = a.src[0].gather;
        const int hw_src0 = a.src[0].hw_src;
        const int Wimg = a.slab_rows;  // > 0: 3x3 / pad 1 / stride 1 over images Wimg wide: arithmetic table
        for (int idx = tid; idx < nt0 * BM; idx += WNT) {
            const int t = idx / BM, row = idx - t * BM;
            const int m = m0 + row;
            int v = -1;
            if (m < a.m) {
                if (g0 && Wimg > 0) {
                    const int b = m / a.hw_out, p = m - b * a.hw_out;
                    const int y = p / Wimg, x = p - y * Wimg;
                    const int ky = t / 3, dy = ky - 1, dx = t - ky * 3 - 1;
                    const int sy = y + dy, sx = x + dx;
                    if (sy >= 0 && sy * Wimg < a.hw_out && sx >= 0 && sx < Wimg) v = b * hw_src0 + sy * Wimg + sx;
                } else if (g0) {
                    const int b = m / a.hw_out, p = m - b * a.hw_out;
                    const int g = g0[t * a.hw_out + p];
                    if (g >= 0) v = b * hw_src0 + g;
                } else {
                    v = m;
                }
            }
            s_tab[idx] = v;
        }
    }
    float* s_aff = reinterpret_cast<float*>(s_tab + 9 * BM);  // A32: [c][2] (scale, shift) of this tile's sample
    if constexpr (A32) {
        const int c0 = a.src[0].c;
        const int b = m0 / a.hw_out;   // (hw_out % BM == 0: one sample per tile)
        const int ngp = c0 / a.a32_pcpg, ratio = a.a32_cpg / a.a32_pcpg;
        for (int c = tid; c < c0; c += WNT) {
            const int g = c / a.a32_cpg;
            double su = 0.0, sq = 0.0;
            for (int k = 0; k < ratio; ++k)
                for (int ck = 0; ck < a.a32_nchunk; ++ck) {
                    const double* pp = a.a32_part + (((long)b * a.a32_nchunk + ck) * ngp + g * ratio + k) * 2;
                    su += pp[0];
                    sq += pp[1];
                }
            const double n = (double)a.src[0].hw_src * a.a32_cpg;
            const double mu = su / n;
            double var = sq / n - mu * mu;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)a.a32_eps));
            const float sc = rstd * a.a32_gamma[c];
            s_aff[2 * c] = sc;
            s_aff[2 * c + 1] = a.a32_beta[c] - (float)mu * sc;
        }
    }
    __syncthreads();
    // Row tiles (16 rows) whose source rows are ALL outside the image for a tap (the top image row under kernel row 0, the bottom one
    // under kernel row 2): their operands are exact zeros and their 15 MFMAs per k-step are left out - a twelfth of the work of a
    // 3x3 convolution over 8 x 32 maps.  Bit 4 tap + i, from the table itself (any gather table, not only the arithmetic one).
    unsigned long long skipmask = 0;
    if (RH == 1 && a.src[0].gather) {
        for (int t = 0; t < a.src[0].ntaps; ++t) {
            const unsigned long long oob = __ballot(s_tab[t * BM + lane] < 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (((oob >> (16 * i)) & 0xFFFFull) == 0xFFFFull) skipmask |= 1ull << (4 * t + i);
        }
    }

    auto make_srd = [](const wd_bf16* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<wd_bf16*>(p), 0, 0x7FFFFFF0, 0x00020000);
    };
    // ---- weights: fragment-major, this wave's five column tiles of a k-step are 5 KB in a row
    const int nct = a.n >> 4;
    // (weight groups, wd_gemm_args::w_ngroups: the run of rows this tile lies in picks the image)
    const long wgrp = a.w_ngroups > 1 ? (long)((m0 % a.hw_out) / (a.hw_out / a.w_ngroups)) * a.w_group_stride : 0;
    const __amdgpu_buffer_rsrc_t srd_w_hi = make_srd(a.w_hi + wgrp), srd_w_lo = make_srd((a.w_lo ? a.w_lo : a.w_hi) + wgrp);
    const uint32_t b_voff = (uint32_t)(((n0 >> 4) + 5 * cg) * 1024 + lane * 16);
    const uint32_t kstep_bytes = (uint32_t)nct * 1024u;

    // ---- A rows: thread -> (row, row + 64; 16-byte chunk) of the BM x 64 stage tile, both planes
    const int arow = tid >> 3, ach = tid & 7;
    const int a_dst = w_lds_off(arow, ach);  // (row + 64: + 8192, the swizzle key repeats every 16 rows)

    const int nk_all = a.ktot / 64;
    // (integer divisions run on the vector ALU: readfirstlane tells the compiler that their results are wave-uniform, else the
    // buffer descriptors and scalar offsets derived from them live in VGPRs and every load sits in a waterfall loop)
    const int k_begin = __builtin_amdgcn_readfirstlane((int)((long)nk_all * sidx / a.ksplit));
    const int k_end = __builtin_amdgcn_readfirstlane((int)((long)nk_all * (sidx + 1) / a.ksplit));
    const int nk = k_end - k_begin;
    const int cpt0 = a.src[0].c >> 6, n0st = a.src[0].ntaps * cpt0;
    // (Every workgroup starts at the same stage on purpose: a rotated start - 32 CUs of an XCD reading 32 different slices of W
    // instead of the same one - measured 6-17 % SLOWER; the L2 serves a line that many CUs ask for at once cheaper than many lines.)
    int s = 0, tap = 0, kc = 0, left;  // left: stages of the current source still to load, this one included
    int cur_ld = a.src[0].ld, cur_cpt = cpt0;
    if (k_begin < n0st) {
        tap = __builtin_amdgcn_readfirstlane(k_begin / cpt0);
        kc = k_begin - tap * cpt0;
        left = n0st - k_begin;
    } else {
        s = 1;
        kc = k_begin - n0st;
        cur_ld = a.src[1].ld;
        cur_cpt = a.src[1].c >> 6;
        left = cur_cpt - kc;
    }
    // (both sources' descriptors are loop constants chosen by a uniform branch at the load: a descriptor REASSIGNED inside the
    // loop ends up in VGPRs and turns every load into a waterfall loop)
    const __amdgpu_buffer_rsrc_t srd0_hi = make_srd(A32 ? reinterpret_cast<const wd_bf16*>(a.a32) : a.src[0].hi),
                                 srd0_lo = make_srd(A32 ? reinterpret_cast<const wd_bf16*>(a.a32) : (a.src[0].lo ? a.src[0].lo : a.src[0].hi));
    const wd_bf16* p1h = a.nsrc > 1 ? a.src[1].hi : a.src[0].hi;
    const wd_bf16* p1l = a.nsrc > 1 ? (a.src[1].lo ? a.src[1].lo : a.src[1].hi) : p1h;
    const __amdgpu_buffer_rsrc_t srd1_hi = make_srd(p1h), srd1_lo = make_srd(p1l);
    uint32_t a_voff[RH];
    auto locate = [&]() {
        // (the lane index is re-derived here by a volatile asm - two VALU instructions once per tap - so that no register has to hold
        // the thread's table address across the whole loop: at 255 VGPRs hipcc spilled it and reloaded it right here, behind a
        // vmcnt(0) that drained the weight loads in flight)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
        for (int j = 0; j < RH; ++j) {
            const int row = wave * 8 + (ln >> 3) + 64 * j;   // (= arow + 64 j)
            int r;
            if (s == 0) r = s_tab[tap * BM + row];
            else r = (m0 + row < a.m) ? m0 + row : -1;  // src[1] is an identity source (1x1 skip)
            if (A32 && s == 0) a_voff[j] = r >= 0 ? (uint32_t)r * (uint32_t)(a.a32_ld * 4) + (uint32_t)(ach * 32) : WD_OOB;
            else a_voff[j] = r >= 0 ? (uint32_t)r * (uint32_t)(cur_ld * 2) + (uint32_t)(ach * 16) : WD_OOB;
        }
    };
    locate();
    auto kabs = [&]() { return s == 0 ? tap * cpt0 + kc : n0st + kc; };  // absolute stage (the W k-steps 2 kabs, 2 kabs + 1)
    auto advance = [&]() {  // (straight-line: an early return inside this lambda made hipcc treat s / kc as divergent - VGPR
        --left;             //  descriptors and scalar offsets, every load in a waterfall loop)
        const bool sw1 = left == 0 && (s + 1 < a.nsrc);
        ++kc;
        bool relocate = false;
        if (kc == cur_cpt) {
            kc = 0;
            ++tap;
            relocate = true;
        }
        if (sw1) {
            s = 1;
            tap = 0;
            kc = 0;
            cur_ld = a.src[1].ld;
            cur_cpt = a.src[1].c >> 6;
            left = cur_cpt;
            relocate = true;
        }
        if (relocate) locate();
    };
    int kb = 0;  // absolute stage of the next W load: the stage A loaded one step earlier

    // Every stage issues the SAME vector-memory instructions in the same order - a load that has nothing left to fetch carries an
    // out-of-range offset (the buffer range check returns zeros without touching memory) instead of being branched around: the
    // compiler's vmcnt bookkeeping merges the two sides of such a branch to the more conservative count, which made every
    // stage wait for the loads it had just issued.
    w_u32x4 ra[RH][NPL];  // the A chunks of the stage after next, on their way to LDS (A32: the eight fp32 values, raw)
    int ra_ch = 0;        // A32: first channel of the chunk in ra, or -1 when ra holds planes of src[1]
    bool ra_ok[RH];       // A32: the row is a real one (zero padding stays zero AFTER the normalisation)
    auto load_a = [&](const bool live) {
        if (A32) ra_ch = (s == 0) ? kc * 64 + ach * 8 : -1;
#pragma unroll
        for (int j = 0; j < RH; ++j) {
            const uint32_t vo = live ? a_voff[j] : WD_OOB;
            if (A32) ra_ok[j] = vo != WD_OOB;
            if (A32 && s == 0) {
                static_assert(!A32 || NPL == 2, "the fp32 source path holds eight floats in the two plane registers");
                ra[j][0] = __builtin_amdgcn_raw_buffer_load_b128(srd0_hi, vo, kc * 256, 0);
                ra[j][NPL - 1] = __builtin_amdgcn_raw_buffer_load_b128(srd0_hi, vo + 16, kc * 256, 0);
            } else {
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    if (s == 0) ra[j][p] = __builtin_amdgcn_raw_buffer_load_b128(p ? srd0_lo : srd0_hi, vo, kc * 128, 0);
                    else ra[j][p] = __builtin_amdgcn_raw_buffer_load_b128(p ? srd1_lo : srd1_hi, vo, kc * 128, 0);
                }
            }
        }
    };
    auto store_a = [&](char* base) {
        if (A32 && ra_ch >= 0) {
            // y = x * scale + shift (GroupNorm with its affine folded per (sample, channel)), SiLU, split into the two planes
            const float4* tab = reinterpret_cast<const float4*>(s_aff + 2 * ra_ch);  // (sc0 sh0 sc1 sh1) ...
            const float4 t0 = tab[0], t1 = tab[1], t2 = tab[2], t3 = tab[3];
            const bool silu = a.a32_silu != 0;
#pragma unroll
            for (int j = 0; j < RH; ++j) {
                const float4 x0 = __builtin_bit_cast(float4, ra[j][0]), x1 = __builtin_bit_cast(float4, ra[j][NPL - 1]);
                float4 y0, y1;
                y0.x = x0.x * t0.x + t0.y; y0.y = x0.y * t0.z + t0.w; y0.z = x0.z * t1.x + t1.y; y0.w = x0.w * t1.z + t1.w;
                y1.x = x1.x * t2.x + t2.y; y1.y = x1.y * t2.z + t2.w; y1.z = x1.z * t3.x + t3.y; y1.w = x1.w * t3.z + t3.w;
                if (silu) {
                    y0.x = wd_silu(y0.x); y0.y = wd_silu(y0.y); y0.z = wd_silu(y0.z); y0.w = wd_silu(y0.w);
                    y1.x = wd_silu(y1.x); y1.y = wd_silu(y1.y); y1.z = wd_silu(y1.z); y1.w = wd_silu(y1.w);
                }
                if (!ra_ok[j]) y0 = y1 = make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 h0, l0, h1, l1;
                wd_split4(y0, h0, l0);
                wd_split4(y1, h1, l1);
                *reinterpret_cast<w_u32x4*>(base + j * 8192 + a_dst) = w_u32x4{h0.x, h0.y, h1.x, h1.y};
                *reinterpret_cast<w_u32x4*>(base + (NPL - 1) * A_PL + j * 8192 + a_dst) = w_u32x4{l0.x, l0.y, l1.x, l1.y};
            }
        } else {
#pragma unroll
            for (int j = 0; j < RH; ++j)
#pragma unroll
                for (int p = 0; p < NPL; ++p) *reinterpret_cast<w_u32x4*>(base + p * A_PL + j * 8192 + a_dst) = ra[j][p];
        }
    };
    auto load_b = [&](bf16x8 (&fb)[5][NPL], const bool live) {
        const uint32_t so = live ? (uint32_t)(2 * kb + kh) * kstep_bytes : 0u;
        const uint32_t vo = live ? b_voff : WD_OOB;
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                fb[t][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(p ? srd_w_lo : srd_w_hi, vo + t * 1024, so, 0));
    };

    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] = 0.0f;

    bf16x8 xb[5][NPL], yb[5][NPL];
    bf16x8 xa[4][NPL];
    auto read_a = [&](const char* base) {
        const int ch = kh * 4 + lq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ao = w_lds_off(rh * 64 + i * 16 + l15, ch);
#pragma unroll
            for (int p = 0; p < NPL; ++p) xa[i][p] = *reinterpret_cast<const bf16x8*>(base + p * A_PL + ao);
        }
    };
    auto mfma_all = [&](const bf16x8 (&fb)[5][NPL], const int skip) {
        if constexpr (RH == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (skip & (1 << i)) continue;   // (wave-uniform)
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    if (NPL == 2) {
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][NPL - 1], fb[t][0], acc[i][t], 0, 0, 0);
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], fb[t][NPL - 1], acc[i][t], 0, 0, 0);
                    }
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], fb[t][0], acc[i][t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 5; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (NPL == 2) {
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][NPL - 1], fb[t][0], acc[i][t], 0, 0, 0);
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], fb[t][NPL - 1], acc[i][t], 0, 0, 0);
                    }
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], fb[t][0], acc[i][t], 0, 0, 0);
                }
            }
        }
    };
    // the all-zero row tiles of stage kit of this K slice (tap = absolute stage / chunks per tap, by multiply-shift: exact below 1024)
    const uint32_t cpt0_inv = (65536u + (uint32_t)cpt0 - 1u) / (uint32_t)cpt0;
    auto skip_of = [&](const int kit) {
        if constexpr (RH != 1) return 0;
        const int ka = k_begin + kit;
        const int tp = (int)(((uint32_t)ka * cpt0_inv) >> 16);
        int sk = (ka < n0st && ka < 1024) ? (int)((skipmask >> (4 * tp)) & 15ull) : 0;
        if (kit >= nk) sk = 15;   // the phantom stage of an odd stage count: all-zero operands
        return __builtin_amdgcn_readfirstlane(sk);
    };

    // ---- prologue: A(0) into buffer 0, A(1) on its way, W(0) on its way
    kb = kabs();
    load_a(nk > 0);
    if (nk > 0) advance();
    load_b(xb, nk > 0);
    store_a(smem);
    kb = kabs();
    load_a(nk > 1);
    if (nk > 1) advance();
    // The two K-half groups share the SIMDs pairwise (waves w and w + 4).  The second group multiplies one stage LATE - right
    // behind the barrier, on the fragments it read before it - so that on every SIMD one wave is in its MFMA phase while the
    // other is in its memory phase (vmcnt wait, ds_write, the load burst, the fragment reads).  The loads stay a BURST on purpose:
    // a 1 KB load holds its wave for 60-190 cycles while the vector memory path is busy (per-stage stamps: ~1400 cycles for twelve
    // from four waves at once, i.e. the ~33 B/clk a CU gets out of L2) - spread between the MFMAs they stalled the MFMA stream of
    // their own wave (8 % slower), as a burst they stall a wave whose SIMD partner is multiplying.
    // (-DWD_GEMMW_STAMPS + a.dbg & 0x100: s_memtime stamps of workgroup 0 into a.ws as u64 [wave][stage][4], tools/gemm_bench.py --stamps)
#ifdef WD_GEMMW_STAMPS
    const bool stamp = (a.dbg & 0x100) && blockIdx.x == 0 && lane == 0 && a.ws;
    unsigned long long* sb = reinterpret_cast<unsigned long long*>(a.ws) + (long)wave * ((nk + 1) & ~1) * 4;
#define WD_STAMP(i) if (stamp) sb[kit * 4 + (i)] = __builtin_amdgcn_s_memtime()
#define WD_STAMP_LGKM() if (stamp) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define WD_STAMP(i)
#define WD_STAMP_LGKM()
#endif
    if (kh == 0 || (a.dbg & 0x4000)) {
        auto half_step = [&](const int kit, const bf16x8 (&bcur)[5][NPL], bf16x8 (&bnext)[5][NPL]) {
            // (the MFMAs are register-only: without these fences hipcc moves them below the next stage's barrier and ds_write,
            // whose vmcnt(0) then waits for loads issued a few instructions earlier - the prefetch distance is gone)
            __builtin_amdgcn_sched_barrier(0);
            WD_STAMP(0);
            __syncthreads();  // A(kit) is visible, the other stage buffer is free
            WD_STAMP(1);
            const char* base = smem + (kit & 1) * STAGE;
            char* nbase = smem + ((kit + 1) & 1) * STAGE;
            store_a(nbase);                  // A(kit + 1) (zeros past the end)
            load_b(bnext, kit + 1 < nk);     // W(kit + 1)
            kb = kabs();
            load_a(kit + 2 < nk);            // A(kit + 2)
            if (kit + 2 < nk) advance();
            read_a(base);
            __builtin_amdgcn_sched_barrier(0);
            WD_STAMP_LGKM();
            WD_STAMP(2);
            mfma_all(bcur, skip_of(kit));
            __builtin_amdgcn_sched_barrier(0);
            WD_STAMP(3);
        };
        // (an odd stage count ends on a phantom stage of all-zero operands - exact zeros added - rather than on a branch)
        for (int kit = 0; kit < nk; kit += 2) {
            half_step(kit, xb, yb);
            half_step(kit + 1, yb, xb);
        }
    } else {
        auto late_step = [&](const int kit, bf16x8 (&bset)[5][NPL]) {  // bset: W(kit - 1) on entry, W(kit + 1) on exit
            __builtin_amdgcn_sched_barrier(0);
            WD_STAMP(0);
            __syncthreads();
            WD_STAMP(1);
            const char* base = smem + (kit & 1) * STAGE;
            char* nbase = smem + ((kit + 1) & 1) * STAGE;
            if (kit > 0) mfma_all(bset, skip_of(kit - 1));     // stage kit - 1, fragments read before the barrier
            __builtin_amdgcn_sched_barrier(0);
            WD_STAMP(2);                     // (late group: [1..2] = MFMA, [2..3] = memory phase)
            store_a(nbase);
            load_b(bset, kit + 1 < nk);
            kb = kabs();
            load_a(kit + 2 < nk);
            if (kit + 2 < nk) advance();
            read_a(base);
            __builtin_amdgcn_sched_barrier(0);
            WD_STAMP_LGKM();
            WD_STAMP(3);
        };
        for (int kit = 0; kit < nk; kit += 2) {
            late_step(kit, yb);
            late_step(kit + 1, xb);
        }
        if (nk > 0) mfma_all(yb, skip_of(((nk + 1) & ~1) - 1));  // the last (possibly phantom) stage: index odd, set yb
    }
#undef WD_STAMP
#undef WD_STAMP_LGKM

    // ---- the two K-halves summed in a fixed order through the fp32 image, then the shared epilogue
    constexpr int LDE = BN + 4;
    float* ep = reinterpret_cast<float*>(smem);
    __syncthreads();
    for (int hh = 0; hh < 2; ++hh) {
        if (kh == hh) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* pe = ep + (rh * 64 + i * 16 + 4 * lq + r) * LDE + cg * 80 + t * 16 + l15;
                        *pe = (hh == 0) ? acc[i][t][r] : *pe + acc[i][t][r];
                    }
        }
        __syncthreads();
    }
    wd_epilogue_tail<BM, BN, WNT>(a, ep, m0, n0, tid, sidx);
#endif
}

// [n][ktot] planes -> fragment-major: block (ks, ct) of 64 lanes x 16 bytes; lane l = W[16 ct + (l & 15)][32 ks + 8 (l >> 4) ..]
__global__ void __launch_bounds__(256) wd_pack_w_kernel(const wd_bf16* __restrict__ hi, const wd_bf16* __restrict__ lo, const int n,
                                                        const int ktot, wd_bf16* __restrict__ out_hi, wd_bf16* __restrict__ out_lo) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // 16-byte chunk of the destination
    const long total = (long)n * ktot / 8;
    if (i >= total) return;
    const int nct = n >> 4;
    const int l = (int)(i & 63);
    const long blk = i >> 6;
    const int ct = (int)(blk % nct);
    const long ks = blk / nct;
    const long src = (long)(ct * 16 + (l & 15)) * ktot + ks * 32 + (l >> 4) * 8;
    reinterpret_cast<uint4*>(out_hi)[i] = *reinterpret_cast<const uint4*>(hi + src);
    if (lo) reinterpret_cast<uint4*>(out_lo)[i] = *reinterpret_cast<const uint4*>(lo + src);
}

template <int NPASS, int RH, bool A32 = false>
int launchw(const wd_gemm_args& a, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int BM = 64 * RH, BN = 320 / RH;
    constexpr int loop_smem = 2 * NPL * BM * 128 + 9 * BM * 4 + (A32 ? 2 * 1024 * 4 : 0);  // (+ the scale / shift table: c <= 1024)
    constexpr int red_smem = BM * (BN + 4) * 4 + WD_STAT_SCRATCH;
    constexpr int smem = loop_smem > red_smem ? loop_smem : red_smem;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemmw_kernel<NPASS, RH, A32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nbn = a.n / BN, nbm = (a.m + BM - 1) / BM;
    {
        WdLaunchScope scope(WD_CLS_GEMM_WDIRECT, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
        hipLaunchKernelGGL((wd_gemmw_kernel<NPASS, RH, A32>), dim3(nbn * nbm * a.ksplit), dim3(WNT), smem, st, a, nbn, nbm);
    }
    if (a.ksplit > 1) return wd_gemm_launch_reduce(a, st, BM);
    return wd_check_launch();
}

}  // namespace

int wd_gemmw_launch(const wd_gemm_args& a, hipStream_t st) {
    if (a.a32) return launchw<3, 1, true>(a, st);  // (wd_gemm checked: npass 3, tile 64320, one sample per tile)
    if (a.tile == 128160) return a.npass == 3 ? launchw<3, 2>(a, st) : launchw<1, 2>(a, st);
    return a.npass == 3 ? launchw<3, 1>(a, st) : launchw<1, 1>(a, st);
}

extern "C" int wd_gemm_pack_w(const wd_bf16* hi, const wd_bf16* lo, int n, int ktot, wd_bf16* out_hi, wd_bf16* out_lo, void* stream) {
    if (!hi || !out_hi || (lo && !out_lo) || n <= 0 || ktot <= 0 || n % 16 || ktot % 32) return WD_EINVAL;
    const long total = (long)n * ktot / 8;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(wd_pack_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, hi, lo, n, ktot, out_hi, out_lo);
    return wd_check_launch();
}
