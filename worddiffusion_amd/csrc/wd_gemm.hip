// Tap-gather GEMM on v_mfma_f32_32x32x16_bf16 with split-bf16 operands (gfx950).
//
//   out[m, n] = act( sum_k A[m, k] W[n, k] + bias[n] + rowvec[m / hw, n] + resid[m, n] )
//
// A is never materialised: for every 3x3 tap (or the single 1x1 "tap") a row of the output tile reads one
// row of a token-major activation plane through a per-position gather table, so same-size 3x3, stride-2 3x3,
// nearest-x2-then-3x3 and 1x1 convolutions, the 1x1 skip projection of a ResBlock (a second source appended
// along K) and plain linears are one kernel.  Replaces nn.Conv2d / nn.Linear of reference unet.py:595,621,632,
// 540,488,364,375,175-183,125,145,1201-1205,611.
//
// Tile: BM x BN x 32 per stage, 4 waves (256 threads), each wave 32 rows x (BN / WN) columns of 32x32 MFMA
// tiles.  Operands are staged global -> VGPR -> LDS (16-byte chunks, XOR-swizzled so that ds_read_b128 of a
// 32-row fragment is bank-conflict free), double buffered with one barrier per K-step.  npass = 3 issues
// lo*hi + hi*lo + hi*hi per fragment pair (fp32-class result), npass = 1 hi*hi only.
#include <stdlib.h>

#include "wd_common.h"
#include "wd_gemm_epi.h"
#include "wd_gemm_priv.h"


namespace {

constexpr int BK = 32;

// byte offset of 16-byte chunk `ch` (0..3) of row `row` inside a [rows][32] bf16 plane (64-byte rows)
__device__ __forceinline__ int lds_off(int row, int ch) { return row * 64 + ((ch ^ ((row >> 2) & 3)) << 4); }
// the same [rows][32] plane for 16-row fragments of v_mfma_f32_16x16x32_bf16 (lane = 16 * chunk + row): a ds_read_b128 is
// served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) - MI355X_MICROARCH.md, LDS - i.e. rows {0-3,12-15}
// of one chunk together with rows {4-11} of the next; with the XOR key -(row >> 2) every group covers the 64 banks once (the key
// (row >> 2) above puts rows 0-3 / 4-7 of neighbouring chunks on the same banks: 8 LDS cycles per read instead of 4)
__device__ __forceinline__ int lds_off16(int row, int ch) { return row * 64 + ((ch ^ ((0 - (row >> 2)) & 3)) << 4); }


template <int BM, int BN, int TN, int NT>
__device__ __forceinline__ void wd_epilogue_lds(const wd_gemm_args& a, const f32x16 (&acc)[TN], char* smem, const int m0,
                                                const int n0, const int wm, const int wn, const int wcols, const int kh,
                                                const int nkh, const int tid, const int sidx = 0) {
    constexpr int LDE = BN + 4;
    float* ep = reinterpret_cast<float*>(smem);
    const int lane = tid & 63;
    const int frow = lane & 31, fhalf = lane >> 5;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is done with the operand buffers
    for (int h = 0; h < nkh; ++h) {
        if (kh == h) {
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* p = ep + (wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * LDE + wn * wcols + t * 32 + frow;
                    *p = (h == 0) ? acc[t][r] : *p + acc[t][r];
                }
        }
        __syncthreads();
    }
    wd_epilogue_tail<BM, BN, NT>(a, ep, m0, n0, tid, sidx);
}

template <int BM, int BN, int NPASS>
__global__ void __launch_bounds__(256, 2) wd_gemm_kernel(const wd_gemm_args a, const int nbn, const int nbm) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int WM = BM / 32, WN = 4 / WM;
    constexpr int WCOLS = BN / WN;
    constexpr int TN = WCOLS / 32;
    static_assert(WCOLS % 32 == 0 && WM * WN == 4, "bad tile");
    constexpr int A_CH = BM * 4 / 256;
    constexpr int B_CH = (BN * 4 + 255) / 256;
    constexpr int A_PL = BM * 64;
    constexpr int B_PL = BN * 64;
    constexpr int STAGE = NPL * (A_PL + B_PL);

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware block order: the 8 XCDs each take a contiguous run of logical tiles, so the n-blocks that
    // share one A row-panel (and neighbouring panels that share gather halos) hit the same L2.
    const int nwg = nbn * nbm;
    int wg;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bn_i = wg % nbn, bm_i = wg / nbn;
    const int m0 = bm_i * BM, n0 = bn_i * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ach = tid & 3;

    // ---- per-thread A rows (fixed for the whole K loop)
    int a_row[A_CH], a_b[A_CH], a_p[A_CH];
    bool a_ok[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        a_row[i] = (tid >> 2) + 64 * i;
        const int m = m0 + a_row[i];
        a_ok[i] = m < a.m;
        a_b[i] = a_ok[i] ? m / a.hw_out : 0;
        a_p[i] = a_ok[i] ? m - a_b[i] * a.hw_out : 0;
    }
    // ---- per-thread W rows
    int b_row[B_CH];
    long b_off[B_CH];
    bool b_ok[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        b_row[i] = (tid >> 2) + 64 * i;
        const int n = n0 + b_row[i];
        b_ok[i] = (b_row[i] < BN) && (n < a.n);
        b_off[i] = (long)n * a.ktot + ach * 8;
    }

    // ---- K iteration state: (source, tap, channel chunk)
    int s = 0, tap = 0, kc = 0;
    const wd_bf16* cur_hi = a.src[0].hi;
    const wd_bf16* cur_lo = a.src[0].lo;
    const int32_t* cur_g = a.src[0].gather;
    int cur_ld = a.src[0].ld, cur_c = a.src[0].c, cur_nt = a.src[0].ntaps, cur_hw = a.src[0].hw_src;
    long a_off[A_CH];  // element offset of the source row, -1 = zero row

    auto locate = [&]() {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            long off = -1;
            if (a_ok[i]) {
                if (cur_g) {
                    const int g = cur_g[tap * a.hw_out + a_p[i]];
                    if (g >= 0) off = ((long)a_b[i] * cur_hw + g) * cur_ld;
                } else {
                    off = (long)(m0 + a_row[i]) * cur_ld;
                }
            }
            a_off[i] = off;
        }
    };
    locate();

    uint4 ra[NPL][A_CH], rb[NPL][B_CH];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto gload = [&](int kit) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const long o = a_off[i] + kc * BK + ach * 8;
            ra[0][i] = a_off[i] >= 0 ? *reinterpret_cast<const uint4*>(cur_hi + o) : zero4;
            if (NPL == 2) ra[NPL - 1][i] = a_off[i] >= 0 ? *reinterpret_cast<const uint4*>(cur_lo + o) : zero4;
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const long o = b_off[i] + (long)kit * BK;
            rb[0][i] = b_ok[i] ? *reinterpret_cast<const uint4*>(a.w_hi + o) : zero4;
            if (NPL == 2) rb[NPL - 1][i] = b_ok[i] ? *reinterpret_cast<const uint4*>(a.w_lo + o) : zero4;
        }
    };
    auto advance = [&]() {
        ++kc;
        if (kc * BK == cur_c) {
            kc = 0;
            ++tap;
            if (tap == cur_nt) {
                tap = 0;
                ++s;
                if (s < a.nsrc) {
                    cur_hi = a.src[1].hi;
                    cur_lo = a.src[1].lo;
                    cur_g = a.src[1].gather;
                    cur_ld = a.src[1].ld;
                    cur_c = a.src[1].c;
                    cur_nt = a.src[1].ntaps;
                    cur_hw = a.src[1].hw_src;
                }
            }
            if (s < a.nsrc) locate();
        }
    };
    auto lstore = [&](int stage) {
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
                *reinterpret_cast<uint4*>(base + p * A_PL + lds_off(a_row[i], ach)) = ra[p][i];
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                if (b_row[i] < BN)
                    *reinterpret_cast<uint4*>(base + NPL * A_PL + p * B_PL + lds_off(b_row[i], ach)) = rb[p][i];
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int nk = a.ktot / BK;
    gload(0);
    advance();
    lstore(0);
    __syncthreads();

    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kit = 0; kit < nk; ++kit) {
        const bool more = kit + 1 < nk;
        if (more) {
            gload(kit + 1);
            advance();
        }
        const char* base = smem + (kit & 1) * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = kk * 2 + fhalf;
            const int ao = lds_off(wm * 32 + frow, ch);
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + ao);
            bf16x8 al;
            if (NPL == 2) al = *reinterpret_cast<const bf16x8*>(base + A_PL + ao);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int bo = NPL * A_PL + lds_off(wn * WCOLS + t * 32 + frow, ch);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(base + bo);
                if (NPL == 2) {
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(base + B_PL + bo);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            }
        }
        if (more) lstore((kit + 1) & 1);
        __syncthreads();
    }

    wd_epilogue_lds<BM, BN, TN, 256>(a, acc, smem, m0, n0, wm, wn, WCOLS, 0, 1, tid);
}

// ======================================================================================================
// v2: BK = 64, operands go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging), two LDS stages,
// one barrier per K-step: the loads of step k+1 are in flight while step k is multiplied.  Full 128-byte lines
// per row and plane.  Requires every source's channel count to be a multiple of 64.
// LDS image of a [rows][64] bf16 plane: 128-byte rows, 16-byte chunk c of row r stored at position
// c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 of 32-row fragments); since the DMA writes lane-linear, the swizzle
// is applied to the per-lane SOURCE address (lane l of an 8-row piece lands at row l>>3, position l&7).
constexpr int BK2 = 64;
__device__ __attribute__((aligned(128))) unsigned int wd_zero_line[32];  // the all-zero row (conv padding, m >= M)

typedef __attribute__((address_space(3))) void* wd_lds_ptr;
typedef __attribute__((address_space(1))) const void* wd_gbl_ptr;

__device__ __forceinline__ int lds_off2(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// KS = 1: 4 waves, each owns a 32 x (BN / WN) output tile for the whole K range.
// KS = 2: 8 waves; waves 4..7 shadow waves 0..3 on the same output tile but multiply the second half of every
//         64-deep stage (k-steps 2,3), so each SIMD hosts two independent MFMA / ds_read streams that cover each
//         other's LDS latency; the two partial accumulators are summed once through LDS before the epilogue.
// PP (KS = 2 only): "ping-pong" - the second K-half group runs its MFMAs one phase late (on the fragments it read in the
//         previous stage), so that on every SIMD one wave is in its LDS-read phase while the other is in its MFMA phase:
//         the LDS port (192 KB of fragment reads + 74 KB of DMA writes per stage) and the MFMA pipe work concurrently
//         instead of alternately.
// M16 (128 x 160 tile, KS = 2): v_mfma_f32_16x16x32_bf16 with a 64 x 80 wave tile (4 x 5 tiles of 16 x 16) instead of
//         32x32x16 with 32 x 160: the same accumulator registers and MFMA cycles, but 18 instead of 24 ds_read_b128 of
//         fragments per wave and stage - the loop is LDS-port bound (per-stage stamps: tools/gemm_bench.py --stamps).
template <int BM, int BN, int NPASS, int KS, bool PP = false, bool M16 = false>
__global__ void __launch_bounds__(256 * KS, 1) wd_gemm2_kernel(const wd_gemm_args a, const int nbn, const int nbm) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-resource type and builtins exist in the device pass only)
    static_assert(!M16 || (BM == 128 && BN == 160 && KS == 2 && !PP), "M16 is the 128x160 KS=2 kernel");
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int NW = 4 * KS;
    constexpr int WM = BM / 32, WN = 4 / WM;
    constexpr int WCOLS = BN / WN;
    constexpr int TN = WCOLS / 32;
    static_assert(WCOLS % 32 == 0 && WM * WN == 4, "bad tile");
    constexpr int A_PIECES = BM / 8, B_PIECES = BN / 8;       // 8-row DMA pieces per plane
    constexpr int A_INS = (A_PIECES + NW - 1) / NW, B_INS = (B_PIECES + NW - 1) / NW;
    constexpr int A_PL = BM * 128;
    constexpr int B_PL = BN * 128;
    constexpr int STAGE = NPL * (A_PL + B_PL);
    constexpr int KK_PER = 4 / KS;

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
    // split-K across workgroups: slice `sidx` of the K stages of tile `wg / ksplit`.  The slices of a tile are neighbours in the
    // XCD-contiguous order above, so they normally share an XCD's L2 - where the in-launch combine (wd_epilogue_tail) reads the
    // slabs fastest; a speed choice only, the combine's release / acquire is placement-independent.
    const int sidx = wg % a.ksplit;
    wg /= a.ksplit;
    const int bn_i = wg % nbn, bm_i = wg / nbn;
    const int m0 = bm_i * BM, n0 = bn_i * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3, kh = wave >> 2;
    const int wm = wq / WN, wn = wq % WN;
    const int lrow = lane >> 3, lpos = lane & 7;

    // ---- source-row table of src[0] for this row panel (one global gather lookup per (tap, row), done once)
    {
        const int nt0 = a.src[0].ntaps;
        const int32_t* g0 = a.src[0].gather;
        const int hw_src0 = a.src[0].hw_src;
        const int Wimg = a.slab_rows;  // > 0: 3x3 / pad 1 / stride 1 over images Wimg wide (wd_gemm() checked): arithmetic table
        for (int idx = tid; idx < nt0 * BM; idx += 256 * KS) {
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
    __syncthreads();

    // ---- the rows this lane feeds: piece (wave + NW * i) of 8 rows, for A and for W
    int a_sw[A_INS];
#pragma unroll
    for (int i = 0; i < A_INS; ++i) {
        const int row = (wave + NW * i) * 8 + lrow;
        a_sw[i] = (lpos ^ ((row >> 1) & 7)) * 8;  // source chunk (in elements) that lands at position lpos
    }
    // Operands are fetched with buffer_load ... lds: a buffer resource per plane in SGPRs, a per-lane 32-bit byte offset that
    // only changes when the tap does, and the stage's position (channel chunk / K step) in the scalar offset - no per-stage
    // vector address arithmetic at all.  Padding and out-of-range rows carry an offset beyond num_records: the hardware
    // returns zeros for them (raw buffer range check on the vector offset), which replaces the zero-line select.
    // (wd_gemm() rejects operands whose plane does not fit 31-bit byte offsets.)
    constexpr uint32_t WD_OOB = 0x80000000u;
    auto make_srd = [](const wd_bf16* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<wd_bf16*>(p), 0, 0x7FFFFFF0, 0x00020000);
    };
    uint32_t b_voff[B_INS];
#pragma unroll
    for (int i = 0; i < B_INS; ++i) {
        const int row = (wave + NW * i) * 8 + lrow;
        const int n = n0 + row;
        b_voff[i] = ((row < BN) && (n < a.n)) ? (uint32_t)(((long)n * a.ktot + (lpos ^ ((row >> 1) & 7)) * 8) * 2) : WD_OOB;
    }
    const __amdgpu_buffer_rsrc_t srd_w_hi = make_srd(a.w_hi), srd_w_lo = make_srd(a.w_lo ? a.w_lo : a.w_hi);

    const int nk_all = a.ktot / BK2;
    const int k_begin = (int)((long)nk_all * sidx / a.ksplit), k_end = (int)((long)nk_all * (sidx + 1) / a.ksplit);
    int s = 0, tap = 0, kc = 0;
    const wd_bf16* cur_hi = a.src[0].hi;
    const wd_bf16* cur_lo = a.src[0].lo;
    int cur_ld = a.src[0].ld, cur_c = a.src[0].c, cur_nt = a.src[0].ntaps;
    {   // (source, tap, chunk) of the first stage of this slice
        const int cpt = a.src[0].c / BK2, n0st = a.src[0].ntaps * cpt;
        if (k_begin < n0st) {
            tap = k_begin / cpt;
            kc = k_begin - tap * cpt;
        } else {
            s = 1;
            kc = k_begin - n0st;
            cur_hi = a.src[1].hi;
            cur_lo = a.src[1].lo;
            cur_ld = a.src[1].ld;
            cur_c = a.src[1].c;
            cur_nt = a.src[1].ntaps;
        }
    }
    __amdgpu_buffer_rsrc_t srd_a_hi = make_srd(cur_hi), srd_a_lo = make_srd(cur_lo ? cur_lo : cur_hi);
    uint32_t a_voff[A_INS];  // byte offset of this lane's chunk of its source row, WD_OOB = zero row

    auto locate = [&]() {
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const int row = (wave + NW * i) * 8 + lrow;
            int r = -1;
            if (row < BM) {
                if (s == 0) r = s_tab[tap * BM + row];
                else r = (m0 + row < a.m) ? m0 + row : -1;  // src[1] is an identity source (1x1 skip)
            }
            a_voff[i] = r >= 0 ? (uint32_t)r * (uint32_t)(cur_ld * 2) + (uint32_t)(a_sw[i] * 2) : WD_OOB;
        }
    };
    locate();

    // One stage = NSLOT DMA pieces for this wave (A hi/lo pieces first, then W hi/lo).  What the next stage needs is captured
    // up front (prep: offsets, scalar offsets, the A resources - advance() may move on to the next tap / source before the
    // pieces are issued), the loads themselves are issued one by one BETWEEN the MFMA groups of the current stage (fire), so
    // that DMA issue hides in the MFMA pipe's shadow instead of preceding it as a burst.
    constexpr int NSLOT = NPL * (A_INS + B_INS);
    uint32_t pva[A_INS];        // captured a_voff of the prepared stage
    int soff_a = 0, soff_b = 0;  // captured scalar byte offsets (channel chunk of A, K step of W)
    __amdgpu_buffer_rsrc_t fa_hi = srd_a_hi, fa_lo = srd_a_lo;
    int pdst[NSLOT];            // wave-uniform LDS byte offset inside the stage, < 0: this wave has no such piece
    auto prep = [&](int kit) {
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const int piece = wave + NW * i;
            pva[i] = a_voff[i];
#pragma unroll
            for (int p = 0; p < NPL; ++p) pdst[i * NPL + p] = piece < A_PIECES ? p * A_PL + piece * 1024 : -1;
        }
#pragma unroll
        for (int i = 0; i < B_INS; ++i) {
            const int piece = wave + NW * i;
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                pdst[A_INS * NPL + i * NPL + p] = piece < B_PIECES ? NPL * A_PL + p * B_PL + piece * 1024 : -1;
        }
        soff_a = kc * BK2 * 2;
        soff_b = kit * BK2 * 2;
        fa_hi = srd_a_hi;
        fa_lo = srd_a_lo;
    };
    auto fire = [&](int sl, char* sbase) {  // sl is a compile-time constant at every call site (unrolled loops)
        if (pdst[sl] < 0) return;
        const int p = sl % NPL;
        if (sl < A_INS * NPL)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? fa_lo : fa_hi, (wd_lds_ptr)(sbase + pdst[sl]), 16, pva[sl / NPL], soff_a, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? srd_w_lo : srd_w_hi, (wd_lds_ptr)(sbase + pdst[sl]), 16,
                                                     b_voff[(sl - A_INS * NPL) / NPL], soff_b, 0, 0);
    };
    // address computation and issue in one go (nothing captured)
    auto prep_fire_all = [&](int kit, char* sbase) {
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const int piece = wave + NW * i;
            if (piece < A_PIECES) {
#pragma unroll
                for (int p = 0; p < NPL; ++p)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? srd_a_lo : srd_a_hi, (wd_lds_ptr)(sbase + p * A_PL + piece * 1024), 16,
                                                             a_voff[i], kc * BK2 * 2, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_INS; ++i) {
            const int piece = wave + NW * i;
            if (piece < B_PIECES) {
#pragma unroll
                for (int p = 0; p < NPL; ++p)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? srd_w_lo : srd_w_hi,
                                                             (wd_lds_ptr)(sbase + NPL * A_PL + p * B_PL + piece * 1024), 16, b_voff[i],
                                                             kit * BK2 * 2, 0, 0);
            }
        }
    };
    auto advance = [&]() {
        ++kc;
        if (kc * BK2 == cur_c) {
            kc = 0;
            ++tap;
            if (tap == cur_nt) {
                tap = 0;
                ++s;
                if (s < a.nsrc) {
                    cur_hi = a.src[1].hi;
                    cur_lo = a.src[1].lo;
                    cur_ld = a.src[1].ld;
                    cur_c = a.src[1].c;
                    cur_nt = a.src[1].ntaps;
                    srd_a_hi = make_srd(cur_hi);
                    srd_a_lo = make_srd(cur_lo ? cur_lo : cur_hi);
                }
            }
            if (s < a.nsrc) locate();
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    f32x4 acc16[M16 ? 4 : 1][M16 ? 5 : 1];
#pragma unroll
    for (int i = 0; i < (M16 ? 4 : 1); ++i)
#pragma unroll
        for (int t = 0; t < (M16 ? 5 : 1); ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[i][t][r] = 0.0f;

    const int nk = k_end - k_begin;
    if (nk > 0) {
        prep(k_begin);
        advance();
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) fire(sl, smem);
    }

    const int frow = lane & 31, fhalf = lane >> 5;
    constexpr int NGRP = KK_PER * TN;  // MFMA groups (one 32x32 tile x one 16-deep k-step) of this wave per stage
    bf16x8 fa[KK_PER][NPL], fb[KK_PER][TN][NPL];
    auto read_frags = [&](const char* base) {
#pragma unroll
        for (int k2 = 0; k2 < KK_PER; ++k2) {
            const int ch = (kh * KK_PER + k2) * 2 + fhalf;
            const int ao = lds_off2(wm * 32 + frow, ch);
#pragma unroll
            for (int p = 0; p < NPL; ++p) fa[k2][p] = *reinterpret_cast<const bf16x8*>(base + p * A_PL + ao);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int bo = NPL * A_PL + lds_off2(wn * WCOLS + t * 32 + frow, ch);
#pragma unroll
                for (int p = 0; p < NPL; ++p) fb[k2][t][p] = *reinterpret_cast<const bf16x8*>(base + p * B_PL + bo);
            }
        }
    };
    auto mfma_stage = [&](bool more, char* nbase) {
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int k2 = g / TN, t = g % TN;
            if (NPL == 2) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k2][NPL - 1], fb[k2][t][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k2][0], fb[k2][t][NPL - 1], acc[t], 0, 0, 0);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k2][0], fb[k2][t][0], acc[t], 0, 0, 0);
            // spread this wave's DMA pieces of the next stage over the MFMA groups
            if (more) {
#pragma unroll
                for (int sl = g * NSLOT / NGRP; sl < (g + 1) * NSLOT / NGRP; ++sl) fire(sl, nbase);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (M16) {
        // wave = (K-half kh, row half wq >> 1, column half wq & 1): 64 rows x 80 columns, one 32-deep k-step per stage
        const int l15 = lane & 15, lq = lane >> 4;
        const int r0w = (wq >> 1) * 64, c0w = (wq & 1) * 80;
        bf16x8 xa[4][NPL], xb[5][NPL];
        auto read16 = [&](const char* base) {
            const int ch = kh * 4 + lq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ao = lds_off2(r0w + i * 16 + l15, ch);
#pragma unroll
                for (int p = 0; p < NPL; ++p) xa[i][p] = *reinterpret_cast<const bf16x8*>(base + p * A_PL + ao);
            }
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int bo = NPL * A_PL + lds_off2(c0w + t * 16 + l15, ch);
#pragma unroll
                for (int p = 0; p < NPL; ++p) xb[t][p] = *reinterpret_cast<const bf16x8*>(base + p * B_PL + bo);
            }
        };
        auto mfma16 = [&](bool more, char* nbase) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    if (NPL == 2) {
                        acc16[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][NPL - 1], xb[t][0], acc16[i][t], 0, 0, 0);
                        acc16[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], xb[t][NPL - 1], acc16[i][t], 0, 0, 0);
                    }
                    acc16[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], xb[t][0], acc16[i][t], 0, 0, 0);
                }
                if (more) {  // this wave's DMA pieces of the next stage, spread over the four row groups
#pragma unroll
                    for (int sl = i * NSLOT / 4; sl < (i + 1) * NSLOT / 4; ++sl) fire(sl, nbase);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        const bool stamp16 = (a.dbg & 0x100) && blockIdx.x == 0 && lane == 0 && a.ws;
        unsigned long long* sb16 = reinterpret_cast<unsigned long long*>(a.ws) + (long)wave * nk * 4;
        if (a.dbg & 0x200 ? kh == 1 : false) {
            // stagger (a.dbg & 0x200): the second K-half group multiplies one stage late, so its LDS reads fall under the
            // first group's MFMAs and vice versa.  It keeps TWO fragment sets and alternates between them, so the reads of
            // stage k never wait for the MFMAs of stage k-1 to have consumed their operands (the loop is unrolled by two to
            // keep the register indices static).
            bf16x8 ya[4][NPL], yb[5][NPL];
            auto read16g = [&](auto& fa_, auto& fb_, const char* base) {
                const int ch = kh * 4 + lq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ao = lds_off2(r0w + i * 16 + l15, ch);
#pragma unroll
                    for (int p = 0; p < NPL; ++p) fa_[i][p] = *reinterpret_cast<const bf16x8*>(base + p * A_PL + ao);
                }
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const int bo = NPL * A_PL + lds_off2(c0w + t * 16 + l15, ch);
#pragma unroll
                    for (int p = 0; p < NPL; ++p) fb_[t][p] = *reinterpret_cast<const bf16x8*>(base + p * B_PL + bo);
                }
            };
            auto mfma16g = [&](auto& fa_, auto& fb_) {
    #pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int t = 0; t < 5; ++t) {
                        if (NPL == 2) {
                            acc16[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[i][NPL - 1], fb_[t][0], acc16[i][t], 0, 0, 0);
                            acc16[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[i][0], fb_[t][NPL - 1], acc16[i][t], 0, 0, 0);
                        }
                        acc16[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[i][0], fb_[t][0], acc16[i][t], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                };
            auto half_step = [&](int kit, auto& rd_a, auto& rd_b, auto& mm_a, auto& mm_b) {
                if (stamp16) sb16[kit * 4 + 0] = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (stamp16) sb16[kit * 4 + 1] = __builtin_amdgcn_s_memtime();
                const char* base = smem + (kit & 1) * STAGE;
                char* nbase = smem + ((kit + 1) & 1) * STAGE;
                // the late group's DMA of the next stage goes out first (the buffer is free as of this barrier and the pieces
                // then have the whole stage to land), then this stage's fragment reads into the fragment set the MFMAs below do not
                // use, then the previous stage's products: the reads are in flight under them (measured: 3220 cycles per stage
                // against 3460 with the products first and 3570 with a single fragment set)
                if (kit + 1 < nk) {
                    prep_fire_all(k_begin + kit + 1, nbase);
                    advance();
                }
                __builtin_amdgcn_sched_barrier(0);
                read16g(rd_a, rd_b, base);
                __builtin_amdgcn_sched_barrier(0);
                if (stamp16) sb16[kit * 4 + 2] = __builtin_amdgcn_s_memtime();
                if (kit > 0) mfma16g(mm_a, mm_b);
                __builtin_amdgcn_sched_barrier(0);
                if (stamp16) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    sb16[kit * 4 + 3] = __builtin_amdgcn_s_memtime();
                }
            };
            for (int kit = 0; kit < nk; kit += 2) {
                half_step(kit, xa, xb, ya, yb);
                if (kit + 1 < nk) half_step(kit + 1, ya, yb, xa, xb);
            }
            if (nk > 0) {
                if (nk & 1) mfma16g(xa, xb);
                else mfma16g(ya, yb);
            }
        } else {
            for (int kit = 0; kit < nk; ++kit) {
                if (stamp16) sb16[kit * 4 + 0] = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // step kit has landed, the other stage buffer is free
                if (stamp16) sb16[kit * 4 + 1] = __builtin_amdgcn_s_memtime();
                const bool more = kit + 1 < nk;
                const char* base = smem + (kit & 1) * STAGE;
                char* nbase = smem + ((kit + 1) & 1) * STAGE;
                if (more) {
                    prep(k_begin + kit + 1);
                    advance();
                }
                read16(base);
                __builtin_amdgcn_sched_barrier(0);
                if (stamp16) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    sb16[kit * 4 + 2] = __builtin_amdgcn_s_memtime();
                }
                mfma16(more, nbase);
                if (stamp16) sb16[kit * 4 + 3] = __builtin_amdgcn_s_memtime();
            }
        }
        // ---- partial tiles -> fp32 LDS image (the two K-halves summed in a fixed order), then the shared epilogue
        constexpr int LDE = BN + 4;
        float* ep = reinterpret_cast<float*>(smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int hh = 0; hh < 2; ++hh) {
            if (kh == hh) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 5; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* pe = ep + (r0w + i * 16 + 4 * lq + r) * LDE + c0w + t * 16 + l15;
                            *pe = (hh == 0) ? acc16[i][t][r] : *pe + acc16[i][t][r];
                        }
            }
            __syncthreads();
        }
        wd_epilogue_tail<BM, BN, 256 * KS>(a, ep, m0, n0, tid, sidx);
        return;
    }
    if (PP && KS == 2 && kh == 1) {
        // late group (its own loop, so that the fragments carried across the barrier do not constrain the early group's
        // register allocation): multiply the fragments of the previous stage while the early group reads this one; only
        // then (fragments dead) compute and issue its share of the next stage's DMA, then read this stage
        const bool stamp_l = (a.dbg & 0x100) && blockIdx.x == 0 && lane == 0 && a.ws;
        unsigned long long* sbuf_l = reinterpret_cast<unsigned long long*>(a.ws) + (long)wave * nk * 4;
        for (int kit = 0; kit < nk; ++kit) {
            if (stamp_l) sbuf_l[kit * 4 + 0] = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (stamp_l) sbuf_l[kit * 4 + 1] = __builtin_amdgcn_s_memtime();
            const bool more = kit + 1 < nk;
            const char* base = smem + (kit & 1) * STAGE;
            char* nbase = smem + ((kit + 1) & 1) * STAGE;
            if (kit > 0) mfma_stage(false, nbase);
            __builtin_amdgcn_sched_barrier(0);
            if (stamp_l) sbuf_l[kit * 4 + 2] = __builtin_amdgcn_s_memtime();   // (late group: [1..2] = MFMA, [2..3] = DMA + reads)
            if (more) {
                prep(k_begin + kit + 1);
                advance();
#pragma unroll
                for (int sl = 0; sl < NSLOT; ++sl) fire(sl, nbase);
            }
            __builtin_amdgcn_sched_barrier(0);
            read_frags(base);
            if (stamp_l) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sbuf_l[kit * 4 + 3] = __builtin_amdgcn_s_memtime();
            }
        }
        if (nk > 0) mfma_stage(false, smem);
    } else {
        // (a.dbg & 0x100: per-stage cycle stamps of workgroup 0 into a.ws as u64 [wave][stage][4] - tools/gemm_bench.py --stamps)
        const bool stamp = (a.dbg & 0x100) && blockIdx.x == 0 && lane == 0 && a.ws;
        unsigned long long* sbuf = reinterpret_cast<unsigned long long*>(a.ws) + (long)wave * nk * 4;
        for (int kit = 0; kit < nk; ++kit) {
            if (stamp) sbuf[kit * 4 + 0] = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // step kit has landed, the other stage buffer is free
            if (stamp) sbuf[kit * 4 + 1] = __builtin_amdgcn_s_memtime();
            const bool more = kit + 1 < nk;
            const char* base = smem + (kit & 1) * STAGE;
            char* nbase = smem + ((kit + 1) & 1) * STAGE;
            if (more) {
                prep(k_begin + kit + 1);
                advance();
            }
            read_frags(base);
            __builtin_amdgcn_sched_barrier(0);
            if (stamp) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sbuf[kit * 4 + 2] = __builtin_amdgcn_s_memtime();
            }
            mfma_stage(more, nbase);
            if (stamp) sbuf[kit * 4 + 3] = __builtin_amdgcn_s_memtime();
        }
    }

    wd_epilogue_lds<BM, BN, TN, 256 * KS>(a, acc, smem, m0, n0, wm, wn, WCOLS, kh, KS, tid, sidx);
#endif
}

// ======================================================================================================
// v4: the 128 x 160 tile with 32-deep stages and FOUR waves (64 x 80 wave tiles of 16x16x32 MFMAs, one MFMA k-step per
// stage).  Two stage buffers are 72 KB, so TWO workgroups share a CU: they run out of phase without sharing a barrier, so
// one's prologue / epilogue / DMA-issue stalls sit under the other's MFMAs - which is what the short-K layers (GEGLU
// projection, 1x1 convolutions, attention projections: 5-20 stages) lack in the one-workgroup-per-CU kernel above.
// LDS rows are 64 bytes (lds_off swizzle); a DMA piece is 16 rows.  The fp32 epilogue image of the whole tile does not
// fit beside a second workgroup, so the epilogue runs over the two 64-row halves in turn; GroupNorm statistics (whose
// layout is tied to 128-row panels) stay with the v2 kernel.
template <int NPASS>
__global__ void __launch_bounds__(256, 2) wd_gemm4_kernel(const wd_gemm_args a, const int nbn, const int nbm) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, BN = 160, BK = 32;
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int A_PL = BM * 64, B_PL = BN * 64;
    constexpr int STAGE = NPL * (A_PL + B_PL);
    constexpr int A_PIECES = BM / 16, B_PIECES = BN / 16;
    constexpr int A_INS = A_PIECES / 4, B_INS = (B_PIECES + 3) / 4;

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
    const int sidx = wg / ntile;
    wg -= sidx * ntile;
    const int bn_i = wg % nbn, bm_i = wg / nbn;
    const int m0 = bm_i * BM, n0 = bn_i * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 2, lpos = lane & 3;
    const int sw = (lpos ^ ((0 - (lane >> 4)) & 3)) * 8;  // source chunk (elements) that lands at position lpos of row lrow

    {
        const int nt0 = a.src[0].ntaps;
        const int32_t* g0 = a.src[0].gather;
        const int hw_src0 = a.src[0].hw_src;
        for (int idx = tid; idx < nt0 * BM; idx += 256) {
            const int t = idx / BM, row = idx - t * BM;
            const int m = m0 + row;
            int v = -1;
            if (m < a.m) {
                if (g0) {
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
    __syncthreads();

    // buffer_load ... lds addressing as in wd_gemm2_kernel: resources in SGPRs, 32-bit per-lane byte offsets, the stage's position
    // in the scalar offset, offsets beyond num_records (padding / out-of-range rows) read as zeros
    constexpr uint32_t WD_OOB = 0x80000000u;
    auto make_srd = [](const wd_bf16* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<wd_bf16*>(p), 0, 0x7FFFFFF0, 0x00020000);
    };
    uint32_t b_voff[B_INS];
#pragma unroll
    for (int i = 0; i < B_INS; ++i) {
        const int piece = wave + 4 * i;
        const int n = n0 + piece * 16 + lrow;
        b_voff[i] = ((piece < B_PIECES) && (n < a.n)) ? (uint32_t)(((long)n * a.ktot + sw) * 2) : WD_OOB;
    }
    const __amdgpu_buffer_rsrc_t srd_w_hi = make_srd(a.w_hi), srd_w_lo = make_srd(a.w_lo ? a.w_lo : a.w_hi);

    const int nk_all = a.ktot / BK;
    const int k_begin = (int)((long)nk_all * sidx / a.ksplit), k_end = (int)((long)nk_all * (sidx + 1) / a.ksplit);
    int s = 0, tap = 0, kc = 0;
    const wd_bf16* cur_hi = a.src[0].hi;
    const wd_bf16* cur_lo = a.src[0].lo;
    int cur_ld = a.src[0].ld, cur_c = a.src[0].c, cur_nt = a.src[0].ntaps;
    {
        const int cpt = a.src[0].c / BK, n0st = a.src[0].ntaps * cpt;
        if (k_begin < n0st) {
            tap = k_begin / cpt;
            kc = k_begin - tap * cpt;
        } else {
            s = 1;
            kc = k_begin - n0st;
            cur_hi = a.src[1].hi;
            cur_lo = a.src[1].lo;
            cur_ld = a.src[1].ld;
            cur_c = a.src[1].c;
            cur_nt = a.src[1].ntaps;
        }
    }
    __amdgpu_buffer_rsrc_t srd_a_hi = make_srd(cur_hi), srd_a_lo = make_srd(cur_lo ? cur_lo : cur_hi);
    uint32_t a_voff[A_INS];
    auto locate = [&]() {
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const int row = (wave + 4 * i) * 16 + lrow;
            int r;
            if (s == 0) r = s_tab[tap * BM + row];
            else r = (m0 + row < a.m) ? m0 + row : -1;  // src[1] is an identity source (1x1 skip)
            a_voff[i] = r >= 0 ? (uint32_t)r * (uint32_t)(cur_ld * 2) + (uint32_t)(sw * 2) : WD_OOB;
        }
    };
    locate();
    auto issue = [&](int kit, char* sbase) {
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const int piece = wave + 4 * i;
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? srd_a_lo : srd_a_hi, (wd_lds_ptr)(sbase + p * A_PL + piece * 1024), 16,
                                                         a_voff[i], kc * BK * 2, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_INS; ++i) {
            const int piece = wave + 4 * i;
            if (piece < B_PIECES) {
#pragma unroll
                for (int p = 0; p < NPL; ++p)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? srd_w_lo : srd_w_hi,
                                                             (wd_lds_ptr)(sbase + NPL * A_PL + p * B_PL + piece * 1024), 16, b_voff[i],
                                                             kit * BK * 2, 0, 0);
            }
        }
    };
    auto advance = [&]() {
        ++kc;
        if (kc * BK == cur_c) {
            kc = 0;
            ++tap;
            if (tap == cur_nt) {
                tap = 0;
                ++s;
                if (s < a.nsrc) {
                    cur_hi = a.src[1].hi;
                    cur_lo = a.src[1].lo;
                    cur_ld = a.src[1].ld;
                    cur_c = a.src[1].c;
                    cur_nt = a.src[1].ntaps;
                    srd_a_hi = make_srd(cur_hi);
                    srd_a_lo = make_srd(cur_lo ? cur_lo : cur_hi);
                }
            }
            if (s < a.nsrc) locate();
        }
    };

    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] = 0.0f;

    const int nk = k_end - k_begin;
    if (nk > 0) {
        issue(k_begin, smem);
        advance();
    }
    const int l15 = lane & 15, lq = lane >> 4;
    const int r0w = (wave >> 1) * 64, c0w = (wave & 1) * 80;
    for (int kit = 0; kit < nk; ++kit) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // stage kit has landed; every wave is done reading the other buffer
        const char* base = smem + (kit & 1) * STAGE;
        // the next stage's DMA goes out first: it then has the whole stage to land, and while this wave is held up in the
        // issue the co-resident workgroup fills the MFMA pipe (issuing between the MFMA groups, or after the fragment
        // reads, both measured 4-7 % slower on the GEGLU projection)
        if (kit + 1 < nk) {
            issue(k_begin + kit + 1, smem + ((kit + 1) & 1) * STAGE);
            advance();
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 xa[4][NPL], xb[5][NPL];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ao = lds_off16(r0w + i * 16 + l15, lq);
#pragma unroll
            for (int p = 0; p < NPL; ++p) xa[i][p] = *reinterpret_cast<const bf16x8*>(base + p * A_PL + ao);
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int bo = NPL * A_PL + lds_off16(c0w + t * 16 + l15, lq);
#pragma unroll
            for (int p = 0; p < NPL; ++p) xb[t][p] = *reinterpret_cast<const bf16x8*>(base + p * B_PL + bo);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (NPL == 2) {
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][NPL - 1], xb[t][0], acc[i][t], 0, 0, 0);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], xb[t][NPL - 1], acc[i][t], 0, 0, 0);
                }
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], xb[t][0], acc[i][t], 0, 0, 0);
            }
    }

    // ---- epilogue over the two 64-row halves of the tile in turn (fp32 LDS image of 64 x 160)
    constexpr int LDE = BN + 4;
    float* ep = reinterpret_cast<float*>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int half = 0; half < 2; ++half) {
        if ((wave >> 1) == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ep[(i * 16 + 4 * lq + r) * LDE + c0w + t * 16 + l15] = acc[i][t][r];
        }
        __syncthreads();
        wd_epilogue_tail<64, BN, 256>(a, ep, m0 + half * 64, n0, tid, sidx);
        __syncthreads();
    }
#endif
}

// ---- opt-in kernels that lost their A/B against the defaults: compiled with -DWDIFF_EXPERIMENTAL only
#ifdef WDIFF_EXPERIMENTAL
// ======================================================================================================
// v3 "slab" kernel: the A operand of a 3x3 convolution is NOT re-fetched per tap.  For a panel of BM output rows
// the union of source rows over all taps is a short contiguous range (a few image rows + halo): that slab is
// DMA'd into LDS once per 32-channel chunk (double buffered, prefetched one chunk ahead) and all nine taps read
// their shifted A fragments from it through a per-(tap,row) LDS table.  Only the weights stream per tap, through
// a 4-deep LDS ring kept full with counted s_waitcnt vmcnt(N) + raw s_barrier (no full drain in the loop).
// Weights are stored in consumption order [chunk][tap][N][32] so that one stage is one contiguous block.
// L2->LDS bytes per MFMA drop ~2.8x against v2; linears (one tap) run through the same code.
constexpr int CK3 = 32;
constexpr int RING3 = 4;

__device__ __forceinline__ void wd_wait_vmcnt(int n) {  // n is wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // stricter than needed, still correct
    }
}

template <int BM, int BN, int NPASS, int SLABR, int KS>
__global__ void __launch_bounds__(256 * KS, 1) wd_gemm3_kernel(const wd_gemm_args a, const int nbn, const int nbm) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int NW = 4 * KS, NT = 256 * KS;
    constexpr int WM = BM / 32, WN = 4 / WM;
    constexpr int WCOLS = BN / WN;
    constexpr int TN = WCOLS / 32;
    static_assert(WCOLS % 32 == 0 && WM * WN == 4, "bad tile");
    constexpr int SLAB_PL = (SLABR + 1) * 64;          // one plane of one slab buffer (+ the all-zero row)
    constexpr int SLAB_BUF = NPL * SLAB_PL;
    constexpr int W_PL = BN * 64;
    constexpr int W_SLOT = NPL * W_PL;
    constexpr int OFF_RING = 2 * SLAB_BUF;
    constexpr int OFF_TAB = OFF_RING + RING3 * W_SLOT;
    constexpr int W_PIECES = NPL * (BN / 16);            // 1 KB DMA pieces per W stage
    constexpr int W_INS = (W_PIECES + NW - 1) / NW;
    constexpr int S_PIECES_MAX = NPL * ((SLABR + 15) / 16);
    constexpr int S_INS = (S_PIECES_MAX + NW - 1) / NW;
    constexpr int KK_PER = 2 / KS;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_tab = reinterpret_cast<int*>(smem + OFF_TAB);   // [ntaps0][BM] slab-local row, SLABR = zero row
    int* s_misc = s_tab + 9 * BM;                          // [0] = min source row of the panel

    const int nwg = nbn * nbm;
    int wg;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bn_i = wg % nbn, bm_i = wg / nbn;
    const int m0 = bm_i * BM, n0 = bn_i * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3, kh = wave >> 2;
    const int wm = wq / WN, wn = wq % WN;
    const int prow = lane >> 2, ppos = lane & 3;  // row / 16-byte position of this lane inside a 16-row DMA piece

    const int dbg = a.dbg;  // timing-only ablation switches (tools/gemm_bench.py --dbg), 0 in production
    const int nt0 = a.src[0].ntaps;
    const int c0chunks = a.src[0].c / CK3;
    const int c1chunks = a.nsrc > 1 ? a.src[1].c / CK3 : 0;
    const int nstage = (dbg & 32) ? 0 : c0chunks * nt0 + c1chunks;
    if (dbg & 128) return;

    // ---- table of source rows, panel minimum, zero rows
    if (tid == 0) s_misc[0] = 0x7fffffff;
    for (int i = tid; i < 2 * NPL * 16; i += NT) {  // the zero row of every slab buffer / plane (64 B each)
        const int which = i >> 4, w = i & 15;
        reinterpret_cast<int*>(smem + which * SLAB_PL + SLABR * 64)[w] = 0;
    }
    __syncthreads();
    {
        const int32_t* g0 = a.src[0].gather;
        const int hw_src0 = a.src[0].hw_src;
        int mn = 0x7fffffff;
        for (int idx = tid; idx < nt0 * BM; idx += NT) {
            const int t = idx / BM, row = idx - t * BM;
            const int m = m0 + row;
            int v = -1;
            if (m < a.m) {
                if (g0) {
                    const int b = m / a.hw_out, p = m - b * a.hw_out;
                    const int g = g0[t * a.hw_out + p];
                    if (g >= 0) v = b * hw_src0 + g;
                } else {
                    v = m;
                }
            }
            s_tab[idx] = v;
            if (v >= 0) mn = min(mn, v);
        }
        atomicMin(&s_misc[0], mn);
    }
    __syncthreads();
    const int rlo = s_misc[0];
    for (int idx = tid; idx < nt0 * BM; idx += NT) {
        const int v = s_tab[idx];
        int l = SLABR;
        if (v >= 0 && v - rlo < SLABR) l = v - rlo;  // (span is validated by the host; out-of-span rows read zeros)
        s_tab[idx] = l;
    }
    // rows of src[0] present in the slab: [rlo, rlo + span0); src[1] (identity): [m0, m0 + BM)
    const int span0 = min(SLABR, a.slab_rows);
    const long rows0 = (long)((a.src[0].gather ? (long)((a.m + a.hw_out - 1) / a.hw_out) * a.src[0].hw_src : (long)a.m));
    const wd_bf16* zline = reinterpret_cast<const wd_bf16*>(wd_zero_line) + ppos * 8;

    // DMA bookkeeping (all wave-uniform scalars): `issued` counts this wave's glds; mw0..mw3 are the values of
    // `issued` right after the W stages st, st+1, st+2, st+3 were issued; ms_cur / ms_nxt the same for the slab of
    // the current / next phase.  s_waitcnt vmcnt(issued - mark) == "everything up to that issue has landed".
    int issued = 0;
    int mw0 = 0, mw1 = 0, mw2 = 0, mw3 = 0;
    int ms_cur = 0, ms_nxt = 0;

    // slab of phase ph (source, chunk) -> buffer ph & 1
    auto issue_slab = [&](int ph) {
        const bool s1 = ph >= c0chunks;
        const int chunk = s1 ? ph - c0chunks : ph;
        const wd_bf16* hi = s1 ? a.src[1].hi : a.src[0].hi;
        const wd_bf16* lo = s1 ? a.src[1].lo : a.src[0].lo;
        const int ld = s1 ? a.src[1].ld : a.src[0].ld;
        const long base_row = s1 ? m0 : rlo;
        const int span = s1 ? BM : span0;
        const long row_lim = s1 ? (long)a.m : rows0;
        const int npieces = NPL * ((span + 15) / 16);
        char* sb = smem + (ph & 1) * SLAB_BUF;
#pragma unroll
        for (int i = 0; i < S_INS; ++i) {
            const int piece = wave + NW * i;  // wave-uniform
            if (piece < npieces) {
                const int pl = (NPL == 2) ? (piece & 1) : 0;
                const int rp = (NPL == 2) ? (piece >> 1) : piece;
                const int r = rp * 16 + prow;                       // slab row
                const long gr = base_row + r;
                const wd_bf16* src = (pl ? lo : hi);
                const wd_bf16* ptr = (r < span && gr < row_lim)
                                         ? src + gr * ld + chunk * CK3 + ((ppos ^ ((r >> 2) & 3)) << 3)
                                         : zline;
                __builtin_amdgcn_global_load_lds((wd_gbl_ptr)ptr, (wd_lds_ptr)(sb + pl * SLAB_PL + rp * 1024), 16, 0, 0);
                ++issued;
            }
        }
        ms_nxt = issued;
    };
    // W stage st (consumption order) -> ring slot st % RING3.  Global block st: [N][32] per plane, contiguous.
    auto issue_w = [&](int st) {
        char* rb = smem + OFF_RING + (st % RING3) * W_SLOT;
        const long blk = (long)st * a.n * CK3;
#pragma unroll
        for (int i = 0; i < W_INS; ++i) {
            const int piece = wave + NW * i;
            if (piece < W_PIECES) {
                const int pl = (NPL == 2) ? (piece & 1) : 0;
                const int rp = (NPL == 2) ? (piece >> 1) : piece;
                const int r = rp * 16 + prow;  // row inside the BN tile
                const int n = n0 + r;
                const wd_bf16* src = pl ? a.w_lo : a.w_hi;
                const wd_bf16* ptr = (n < a.n) ? src + blk + (long)n * CK3 + ((ppos ^ ((r >> 2) & 3)) << 3) : zline;
                __builtin_amdgcn_global_load_lds((wd_gbl_ptr)ptr, (wd_lds_ptr)(rb + pl * W_PL + rp * 1024), 16, 0, 0);
                ++issued;
            }
        }
        mw3 = issued;
    };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    __syncthreads();  // table finalised (also orders the zero-row writes before any fragment read)
    issue_slab(0);
    ms_cur = ms_nxt;
    if (0 < nstage) { issue_w(0); mw0 = mw3; }
    if (1 < nstage) { issue_w(1); mw1 = mw3; }
    if (2 < nstage) { issue_w(2); mw2 = mw3; }

    const int frow = lane & 31, fhalf = lane >> 5;
    const int arow = wm * 32 + frow;
    int ph = 0, tap = 0;       // phase / tap of the current stage
    int ntap_ph = nt0;         // taps in the current phase
    auto next_stage = [&]() {
        mw0 = mw1;
        mw1 = mw2;
        mw2 = mw3;
        if (++tap == ntap_ph) {
            tap = 0;
            ++ph;
            ntap_ph = ph < c0chunks ? nt0 : 1;
            ms_cur = ms_nxt;
        }
    };
    for (int st = 0; st < nstage; ++st) {
        // ---- wait until stage st's weights and its phase's slab have landed (everything younger may stay in flight)
        wd_wait_vmcnt(issued - max(mw0, ms_cur));
        if (!(dbg & 16)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- refill: the slot of stage st-1 and (at a phase start) the other slab buffer are free now
        if (tap == 0 && ph + 1 < c0chunks + c1chunks && !(dbg & 8)) issue_slab(ph + 1);
        if (st + RING3 - 1 < nstage && !(dbg & 4)) issue_w(st + RING3 - 1);

        // ---- multiply stage st
        const char* sb = smem + (ph & 1) * SLAB_BUF;
        const char* rb = smem + OFF_RING + (st % RING3) * W_SLOT;
        int idx;
        if (ph < c0chunks) idx = s_tab[tap * BM + arow];
        else idx = (m0 + arow < a.m) ? arow : SLABR;
        const int abase = idx * 64, asw = (idx >> 2) & 3;
        // all fragments of this wave's k-steps first (the MFMAs then drain them behind counted lgkmcnt waits) ...
        bf16x8 fa[KK_PER][NPL], fb[KK_PER][TN][NPL];
        if (dbg & 2) { next_stage(); continue; }
#pragma unroll
        for (int k2 = 0; k2 < KK_PER; ++k2) {
            const int ch = (kh * KK_PER + k2) * 2 + fhalf;
            const int ao = abase + ((ch ^ asw) << 4);
#pragma unroll
            for (int p = 0; p < NPL; ++p) fa[k2][p] = *reinterpret_cast<const bf16x8*>(sb + p * SLAB_PL + ao);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int bo = lds_off(wn * WCOLS + t * 32 + frow, ch);
#pragma unroll
                for (int p = 0; p < NPL; ++p) fb[k2][t][p] = *reinterpret_cast<const bf16x8*>(rb + p * W_PL + bo);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (dbg & 1) {
#pragma unroll
            for (int k2 = 0; k2 < KK_PER; ++k2) {
                asm volatile("" ::"v"(fa[k2][0]), "v"(fa[k2][NPL - 1]));
#pragma unroll
                for (int t = 0; t < TN; ++t) asm volatile("" ::"v"(fb[k2][t][0]), "v"(fb[k2][t][NPL - 1]));
            }
            next_stage();
            continue;
        }
        // ... then the multiply block
#pragma unroll
        for (int k2 = 0; k2 < KK_PER; ++k2) {
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (NPL == 2) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k2][NPL - 1], fb[k2][t][0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k2][0], fb[k2][t][NPL - 1], acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k2][0], fb[k2][t][0], acc[t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        next_stage();
    }
    if (dbg & 64) {
        if (acc[0][0] == 123.456f) a.out_f32[0] = acc[0][1];
        return;
    }
    wd_epilogue_lds<BM, BN, TN, NT>(a, acc, smem, m0, n0, wm, wn, WCOLS, kh, KS, tid);
}

template <int BM, int BN, int NPASS, int SLABR, int KS>
int launch3(const wd_gemm_args& a, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int loop_smem = 2 * NPL * (SLABR + 1) * 64 + RING3 * NPL * BN * 64 + 9 * BM * 4 + 64;
    constexpr int red_smem = BM * (BN + 4) * 4 + WD_STAT_SCRATCH;
    constexpr int smem = loop_smem > red_smem ? loop_smem : red_smem;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemm3_kernel<BM, BN, NPASS, SLABR, KS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nbn = (a.n + BN - 1) / BN, nbm = (a.m + BM - 1) / BM;
    WdLaunchScope scope(WD_CLS_GEMM_OTHER, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
    hipLaunchKernelGGL((wd_gemm3_kernel<BM, BN, NPASS, SLABR, KS>), dim3(nbn * nbm), dim3(256 * KS), smem, st, a, nbn,
                       nbm);
    return wd_check_launch();
}

#endif  // WDIFF_EXPERIMENTAL

template <int BM, int BN, int NPASS>
int launch(const wd_gemm_args& a, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int loop_smem = 2 * NPL * (BM + BN) * 64;
    constexpr int red_smem = BM * (BN + 4) * 4 + WD_STAT_SCRATCH;
    constexpr int smem = loop_smem > red_smem ? loop_smem : red_smem;
    static bool attr_done = false;  // one process = one device, set once per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemm_kernel<BM, BN, NPASS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nbn = (a.n + BN - 1) / BN, nbm = (a.m + BM - 1) / BM;
    WdLaunchScope scope(WD_CLS_GEMM_OTHER, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
    hipLaunchKernelGGL((wd_gemm_kernel<BM, BN, NPASS>), dim3(nbn * nbm), dim3(256), smem, st, a, nbn, nbm);
    return wd_check_launch();
}

// out = epilogue(sum of the ksplit partial slabs): one workgroup per BM x BN tile sums the slabs in a fixed order into
// the same fp32 LDS image the in-kernel epilogue uses and then runs that epilogue (statistics included).
template <int BM, int BN>
__global__ void __launch_bounds__(256) wd_gemm_reduce_kernel(const wd_gemm_args a, const int nbn) {
    constexpr int LDE = BN + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ep = reinterpret_cast<float*>(smem);
    const int bn_i = blockIdx.x % nbn, bm_i = blockIdx.x / nbn;
    const int m0 = bm_i * BM, n0 = bn_i * BN;
    const int tid = threadIdx.x;
    const long total = (long)a.m * a.n;
    const bool v4 = (a.n & 3) == 0;
    if constexpr (BM == 64 && BN == 40) {
        if (a.gn_gamma) {  // (the host has checked shapes and alignment: wd_gemm)
            wd_gn_tile<40, 256, true>(a, nullptr, ep, m0, n0, tid);
            return;
        }
    }
    {
        // the epilogue's vector path can sum the slabs itself (no LDS image, one barrier less, the slab / residual / row-vector
        // loads of a row in flight together); same conditions as its own `vec` test plus aligned slabs
        const bool direct = v4 && (((a.out_ld | a.rowvec_ld | a.resid_ld | a.out_pl_ld) & 3) == 0) &&
                            (((reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.rowvec) |
                               reinterpret_cast<uintptr_t>(a.resid) | reinterpret_cast<uintptr_t>(a.out_f32) |
                               reinterpret_cast<uintptr_t>(a.ws)) & 15) == 0) &&
                            (((reinterpret_cast<uintptr_t>(a.out_hi) | reinterpret_cast<uintptr_t>(a.out_lo)) & 7) == 0) &&
                            n0 + BN <= a.n;
        if (direct) {
            wd_gemm_args b = a;
            b.ksplit = 1;
            wd_epilogue_from_image<BM, BN, 256, true>(b, ep, m0, n0, tid, a.ksplit);
            return;
        }
    }
    if (v4) {
        // slab-major with all of the thread's elements in flight per slab: the pass is latency-bound otherwise (one
        // workgroup's worth of loads per CU).  Per element the slabs are still summed in ascending order.
        constexpr int NI = (BM * (BN / 4) + 255) / 256;
        float4 v[NI];
        long off[NI];
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            const int i = tid + ii * 256;
            const int row = i / (BN / 4), c = (i - row * (BN / 4)) * 4;
            const int m = m0 + row, n = n0 + c;
            v[ii] = make_float4(0.f, 0.f, 0.f, 0.f);
            off[ii] = (i < BM * (BN / 4) && m < a.m && n < a.n) ? (long)m * a.n + n : -1;
        }
        for (int sp = 0; sp < a.ksplit; ++sp) {
            const float* slab = a.ws + (long)sp * total;
            float4 q[NI];
#pragma unroll
            for (int ii = 0; ii < NI; ++ii)
                q[ii] = off[ii] >= 0 ? *reinterpret_cast<const float4*>(slab + off[ii]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) {
                v[ii].x += q[ii].x; v[ii].y += q[ii].y; v[ii].z += q[ii].z; v[ii].w += q[ii].w;
            }
        }
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            const int i = tid + ii * 256;
            if (i < BM * (BN / 4)) {
                const int row = i / (BN / 4), c = (i - row * (BN / 4)) * 4;
                *reinterpret_cast<float4*>(ep + row * LDE + c) = v[ii];
            }
        }
    } else {
        for (int i = tid; i < BM * (BN / 4); i += 256) {
            const int row = i / (BN / 4), c = (i - row * (BN / 4)) * 4;
            const int m = m0 + row, n = n0 + c;
            float e[4] = {0.f, 0.f, 0.f, 0.f};
            if (m < a.m && n < a.n)
                for (int sp = 0; sp < a.ksplit; ++sp)
                    for (int j = 0; j < 4 && n + j < a.n; ++j) e[j] += a.ws[(long)sp * total + (long)m * a.n + n + j];
            *reinterpret_cast<float4*>(ep + row * LDE + c) = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
    __syncthreads();
    wd_gemm_args b = a;
    b.ksplit = 1;
    wd_epilogue_from_image<BM, BN, 256>(b, ep, m0, n0, tid);
}

template <int RM, int RN>
int launch_reduce(const wd_gemm_args& a, hipStream_t st) {
    constexpr int rsmem = RM * (RN + 4) * 4 + WD_STAT_SCRATCH;
    static bool rattr = false;
    if (!rattr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemm_reduce_kernel<RM, RN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, rsmem) != hipSuccess)
            return WD_ELAUNCH;
        rattr = true;
    }
    const int rbn = (a.n + RN - 1) / RN, rbm = (a.m + RM - 1) / RM;
    WdLaunchScope scope(WD_CLS_GEMM_REDUCE, st);
    hipLaunchKernelGGL((wd_gemm_reduce_kernel<RM, RN>), dim3(rbn * rbm), dim3(256), rsmem, st, a, rbn);
    return wd_check_launch();
}

// the combine pass is pure streaming: narrow column tiles so that it fills the chip (whole statistics groups per tile
// when the GroupNorm sums are fused in).  The statistics layout is indexed by chunks of BM rows when a sample has more
// rows than that, so the 64-row tile is only used where it leaves that layout unchanged (samples of <= 64 rows).
// The combine for plain fp32 outputs (no statistics, row vector, activation or planes - every weight-gradient GEMM of the
// training step): a pure stream, one float4 of the output per thread, the slabs summed in ascending order; bias and the
// residual (gradient accumulation) are the only epilogue terms.  Half the time of the tile-shaped combine above.
__global__ void __launch_bounds__(256) wd_gemm_reduce_flat_kernel(const wd_gemm_args a) {
    const long total = (long)a.m * a.n, n4 = a.n >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.m * n4) return;
    const long m = i / n4;
    const int n = (int)(i - m * n4) * 4;
    const float* p = a.ws + m * a.n + n;
    float4 v = *reinterpret_cast<const float4*>(p);
    for (int sp = 1; sp < a.ksplit; ++sp) {
        const float4 q = *reinterpret_cast<const float4*>(p + (long)sp * total);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    if (a.bias) {
        const float4 q = *reinterpret_cast<const float4*>(a.bias + n);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    if (a.resid) {
        const float4 q = *reinterpret_cast<const float4*>(a.resid + m * a.resid_ld + n);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *reinterpret_cast<float4*>(a.out_f32 + m * a.out_ld + n) = v;
}

static bool reduce_flat_ok(const wd_gemm_args& a) {
    return !a.stat_part && !a.rowvec && !a.resid_rows && a.act == WD_ACT_NONE && !a.out_hi && a.out_f32 && (a.n & 3) == 0 &&
           (a.out_ld & 3) == 0 && (!a.resid || (a.resid_ld & 3) == 0) &&
           ((reinterpret_cast<uintptr_t>(a.out_f32) | reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.resid) |
             reinterpret_cast<uintptr_t>(a.ws)) & 15) == 0;
}

template <int BM, int BN>
int launch_reduce_any(const wd_gemm_args& a, hipStream_t st) {
    if (reduce_flat_ok(a)) {
        WdLaunchScope scope(WD_CLS_GEMM_REDUCE, st);
        const long items = (long)a.m * (a.n >> 2);
        hipLaunchKernelGGL(wd_gemm_reduce_flat_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, a);
        return wd_check_launch();
    }
    const int cpg = a.stat_part ? a.stat_cpg : 1;
    const bool half_rows = BM == 128 && (!a.stat_part || (a.hw_out <= 64 && 64 % a.hw_out == 0));
    if (40 % cpg == 0) return half_rows ? launch_reduce<64, 40>(a, st) : launch_reduce<128, 40>(a, st);
    if (32 % cpg == 0) return half_rows ? launch_reduce<64, 32>(a, st) : launch_reduce<128, 32>(a, st);
    return launch_reduce<BM, BN>(a, st);
}

template <int NPASS>
int launch4(const wd_gemm_args& a, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int smem = 2 * NPL * (128 + 160) * 64 + 9 * 128 * 4;
    static_assert(64 * 164 * 4 <= 2 * NPL * (128 + 160) * 64 || NPASS == 1, "epilogue image must fit the stage buffers");
    constexpr int smem_need = smem > 64 * 164 * 4 ? smem : 64 * 164 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemm4_kernel<NPASS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem_need) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nbn = (a.n + 159) / 160, nbm = (a.m + 127) / 128;
    {
        WdLaunchScope scope(WD_CLS_GEMM_2CU, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
        hipLaunchKernelGGL((wd_gemm4_kernel<NPASS>), dim3(nbn * nbm * a.ksplit), dim3(256), smem_need, st, a, nbn, nbm);
    }
    if (a.ksplit > 1) return launch_reduce_any<128, 160>(a, st);
    return wd_check_launch();
}

template <int BM, int BN, int NPASS, int KS, bool PP = false, bool M16 = false>
int launch2(const wd_gemm_args& a, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int loop_smem = 2 * NPL * (BM + BN) * 128 + 9 * BM * 4;
    constexpr int red_smem = BM * (BN + 4) * 4 + WD_STAT_SCRATCH;  // fp32 epilogue image (+ statistics scratch) overlays the stages
    constexpr int smem = loop_smem > red_smem ? loop_smem : red_smem;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemm2_kernel<BM, BN, NPASS, KS, PP, M16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nbn = (a.n + BN - 1) / BN, nbm = (a.m + BM - 1) / BM;
    {
        WdLaunchScope scope((BM == 128 && BN == 160) ? WD_CLS_GEMM : WD_CLS_GEMM_OTHER, st,
                            2.0 * (double)a.m * (double)a.n * (double)a.ktot);
        hipLaunchKernelGGL((wd_gemm2_kernel<BM, BN, NPASS, KS, PP, M16>), dim3(nbn * nbm * a.ksplit), dim3(256 * KS), smem, st, a,
                           nbn, nbm);
    }
    if (a.ksplit > 1 && !a.tickets) {
        // the combine pass is pure streaming: narrow column tiles so that it fills the chip (whole statistics groups
        // per tile when the GroupNorm sums are fused in)
        return launch_reduce_any<BM, BN>(a, st);
    }
    return wd_check_launch();
}


#ifdef WDIFF_EXPERIMENTAL
// ======================================================================================================
// Row-shared-taps kernel for the 3x3 / pad 1 / stride 1 convolutions (w_layout == 2; slab_rows = image width W).
// The per-stage cycle stamps of the kernel above show that a global_load_lds piece costs its wave 100-185 issue cycles while
// LDS reads are in flight, and that the LDS port is the bound: bytes through DMA per MFMA is what counts.  The three taps of
// one kernel row (dy; dx = -1, 0, +1) read the SAME source rows shifted by one token, so their A tile is loaded once: rows
// m0 + dy*W - 1 .. m0 + dy*W + 128 (130 rows + padding to 136, the last one all zero) per (dy, 64-channel chunk), used by three
// stages whose fragment reads go through row index r + dx + 1, or the zero row where the gather table says "padding" (image
// edges, sample boundaries).  K order: (dy, chunk, dx) - the weight tile of a stage is simply addressed at column
// (3*dy + dx + 4... tap)*C + chunk*64 of the ordinary [n][tap*C + c] operand, no re-layout.  DMA pieces per three taps:
// 34 (A) + 120 (W) instead of 96 + 120.  An identity second source (the 1x1 skip of a ResBlock) follows as ordinary stages.
// Tile 128 x 160, 8 waves = 2 K-halves x (2 x 2) waves of 64 x 80 on v_mfma_f32_16x16x32_bf16, the second K-half group one
// stage late (stagger), epilogue shared with the kernel above.
template <int NPASS>
__global__ void __launch_bounds__(512, 1) wd_conv3_kernel(const wd_gemm_args a, const int nbn, const int nbm) {
    constexpr int BM = 128, BN = 160;
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int AR = 136;                 // rows of an A tile (130 used, row 135 = zero row)
    constexpr int A_PL = AR * 128, B_PL = BN * 128;
    constexpr int ABUF = NPL * A_PL, WBUF = NPL * B_PL;
    constexpr int APCS = NPL * (AR / 8), WPCS = NPL * (BN / 8);   // DMA pieces per tile
    constexpr int A_INS = (APCS + 7) / 8, W_INS = (WPCS + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                        // [2][NPL][AR][128 B]
    char* sW = smem + 2 * ABUF;             // [2][NPL][BN][128 B]
    int* s_tab = reinterpret_cast<int*>(smem + 2 * ABUF + 2 * WBUF);  // [9][BM] source row of src[0] per tap, -1 = padding

    const int ntile = nbn * nbm;
    const int nwg = ntile * a.ksplit;
    int wg;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int sidx = wg / ntile;
    wg -= sidx * ntile;
    const int bn_i = wg % nbn, bm_i = wg / nbn;
    const int m0 = bm_i * BM, n0 = bn_i * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3, kh = wave >> 2;
    const int lrow = lane >> 3, lpos = lane & 7;
    const int Wimg = a.slab_rows;

    {
        const int32_t* g0 = a.src[0].gather;
        const int hw_src0 = a.src[0].hw_src;
        for (int idx = tid; idx < 9 * BM; idx += 512) {
            const int t = idx / BM, row = idx - t * BM;
            const int m = m0 + row;
            int v = -1;
            if (m < a.m) {
                const int b = m / a.hw_out, p = m - b * a.hw_out;
                const int g = g0[t * a.hw_out + p];
                if (g >= 0) v = b * hw_src0 + g;
            }
            s_tab[idx] = v;
        }
    }
    const wd_bf16* zline = reinterpret_cast<const wd_bf16*>(wd_zero_line) + lpos * 8;
    const int C0 = a.src[0].c, nc0 = C0 / 64;
    const int nq0 = 9 * nc0;                              // stages of the convolution
    const int nk_all = a.ktot / 64;                       // + C1 / 64 identity stages
    const int k_begin = (int)((long)nk_all * sidx / a.ksplit), k_end = (int)((long)nk_all * (sidx + 1) / a.ksplit);
    const int nk = k_end - k_begin;
    const long rows_src0 = (long)(a.m / a.hw_out) * a.src[0].hw_src;   // rows of the source planes (== a.m for a 'same' conv)

    // stage state (wave-uniform, advanced incrementally - no divisions in the loop): conv stages run (dy, chunk, dx), then the
    // identity stages; G = group id (one A tile per group), kcol = first weight column of the stage
    struct Stg { int conv, dyi, chunk, dxi, G; long kcol; };
    auto stage_at = [&](int q) {
        Stg t;
        if (q < nq0) {
            t.conv = 1;
            t.dyi = q / (3 * nc0);
            const int rem = q - t.dyi * 3 * nc0;
            t.chunk = rem / 3;
            t.dxi = rem - t.chunk * 3;
            t.G = q / 3;
            t.kcol = (long)(t.dyi * 3 + t.dxi) * C0 + t.chunk * 64;
        } else {
            t.conv = 0; t.dyi = 1; t.chunk = q - nq0; t.dxi = 0;
            t.G = nq0 / 3 + t.chunk;
            t.kcol = (long)9 * C0 + (long)t.chunk * 64;
        }
        return t;
    };
    auto next_of = [&](Stg t) {   // the stage after t
        if (t.conv) {
            if (++t.dxi == 3) {
                t.dxi = 0;
                ++t.G;
                if (++t.chunk == nc0) {
                    t.chunk = 0;
                    if (++t.dyi == 3) { t.conv = 0; t.dyi = 1; }
                }
            }
            t.kcol = t.conv ? (long)(t.dyi * 3 + t.dxi) * C0 + t.chunk * 64 : (long)9 * C0;
        } else {
            ++t.chunk; ++t.G; t.kcol += 64;
        }
        return t;
    };
    // static part of this lane's DMA slots: slot i = piece wave + 8 i -> (plane, 8-row piece)
    int a_j[A_INS];                     // tile row of the A slots
    long a_ro0[A_INS], a_ro1[A_INS];    // tile row * ld + swizzled chunk, for src[0] / src[1]
    long w_off[W_INS];                  // n * ktot + swizzled chunk of the W slots, -1 = zero line
#pragma unroll
    for (int i = 0; i < A_INS; ++i) {
        const int pc = wave + 8 * i, rp = pc % (AR / 8);
        a_j[i] = rp * 8 + lrow;
        const int sw = (lpos ^ ((a_j[i] >> 1) & 7)) * 8;
        a_ro0[i] = (long)a_j[i] * a.src[0].ld + sw;
        a_ro1[i] = a.nsrc > 1 ? (long)a_j[i] * a.src[1].ld + sw : 0;
    }
#pragma unroll
    for (int i = 0; i < W_INS; ++i) {
        const int pc = wave + 8 * i, rp = pc % (BN / 8);
        const int row = rp * 8 + lrow, n = n0 + row;
        w_off[i] = (pc < WPCS && n < a.n) ? (long)n * a.ktot + (lpos ^ ((row >> 1) & 7)) * 8 : -1;
    }
    // source pointer of one DMA slot of the A tile of stage-state t / of its W tile (zero line where there is nothing to load)
    auto ptr_A = [&](const Stg& t, int i) -> const wd_bf16* {
        const int pc = wave + 8 * i;
        const int pl = pc / (AR / 8);
        const long srow0 = t.conv ? (long)m0 + (long)(t.dyi - 1) * Wimg - 1 : (long)m0;   // source row of tile row 0 (uniform)
        const long sr = srow0 + a_j[i];
        const bool ok = pc < APCS && a_j[i] < (t.conv ? 130 : 128) && sr >= 0 && sr < (t.conv ? rows_src0 : (long)a.m);
        const wd_bf16* base = t.conv ? (pl ? a.src[0].lo : a.src[0].hi) + srow0 * a.src[0].ld + t.chunk * 64
                                     : (pl ? a.src[1].lo : a.src[1].hi) + srow0 * a.src[1].ld + t.chunk * 64;
        return ok ? base + (t.conv ? a_ro0[i] : a_ro1[i]) : zline;
    };
    auto ptr_W = [&](long kcol, int i) -> const wd_bf16* {
        const int pc = wave + 8 * i;
        const int pl = pc / (BN / 8);
        return w_off[i] >= 0 ? (pl ? a.w_lo : a.w_hi) + kcol + w_off[i] : zline;
    };
    // issue: destination of slot i is fixed by (wave, i): buffer G & 1 of sA / buffer q & 1 of sW
    auto issue_A = [&](int G, int i, const wd_bf16* src) {
        const int pc = wave + 8 * i;
        if (pc >= APCS) return;
        const int pl = pc / (AR / 8), rp = pc - pl * (AR / 8);
        __builtin_amdgcn_global_load_lds((wd_gbl_ptr)src, (wd_lds_ptr)(sA + (G & 1) * ABUF + pl * A_PL + rp * 1024), 16, 0, 0);
    };
    auto issue_W = [&](int q, int i, const wd_bf16* src) {
        const int pc = wave + 8 * i;
        if (pc >= WPCS) return;
        const int pl = pc / (BN / 8), rp = pc - pl * (BN / 8);
        __builtin_amdgcn_global_load_lds((wd_gbl_ptr)src, (wd_lds_ptr)(sW + (q & 1) * WBUF + pl * B_PL + rp * 1024), 16, 0, 0);
    };
    auto fire_A1 = [&](const Stg& t, int i) { issue_A(t.G, i, ptr_A(t, i)); };
    auto fire_W1 = [&](int q, long kcol, int i) { issue_W(q, i, ptr_W(kcol, i)); };
    // slot of everything stage q+1 (state tn) needs that is not in LDS yet: W slots first, then (new group only) the A slots
    auto prefetch_slot = [&](int q, const Stg& tn, bool newA, int sl) {
        if (sl < W_INS) fire_W1(q + 1, tn.kcol, sl);
        else if (newA) fire_A1(tn, sl - W_INS);
    };
    constexpr int NSL = W_INS + A_INS;

    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] = 0.0f;
    Stg cur = stage_at(k_begin < nk_all ? k_begin : 0);
    if (nk > 0) {
#pragma unroll
        for (int i = 0; i < A_INS; ++i) fire_A1(cur, i);
#pragma unroll
        for (int i = 0; i < W_INS; ++i) fire_W1(k_begin, cur.kcol, i);
    }
    const int l15 = lane & 15, lq = lane >> 4;
    const int r0w = (wq >> 1) * 64, c0w = (wq & 1) * 80;
    const int ch = kh * 4 + lq;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the gather table is complete
    // which (tap, row) pairs of this lane's four fragment rows are real pixels: bit tap * 4 + i
    unsigned long long vmask = 0ull;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (s_tab[t * BM + r0w + i * 16 + l15] >= 0) vmask |= 1ull << (t * 4 + i);
    bf16x8 xa[4][NPL], xb[5][NPL];
    auto read16 = [&](int q, const Stg& t) {
        const char* ab = sA + (t.G & 1) * ABUF;
        const char* wb = sW + (q & 1) * WBUF;
        const unsigned vm = t.conv ? (unsigned)(vmask >> ((t.dyi * 3 + t.dxi) * 4)) & 15u : 15u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0w + i * 16 + l15;
            const int j = (vm >> i) & 1u ? r + t.dxi : AR - 1;   // (identity stages: dxi = 0, rows beyond M are zero-filled)
            const int ao = lds_off2(j, ch);
#pragma unroll
            for (int p = 0; p < NPL; ++p) xa[i][p] = *reinterpret_cast<const bf16x8*>(ab + p * A_PL + ao);
        }
#pragma unroll
        for (int t5 = 0; t5 < 5; ++t5) {
            const int bo = lds_off2(c0w + t5 * 16 + l15, ch);
#pragma unroll
            for (int p = 0; p < NPL; ++p) xb[t5][p] = *reinterpret_cast<const bf16x8*>(wb + p * B_PL + bo);
        }
    };
    // the products of one stage; `pf`: issue the (pre-computed) DMA slots of stage q+1 between the four row groups - no address
    // arithmetic between the MFMAs, a 16x16x32 MFMA leaves room for about two vector instructions
    const wd_bf16* pw[W_INS];
    const wd_bf16* pa[A_INS];
    auto mfma16 = [&](bool pf, int q, int Gn, bool newA) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (NPL == 2) {
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][NPL - 1], xb[t][0], acc[i][t], 0, 0, 0);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], xb[t][NPL - 1], acc[i][t], 0, 0, 0);
                }
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], xb[t][0], acc[i][t], 0, 0, 0);
            }
            if (pf) {
#pragma unroll
                for (int sl = i * NSL / 4; sl < (i + 1) * NSL / 4; ++sl) {
                    if (sl < W_INS) issue_W(q + 1, sl, pw[sl]);
                    else if (newA) issue_A(Gn, sl - W_INS, pa[sl - W_INS]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const bool stamp = (a.dbg & 0x100) && blockIdx.x == 0 && lane == 0 && a.ws && a.ksplit == 1;
    unsigned long long* sb = reinterpret_cast<unsigned long long*>(a.ws) + (long)wave * nk * 4;
    if (kh == 1) {
        // late group: DMA of the next stage first, then the previous stage's products, then this stage's fragments
        for (int q = k_begin; q < k_end; ++q) {
            if (stamp) sb[(q - k_begin) * 4 + 0] = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces have landed (the fence of
            __syncthreads();                                  // __syncthreads does not have to wait for loads)
            if (stamp) sb[(q - k_begin) * 4 + 1] = __builtin_amdgcn_s_memtime();
            const Stg tn = next_of(cur);
            if (q + 1 < k_end) {
                const bool newA = tn.G != cur.G;
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) prefetch_slot(q, tn, newA, sl);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q > k_begin) mfma16(false, q, 0, false);
            __builtin_amdgcn_sched_barrier(0);
            if (stamp) sb[(q - k_begin) * 4 + 2] = __builtin_amdgcn_s_memtime();
            read16(q, cur);
            if (stamp) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sb[(q - k_begin) * 4 + 3] = __builtin_amdgcn_s_memtime();
            }
            cur = tn;
        }
        if (nk > 0) mfma16(false, 0, 0, false);
    } else {
        for (int q = k_begin; q < k_end; ++q) {
            if (stamp) sb[(q - k_begin) * 4 + 0] = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // stage q has landed, the buffers of stage q-1 are free
            if (stamp) sb[(q - k_begin) * 4 + 1] = __builtin_amdgcn_s_memtime();
            const bool more = q + 1 < k_end;
            const Stg tn = next_of(cur);
            const bool newA = more && tn.G != cur.G;
            read16(q, cur);
            if (more) {  // addresses of the next stage's pieces, in the shadow of the fragment reads
#pragma unroll
                for (int i = 0; i < W_INS; ++i) pw[i] = ptr_W(tn.kcol, i);
                if (newA) {
#pragma unroll
                    for (int i = 0; i < A_INS; ++i) pa[i] = ptr_A(tn, i);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (stamp) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sb[(q - k_begin) * 4 + 2] = __builtin_amdgcn_s_memtime();
            }
            mfma16(more, q, tn.G, newA);
            if (stamp) sb[(q - k_begin) * 4 + 3] = __builtin_amdgcn_s_memtime();
            cur = tn;
        }
    }
    // ---- partial tiles -> fp32 LDS image (the two K-halves summed in a fixed order), then the shared epilogue
    constexpr int LDE = BN + 4;
    float* ep = reinterpret_cast<float*>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int hh = 0; hh < 2; ++hh) {
        if (kh == hh) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* pe = ep + (r0w + i * 16 + 4 * lq + r) * LDE + c0w + t * 16 + l15;
                        *pe = (hh == 0) ? acc[i][t][r] : *pe + acc[i][t][r];
                    }
        }
        __syncthreads();
    }
    wd_epilogue_tail<BM, BN, 512>(a, ep, m0, n0, tid, sidx);
}

template <int NPASS>
int launch_conv3(const wd_gemm_args& a, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int loop_smem = 2 * NPL * (136 + 160) * 128 + 9 * 128 * 4;
    constexpr int red_smem = 128 * (160 + 4) * 4 + WD_STAT_SCRATCH;
    constexpr int smem = loop_smem > red_smem ? loop_smem : red_smem;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_conv3_kernel<NPASS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nbn = (a.n + 159) / 160, nbm = (a.m + 127) / 128;
    {
        WdLaunchScope scope(WD_CLS_GEMM, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
        hipLaunchKernelGGL((wd_conv3_kernel<NPASS>), dim3(nbn * nbm * a.ksplit), dim3(512), smem, st, a, nbn, nbm);
    }
    if (a.ksplit > 1) {
        return launch_reduce_any<128, 160>(a, st);
    }
    return wd_check_launch();
}

// ======================================================================================================
// v8: a RING kernel (32-deep stages, four LDS buffers, one raw barrier per stage, counted vmcnt) with the DMA moved to FOUR
// DEDICATED LOADER WAVES (12 waves per workgroup, three per SIMD at <= 168 registers).  Opt-in (WDIFF_GEMM_V8=1): within
// +-5 % of wd_gemm2_kernel on every layer shape (same box: 88-90 vs 85-86 us on the 320 x 3 x 3 convolution at B = 64, 19.9-20.3 vs
// 20.3-20.5 us on the 5-stage 1x1 layers).  What it was built to find out (round 2): with the parts of wd_gemm2_kernel switched off
// one at a time its MFMA, fragment-read and DMA costs ADD instead of overlapping - a wave that issues buffer_load ... lds
// cannot feed the MFMA pipe meanwhile.  Here the eight compute waves (32 x 80 blocks of 2 x 5 tiles of
// v_mfma_f32_16x16x32_bf16, 40 accumulator registers, which pays for TWO fragment sets: the fragments of stage k+1 are read
// while stage k is multiplied, pass-major so that dependent MFMAs are nine apart) only read fragments and multiply; loader
// wave L (one per SIMD) fetches, per stage, A pieces 2L, 2L+1 and W pieces L, L+4 of every plane plus one of the four leftover
// W pieces - nine pieces (five with one plane), so `s_waitcnt vmcnt(9)` at the top of its step leaves exactly the youngest
// stage in flight.  Result of the same switch-off experiment here: the loaders are NOT the limit (compute waves alone: 78 of
// 85 us); the MFMAs alone run at 14 cycles each (near the pipe's 16), but the barrier every 32-deep stage and the fragment
// waits stretch a stage from ~850 to ~1800 cycles - two waves per SIMD that meet at the same barrier cannot cover each other.
// Orders: RAW - the loaders wait for their own pieces of stage k+1, then the barrier every wave passes, then the compute
// waves read that buffer; WAR - the buffer refilled after barrier k held stage k-1, whose fragment reads the compute waves
// consumed (lgkmcnt) before the MFMAs of stage k-1, i.e. before they arrived at barrier k.
template <int NPASS>
__global__ void __launch_bounds__(768, 1) wd_gemm8_kernel(const wd_gemm_args a, const int nbn, const int nbm) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, BN = 160, BKS = 32, NST = 4;
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int A_PL = BM * 64, B_PL = BN * 64;
    constexpr int STAGE = NPL * (A_PL + B_PL);
    constexpr int RING = NST * STAGE;
    constexpr int TAB_OFF = RING;
    constexpr uint32_t WD_OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_tab = reinterpret_cast<int*>(smem + TAB_OFF);  // [ntaps0][BM] source row of src[0] per tap, -1 = zero row

    const int ntile = nbn * nbm;
    const int nwg = ntile * a.ksplit;
    int wg;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int sidx = wg / ntile;
    wg -= sidx * ntile;
    const int bn_i = wg % nbn, bm_i = wg / nbn;
    const int m0 = bm_i * BM, n0 = bn_i * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    {
        const int nt0 = a.src[0].ntaps;
        const int32_t* g0 = a.src[0].gather;
        const int hw_src0 = a.src[0].hw_src;
        const int Wimg = a.slab_rows;  // > 0: 3x3 / pad 1 / stride 1 over images Wimg wide: the table is computed
        for (int idx = tid; idx < nt0 * BM; idx += 768) {
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
    __syncthreads();

    const int nk_all = a.ktot / BKS;
    const int k_begin = (int)((long)nk_all * sidx / a.ksplit), k_end = (int)((long)nk_all * (sidx + 1) / a.ksplit);
    const int nk = k_end - k_begin;

    f32x4 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] = 0.0f;
    const int l15 = lane & 15, lq = lane >> 4;
    const int cw = wave & 7;
    const int r0w = (cw >> 1) * 32, c0w = (cw & 1) * 80;

    if (wave >= 8) {
        // =========================== loader wave L: every DMA piece of the workgroup ===========================
        const int L = wave - 8;
        const int lrow = lane >> 2, lpos = lane & 3;              // a DMA piece = 16 rows x 64 bytes
        const int sw = (lpos ^ ((0 - (lane >> 4)) & 3)) * 8;      // source chunk (elements) that lands at position lpos of row lrow
        auto make_srd = [](const wd_bf16* p) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<wd_bf16*>(p), 0, 0x7FFFFFF0, 0x00020000);
        };
        // W pieces L, L + 4 (both planes) and leftover item L: piece 8 + (L >> (NPL - 1)) of plane L & (NPL - 1); one plane: loaders
        // 0, 1 take pieces 8, 9 and loaders 2, 3 re-fetch pieces 8, 9 as well (same bytes to the same place: harmless, and every
        // loader issues the same number of pieces)
        const int x_piece = NPL == 2 ? 8 + (L >> 1) : 8 + (L & 1), x_plane = NPL == 2 ? (L & 1) : 0;
        uint32_t b_voff[3];
        {
            const int pcs[3] = {L, L + 4, x_piece};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int n = n0 + pcs[i] * 16 + lrow;
                b_voff[i] = n < a.n ? (uint32_t)(((long)n * a.ktot + sw) * 2) : WD_OOB;
            }
        }
        const __amdgpu_buffer_rsrc_t srd_w_hi = make_srd(a.w_hi), srd_w_lo = make_srd(a.w_lo ? a.w_lo : a.w_hi);
        const __amdgpu_buffer_rsrc_t srd_w_x = (x_plane ? srd_w_lo : srd_w_hi);
        int s = 0, tap = 0, kc = 0;
        int cur_ld = a.src[0].ld, cur_cpt = a.src[0].c / BKS, cur_nt = a.src[0].ntaps;
        const wd_bf16* cur_hi = a.src[0].hi;
        const wd_bf16* cur_lo = a.src[0].lo;
        {
            const int cpt = a.src[0].c / BKS, n0st = a.src[0].ntaps * cpt;
            if (k_begin < n0st) {
                tap = k_begin / cpt;
                kc = k_begin - tap * cpt;
            } else {
                s = 1;
                kc = k_begin - n0st;
                cur_hi = a.src[1].hi;
                cur_lo = a.src[1].lo;
                cur_ld = a.src[1].ld;
                cur_cpt = a.src[1].c / BKS;
                cur_nt = a.src[1].ntaps;
            }
        }
        __amdgpu_buffer_rsrc_t srd_a_hi = make_srd(cur_hi), srd_a_lo = make_srd(cur_lo ? cur_lo : cur_hi);
        uint32_t a_voff[2];
        auto locate = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (2 * L + i) * 16 + lrow;
                int r;
                if (s == 0) r = s_tab[tap * BM + row];
                else r = (m0 + row < a.m) ? m0 + row : -1;  // src[1] is an identity source (1x1 skip)
                a_voff[i] = r >= 0 ? (uint32_t)r * (uint32_t)(cur_ld * 2) + (uint32_t)(sw * 2) : WD_OOB;
            }
        };
        locate();
        int soff_a = kc * BKS * 2, soff_b = k_begin * BKS * 2;
        int wr_off = 0;
        auto issue_stage = [&]() {
            char* sbase = smem + wr_off;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a_hi, (wd_lds_ptr)(sbase + (2 * L + i) * 1024), 16, a_voff[i], soff_a, 0, 0);
                if (NPL == 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a_lo, (wd_lds_ptr)(sbase + A_PL + (2 * L + i) * 1024), 16, a_voff[i],
                                                             soff_a, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w_hi, (wd_lds_ptr)(sbase + NPL * A_PL + (L + 4 * i) * 1024), 16, b_voff[i],
                                                         soff_b, 0, 0);
                if (NPL == 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w_lo, (wd_lds_ptr)(sbase + NPL * A_PL + B_PL + (L + 4 * i) * 1024), 16,
                                                             b_voff[i], soff_b, 0, 0);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w_x, (wd_lds_ptr)(sbase + NPL * A_PL + x_plane * B_PL + x_piece * 1024), 16,
                                                     b_voff[2], soff_b, 0, 0);
            // the issue state moves on by one stage
            soff_b += BKS * 2;
            soff_a += BKS * 2;
            wr_off += STAGE;
            if (wr_off == RING) wr_off = 0;
            ++kc;
            if (kc == cur_cpt) {
                kc = 0;
                soff_a = 0;
                ++tap;
                if (tap == cur_nt) {
                    tap = 0;
                    ++s;
                    if (s < a.nsrc) {
                        cur_hi = a.src[1].hi;
                        cur_lo = a.src[1].lo;
                        cur_ld = a.src[1].ld;
                        cur_cpt = a.src[1].c / BKS;
                        cur_nt = a.src[1].ntaps;
                        srd_a_hi = make_srd(cur_hi);
                        srd_a_lo = make_srd(cur_lo ? cur_lo : cur_hi);
                    }
                }
                if (s < a.nsrc) locate();
            }
        };
        auto wait_keep1 = [&]() {  // all but the youngest issued stage have landed
            if (NPL == 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        };
        __builtin_amdgcn_s_setprio(2);
        int issued = 0;
        for (; issued < 3 && issued < nk; ++issued) issue_stage();
        if (nk > 0) {  // stage 0 landed
            if (issued >= 2) {
                if (issued == 3) {
                    if (NPL == 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                } else {
                    wait_keep1();
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        int kit = 0;
        for (; kit + 3 < nk; ++kit) {  // steady state: stage kit + 1 landed -> barrier -> fetch stage kit + 3
            wait_keep1();
            __builtin_amdgcn_s_barrier();
            issue_stage();
        }
        for (; kit < nk; ++kit) {      // drain: nothing left to fetch
            if (kit + 1 < nk) {
                if (kit + 2 < nk) wait_keep1();
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // =========================== compute wave: fragments and products only ===========================
        const int ao0 = lds_off16(r0w + l15, lq), bo0 = NPL * A_PL + lds_off16(c0w + l15, lq);
        bf16x8 fa0[2][NPL], fb0[5][NPL], fa1[2][NPL], fb1[5][NPL];
        // (16 rows further down the XOR key of lds_off16 differs - rows r0w + 16 i, c0w + 16 t: (row >> 2) & 3 advances by 4 = 0 mod 4,
        // so the key repeats and a wave's fragments are one lane offset + immediates)
        auto read_a = [&](auto& fa_, const char* base, int i) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) fa_[i][p] = *reinterpret_cast<const bf16x8*>(base + ao0 + p * A_PL + i * 1024);
        };
        auto read_b = [&](auto& fb_, const char* base, int t) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) fb_[t][p] = *reinterpret_cast<const bf16x8*>(base + bo0 + p * B_PL + t * 1024);
        };
        // The three products of a tile (lo*hi, hi*lo, hi*hi) go to the same accumulator: issued back to back each waits for the
        // previous one's result (measured: 30.7 cycles per MFMA in this loop with the DMA switched off, against 16 at the pipe's
        // rate).  So the step runs pass-major: the first product of all ten tiles, then the second, then the third - nine
        // independent MFMAs between two dependent ones; per tile the order of the three products, hence the result, is unchanged.
        auto mfma_pass = [&](auto& fa_, auto& fb_, int pass, int g) {
            const int i = g / 5, t = g % 5;
            if (NPL == 2) {
                if (pass == 0) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[i][NPL - 1], fb_[t][0], acc[i][t], 0, 0, 0);
                else if (pass == 1) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[i][0], fb_[t][NPL - 1], acc[i][t], 0, 0, 0);
                else acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[i][0], fb_[t][0], acc[i][t], 0, 0, 0);
            } else {
                if (pass == 2) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[i][0], fb_[t][0], acc[i][t], 0, 0, 0);
            }
        };
        if (nk > 0) {
            __builtin_amdgcn_s_barrier();  // stage 0 has landed
#pragma unroll
            for (int i = 0; i < 2; ++i) read_a(fa0, smem, i);
#pragma unroll
            for (int t = 0; t < 5; ++t) read_b(fb0, smem, t);
        }
        int rd_off = STAGE;  // ring offset of the stage whose fragments are read next (stage kit + 1)
        auto step = [&](bool have1, auto& ma, auto& mb, auto& ra, auto& rb) {
            __builtin_amdgcn_s_barrier();
            const char* nb = smem + rd_off;
            // Per-step stamps of this loop (clock64, one compute wave): barrier wait ~100-250 cycles, the first ten MFMAs of the
            // step ~700 (the two compute waves of a SIMD together: 36 cycles per MFMA), the other twenty ~500 (14 per MFMA, the
            // pipe's rate), 1490-1620 cycles per step - whether the 14 fragment reads of the next stage are issued at once, as here,
            // or spread over the step (which costs registers: 168 + spills instead of 159); the loaders wait ~800 cycles per step
            // at the barrier, i.e. they are not the limit.
#pragma unroll
            for (int pass = (NPL == 2 ? 0 : 2); pass < 3; ++pass) {
#pragma unroll
                for (int g = 0; g < 10; ++g) {
                    // the 14 fragment reads of the next stage, two per slot of the first pass (the only pass with one plane)
                    if (have1 && pass == (NPL == 2 ? 0 : 2)) {
                        if (g < 2) read_a(ra, nb, g);
                        else if (g < 7) read_b(rb, nb, g - 2);
                    }
                    mfma_pass(ma, mb, pass, g);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            rd_off += STAGE;
            if (rd_off == RING) rd_off = 0;
        };
        for (int kit = 0; kit < nk; kit += 2) {  // (static roles of the two fragment sets: unrolled by two)
            step(kit + 1 < nk, fa0, fb0, fa1, fb1);
            if (kit + 1 < nk) step(kit + 2 < nk, fa1, fb1, fa0, fb0);
        }
    }

    // ---- epilogue: accumulators -> fp32 LDS image of the tile, then the shared epilogue (all twelve waves)
    constexpr int LDE = BN + 4;
    float* ep = reinterpret_cast<float*>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave < 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(r0w + i * 16 + 4 * lq + r) * LDE + c0w + t * 16 + l15] = acc[i][t][r];
    }
    __syncthreads();
    wd_epilogue_tail<BM, BN, 768>(a, ep, m0, n0, tid, sidx);
#endif
}

template <int NPASS>
int launch8(const wd_gemm_args& a, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int loop_smem = 4 * NPL * (128 + 160) * 64 + 9 * 128 * 4;
    constexpr int red_smem = 128 * (160 + 4) * 4 + WD_STAT_SCRATCH;
    constexpr int smem = loop_smem > red_smem ? loop_smem : red_smem;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemm8_kernel<NPASS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nbn = (a.n + 159) / 160, nbm = (a.m + 127) / 128;
    {
        WdLaunchScope scope(WD_CLS_GEMM, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
        hipLaunchKernelGGL((wd_gemm8_kernel<NPASS>), dim3(nbn * nbm * a.ksplit), dim3(768), smem, st, a, nbn, nbm);
    }
    if (a.ksplit > 1) return launch_reduce_any<128, 160>(a, st);
    return wd_check_launch();
}

#endif  // WDIFF_EXPERIMENTAL

}  // namespace

int wd_gemm_launch_reduce(const wd_gemm_args& a, hipStream_t st, int bm) {
    if (bm == 64) {
        if (reduce_flat_ok(a)) return launch_reduce_any<64, 64>(a, st);
        const int cpg = a.stat_part ? a.stat_cpg : 1;
        if (40 % cpg == 0 && a.n % 40 == 0) return launch_reduce<64, 40>(a, st);
        if (32 % cpg == 0 && a.n % 32 == 0) return launch_reduce<64, 32>(a, st);
        return launch_reduce<64, 64>(a, st);
    }
    return launch_reduce_any<128, 160>(a, st);
}

// the K cut wd_gemm picks by itself: fewer than 128 tiles (or exactly 128 on a long K loop) and at least 8 stages of 64
static int wd_auto_ksplit(const int tile, const int m, const int n, const int nk64, const long ws_floats) {
    const int bn = tile % 1000, bm = tile / 1000;
    const long tiles = (long)((m + bm - 1) / bm) * ((n + bn - 1) / bn);
    if (!((tiles < 128 || (tiles == 128 && nk64 >= 32)) && nk64 >= 8)) return 1;
    long sp = 256 / tiles;
    if (sp > nk64 / 4) sp = nk64 / 4;
    // reductions over the token dimension (the weight-gradient GEMMs of the training step: K = 4096 ... 16384 tokens, a handful of
    // output tiles) are cut as finely as it takes to put a workgroup on every CU; the short-K layers of the denoising step keep <= 8
    static const long cap_env = getenv("WDIFF_KSPLIT_CAP") ? atol(getenv("WDIFF_KSPLIT_CAP")) : WD_MAX_KSPLIT;
    const long cap = nk64 >= 64 ? cap_env : 8;
    if (sp > cap) sp = cap;
    while (sp > 1 && sp * m * n > ws_floats) --sp;
    return sp > 1 ? (int)sp : 1;
}

static int wd_auto_tile(const int m, const int n, const int nk64, const bool can_split_base, const long ws_floats) {
    // 128x160 when N allows it (A is re-read only N/160 times), else 128x64; when that leaves most CUs idle either
    // cut K across workgroups (needs a workspace and enough K per slice) or fall back to 64x64 tiles.
    int tile = (n % 160 == 0) ? 128160 : 128064;
    const long tiles = (long)((m + 127) / 128) * ((n + (tile % 1000) - 1) / (tile % 1000));
    const bool can_split = can_split_base && nk64 >= 8 && (long)2 * m * n <= ws_floats;
    if (tiles < 128 && !can_split && m > 64) tile = 64064;
    return tile;
}

extern "C" int wd_gemm_experimental(void) {  // 1: built with -DWDIFF_EXPERIMENTAL (slab / row-shared-taps / ring kernels present)
#ifdef WDIFF_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}

extern "C" int wd_gemm_args_bytes(void) { return (int)sizeof(wd_gemm_args); }  // (callers that mirror the struct check their layout)

extern "C" int wd_gemm_auto_ksplit(int m, int n, int ktot, int64_t ws_floats) {
    if (m <= 0 || n <= 0 || ktot <= 0 || ktot % BK2 || ws_floats <= 0) return 1;
    const int nk64 = ktot / BK2;
    return wd_auto_ksplit(wd_auto_tile(m, n, nk64, true, (long)ws_floats), m, n, nk64, (long)ws_floats);
}

extern "C" int wd_gemm(const wd_gemm_args* pa, void* stream) {
    if (!pa) return WD_EINVAL;
    wd_gemm_args a = *pa;
    if (a.ksplit < 0 || a.ksplit > WD_MAX_KSPLIT) return WD_EINVAL;
    if (a.w_layout == 3) {
        // fragment-major weights (wd_gemm_pack_w): the 64 x 320 weights-to-registers kernel only
        if (a.n % 80 || a.act == WD_ACT_GEGLU || (a.tile && a.tile != 64320 && a.tile != 128160 && a.tile != 64080) || a.ktot % 64)
            return WD_EINVAL;
        if (a.tile == 0) a.tile = (a.a32 || a.ln_gamma) ? 64320 : 128160;
        if (a.n % (a.tile % 1000)) return WD_EINVAL;
        if (a.tile == 64080) a.ksplit = 1;  // (all of K inside the workgroup: wd_gemmq_kernel; wd_gemmq_applies() is checked at the dispatch)
    }
    if (a.w_ngroups > 1) {
        // weight groups: the 64 x 320 weights-to-registers kernel only, a tile inside one run of rows, no K cut
        if (a.w_layout != 3 || a.tile != 64320 || a.ksplit > 1 || a.hw_out % (64 * a.w_ngroups) || a.w_group_stride <= 0 || (a.w_group_stride & 7))
            return WD_EINVAL;
        a.ksplit = 1;
    }
    if (a.stat_part) {
        // fused GroupNorm statistics need row panels that tile the samples and whole groups inside a column tile
        if (a.stat_cpg <= 0 || a.n % a.stat_cpg || a.act == WD_ACT_GEGLU) return WD_EINVAL;
        const int bn = (a.tile ? a.tile % 1000 : (a.n % 160 == 0 ? 160 : 64));
        const int bm = (a.tile ? a.tile / 1000 : 128);
        if (!((a.hw_out % bm == 0) || (bm % a.hw_out == 0 && bm / a.hw_out <= WD_STAT_MAXNS))) return WD_EINVAL;
        if ((bm != 128 && a.w_layout != 3) || bn % a.stat_cpg) return WD_EINVAL;
        if ((a.out_ld | a.rowvec_ld | a.resid_ld | a.out_pl_ld) & 3) return WD_EINVAL;  // vector epilogue only
        if (a.tile == 0) a.tile = bm * 1000 + bn;  // keep 128-row panels (no 64x64 fallback)
    }
    if (a.ksplit > 1 && (!a.ws || a.act == WD_ACT_GEGLU || a.w_layout == 1)) return WD_EINVAL;
    if (a.ln_gamma) {
        // row LayerNorm of the result: the tile must hold whole rows and the epilogue must run in the GEMM launch itself
        if (a.w_layout != 3 || (a.tile && a.tile != 64320) || a.n != 320 || a.ksplit > 1 || !a.ln_beta || !a.out_hi || (a.out_pl_ld & 3) ||
            ((reinterpret_cast<uintptr_t>(a.ln_gamma) | reinterpret_cast<uintptr_t>(a.ln_beta)) & 15) ||
            ((reinterpret_cast<uintptr_t>(a.out_hi) | reinterpret_cast<uintptr_t>(a.out_lo)) & 7) || a.act == WD_ACT_GEGLU)
            return WD_EINVAL;
        a.tile = 64320;
        a.ksplit = 1;
    }
    if (a.gn_gamma) {
        // GroupNorm of the result in the combine launch (wd_reduce_gn_tile): whole (sample, group) blocks per 64 x 40 tile
        if (!a.gn_beta || !a.stat_part || !a.out_hi || (!a.ws && a.tile != 64080) || a.hw_out != 64 || a.m % 64 || a.n % 160 || a.gn_cpg <= 0 ||
            40 % a.gn_cpg || a.stat_cpg <= 0 || a.gn_cpg % a.stat_cpg || a.act != WD_ACT_NONE || a.resid_rows || a.w_layout == 1)
            return WD_EINVAL;
        if (((a.out_ld | a.rowvec_ld | a.resid_ld | a.out_pl_ld) & 3) ||
            ((reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.rowvec) | reinterpret_cast<uintptr_t>(a.resid) |
              reinterpret_cast<uintptr_t>(a.out_f32) | reinterpret_cast<uintptr_t>(a.ws) | reinterpret_cast<uintptr_t>(a.gn_gamma) |
              reinterpret_cast<uintptr_t>(a.gn_beta)) & 15) ||
            ((reinterpret_cast<uintptr_t>(a.out_hi) | reinterpret_cast<uintptr_t>(a.out_lo)) & 7))
            return WD_EINVAL;
        a.tickets = nullptr;  // the norm lives in the combine launch
    }
    if (a.w_layout == 1) {
        a.ksplit = 1;
        a.tickets = nullptr;
    }
    if (a.nsrc < 1 || a.nsrc > 2 || (a.npass != 1 && a.npass != 3)) return WD_EINVAL;
    if (a.m <= 0 || a.n <= 0 || a.hw_out <= 0 || !a.w_hi || (a.npass == 3 && !a.w_lo)) return WD_EINVAL;
    if (!a.out_f32 && !a.out_hi) return WD_EINVAL;
    long k = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        const wd_src& q = a.src[s];
        if (s == 0 && a.a32) {
            // src[0] staged from the fp32 map with the consumer's GroupNorm: the weights-to-registers kernel, 64-row tiles inside one sample
            if (a.w_layout != 3 || a.npass != 3 || (a.tile && a.tile != 64320) || a.n % 320 || a.hw_out % 64 || !a.a32_part || !a.a32_gamma ||
                !a.a32_beta || a.a32_nchunk <= 0 || a.a32_pcpg <= 0 || a.a32_cpg <= 0 || a.a32_cpg % a.a32_pcpg || q.c % a.a32_cpg ||
                q.c > 1024 || a.a32_ld < q.c || (a.a32_ld & 3) || (reinterpret_cast<uintptr_t>(a.a32) & 15) || q.hw_src <= 0)
                return WD_EINVAL;
        } else if (!q.hi || (a.npass == 3 && !q.lo)) return WD_EINVAL;
        if (q.c <= 0 || q.c % BK || q.ld % 8 || q.ntaps < 1) return WD_EINVAL;
        if (!q.gather && q.ntaps != 1) return WD_EINVAL;
        if (q.gather && q.hw_src <= 0) return WD_EINVAL;
        k += (long)q.ntaps * q.c;
    }
    if (k != a.ktot) return WD_EINVAL;
    {   // the main kernel addresses every operand plane with 31-bit byte offsets (buffer_load ... lds)
        const long lim = 0x7FFFFFF0L;
        if ((long)a.n * a.ktot * 2 >= lim) return WD_EINVAL;
        for (int s = 0; s < a.nsrc; ++s) {
            const wd_src& q = a.src[s];
            const long rows = q.gather ? (long)((a.m + a.hw_out - 1) / a.hw_out) * q.hw_src : (long)a.m;
            if ((s == 0 && a.a32) ? rows * a.a32_ld * 4 >= lim : rows * q.ld * 2 >= lim) return WD_EINVAL;
        }
    }
    if (a.act == WD_ACT_GEGLU && (a.n % 64 || a.tile == 0)) return WD_EINVAL;  // the tile fixes the x|gate packing
    if (a.rowvec && a.rowvec_ld <= 0) return WD_EINVAL;
    if (a.resid && a.resid_ld <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    if (a.w_layout == 3) {
        bool ok = a.src[0].ntaps <= 9 && (a.nsrc == 1 || (a.src[1].gather == nullptr && a.src[1].ntaps == 1));
        for (int s = 0; s < a.nsrc; ++s) ok = ok && (a.src[s].c % 64 == 0);
        if (!ok) return WD_EINVAL;
        const wd_src& q0 = a.src[0];
        const bool same3 = q0.ntaps == 9 && q0.gather && a.slab_rows > 0 && q0.hw_src == a.hw_out && a.hw_out % a.slab_rows == 0;
        // (slab_rows: the image width of a 3x3 / pad 1 / stride 1 source, as for w_layout 2; tile 64080 also takes the stride-2 table
        // of a Downsample - hw_src == 4 hw_out -, slab_rows then is the OUTPUT width)
        const bool down3 = a.tile == 64080 && q0.ntaps == 9 && q0.gather && a.slab_rows > 0 && q0.hw_src == 4 * a.hw_out;
        if (!same3 && !down3) a.slab_rows = 0;
        a.tickets = nullptr;
        if (a.tile == 64080) {
            if (!wd_gemmq_applies(a)) return WD_EINVAL;
            if (a.gn_gamma && (40 % a.gn_cpg || 80 % a.gn_cpg)) return WD_EINVAL;
            return wd_gemmq_launch(a, st);
        }
        const int nk64 = a.ktot / 64;
        if (a.ksplit == 0) a.ksplit = a.ws ? wd_auto_ksplit(a.tile, a.m, a.n, nk64, a.ws_floats) : 1;
        if (a.ksplit > 1 && (nk64 < a.ksplit || (long)a.ksplit * a.m * a.n > a.ws_floats)) a.ksplit = 1;
        if (a.gn_gamma && a.ksplit <= 1) return WD_EINVAL;
        return wd_gemmw_launch(a, st);
    }
#ifndef WDIFF_EXPERIMENTAL
    if (a.w_layout == 1) return WD_EINVAL;  // (the slab kernel is an experimental build option)
#else
    if (a.w_layout == 1) {
        // slab-order weights: v3 kernel only.  Contract: c % 32 == 0 (checked above), src[1] identity, slab_rows set.
        if (a.src[0].ntaps > 9 || a.slab_rows <= 0) return WD_EINVAL;
        if (a.nsrc == 2 && (a.src[1].gather || a.src[1].ntaps != 1)) return WD_EINVAL;
        int t3 = a.tile;
        if (t3 == 0) t3 = (a.act == WD_ACT_GEGLU || a.n % 160) ? 128064 : 128160;
        if (a.act == WD_ACT_GEGLU && a.n % (t3 % 1000)) return WD_EINVAL;
        if (a.slab_rows > 192) return WD_EINVAL;
        static const int ks3_env = getenv("WDIFF_GEMM_KS") ? atoi(getenv("WDIFF_GEMM_KS")) : 2;
#define WD_DISPATCH3(BM_, BN_)                                                                                   \
    if (ks3_env == 1) return a.npass == 3 ? launch3<BM_, BN_, 3, 192, 1>(a, st) : launch3<BM_, BN_, 1, 192, 1>(a, st); \
    return a.npass == 3 ? launch3<BM_, BN_, 3, 192, 2>(a, st) : launch3<BM_, BN_, 1, 192, 2>(a, st)
        switch (t3) {
            case 128064: WD_DISPATCH3(128, 64);
            case 128128: WD_DISPATCH3(128, 128);
            case 128160: WD_DISPATCH3(128, 160);
            default: return WD_EINVAL;
        }
#undef WD_DISPATCH3
    }
#endif
    bool conv3 = false;
    if (a.w_layout == 2) {
        // row-shared taps (wd_conv3_kernel): 3x3 / pad 1 / stride 1 over images of width slab_rows, optional identity source.
        // The operand layout is the ordinary one, so this is a hint: shapes the kernel does not cover take the generic path.
        const wd_src& q0 = a.src[0];
        static const int conv3_mode = getenv("WDIFF_CONV3") ? atoi(getenv("WDIFF_CONV3")) : 0;
#ifdef WDIFF_EXPERIMENTAL
        const bool conv3_env = conv3_mode != 0 || (a.dbg & 0x1000);  // (dbg 0x1000: the parity tests pick the kernel)
#else
        const bool conv3_env = false;
        (void)conv3_mode;
#endif
        const bool same3 = q0.ntaps == 9 && q0.gather && a.slab_rows > 0 && q0.hw_src == a.hw_out && a.hw_out % a.slab_rows == 0;
        conv3 = conv3_env && same3 && q0.c % 64 == 0 && a.act != WD_ACT_GEGLU && (a.tile == 0 || a.tile == 128160) &&
                (a.nsrc == 1 || (!a.src[1].gather && a.src[1].ntaps == 1 && a.src[1].c % 64 == 0));
        if (conv3) a.tile = 128160;
        a.w_layout = 0;
        // generic kernel: slab_rows > 0 now means "src[0] is a 3x3 / pad 1 / stride 1 window over images slab_rows wide" - the
        // source-row table of a panel is then computed, not looked up (saves the dependent global loads of the prologue)
        static const bool arith_env = getenv("WDIFF_GEMM_ARITH_TAB") ? atoi(getenv("WDIFF_GEMM_ARITH_TAB")) != 0 : true;
        if (!same3 || !arith_env) a.slab_rows = 0;
    } else if (a.w_layout == 0) {
        a.slab_rows = 0;
    } else if (a.w_layout != 0) {
        return WD_EINVAL;
    }
    bool v2ok = (a.ktot % BK2 == 0) && !getenv("WDIFF_GEMM_V1");
    for (int s = 0; s < a.nsrc; ++s) v2ok = v2ok && (a.src[s].c % BK2 == 0);
    v2ok = v2ok && a.src[0].ntaps <= 9 && (a.nsrc == 1 || (a.src[1].gather == nullptr && a.src[1].ntaps == 1));
    const int nk64 = a.ktot / BK2;
    int tile = a.tile;
    if (tile == 0)
        tile = wd_auto_tile(a.m, a.n, nk64, v2ok && a.ws && a.ksplit == 0 && a.act != WD_ACT_GEGLU, a.ws_floats);
    if (a.ksplit == 0) {
        // (half-filled chip - exactly 128 tiles, e.g. batch 32 -: a two-way cut pays on the long-K convolutions only)
        a.ksplit = (v2ok && a.ws && a.act != WD_ACT_GEGLU) ? wd_auto_ksplit(tile, a.m, a.n, nk64, a.ws_floats) : 1;
    }
    if (a.act == WD_ACT_GEGLU && a.n % (tile % 1000)) return WD_EINVAL;
    if (a.ksplit > 1 && (!v2ok || nk64 < a.ksplit || (long)a.ksplit * a.m * a.n > a.ws_floats)) a.ksplit = 1;
    bool use_v4 = false;
    if (tile == 128160) {
        // two co-resident 4-wave workgroups per CU for the layers without fused statistics (see wd_gemm4_kernel)
        // WDIFF_GEMM_V4: 0 never, 1 always (where legal), default 2 = when every CU gets at least two workgroups
        static const int v4_env = getenv("WDIFF_GEMM_V4") ? atoi(getenv("WDIFF_GEMM_V4")) : 2;
        const long wgs = (long)((a.m + 127) / 128) * ((a.n + 159) / 160) * a.ksplit;
        use_v4 = v2ok && !a.stat_part && (v4_env == 1 || (v4_env == 2 && wgs >= 512) || (a.dbg & 0x400));
    }
    {
        // in-launch split-K combine (tickets): v2 kernels only, vector epilogue (16-byte aligned everything), whole column tiles
        // Measured on the 4x16-level convolutions (M = 4096, 64 tiles x 4 slices of 80 KB): 46.0 us against 33.8 us for GEMM +
        // combine launch - the slabs of a tile are combined by ONE workgroup (64 busy CUs) behind an agent-scope release /
        // acquire, the combine launch spreads the same bytes over 512 workgroups - so the default is the second launch
        // (the guide's rule: in-launch pays up to a few tens of KB of slabs per tile).  dbg 0x2000 forces it (parity tests).
        static const bool fuse_env = getenv("WDIFF_GEMM_FUSE_COMBINE") ? atoi(getenv("WDIFF_GEMM_FUSE_COMBINE")) != 0 : false;
        const int bn = tile % 1000, bm = tile / 1000;
        const long ntile = (long)((a.m + bm - 1) / bm) * ((a.n + bn - 1) / bn);
        const bool aligned = (a.n & 3) == 0 && a.n % bn == 0 && (((a.out_ld | a.rowvec_ld | a.resid_ld | a.out_pl_ld) & 3) == 0) &&
                             (((reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.rowvec) |
                                reinterpret_cast<uintptr_t>(a.resid) | reinterpret_cast<uintptr_t>(a.out_f32) |
                                reinterpret_cast<uintptr_t>(a.ws)) & 15) == 0) &&
                             (((reinterpret_cast<uintptr_t>(a.out_hi) | reinterpret_cast<uintptr_t>(a.out_lo)) & 7) == 0);
        if (!((fuse_env || (a.dbg & 0x2000)) && a.tickets && a.ksplit > 1 && v2ok && !use_v4 && !conv3 && aligned && ntile <= a.ntickets))
            a.tickets = nullptr;
    }
    // the GroupNorm epilogue exists in the combine launch of the 128 x 160 kernels only: anything else is an error, not a skipped norm
    if (a.gn_gamma && !(a.ksplit > 1 && tile == 128160 && v2ok && !use_v4 && !conv3)) return WD_EINVAL;
    static const int ks_env = getenv("WDIFF_GEMM_KS") ? atoi(getenv("WDIFF_GEMM_KS")) : 2;
    const int ks = ks_env == 1 ? 1 : 2;
    static const int stagger_min = getenv("WDIFF_GEMM_STAGGER_MIN") ? atoi(getenv("WDIFF_GEMM_STAGGER_MIN")) : 10;
    static const bool stagger = getenv("WDIFF_GEMM_STAGGER") ? atoi(getenv("WDIFF_GEMM_STAGGER")) != 0 : true;
    static const bool m16 = getenv("WDIFF_GEMM_M16") ? atoi(getenv("WDIFF_GEMM_M16")) != 0 : true;
    static const bool pp = getenv("WDIFF_GEMM_PP") ? atoi(getenv("WDIFF_GEMM_PP")) != 0 : false;  // measured: no gain
#ifdef WDIFF_EXPERIMENTAL
#define WD_DISPATCH_PP(BM_, BN_) \
    if (v2ok && ks == 2 && pp) return a.npass == 3 ? launch2<BM_, BN_, 3, 2, true>(a, st) : launch2<BM_, BN_, 1, 2, true>(a, st);
#else
#define WD_DISPATCH_PP(BM_, BN_) (void)pp;
#endif
#define WD_DISPATCH(BM_, BN_)                                                                      \
    WD_DISPATCH_PP(BM_, BN_)                                                                       \
    if (v2ok && ks == 2) return a.npass == 3 ? launch2<BM_, BN_, 3, 2>(a, st) : launch2<BM_, BN_, 1, 2>(a, st); \
    if (v2ok) return a.npass == 3 ? launch2<BM_, BN_, 3, 1>(a, st) : launch2<BM_, BN_, 1, 1>(a, st); \
    a.ksplit = 1;                                                                                   \
    a.tickets = nullptr;                                                                            \
    return a.npass == 3 ? launch<BM_, BN_, 3>(a, st) : launch<BM_, BN_, 1>(a, st)
#ifdef WDIFF_EXPERIMENTAL
    if (conv3 && v2ok) return a.npass == 3 ? launch_conv3<3>(a, st) : launch_conv3<1>(a, st);
#endif
    switch (tile) {
        case 128064: WD_DISPATCH(128, 64);
        case 128160:
            // second K-half group runs one stage late (see the kernel); the extra drain phase only pays on long K loops
            if (use_v4) return a.npass == 3 ? launch4<3>(a, st) : launch4<1>(a, st);
#ifdef WDIFF_EXPERIMENTAL
            {
                // ring kernel with dedicated loader waves (wd_gemm8_kernel): within +-5 % of the default kernel on every shape; no
                // fused GroupNorm statistics (its 768-thread epilogue would need a larger statistics scratch).  WDIFF_GEMM_V8=1 takes it wherever legal, dbg 0x80000 forces it (parity tests).
                static const int v8_env = getenv("WDIFF_GEMM_V8") ? atoi(getenv("WDIFF_GEMM_V8")) : 0;
                bool v8ok = a.ktot % 32 == 0 && a.src[0].ntaps <= 9 && !a.stat_part && !a.gn_gamma && a.act != WD_ACT_GEGLU &&
                            (a.nsrc == 1 || (a.src[1].gather == nullptr && a.src[1].ntaps == 1));
                for (int s8 = 0; s8 < a.nsrc; ++s8) v8ok = v8ok && (a.src[s8].c % 32 == 0);
                if (v8ok && (v8_env == 1 || (a.dbg & 0x80000))) {
                    if (a.ksplit > 1 && (a.ktot / 32 < a.ksplit || (long)a.ksplit * a.m * a.n > a.ws_floats)) a.ksplit = 1;
                    a.tickets = nullptr;
                    return a.npass == 3 ? launch8<3>(a, st) : launch8<1>(a, st);
                }
            }
#endif
            if (v2ok && ks == 2 && m16 && stagger && nk64 / a.ksplit >= stagger_min) a.dbg |= 0x200;
            if (v2ok && ks == 2 && m16)
                return a.npass == 3 ? launch2<128, 160, 3, 2, false, true>(a, st) : launch2<128, 160, 1, 2, false, true>(a, st);
            WD_DISPATCH(128, 160);
        case 128128: WD_DISPATCH(128, 128);
        case 64064: WD_DISPATCH(64, 64);
        default: return WD_EINVAL;
    }
#undef WD_DISPATCH
}
