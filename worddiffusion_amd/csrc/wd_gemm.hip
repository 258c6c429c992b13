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

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int BK = 32;

// byte offset of 16-byte chunk `ch` (0..3) of row `row` inside a [rows][32] bf16 plane (64-byte rows)
__device__ __forceinline__ int lds_off(int row, int ch) { return row * 64 + ((ch ^ ((row >> 2) & 3)) << 4); }

// ---- epilogue shared by both kernels.  C/D map of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
template <int TN>
__device__ __forceinline__ void wd_epilogue(const wd_gemm_args& a, const f32x16 (&acc)[TN], const int mrow0,
                                            const int ncolbase, const int lane) {
    const int frow = lane & 31, fhalf = lane >> 5;
    const int ncol0 = ncolbase + frow;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        if (m >= a.m) continue;
        const float* rv = nullptr;
        if (a.rowvec) rv = a.rowvec + (long)(m / a.hw_out) * a.rowvec_ld;
        const float* rs = nullptr;
        if (a.resid) rs = a.resid + (a.resid_rows ? (long)a.resid_rows[m] : (long)m) * a.resid_ld;
        if (a.act == WD_ACT_GEGLU) {
#pragma unroll
            for (int t = 0; t + 1 < TN; t += 2) {
                const int nx = ncol0 + t * 32, ng = nx + 32;
                const int no = (ncolbase >> 1) + (t >> 1) * 32 + frow;
                if (ng >= a.n) continue;
                float x = acc[t][r], g = acc[t + 1][r];
                if (a.bias) {
                    x += a.bias[nx];
                    g += a.bias[ng];
                }
                float v = x * wd_gelu_erf(g);
                if (rv) v += rv[no];
                if (rs) v += rs[no];
                if (a.out_f32) a.out_f32[(long)m * a.out_ld + no] = v;
                if (a.out_hi) {
                    uint32_t h, l;
                    wd_split1(v, h, l);
                    a.out_hi[(long)m * a.out_pl_ld + no] = (wd_bf16)h;
                    if (a.out_lo) a.out_lo[(long)m * a.out_pl_ld + no] = (wd_bf16)l;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int n = ncol0 + t * 32;
                if (n >= a.n) continue;
                float v = acc[t][r];
                if (a.bias) v += a.bias[n];
                if (rv) v += rv[n];
                if (rs) v += rs[n];
                if (a.act == WD_ACT_SILU) v = wd_silu(v);
                if (a.out_f32) a.out_f32[(long)m * a.out_ld + n] = v;
                if (a.out_hi) {
                    uint32_t h, l;
                    wd_split1(v, h, l);
                    a.out_hi[(long)m * a.out_pl_ld + n] = (wd_bf16)h;
                    if (a.out_lo) a.out_lo[(long)m * a.out_pl_ld + n] = (wd_bf16)l;
                }
            }
        }
    }
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

    wd_epilogue<TN>(a, acc, m0 + wm * 32, n0 + wn * WCOLS, lane);
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

template <int BM, int BN, int NPASS>
__global__ void __launch_bounds__(256, 1) wd_gemm2_kernel(const wd_gemm_args a, const int nbn, const int nbm) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int WM = BM / 32, WN = 4 / WM;
    constexpr int WCOLS = BN / WN;
    constexpr int TN = WCOLS / 32;
    static_assert(WCOLS % 32 == 0 && WM * WN == 4, "bad tile");
    constexpr int A_INS = BM / 32;  // 8-row DMA pieces per plane per wave
    constexpr int B_INS = BN / 32;
    static_assert(BM % 32 == 0 && BN % 32 == 0, "bad tile");
    constexpr int A_PL = BM * 128;
    constexpr int B_PL = BN * 128;
    constexpr int STAGE = NPL * (A_PL + B_PL);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_tab = reinterpret_cast<int*>(smem + 2 * STAGE);  // [ntaps0][BM] source row of src[0] per tap, -1 = zero row

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
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane >> 3, lpos = lane & 7;

    // ---- source-row table of src[0] for this row panel (one global gather lookup per (tap, row), done once)
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

    // ---- the rows this lane feeds (A_INS pieces of 8 rows per plane for A, B_INS for W)
    int a_sw[A_INS];
#pragma unroll
    for (int i = 0; i < A_INS; ++i) {
        const int row = wave * (BM / 4) + 8 * i + lrow;
        a_sw[i] = (lpos ^ ((row >> 1) & 7)) * 8;  // source chunk (in elements) that lands at position lpos
    }
    long b_off[B_INS];
    bool b_ok[B_INS];
#pragma unroll
    for (int i = 0; i < B_INS; ++i) {
        const int row = wave * (BN / 4) + 8 * i + lrow;
        const int n = n0 + row;
        b_ok[i] = n < a.n;
        b_off[i] = (long)n * a.ktot + (lpos ^ ((row >> 1) & 7)) * 8;
    }
    const wd_bf16* zline = reinterpret_cast<const wd_bf16*>(wd_zero_line) + lpos * 8;

    int s = 0, tap = 0, kc = 0;
    const wd_bf16* cur_hi = a.src[0].hi;
    const wd_bf16* cur_lo = a.src[0].lo;
    int cur_ld = a.src[0].ld, cur_c = a.src[0].c, cur_nt = a.src[0].ntaps;
    long a_off[A_INS];

    auto locate = [&]() {
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const int row = wave * (BM / 4) + 8 * i + lrow;
            int r;
            if (s == 0) r = s_tab[tap * BM + row];
            else r = (m0 + row < a.m) ? m0 + row : -1;  // src[1] is an identity source (1x1 skip)
            a_off[i] = r >= 0 ? (long)r * cur_ld + a_sw[i] : -1;
        }
    };
    locate();

    auto issue = [&](int kit, int stage) {
        char* base = smem + stage * STAGE + wave * (BM / 4) * 128;
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            const wd_bf16* ph = a_off[i] >= 0 ? cur_hi + a_off[i] + kc * BK2 : zline;
            __builtin_amdgcn_global_load_lds((wd_gbl_ptr)ph, (wd_lds_ptr)(base + i * 1024), 16, 0, 0);
            if (NPL == 2) {
                const wd_bf16* pl = a_off[i] >= 0 ? cur_lo + a_off[i] + kc * BK2 : zline;
                __builtin_amdgcn_global_load_lds((wd_gbl_ptr)pl, (wd_lds_ptr)(base + A_PL + i * 1024), 16, 0, 0);
            }
        }
        char* bb = smem + stage * STAGE + NPL * A_PL + wave * (BN / 4) * 128;
#pragma unroll
        for (int i = 0; i < B_INS; ++i) {
            const wd_bf16* ph = b_ok[i] ? a.w_hi + b_off[i] + (long)kit * BK2 : zline;
            __builtin_amdgcn_global_load_lds((wd_gbl_ptr)ph, (wd_lds_ptr)(bb + i * 1024), 16, 0, 0);
            if (NPL == 2) {
                const wd_bf16* pl = b_ok[i] ? a.w_lo + b_off[i] + (long)kit * BK2 : zline;
                __builtin_amdgcn_global_load_lds((wd_gbl_ptr)pl, (wd_lds_ptr)(bb + B_PL + i * 1024), 16, 0, 0);
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

    const int nk = a.ktot / BK2;
    issue(0, 0);
    advance();

    const int frow = lane & 31, fhalf = lane >> 5;
    // fragment registers, double buffered over the four 16-deep k-steps of a stage
    bf16x8 fa[2][NPL], fb[2][TN][NPL];
    auto lfrag = [&](const char* base, int kk, int set) {
        const int ch = kk * 2 + fhalf;
        const int ao = lds_off2(wm * 32 + frow, ch);
#pragma unroll
        for (int p = 0; p < NPL; ++p) fa[set][p] = *reinterpret_cast<const bf16x8*>(base + p * A_PL + ao);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int bo = NPL * A_PL + lds_off2(wn * WCOLS + t * 32 + frow, ch);
#pragma unroll
            for (int p = 0; p < NPL; ++p) fb[set][t][p] = *reinterpret_cast<const bf16x8*>(base + p * B_PL + bo);
        }
    };

    for (int kit = 0; kit < nk; ++kit) {
        __syncthreads();  // vmcnt(0) + barrier: step kit has landed, everybody is done reading the other stage
        if (kit + 1 < nk) {
            issue(kit + 1, (kit + 1) & 1);
            advance();
        }
        const char* base = smem + (kit & 1) * STAGE;
        lfrag(base, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1;
            if (kk < 3) lfrag(base, kk + 1, cur ^ 1);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (NPL == 2) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][NPL - 1], fb[cur][t][0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][t][NPL - 1], acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][t][0], acc[t], 0, 0, 0);
            }
        }
    }

    wd_epilogue<TN>(a, acc, m0 + wm * 32, n0 + wn * WCOLS, lane);
}

template <int BM, int BN, int NPASS>
int launch(const wd_gemm_args& a, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int smem = 2 * NPL * (BM + BN) * 64;
    static bool attr_done = false;  // one process = one device, set once per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemm_kernel<BM, BN, NPASS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nbn = (a.n + BN - 1) / BN, nbm = (a.m + BM - 1) / BM;
    WdLaunchScope scope(WD_CLS_GEMM, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
    hipLaunchKernelGGL((wd_gemm_kernel<BM, BN, NPASS>), dim3(nbn * nbm), dim3(256), smem, st, a, nbn, nbm);
    return wd_check_launch();
}

template <int BM, int BN, int NPASS>
int launch2(const wd_gemm_args& a, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int smem = 2 * NPL * (BM + BN) * 128 + 9 * BM * 4;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemm2_kernel<BM, BN, NPASS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nbn = (a.n + BN - 1) / BN, nbm = (a.m + BM - 1) / BM;
    WdLaunchScope scope(WD_CLS_GEMM, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
    hipLaunchKernelGGL((wd_gemm2_kernel<BM, BN, NPASS>), dim3(nbn * nbm), dim3(256), smem, st, a, nbn, nbm);
    return wd_check_launch();
}

}  // namespace

extern "C" int wd_gemm(const wd_gemm_args* pa, void* stream) {
    if (!pa) return WD_EINVAL;
    const wd_gemm_args& a = *pa;
    if (a.nsrc < 1 || a.nsrc > 2 || (a.npass != 1 && a.npass != 3)) return WD_EINVAL;
    if (a.m <= 0 || a.n <= 0 || a.hw_out <= 0 || !a.w_hi || (a.npass == 3 && !a.w_lo)) return WD_EINVAL;
    if (!a.out_f32 && !a.out_hi) return WD_EINVAL;
    long k = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        const wd_src& q = a.src[s];
        if (!q.hi || (a.npass == 3 && !q.lo)) return WD_EINVAL;
        if (q.c <= 0 || q.c % BK || q.ld % 8 || q.ntaps < 1) return WD_EINVAL;
        if (!q.gather && q.ntaps != 1) return WD_EINVAL;
        if (q.gather && q.hw_src <= 0) return WD_EINVAL;
        k += (long)q.ntaps * q.c;
    }
    if (k != a.ktot) return WD_EINVAL;
    if (a.act == WD_ACT_GEGLU && (a.n % 64)) return WD_EINVAL;
    if (a.rowvec && a.rowvec_ld <= 0) return WD_EINVAL;
    if (a.resid && a.resid_ld <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    int tile = a.tile;
    if (tile == 0) {
        // heuristics: keep >= ~1 workgroup per CU; GEGLU pairs column tiles inside one wave (needs 128x64 / 64x64 ...)
        const long big = (long)((a.m + 127) / 128) * ((a.n + 159) / 160);
        if (a.act == WD_ACT_GEGLU) tile = ((long)((a.m + 127) / 128) * (a.n / 64) >= 192) ? 128064 : 128064;
        else if (a.n % 160 == 0 && big >= 192) tile = 128160;
        else if ((long)((a.m + 127) / 128) * ((a.n + 63) / 64) >= 192) tile = 128064;
        else tile = (a.m > 64 * 2) ? 64064 : 128064;
    }
    if (a.act == WD_ACT_GEGLU && tile != 128064) return WD_EINVAL;
    bool v2ok = (a.ktot % BK2 == 0) && !getenv("WDIFF_GEMM_V1");
    for (int s = 0; s < a.nsrc; ++s) v2ok = v2ok && (a.src[s].c % BK2 == 0);
    v2ok = v2ok && a.src[0].ntaps <= 9 && (a.nsrc == 1 || (a.src[1].gather == nullptr && a.src[1].ntaps == 1));
#define WD_DISPATCH(BM_, BN_)                                                                      \
    if (v2ok) return a.npass == 3 ? launch2<BM_, BN_, 3>(a, st) : launch2<BM_, BN_, 1>(a, st); \
    return a.npass == 3 ? launch<BM_, BN_, 3>(a, st) : launch<BM_, BN_, 1>(a, st)
    switch (tile) {
        case 128064: WD_DISPATCH(128, 64);
        case 128160: WD_DISPATCH(128, 160);
        case 64064: WD_DISPATCH(64, 64);
        default: return WD_EINVAL;
    }
#undef WD_DISPATCH
}
