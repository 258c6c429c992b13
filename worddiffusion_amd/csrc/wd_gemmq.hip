// 3x3 / pad 1 / stride 1 convolution over SMALL maps (the 4 x 16 level of the UNet: 64 positions per sample, M = 4096 rows at batch 64;
// reference unet.py:595,621,632 at the second resolution) - tile 64 x 80 of wd_gemm_args.tile == 64080, all of K inside the workgroup.
//
// The 128 x 160 / 64 x 320 kernels leave most CUs idle on a 4096 x 320 output (32 / 64 tiles), so wd_gemm cut K over workgroups: 29 us
// of GEMM for partial slabs + a 12 us combine launch per layer, 30 % of the denoising step at a quarter of its FLOPs.  A 64 x 80 tile
// fills the chip (256 tiles) but every CU then has to pull 64 rows of A NINE times (once per tap) and 80 columns of W - 1.66 MB for
// K = 2880 - through its ~70 GB/s vector-memory path: a first kernel of that shape (LDS-staged operands) took exactly that long.
// This one moves 1.04 MB and has no barrier in its K loop:
//
//   * the tile's rows are ONE sample (64 positions = 64 / W image rows); the input rows it needs - those image rows, one above and one
//     below, W + 2 tokens per row with a zero token at either end (the padding of the convolution: no masks in the loop), all channels
//     of the pass, hi and lo planes - are copied into LDS ONCE (123 KB for 320 channels at W = 16), and a tap is a constant
//     shift of the slab row a lane reads its A fragment from: every input element crosses the vector-memory path once, not nine times;
//   * the weights never touch LDS: fragment-major (wd_gemm_pack_w), a wave loads its five column tiles of a k-step as five
//     consecutive kilobytes straight into the MFMA B-operand registers, one k-step ahead;
//   * the eight waves split K, not the tile: wave w owns k-steps w, w + 8, ... (32 deep each, chunk-major: the 18 k-steps of a
//     64-channel chunk are consecutive) and accumulates the WHOLE 64 x 80 tile (80 registers).  No wave ever waits for another inside
//     the loop - no barrier, no shared stage buffer - so their load and MFMA phases interleave by themselves;
//   * the eight partial tiles are summed in a fixed order through LDS (two rounds), then the shared epilogue runs on the finished fp32
//     image: bias, FiLM row, residual, statistics - and, since a tile holds whole (sample, group) blocks, the CONSUMER's GroupNorm
//     (+ SiLU) and its operand planes (wd_gemm_args.gn_*, wd_gn_tile): no combine launch, no wd_gn_apply launch;
//   * more than five chunks (the 640-channel decoder layers): passes of five chunks, the slab refilled between two barriers.
#include "wd_gemm_epi.h"
#include "wd_gemm_priv.h"

namespace {

constexpr int QNT = 512;
constexpr int QBM = 64, QBN = 80;
constexpr int QLDE = QBN + 4;
constexpr uint32_t Q_OOB = 0x80000000u;

typedef __attribute__((ext_vector_type(4))) unsigned q_u32x4;

__device__ __forceinline__ int q_lds_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

__global__ void __launch_bounds__(QNT, 1) wd_gemmq_kernel(const wd_gemm_args a, const int nbn, const int nbm, const int nchp) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntile = nbn * nbm;
    int wg;
    {
        const int bid = blockIdx.x;
        const int q = ntile >> 3, r = ntile & 7, xcd = bid & 7, loc = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bn_i = wg % nbn, bm_i = wg / nbn;  // the column tiles of a row panel are neighbours: they share its rows in L2
    const int m0 = bm_i * QBM, n0 = bn_i * QBN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;

    // geometry of a 3x3 source (W = 0: there is none - 1x1 / linear layers).  ST = 2: the stride-2 convolution of a Downsample
    // (hw_src == 4 hw_out: the source image is twice as wide and high; only its top / left padding is ever read)
    const int W = a.src[0].ntaps == 9 ? a.slab_rows : 0;   // OUTPUT image width
    const int wsh = W ? __builtin_ctz(W) : 0;
    const int ST = (W && a.src[0].hw_src == 4 * a.hw_out) ? 2 : 1;
    const int Ws = W * ST, wss = wsh + ST - 1;              // source image width
    const int Hs = W ? (a.hw_out >> wsh) * ST : 0;          // source image height
    const int TR = W ? QBM >> wsh : 0;         // output image rows of the tile
    const int NR = W ? (TR - 1) * ST + 3 : 0;  // source image rows of the slab: ST y0 - 1 ... ST (y0 + TR - 1) + 1
    const int SW = Ws + 2;                     // slab tokens per image row (a zero token at either end)
    const int bsm = m0 / a.hw_out;             // sample of the tile
    const int y0 = W ? (m0 - bsm * a.hw_out) >> wsh : 0;

    auto make_srd = [](const void* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFF0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t srd_w_hi = make_srd(a.w_hi), srd_w_lo = make_srd(a.w_lo);

    // ---- weights: fragment-major, this tile's five column tiles of a k-step are 5 KB in a row
    const int nct = a.n >> 4;
    const uint32_t b_voff = (uint32_t)((n0 >> 4) * 1024 + lane * 16);
    const uint32_t kstep_bytes = (uint32_t)nct * 1024u;
    auto load_b = [&](bf16x8 (&fb)[5][2], const int kabs, const bool live) {
        const uint32_t so = live ? (uint32_t)kabs * kstep_bytes : 0u;
        const uint32_t vo = live ? b_voff : Q_OOB;
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                fb[t][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(p ? srd_w_lo : srd_w_hi, vo + t * 1024, so, 0));
    };
    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] = 0.0f;
    // one k-step: A fragments of row tile i from slab rows rb[i] + shift + (lane row), read right before its 15 MFMAs (one tile ahead)
    // `skip`: bit i = every source row of row tile i lies above / below the image for this tap (zero rows: their 15 MFMAs add exact
    // zeros and are left out - a sixth of a 4 x 16 map's convolution work, a third of a 2 x 32 map's)
    auto mfma_step = [&](const char* base, const int cpl, const int (&rb)[4], const int shift0, const int lm, const int half,
                         const bf16x8 (&fb)[5][2], const int skip) {
        const int ach = half * 4 + lq;
        const int shift = shift0 + lm * l15 - l15;   // (the reads below add l15)
        bf16x8 xa[2][2];
        {
            const int ao = q_lds_off(rb[0] + shift + l15, ach);
            xa[0][0] = *reinterpret_cast<const bf16x8*>(base + ao);
            xa[0][1] = *reinterpret_cast<const bf16x8*>(base + cpl + ao);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) {
                const int ao = q_lds_off(rb[i + 1] + shift + l15, ach);
                xa[(i + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(base + ao);
                xa[(i + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(base + cpl + ao);
            }
            if (skip & (1 << i)) continue;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i & 1][1], fb[t][0], acc[i][t], 0, 0, 0);
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i & 1][0], fb[t][1], acc[i][t], 0, 0, 0);
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i & 1][0], fb[t][0], acc[i][t], 0, 0, 0);
            }
        }
    };
#ifdef WD_Q_STAMPS
    unsigned long long* qst = (a.dbg & 0x100) && a.ws && blockIdx.x == 0 ? reinterpret_cast<unsigned long long*>(a.ws) + wave * 8 : nullptr;
#define Q_STAMP(i) if (qst && lane == 0) qst[i] = __builtin_amdgcn_s_memtime()
#else
#define Q_STAMP(i)
#endif
    Q_STAMP(0);
    bf16x8 xb[5][2], yb[5][2];
    bool first_pass = true;

    // One source = passes of up to `pchunks` 64-channel chunks whose rows fit LDS.  CONV: the 3x3 slab ((TR + 2) image rows of W + 2
    // tokens, zero tokens at the row ends, 18 k-steps per chunk: 2 tap + half).  Identity (1x1 / linear / the skip source of a
    // decoder block): the tile's own 64 rows, 2 k-steps per chunk.  kbase: the source's first k-step in the weights.
    auto run_source = [&](const wd_src& q, const bool conv, const int kbase) {
        const int cpt = q.c >> 6;
        const int slr = conv ? NR * SW : QBM;            // slab rows per chunk
        const int cpl = slr * 128, chb = 2 * cpl;
        const int pchunks = min(conv ? 5 : 8, (140 * 1024) / chb);
        const int kpc = conv ? 18 : 2;                   // k-steps per chunk
        const int ntok = conv ? NR * Ws : QBM;           // real tokens of a slab
        const int lm = conv ? ST : 1;                    // slab rows between neighbouring output positions
        const __amdgpu_buffer_rsrc_t srd_a_hi = make_srd(q.hi), srd_a_lo = make_srd(q.lo);
        int rb[4];
        int skm0 = 0, skm2 = 0;   // row tiles whose source rows are all outside the image for the taps of kernel row 0 / 2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = 16 * i;
            rb[i] = conv ? ST * (p >> wsh) * SW + ST * (p & (W - 1)) : p;   // tap (0, 0) of the tile's first position
            if (conv) {
                const int ys = ST * (y0 + (p >> wsh)) - 1;   // source row of kernel row 0 (a row tile of 16 positions lies in one row)
                if (ys < 0) skm0 |= 1 << i;
                if (ys + 2 >= Hs) skm2 |= 1 << i;
            }
        }
        for (int c_lo = 0; c_lo < cpt; c_lo += pchunks) {
            const int nch = min(pchunks, cpt - c_lo);
            const int nks = kpc * nch;  // k-steps of the pass, chunk-major
            if (!first_pass) __syncthreads();  // every wave is done with the previous slab
            first_pass = false;
            // this wave's k-steps: wave, wave + 8, ...  (c, r): chunk of the pass and position in the chunk (CONV: 2 tap + half)
            auto kabs_of = [&](const int c, const int r) {
                return kbase + (conv ? 2 * ((r >> 1) * cpt + c_lo + c) + (r & 1) : 2 * (c_lo + c) + r);
            };
            auto advance = [&](int& c, int& r) {
                if (conv) {
                    r += 8;
                    const bool wrap = r >= 18;
                    r = __builtin_amdgcn_readfirstlane(wrap ? r - 18 : r);
                    c = __builtin_amdgcn_readfirstlane(wrap ? c + 1 : c);
                } else {
                    c += 4;
                }
            };
            auto skip_of = [&](const int r) {   // (wave-uniform: r is)
                if (!conv) return 0;
                const int ky = (r >> 1) / 3;
                return __builtin_amdgcn_readfirstlane(ky == 0 ? skm0 : ky == 2 ? skm2 : 0);
            };
            auto shift_of = [&](const int r) {
                if (!conv) return 0;
                const int tap = r >> 1, ky = tap / 3;
                return ky * SW + (tap - 3 * ky);
            };
            int nc = conv ? 0 : wave >> 1, nr = conv ? wave : wave & 1;   // the next step to load
            int cc = nc, cr = nr;                                          // the step to multiply
            load_b(xb, kabs_of(nc, nr), kpc * nc + nr < nks);   // (the first weights are on their way while the slab is filled)
            advance(nc, nr);
            if (conv) {  // the padding tokens (left / right end of every slab image row), both planes
                for (int i = tid; i < nch * 2 * NR * 2 * 8; i += QNT) {
                    const int pc = i & 7, side = (i >> 3) & 1;
                    int rest = i >> 4;
                    const int ry = rest % NR;
                    rest /= NR;
                    const int pl = rest & 1, ch = rest >> 1;
                    const int row = ry * SW + (side ? SW - 1 : 0);
                    *reinterpret_cast<q_u32x4*>(smem + ch * chb + pl * cpl + row * 128 + pc * 16) = q_u32x4{0u, 0u, 0u, 0u};
                }
            }
            // slab fill through registers (ordinary loads and LDS stores: an LDS-DMA here would make hipcc guard every LDS read of the
            // loop with vmcnt(0) - and the loop keeps weight loads in flight).  All the loads of a round are in flight before its
            // first store.  Rows above / below the image carry an out-of-range offset: zeros.
            const int nitems = nch * ntok * 8;  // (chunk, token, 16-byte piece) per plane
            for (int i0 = 0; i0 < nitems; i0 += 8 * QNT) {
                q_u32x4 v[2][8];
                int dst[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k * QNT + tid;
                    const int pc = i & 7;
                    const int rest = i >> 3;
                    const int chl = rest / ntok, tq = rest - chl * ntok;
                    int grow, srow;
                    bool ok = i < nitems;
                    if (conv) {
                        const int ry = tq >> wss, x = tq & (Ws - 1);
                        const int gy = ST * y0 - 1 + ry;
                        ok = ok && gy >= 0 && gy < Hs;
                        grow = bsm * q.hw_src + gy * Ws + x;
                        srow = ry * SW + x + 1;
                    } else {
                        grow = m0 + tq;
                        srow = tq;
                    }
                    const uint32_t vo = ok ? (uint32_t)grow * (uint32_t)(q.ld * 2) + (uint32_t)((c_lo + chl) * 128 + pc * 16) : Q_OOB;
                    v[0][k] = __builtin_amdgcn_raw_buffer_load_b128(srd_a_hi, vo, 0, 0);
                    v[1][k] = __builtin_amdgcn_raw_buffer_load_b128(srd_a_lo, vo, 0, 0);
                    dst[k] = i < nitems ? chl * chb + q_lds_off(srow, pc) : -1;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (dst[k] >= 0) {
                        *reinterpret_cast<q_u32x4*>(smem + dst[k]) = v[0][k];
                        *reinterpret_cast<q_u32x4*>(smem + dst[k] + cpl) = v[1][k];
                    }
            }
            __syncthreads();
            Q_STAMP(1);
            for (int ks = wave; ks < nks; ks += 16) {
                load_b(yb, kabs_of(nc, nr), kpc * nc + nr < nks);
                advance(nc, nr);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(smem + cc * chb, cpl, rb, shift_of(cr), lm, cr & 1, xb, skip_of(cr));
                __builtin_amdgcn_sched_barrier(0);
                advance(cc, cr);
                load_b(xb, kabs_of(nc, nr), kpc * nc + nr < nks);
                advance(nc, nr);
                __builtin_amdgcn_sched_barrier(0);
                if (kpc * cc + cr < nks) mfma_step(smem + cc * chb, cpl, rb, shift_of(cr), lm, cr & 1, yb, skip_of(cr));
                __builtin_amdgcn_sched_barrier(0);
                advance(cc, cr);
            }
        }
    };
    const bool conv0 = a.src[0].ntaps == 9;
    run_source(a.src[0], conv0, 0);
    if (a.nsrc > 1) run_source(a.src[1], false, 2 * a.src[0].ntaps * (a.src[0].c >> 6));

    Q_STAMP(4);
    // ---- the eight partial tiles summed in a fixed order: s_j = w_j + w_{j+4}, result = ((s_0 + s_1) + s_2) + s_3.  The partial
    // images are COLUMN-major ([column][64 rows + 4]: a lane's four rows of a column are one 16-byte access); the last step - all 512
    // threads - sums the four s_j and lays the result out row-major for the epilogue.
    __syncthreads();
    Q_STAMP(5);
    constexpr int IMG = QBM * QLDE;   // floats of the row-major result image
    constexpr int CLD = QBM + 4;      // column pitch of a partial image
    constexpr int PIMG = QBN * CLD;   // floats of one partial image
    float* res = reinterpret_cast<float*>(smem);        // result image
    float* part = res + IMG;                             // four partial images
    auto pcol = [&](float* b, const int i, const int t) -> f32x4* {
        return reinterpret_cast<f32x4*>(b + (t * 16 + l15) * CLD + i * 16 + 4 * lq);
    };
    if (wave >= 4) {
        float* b = part + (wave - 4) * PIMG;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 5; ++t) *pcol(b, i, t) = acc[i][t];
    }
    __syncthreads();
    if (wave < 4) {
        float* b = part + wave * PIMG;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                acc[i][t] += *pcol(b, i, t);
                *pcol(b, i, t) = acc[i][t];   // (same thread, same address: s_j in place)
            }
    }
    __syncthreads();
    for (int e = tid; e < QBN * (QBM / 4); e += QNT) {   // (column, four rows) per thread and round
        const int col = e >> 4, r4 = (e & 15) * 4;
        const float* p = part + col * CLD + r4;
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
        v += *reinterpret_cast<const f32x4*>(p + PIMG);
        v += *reinterpret_cast<const f32x4*>(p + 2 * PIMG);
        v += *reinterpret_cast<const f32x4*>(p + 3 * PIMG);
#pragma unroll
        for (int r = 0; r < 4; ++r) res[(r4 + r) * QLDE + col] = v[r];
    }
    __syncthreads();
    float* buf = res;
    Q_STAMP(6);
    // ---- epilogue on the finished image (scratch behind the four images)
    if (a.gn_gamma) wd_gn_tile<QBN, QNT, false>(a, buf, buf + IMG, m0, n0, tid);
    else wd_epilogue_tail<QBM, QBN, QNT>(a, buf, m0, n0, tid, 0);
    Q_STAMP(7);
#endif
}

}  // namespace

// Shapes the kernel serves: one 3x3 / pad 1 / stride 1 source given as planes, 64-position samples of width 16 or 32, 80-column
// tiles, split-bf16, no K cut, vector epilogue.
bool wd_gemmq_applies(const wd_gemm_args& a) {
    const int W = a.slab_rows;
    if (!(a.w_layout == 3 && a.tile == 64080 && !a.a32 && !a.ln_gamma && a.npass == 3 && a.ksplit <= 1 && a.m % 64 == 0 && a.n % QBN == 0 &&
          a.act == WD_ACT_NONE && !a.resid_rows && a.nsrc >= 1 && a.nsrc <= 2))
        return false;
    for (int s = 0; s < a.nsrc; ++s) {
        const wd_src& q = a.src[s];
        if (!q.hi || !q.lo || q.c % 64 || (q.ld & 7)) return false;
        const bool conv = s == 0 && q.ntaps == 9 && q.gather && a.hw_out == 64 &&
                          (((W == 16 || W == 32) && q.hw_src == 64) || (W == 16 && q.hw_src == 256));   // stride 1 / the stride-2 table
        const bool ident = q.ntaps == 1 && !q.gather;
        if (!conv && !ident) return false;
    }
    return true;
}

int wd_gemmq_launch(const wd_gemm_args& a, hipStream_t st) {
    const int loop_smem = 140 * 1024;   // (slabs of a pass: at most 140 KB by construction in the kernel)
    const int red_smem = (QBM * QLDE + 4 * QBN * (QBM + 4)) * 4 + WD_STAT_SCRATCH;
    const int smem = loop_smem > red_smem ? loop_smem : red_smem;
    const int nchp = 0;
    static int attr_max = 0;
    if (smem > attr_max) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemmq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_max = smem;
    }
    const int nbn = a.n / QBN, nbm = a.m / QBM;
    {
        WdLaunchScope scope(WD_CLS_GEMM_Q, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
        hipLaunchKernelGGL(wd_gemmq_kernel, dim3(nbn * nbm), dim3(QNT), smem, st, a, nbn, nbm, nchp);
    }
    return wd_check_launch();
}
