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

    const int W = a.slab_rows, H = a.hw_out / W;
    const int wsh = __builtin_ctz(W);
    const int TR = QBM >> wsh;                 // image rows of the tile
    const int SW = W + 2;                      // slab tokens per image row
    const int SLR = (TR + 2) * SW;             // slab rows per chunk
    const int CPL = SLR * 128;                 // one plane of one chunk
    const int CHB = 2 * CPL;                   // one chunk: hi plane, lo plane
    const int bsm = m0 / a.hw_out;             // sample of the tile (hw_out == 64)
    const int y0 = (m0 - bsm * a.hw_out) >> wsh;
    const int cpt0 = a.src[0].c >> 6;          // 64-channel chunks of the source
    const int npass = (cpt0 + nchp - 1) / nchp;

    auto make_srd = [](const void* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFF0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t srd_w_hi = make_srd(a.w_hi), srd_w_lo = make_srd(a.w_lo);
    const __amdgpu_buffer_rsrc_t srd_a_hi = make_srd(a.src[0].hi), srd_a_lo = make_srd(a.src[0].lo);

    // ---- the padding tokens of every chunk (left / right end of every slab image row), both planes: zero for the whole launch
    for (int i = tid; i < nchp * 2 * (TR + 2) * 2 * 8; i += QNT) {
        const int pc = i & 7, side = (i >> 3) & 1;
        int rest = i >> 4;
        const int ry = rest % (TR + 2);
        rest /= (TR + 2);
        const int pl = rest & 1, ch = rest >> 1;
        const int row = ry * SW + (side ? SW - 1 : 0);
        *reinterpret_cast<q_u32x4*>(smem + ch * CHB + pl * CPL + row * 128 + pc * 16) = q_u32x4{0u, 0u, 0u, 0u};
    }

    // ---- slab fill: item = (chunk, plane, slab token, 16-byte piece), one per thread and round, through registers (ordinary loads
    // and LDS stores: an LDS-DMA here would make hipcc guard every LDS read of the loop with vmcnt(0) - and the loop keeps weight
    // loads in flight).  Rows above / below the image carry an out-of-range offset: zeros.
    const int ntok = (TR + 2) * W;             // real tokens of a slab
    auto fill = [&](const int pass) {
        const int c_lo = pass * nchp, nch = min(nchp, cpt0 - c_lo);
        const int nitems = nch * ntok * 8;     // per plane
        // (all the loads of a round are in flight before the first store: at most 8 items per thread and plane - 5 chunks x 136
        // tokens x 8 pieces / 512 - i.e. one round trip for the whole slab instead of one per four items)
        for (int i0 = 0; i0 < nitems; i0 += 8 * QNT) {
            q_u32x4 v[2][8];
            int dst[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * QNT + tid;
                const int pc = i & 7;
                const int rest = i >> 3;
                const int chl = rest / ntok, tq = rest - chl * ntok;
                const int ry = tq >> wsh, x = tq & (W - 1);
                const int gy = y0 - 1 + ry;
                const bool ok = i < nitems && gy >= 0 && gy < H;
                const uint32_t vo = ok ? (uint32_t)(bsm * a.src[0].hw_src + gy * W + x) * (uint32_t)(a.src[0].ld * 2) +
                                             (uint32_t)((c_lo + chl) * 128 + pc * 16)
                                       : Q_OOB;
                v[0][k] = __builtin_amdgcn_raw_buffer_load_b128(srd_a_hi, vo, 0, 0);
                v[1][k] = __builtin_amdgcn_raw_buffer_load_b128(srd_a_lo, vo, 0, 0);
                dst[k] = i < nitems ? chl * CHB + q_lds_off(ry * SW + x + 1, pc) : -1;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (dst[k] >= 0) {
                    *reinterpret_cast<q_u32x4*>(smem + dst[k]) = v[0][k];
                    *reinterpret_cast<q_u32x4*>(smem + dst[k] + CPL) = v[1][k];
                }
        }
    };

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
    // slab row of (row tile i, lane row 0) for the centre tap - wave-uniform: 16 consecutive output positions are 16 consecutive slab rows
    int srs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = 16 * i;
        srs[i] = ((p >> wsh) + 1) * SW + (p & (W - 1)) + 1;
    }
    auto mfma_step = [&](const int chl, const int tap, const int half, const bf16x8 (&fb)[5][2]) {
        const int ky = tap / 3;
        const int shift = (ky - 1) * SW + (tap - 3 * ky - 1) + l15;
        const char* base = smem + chl * CHB;
        const int ach = half * 4 + lq;
        bf16x8 xa[2][2];
        {
            const int ao = q_lds_off(srs[0] + shift, ach);
            xa[0][0] = *reinterpret_cast<const bf16x8*>(base + ao);
            xa[0][1] = *reinterpret_cast<const bf16x8*>(base + CPL + ao);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) {
                const int ao = q_lds_off(srs[i + 1] + shift, ach);
                xa[(i + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(base + ao);
                xa[(i + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(base + CPL + ao);
            }
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
    bf16x8 xb[5][2], yb[5][2], zb[5][2];
    for (int pass = 0; pass < npass; ++pass) {
        const int c_lo = pass * nchp, nch = min(nchp, cpt0 - c_lo);
        const int nks = 18 * nch;  // k-steps of the pass, chunk-major: ks = 18 chunk + 2 tap + half
        if (pass > 0) __syncthreads();  // every wave is done with the previous pass's slab
        fill(pass);
        __syncthreads();
        Q_STAMP(1 + 2 * (pass & 1));
        // this wave's k-steps: wave, wave + 8, ...  State of the NEXT step to load: (chunk nc, position nr = 2 tap + half in the chunk)
        int nc = 0, nr = wave;     // (wave < 18)
        auto kabs_of = [&](const int chl, const int r) {  // absolute k-step of the weights: 2 (tap cpt0 + chunk) + half
            return 2 * ((r >> 1) * cpt0 + c_lo + chl) + (r & 1);
        };
        auto advance = [&](int& c, int& r) {
            r += 8;
            const bool wrap = r >= 18;
            r = __builtin_amdgcn_readfirstlane(wrap ? r - 18 : r);
            c = __builtin_amdgcn_readfirstlane(wrap ? c + 1 : c);
        };
        int cc = 0, cr = wave;     // the step to multiply
        // weights two k-steps ahead (three register sets): one step of lead left the waves waiting on L2 a third of the loop
        load_b(xb, kabs_of(nc, nr), 18 * nc + nr < nks);
        advance(nc, nr);
        load_b(yb, kabs_of(nc, nr), 18 * nc + nr < nks);
        advance(nc, nr);
        for (int ks = wave; ks < nks; ks += 24) {
            load_b(zb, kabs_of(nc, nr), 18 * nc + nr < nks);
            advance(nc, nr);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(cc, cr >> 1, cr & 1, xb);
            __builtin_amdgcn_sched_barrier(0);
            advance(cc, cr);
            load_b(xb, kabs_of(nc, nr), 18 * nc + nr < nks);
            advance(nc, nr);
            __builtin_amdgcn_sched_barrier(0);
            if (18 * cc + cr < nks) mfma_step(cc, cr >> 1, cr & 1, yb);
            __builtin_amdgcn_sched_barrier(0);
            advance(cc, cr);
            load_b(yb, kabs_of(nc, nr), 18 * nc + nr < nks);
            advance(nc, nr);
            __builtin_amdgcn_sched_barrier(0);
            if (18 * cc + cr < nks) mfma_step(cc, cr >> 1, cr & 1, zb);
            __builtin_amdgcn_sched_barrier(0);
            advance(cc, cr);
        }
    }

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

int q_chunks_per_pass(const int W) {
    const int chb = 2 * ((64 / W + 2) * (W + 2)) * 128;
    int n = (140 * 1024) / chb;
    return n > 5 ? 5 : n;
}

}  // namespace

// Shapes the kernel serves: one 3x3 / pad 1 / stride 1 source given as planes, 64-position samples of width 16 or 32, 80-column
// tiles, split-bf16, no K cut, vector epilogue.
bool wd_gemmq_applies(const wd_gemm_args& a) {
    const int W = a.slab_rows;
    return a.w_layout == 3 && a.tile == 64080 && !a.a32 && !a.ln_gamma && a.nsrc == 1 && a.src[0].ntaps == 9 && a.src[0].gather &&
           a.src[0].hi && a.src[0].lo && a.npass == 3 && a.ksplit <= 1 && (W == 16 || W == 32) && a.hw_out == 64 &&
           a.src[0].hw_src == 64 && a.m % 64 == 0 && a.n % QBN == 0 && a.src[0].c % 64 == 0 && a.act == WD_ACT_NONE && !a.resid_rows &&
           (a.src[0].ld & 7) == 0;
}

int wd_gemmq_launch(const wd_gemm_args& a, hipStream_t st) {
    const int W = a.slab_rows;
    const int nchp = q_chunks_per_pass(W);
    const int loop_smem = nchp * 2 * ((64 / W + 2) * (W + 2)) * 128;
    const int red_smem = (QBM * QLDE + 4 * QBN * (QBM + 4)) * 4 + WD_STAT_SCRATCH;
    const int smem = loop_smem > red_smem ? loop_smem : red_smem;
    static int attr_max = 0;
    if (smem > attr_max) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_gemmq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_max = smem;
    }
    const int nbn = a.n / QBN, nbm = a.m / QBM;
    {
        WdLaunchScope scope(WD_CLS_GEMM_OTHER, st, 2.0 * (double)a.m * (double)a.n * (double)a.ktot);
        hipLaunchKernelGGL(wd_gemmq_kernel, dim3(nbn * nbm), dim3(QNT), smem, st, a, nbn, nbm, nchp);
    }
    return wd_check_launch();
}
