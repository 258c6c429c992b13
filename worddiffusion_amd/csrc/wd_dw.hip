// d(weight) of a convolution / linear layer of the training step, straight from the ROW-MAJOR operand planes:
//
//     grad[n][ci * T + t] (+)= sum over tokens m of  dout[m][n] * x[src_t(m)][ci]            (reference: autograd of unet.py's
//                                                                                            conv_nd / linear layers, train.py:293)
//
// The reduction runs over the TOKEN index, which is the slow index of both operands as they sit in memory.  The first training
// engine made it the fast one by writing transposed copies - d(out)^T once per layer and x^T once PER TAP (wd_transpose_planes:
// nine gathers of a 3x3 layer's input, 189 MB written for a 320-channel layer at batch 64) - and ran wd_gemm over them: 1.03 ms
// of transposes and 0.3 ms of d(out)^T per step beside 2.9 ms of GEMM.  gfx950 can transpose on the way out of LDS instead:
// ds_read_b64_tr_b16 hands lane i of a 16-lane group COLUMN i of a 4-row x 16-column block, i.e. four consecutive tokens of one
// channel - half of a v_mfma_f32_16x16x32_bf16 operand.  So both operands are staged as they are, rows of tokens:
//
//   * a workgroup owns a 160 (output channels) x 160 (input channels) tile of ONE tap and a slice of the tokens; 8 waves = 2 groups
//     x (2 x 2) wave tiles of 80 x 80 (5 x 5 MFMA tiles, 100 accumulator registers);
//   * tokens move in SUB-STAGES of 32: 32 rows of d(out) (160 channels of the tile) and the 32 rows of x the tap's gather table
//     names, hi and lo planes, 41 KB, by LDS-DMA (buffer_load ... lds, 16 bytes per lane) into a ring of three slots.  An operand
//     plane is kept as two half-tile images [32 tokens][80 channels] - the halves the 2 x 2 wave tiles read - whose 160-byte rows
//     start 40 banks apart: rows 0..7 sit at banks 0, 40, 16, 56, 32, 8, 48, 24, so the transposed read of a 32-lane half (8 rows x
//     8 banks) is conflict-free without padding;
//   * the two wave groups work in ANTIPHASE: in a half-step one group issues the DMA of its next sub-stage and pulls its 20 operand
//     fragments (40 transposed reads per plane pair) out of the slot that landed a half-step ago, while the other runs the 75
//     MFMAs of the sub-stage it read a half-step earlier - each SIMD hosts one wave of each group, so its MFMA pipe has exactly
//     one wave feeding it and the LDS reads, the DMA issue and the barrier sit in that wave's shadow;
//   * MFMA operand k-slot j of lane group g holds token 4 g + j (j < 4) or 16 + 4 g + j - 4: any assignment works as long as
//     both operands use the same one, and this one makes the four lane groups of a read touch 16 consecutive rows;
//   * split-bf16: hi.hi + hi.lo + lo.hi in fp32 accumulators (npass 3) or hi.hi alone (npass 1), as everywhere else;
//   * the token slices' partial tiles go to the workspace in [slice][n][tap][ci] order (coalesced), and one small combine launch
//     sums them in fixed order into the OIHW gradient (+ the accumulation into a gradient that already holds a value).
//     No atomics: a replayed step is bit-identical.
// Per sub-stage and CU: 41 KB through the vector memory path (40 kilobyte instructions at ~25 cycles each, ~1000 cycles of the
// reading group's half-step, + ~450 of transposed reads) against 1220 cycles of MFMAs in the other group.
#include "wd_gemm_epi.h"

namespace {

constexpr int DW_NT = 512;
constexpr int DW_TB = 160;                 // tile edge (channels), both ways
constexpr int DW_ROWB = 160;               // bytes between token rows of a half-tile image in LDS: 80 channels, no padding
constexpr int DW_HB = 32 * DW_ROWB;        // one half-tile image: 32 token rows x 80 channels = 5 DMA instructions of a kilobyte
constexpr int DW_PL = 2 * DW_HB;           // one plane of one operand of a sub-stage: [channel half][32][80]
constexpr int DW_MAXHW = 1024;             // largest hw_out (entries of the tap's row of the gather table kept in LDS)
constexpr int DW_LDE = DW_TB + 4;          // row pitch of the fp32 epilogue image in floats
constexpr uint32_t DW_OOB = 0x80000000u;

typedef __attribute__((address_space(3))) void* dw_lds_ptr;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short dw_s16x4;
typedef __attribute__((address_space(3))) dw_s16x4* dw_lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short dw_s16x8;

// LDS reads of the main loop are inline assembly on purpose: hipcc treats an LDS-DMA (buffer_load ... lds) as a store to LDS that any
// later ds_read it can see may alias, and puts s_waitcnt vmcnt(0) in front of that read - which would serialise the DMA ring.
// hipcc does not count an asm load either: its destination registers count as written when the statement ends, so nothing may
// touch them (not even a copy) before a wait that NAMES them: dw_tr_read's results go through dw_landed() before any use, the
// table look-ups load and wait in one statement.
template <int OFF>
__device__ __forceinline__ dw_s16x4 dw_tr_read(const uint32_t addr) {
    dw_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// s_waitcnt lgkmcnt(0) that the compiler sees as rewriting the twenty values (asm statements take at most 30 operands)
__device__ __forceinline__ void dw_landed(dw_s16x4 (&v)[20]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                   "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]),
                   "+v"(v[19]));
}
__device__ __forceinline__ void dw_lds_read6(const uint32_t (&ad)[6], int (&r)[6]) {
    asm volatile(
        "ds_read_b32 %0, %6\n\tds_read_b32 %1, %7\n\tds_read_b32 %2, %8\n\tds_read_b32 %3, %9\n\tds_read_b32 %4, %10\n\t"
        "ds_read_b32 %5, %11\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5])
        : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]));
}

template <int NPASS>
__global__ void __launch_bounds__(DW_NT, 1) wd_dw_kernel(const wd_dw_args a0, const int ntn, const int ntc, const int nslice) {
#if defined(__HIP_DEVICE_COMPILE__)
    wd_dw_args a = a0;
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int OPB = NPL * DW_PL;       // one operand of a sub-stage
    constexpr int SLOT = 2 * OPB;          // d(out) rows, then x rows

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_tab = reinterpret_cast<int*>(smem + 3 * SLOT);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;

    // workgroup -> (token slice, tile): consecutive workgroups of an XCD (blockIdx % 8) share a slice, so its rows of both operands
    // are fetched into that XCD's L2 once
    const int ntile = ntn * ntc * a.ntaps;
    const int nprob = a.items ? a.nitems : 1;  // grouped launch: same shapes, nprob sets of operands (wd_dw_group)
    const int nwg = ntile * nprob * nslice;
    int wg;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int sidx = __builtin_amdgcn_readfirstlane(wg / (ntile * nprob));
    int tl = wg - sidx * ntile * nprob;
    const int pi = __builtin_amdgcn_readfirstlane(tl / ntile);
    tl -= pi * ntile;
    if (a.items) {
        const wd_dw_item it = a.items[pi];
        a.d_hi = it.d_hi; a.d_lo = it.d_lo; a.x_hi = it.x_hi; a.x_lo = it.x_lo;
        a.d_ld = it.d_ld; a.x_ld = it.x_ld;
    }
    const int nb = tl % ntn;
    tl /= ntn;
    const int cb = __builtin_amdgcn_readfirstlane(tl % ntc);
    const int tap = __builtin_amdgcn_readfirstlane(tl / ntc);
    const int n0 = nb * DW_TB, c0 = cb * DW_TB;

    const int units = a.m >> 6;
    const int u0 = __builtin_amdgcn_readfirstlane((int)((long)units * sidx / nslice));
    const int u1 = __builtin_amdgcn_readfirstlane((int)((long)units * (sidx + 1) / nslice));
    const int U = 2 * (u1 - u0);           // sub-stages of this workgroup (even, >= 2)
    const int tok0 = u0 * 64;

    // (the identity when there is no table: one code path in the loop)
    for (int p = tid; p < a.hw_out; p += DW_NT) s_tab[p] = a.gather ? a.gather[tap * a.hw_out + p] : p;

    auto make_srd = [](const wd_bf16* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<wd_bf16*>(p), 0, 0x7FFFFFF0, 0x00020000);
    };
    // ---- DMA plan.  A sub-stage is 20 (operand, kilobyte) pairs - P < 10: rows of d(out), else rows of x; a pair = the hi and the lo
    // kilobyte; kilobyte i of an operand plane = piece i % 5 of channel half i / 5.  Lane l of piece pc moves chunk L = 64 pc + l of the
    // [32 tokens][10 chunks] half-tile image: token L / 10, chunk L % 10 - every lane moves data (a padded pitch cost 15 % of the
    // vector-memory issue time in lanes that fetched nothing).  The four waves of a group move the sub-stages that group reads:
    // wave w takes pairs (w & 3) + 4 j, j = 0..4.
    constexpr int NP = 5;
    bool pair_x[NP];
    int pair_tok[NP];
    uint32_t pair_voff[NP];  // d(out) pairs: constant byte offset below the sub-stage's first row.  x pairs: column byte offset.
    int pair_dst[NP];        // byte offset of the hi kilobyte inside the slot
    __amdgpu_buffer_rsrc_t srd_hi[NP], srd_lo[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int P = (wave & 3) + 4 * j;
        const bool isx = P >= 10;
        const int i = isx ? P - 10 : P;           // kilobyte of the operand: channel half i / 5, piece i % 5
        const int hf = i / 5, pc = i - 5 * hf;
        const int L = 64 * pc + lane;             // chunk of the [32 tokens][10 chunks] half-tile image
        const int tok = L / 10, ch = L - tok * 10;
        pair_x[j] = isx;
        pair_tok[j] = tok;
        pair_dst[j] = (isx ? OPB : 0) + hf * DW_HB + pc * 1024;
        if (isx) pair_voff[j] = (uint32_t)((c0 + 80 * hf) * 2 + ch * 16);
        else pair_voff[j] = (uint32_t)tok * (uint32_t)(a.d_ld * 2) + (uint32_t)((n0 + 80 * hf) * 2 + ch * 16);
        // (loop-constant descriptors chosen by wave-uniform selects: a descriptor picked inside the loop ends up in VGPRs)
        srd_hi[j] = make_srd(isx ? a.x_hi : a.d_hi);
        srd_lo[j] = make_srd(isx ? (a.x_lo ? a.x_lo : a.x_hi) : (a.d_lo ? a.d_lo : a.d_hi));
    }

    // position of this group's next sub-stage to issue (64 | hw_out: a 64-token unit never straddles two samples)
    int is_tok = tok0 + 32 * grp;
    int is_p0 = __builtin_amdgcn_readfirstlane(is_tok % a.hw_out);
    int is_b = __builtin_amdgcn_readfirstlane(is_tok / a.hw_out);
    __syncthreads();  // s_tab

    const uint32_t lds0 = (uint32_t)(uintptr_t)(dw_lds_ptr)smem;
    const uint32_t tab_addr = lds0 + 3 * SLOT;
    auto issue = [&](const int slot, const bool live) {   // straight-line: selects, no branches around the DMA instructions
        uint32_t vo[NP];
        int so[NP], dst[NP], dlo[NP], rr[NP];
        {
            uint32_t ta[6];
            int r6[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) ta[j] = tab_addr + (uint32_t)((is_p0 + pair_tok[j < NP ? j : 0]) * 4);
            dw_lds_read6(ta, r6);
#pragma unroll
            for (int j = 0; j < NP; ++j) rr[j] = r6[j];
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int r = rr[j];
            const uint32_t vx = (uint32_t)(r + is_b * a.hw_src) * (uint32_t)(a.x_ld * 2) + pair_voff[j];
            const uint32_t v = pair_x[j] ? vx : pair_voff[j];
            vo[j] = (!live || (pair_x[j] && r < 0)) ? DW_OOB : v;
            so[j] = pair_x[j] ? 0 : is_tok * a.d_ld * 2;
            dst[j] = slot * SLOT + pair_dst[j];
            dlo[j] = DW_PL;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_hi[j], (dw_lds_ptr)(smem + dst[j]), 16, vo[j], so[j], 0, 0);
            if (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_lo[j], (dw_lds_ptr)(smem + dst[j] + dlo[j]), 16, vo[j], so[j], 0, 0);
        }
        is_tok += 64;
        is_p0 += 64;
        const bool wrap = is_p0 >= a.hw_out;
        is_p0 = wrap ? is_p0 - a.hw_out : is_p0;
        is_b += wrap ? 1 : 0;
    };

    // ---- operand fragments: transposed reads.  Lane (g = lane / 16, q = (lane % 16) / 4, p = lane % 4) supplies the address of token row
    // 4 g + q, channels 4 p .. 4 p + 3 of a 16-channel tile and receives channel lane % 16 of token rows 4 g .. 4 g + 3.
    const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
    const int fr_d = wr * DW_HB + (4 * lg + lq) * DW_ROWB + 8 * lp;
    const int fr_x = OPB + wc * DW_HB + (4 * lg + lq) * DW_ROWB + 8 * lp;
    bf16x8 fd[NPL][5], fx[NPL][5];
    auto read_frags = [&](const int slot) {
        const uint32_t ad = lds0 + (uint32_t)(slot * SLOT + fr_d), ax = lds0 + (uint32_t)(slot * SLOT + fr_x);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            dw_s16x4 v[20];  // [tile][d k-half 0, d k-half 1, x k-half 0, x k-half 1]
#define DW_RD(T)                                                                                                        \
    v[4 * T + 0] = pl ? dw_tr_read<DW_PL + T * 32>(ad) : dw_tr_read<T * 32>(ad);                                        \
    v[4 * T + 1] = pl ? dw_tr_read<DW_PL + T * 32 + 16 * DW_ROWB>(ad) : dw_tr_read<T * 32 + 16 * DW_ROWB>(ad);          \
    v[4 * T + 2] = pl ? dw_tr_read<DW_PL + T * 32>(ax) : dw_tr_read<T * 32>(ax);                                        \
    v[4 * T + 3] = pl ? dw_tr_read<DW_PL + T * 32 + 16 * DW_ROWB>(ax) : dw_tr_read<T * 32 + 16 * DW_ROWB>(ax);
            DW_RD(0) DW_RD(1) DW_RD(2) DW_RD(3) DW_RD(4)
#undef DW_RD
            dw_landed(v);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                fd[pl][t] = __builtin_bit_cast(bf16x8, (dw_s16x8)__builtin_shufflevector(v[4 * t], v[4 * t + 1], 0, 1, 2, 3, 4, 5, 6, 7));
                fx[pl][t] = __builtin_bit_cast(bf16x8, (dw_s16x8)__builtin_shufflevector(v[4 * t + 2], v[4 * t + 3], 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
    };
    f32x4 acc[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfma_all = [&]() {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int pd = ps == 2 ? 1 : 0, px = ps == 1 ? 1 : 0;   // hi.hi, hi.lo, lo.hi
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[NPL == 2 ? pd : 0][i], fx[NPL == 2 ? px : 0][t], acc[i][t], 0, 0, 0);
        }
    };
    // ---- prologue: each group fetches its first sub-stage (0 -> slot 0, 1 -> slot 1)
    issue(grp, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- half-steps k = 0 .. U: group (k & 1) READS sub-stage k - after it has put the DMA of sub-stage k + 2 (its own next one) on the
    // way into the slot the other group finished with a half-step ago - while the other group MULTIPLIES sub-stage k - 1.  DMA issue
    // is expensive for the issuing wave (the vector-memory front end takes ~18 cycles per kilobyte instruction, 44 of them per
    // sub-stage): issued by all eight waves at the top of every half-step it held the MFMA waves back 650 - 1000 cycles of a
    // 2300-cycle half-step; the reading group has those cycles to spare.  A wave waits for its DMA (vmcnt(0)) at the end of its
    // multiply half-step, one half-step before the data is read.
    // Each group runs its own straight-line loop (a branch around the MFMA block inside one shared loop made hipcc copy the 100
    // accumulator registers at the join); the barriers pair up by count: both groups execute U + 1 half-steps.
    // (-DWD_DW_STAMPS + a.dbg & 0x100: s_memtime sums of workgroup 0's waves 0 and 4 into a.stamps - per kind of half-step
    // (0 read, 1 multiply) the ticks spent in DMA issue, the work, the waits and the barrier; tools/dw_bench.py --stamps)
#ifdef WD_DW_STAMPS
    const bool stamp = (a.dbg & 0x100) && blockIdx.x == 0 && a.stamps;
    unsigned long long tsum[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#define DW_STAMP(v) if (stamp) v = __builtin_amdgcn_s_memtime()
#define DW_STAMP_SUM(kind, ksub)                                     \
    if (stamp && (ksub) >= 2 && (ksub) + 3 < U) {                    \
        const unsigned long long t4 = __builtin_amdgcn_s_memtime(); \
        tsum[kind][0] += t1 - t0;                                    \
        tsum[kind][1] += t2 - t1;                                    \
        tsum[kind][2] += t3 - t2;                                    \
        tsum[kind][3] += t4 - t3;                                    \
    }
#else
#define DW_STAMP(v)
#define DW_STAMP_SUM(kind, ksub)
#endif
    int slot_r = grp;  // slot of this group's current sub-stage; its next one goes two slots on (= one back)
    auto hs_read = [&](const int ksub) {
        const int slot_w = slot_r == 0 ? 2 : slot_r - 1;
        DW_STAMP(t0);
        issue(slot_w, ksub + 2 < U);
        __builtin_amdgcn_sched_barrier(0);
        DW_STAMP(t1);
        read_frags(slot_r);
        __builtin_amdgcn_sched_barrier(0);
        DW_STAMP(t2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the slot are done before anyone refills it
        DW_STAMP(t3);
        __builtin_amdgcn_s_barrier();  // (not __syncthreads(): its fence makes hipcc wait for ALL the DMA in flight)
        __builtin_amdgcn_sched_barrier(0);
        DW_STAMP_SUM(0, ksub);
        slot_r = slot_w;
    };
    auto hs_mfma = [&](const int ksub) {
        DW_STAMP(t0);
        DW_STAMP(t1);
        mfma_all();
        __builtin_amdgcn_sched_barrier(0);
        DW_STAMP(t2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the group's next sub-stage has landed
        DW_STAMP(t3);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        DW_STAMP_SUM(1, ksub);
    };
    float* ep = reinterpret_cast<float*>(smem);
    const int er = wr * 80 + 4 * lg, ec = wc * 80 + (lane & 15);
    if (grp == 0) {
#pragma unroll 1
        for (int kk = 0; kk < U; kk += 2) {
            hs_read(kk);
            hs_mfma(kk);
        }
        __builtin_amdgcn_s_barrier();  // half-step U: group 1 multiplies its last sub-stage
        __syncthreads();               // group 1's tile is in the image
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(er + 16 * i + r) * DW_LDE + ec + 16 * t] += acc[i][t][r];
    } else {
        __builtin_amdgcn_s_barrier();  // half-step 0: group 0 reads sub-stage 0
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int kk = 1; kk < U; kk += 2) {
            hs_read(kk);
            hs_mfma(kk);
        }
        // (hs_mfma's vmcnt(0) covered this group's trailing no-op DMA as well: nothing lands in what becomes the image)
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(er + 16 * i + r) * DW_LDE + ec + 16 * t] = acc[i][t][r];
        __syncthreads();
    }
    __syncthreads();
#ifdef WD_DW_STAMPS
    if (stamp && lane == 0 && (wave & 3) == 0) {
        unsigned long long* sb = reinterpret_cast<unsigned long long*>(a.stamps) + grp * 8;
        for (int q = 0; q < 8; ++q) sb[q] = tsum[q >> 2][q & 3];
    }
#endif
    // ---- everybody streams the summed rows to the slice's slab
    const long kt = (long)a.ntaps * a.c;
    float* slab = a.ws + ((long)sidx * nprob + pi) * a.n * kt;
    for (int idx = tid; idx < DW_TB * (DW_TB / 4); idx += DW_NT) {
        const int row = idx / (DW_TB / 4), c4 = idx - row * (DW_TB / 4);
        const float4 v = *reinterpret_cast<const float4*>(ep + row * DW_LDE + c4 * 4);
        *reinterpret_cast<float4*>(slab + (long)(n0 + row) * kt + (long)tap * a.c + c0 + c4 * 4) = v;
    }
#endif
}

// grad[n][ci * T + t] (+)= sum over slices of ws[slice][n][t * C + ci].  One workgroup per (n, 64 input channels): the slabs are
// read tap row by tap row (64 consecutive floats), the sums turned to (ci, t) order through LDS and written as one contiguous run of
// 64 T floats - both sides coalesced (a thread per (n, t, four ci) wrote its four results 4 T bytes apart).
__global__ void __launch_bounds__(256) wd_dw_combine_kernel(const wd_dw_args a0, const int nslice) {
    __shared__ float s_out[64 * 9];
    wd_dw_args a = a0;
    const int nprob = a.items ? a.nitems : 1, pi = blockIdx.y;
    if (a.items) {
        const wd_dw_item it = a.items[pi];
        a.grad = it.grad; a.grad_ld = it.grad_ld; a.accumulate = it.accumulate;
    }
    const int T = a.ntaps;
    const int nblk = (a.c + 63) >> 6;          // (c % 160 == 0 does not make c a multiple of 64: the last block may be short)
    const int n = blockIdx.x / nblk, c0 = (blockIdx.x - n * nblk) * 64;
    const int cw = a.c - c0 < 64 ? a.c - c0 : 64;
    const long kt = (long)T * a.c, slab = (long)nprob * a.n * kt;
    const float* p = a.ws + (long)pi * a.n * kt + (long)n * kt + c0;
    for (int j = threadIdx.x; j < 64 * T; j += 256) {
        const int t = j >> 6, cl = j & 63;
        if (cl < cw) {
            const float* q = p + (long)t * a.c + cl;
            float v = q[0];
            for (int s = 1; s < nslice; ++s) v += q[s * slab];
            s_out[cl * T + t] = v;
        }
    }
    __syncthreads();
    float* g = a.grad + (long)n * a.grad_ld + (long)c0 * T;
    for (int j = threadIdx.x; j < cw * T; j += 256) g[j] = a.accumulate ? g[j] + s_out[j] : s_out[j];
}

// the same for one tap (linear layers, 1x1 convolutions): no reordering, a float4 of the gradient per thread
__global__ void __launch_bounds__(256) wd_dw_combine1_kernel(const wd_dw_args a0, const int nslice) {
    wd_dw_args a = a0;
    const int nprob = a.items ? a.nitems : 1, pi = blockIdx.y;
    if (a.items) {
        const wd_dw_item it = a.items[pi];
        a.grad = it.grad; a.grad_ld = it.grad_ld; a.accumulate = it.accumulate;
    }
    const int c4n = a.c >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.n * c4n) return;
    const int n = (int)(i / c4n), ci = (int)(i - (long)n * c4n) * 4;
    const float* p = a.ws + (long)pi * a.n * a.c + (long)n * a.c + ci;
    float4 v = *reinterpret_cast<const float4*>(p);
    const long slab = (long)nprob * a.n * a.c;
    for (int s = 1; s < nslice; ++s) {
        const float4 q = *reinterpret_cast<const float4*>(p + s * slab);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    float* g = a.grad + (long)n * a.grad_ld + ci;
    if (a.accumulate) {
        v.x += g[0]; v.y += g[1]; v.z += g[2]; v.w += g[3];
    }
    if (((reinterpret_cast<uintptr_t>(g)) & 15) == 0) {
        *reinterpret_cast<float4*>(g) = v;
    } else {
        g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
    }
}

int dw_auto_slices(const int tiles, const int units) {
    // as many token slices as it takes to put a workgroup on every CU, whole rounds of 256 preferred, >= 2 units (128 tokens) a slice
    int best = 1;
    double best_cost = 1e30;
    const int smax = units / 2 < 64 ? units / 2 : 64;
    for (int s = 1; s <= smax; ++s) {
        const long wgs = (long)tiles * s;
        const long rounds = (wgs + 255) / 256;
        // time ~ rounds x (sub-stages of a slice + fixed prologue / epilogue of ~6 sub-stages)
        const double cost = (double)rounds * (2.0 * ((units + s - 1) / s) + 6.0);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
}

template <int NPASS>
int dw_launch(const wd_dw_args& a, const int nslice, hipStream_t st) {
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int ring = 3 * 2 * NPL * DW_PL + DW_MAXHW * 4;
    constexpr int image = DW_TB * DW_LDE * 4;
    constexpr int smem = ring > image ? ring : image;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_dw_kernel<NPASS>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
            hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int ntn = a.n / DW_TB, ntc = a.c / DW_TB;
    const int nprob = a.items ? a.nitems : 1;
    {
        WdLaunchScope scope(WD_CLS_DW, st, 2.0 * (double)nprob * (double)a.m * (double)a.n * (double)a.c * (double)a.ntaps);
        hipLaunchKernelGGL((wd_dw_kernel<NPASS>), dim3(ntn * ntc * a.ntaps * nprob * nslice), dim3(DW_NT), smem, st, a, ntn, ntc, nslice);
    }
    {
        WdLaunchScope scope(WD_CLS_GEMM_REDUCE, st);
        if (a.ntaps == 1)
            hipLaunchKernelGGL(wd_dw_combine1_kernel, dim3((unsigned)(((long)a.n * (a.c >> 2) + 255) / 256), nprob), dim3(256), 0, st, a,
                               nslice);
        else
            hipLaunchKernelGGL(wd_dw_combine_kernel, dim3((unsigned)(a.n * ((a.c + 63) / 64)), nprob), dim3(256), 0, st, a, nslice);
    }
    return wd_check_launch();
}

}  // namespace

extern "C" int wd_dw_supported(int m, int n, int c, int ntaps, int hw_out) {
    return m > 0 && m % 64 == 0 && n > 0 && n % DW_TB == 0 && c > 0 && c % DW_TB == 0 && ntaps >= 1 && ntaps <= 9 && hw_out > 0 &&
           hw_out % 64 == 0 && hw_out <= DW_MAXHW && m % hw_out == 0;
}

extern "C" int wd_dw_slices(int m, int n, int c, int ntaps) {
    if (m < 128 || n < DW_TB || c < DW_TB) return 1;
    return dw_auto_slices((n / DW_TB) * (c / DW_TB) * ntaps, m / 64);
}

extern "C" int wd_dw_args_bytes(void) { return (int)sizeof(wd_dw_args); }

static int dw_check_operands(const wd_dw_args& a, const wd_bf16* d_hi, const wd_bf16* d_lo, const wd_bf16* x_hi, const wd_bf16* x_lo,
                             const float* grad, int d_ld, int x_ld, int grad_ld) {
    if (!d_hi || !x_hi || !grad) return WD_EINVAL;
    if (a.npass == 3 && (!d_lo || !x_lo)) return WD_EINVAL;
    if (d_ld < a.n || x_ld < a.c || (d_ld & 7) || (x_ld & 7) || grad_ld < a.c * a.ntaps) return WD_EINVAL;
    if (((reinterpret_cast<uintptr_t>(d_hi) | reinterpret_cast<uintptr_t>(d_lo) | reinterpret_cast<uintptr_t>(x_hi) |
          reinterpret_cast<uintptr_t>(x_lo)) & 15))
        return WD_EINVAL;
    // (operand planes are addressed with 32-bit byte offsets below 2 GiB)
    if ((long)a.m * d_ld * 2 >= 0x7FFFFFF0L || (long)(a.m / a.hw_out) * a.hw_src * x_ld * 2 >= 0x7FFFFFF0L) return WD_EINVAL;
    return WD_OK;
}

static int dw_run(wd_dw_args& a, const int nprob, void* stream) {
    if (!a.ws || (reinterpret_cast<uintptr_t>(a.ws) & 15) || (a.npass != 1 && a.npass != 3)) return WD_EINVAL;
    if (!a.gather && (a.ntaps != 1 || a.hw_src != a.hw_out)) return WD_EINVAL;
    if (a.gather && a.hw_src <= 0) return WD_EINVAL;
    const int units = a.m / 64;
    const long per_slice = (long)nprob * a.n * a.c * a.ntaps;
    int nslice = a.nslice;
    if (nslice <= 0) nslice = dw_auto_slices((a.n / DW_TB) * (a.c / DW_TB) * a.ntaps * nprob, units);
    if (nslice > units) nslice = units;
    while (nslice > 1 && nslice * per_slice > a.ws_floats) --nslice;
    if (nslice * per_slice > a.ws_floats) return WD_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return a.npass == 3 ? dw_launch<3>(a, nslice, st) : dw_launch<1>(a, nslice, st);
}

extern "C" int wd_dw_group_slices(int m, int n, int c, int ntaps, int nitems) {
    if (m < 128 || n < DW_TB || c < DW_TB || nitems < 1) return 1;
    return dw_auto_slices((n / DW_TB) * (c / DW_TB) * ntaps * nitems, m / 64);
}

extern "C" int wd_dw_item_bytes(void) { return (int)sizeof(wd_dw_item); }

extern "C" int wd_dw_group(const wd_dw_args* pa, const wd_dw_item* items_host, const wd_dw_item* items_dev, int nitems, void* stream) {
    if (!pa || !items_host || !items_dev || nitems < 1 || nitems > 64) return WD_EINVAL;
    wd_dw_args a = *pa;
    if (!wd_dw_supported(a.m, a.n, a.c, a.ntaps, a.hw_out)) return WD_EINVAL;
    for (int i = 0; i < nitems; ++i) {
        const wd_dw_item& it = items_host[i];
        const int rc = dw_check_operands(a, it.d_hi, it.d_lo, it.x_hi, it.x_lo, it.grad, it.d_ld, it.x_ld, it.grad_ld);
        if (rc != WD_OK) return rc;
    }
    a.items = items_dev;
    a.nitems = nitems;
    return dw_run(a, nitems, stream);
}

extern "C" int wd_dw(const wd_dw_args* pa, void* stream) {
    if (!pa) return WD_EINVAL;
    wd_dw_args a = *pa;
    if (!wd_dw_supported(a.m, a.n, a.c, a.ntaps, a.hw_out)) return WD_EINVAL;
    const int rc = dw_check_operands(a, a.d_hi, a.d_lo, a.x_hi, a.x_lo, a.grad, a.d_ld, a.x_ld, a.grad_ld);
    if (rc != WD_OK) return rc;
    a.items = nullptr;
    a.nitems = 0;
    return dw_run(a, 1, stream);
}
