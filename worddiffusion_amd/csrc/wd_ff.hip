// The tail of a transformer block in ONE launch per 64-token panel (gfx950):
//
//     h   = GEGLU(LN3(x) W1^T + b1)              FeedForward / GEGLU, reference unet.py:122-149 (unetPhosc.py:103-130)
//     x'  = x + h W2^T + b2                      BasicTransformerBlock residual, unet.py:343-344
//     out = x_in + x' Wo^T + bo   (optional)     SpatialTransformer.proj_out + residual, unet.py:406-412
//
// As three wd_gemm launches the hidden activations made a round trip through HBM as split-bf16 planes (84 MB written and read
// per layer at 64 x 8 x 32 tokens), the 2048 five-stage tiles of the first projection paid a prologue and an epilogue each, and
// the two short-K products that follow are bound by their own activation traffic.  Here a workgroup keeps its 64 normalised
// token rows resident in LDS (80 KB as split-bf16 planes), walks the hidden dimension in chunks of 128 units and never stores
// h: per chunk the 8 waves each own 16 hidden units (one x tile + one gate tile of v_mfma_f32_16x16x32_bf16, so x and gate of
// an element sit in the same lane and register), GEGLU runs on the accumulators, h goes to a 32 KB LDS image as the A operand
// of the second product, whose 64 x 320 result stays in 80 accumulator registers per wave (2 K-halves x 4 column groups, as
// in wd_gemmw_kernel) over all chunks.  Every weight is read exactly once per workgroup, straight from its FRAGMENT-MAJOR image
// (wd_gemm_pack_w) into the MFMA operand registers through a ring of six two-load groups (12 KB per wave in flight).
#include "wd_gemm_epi.h"

namespace {

constexpr int FNT = 512;
constexpr int FBM = 64;
constexpr int FC = 320;          // token width (model_channels of every reference configuration that reaches this kernel)
constexpr int FCH = 128;         // hidden units per chunk
constexpr int FRING = 6;         // weight-fragment groups in flight per wave
constexpr int F_MAX_INNER = 2048; // hidden width the LDS copy of b1 leaves room for (144 KB of rows and h images + 8 bytes per unit)
constexpr int FGRP = 30;         // groups per chunk: 10 k-steps x (x, gate) + 2 k-steps x 5 column tiles
constexpr uint32_t F_OOB = 0x80000000u;

typedef __attribute__((ext_vector_type(4))) unsigned f_u32x4;

__device__ __forceinline__ int f_lds_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

template <int NPASS, bool PROJ>
__global__ void __launch_bounds__(FNT, 1) wd_ff_kernel(const wd_ff_args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NPL = (NPASS == 1) ? 1 : 2;
    constexpr int SLAB = 2 * FBM * 128;            // one 64-deep K slab of 64 rows, both planes (plane stride FBM * 128)
    constexpr int A_BYTES = 5 * SLAB;              // the resident token rows: 320 channels
    constexpr int H_BYTES = 2 * SLAB;              // one h image: 128 hidden units
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_a = smem;
    char* s_h = smem + A_BYTES;                    // two h images
    float* s_b1 = reinterpret_cast<float*>(smem + A_BYTES + 2 * H_BYTES);   // b1 (2 * inner floats): a global load of the bias inside
    //                                                 the chunk loop drains the weight ring once per chunk (vmcnt counts in order)

    const int m0 = blockIdx.x * FBM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, cg = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    const int nchunk = a.inner / FCH;

    auto make_srd = [](const wd_bf16* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<wd_bf16*>(p), 0, 0x7FFFFFF0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t srd1_hi = make_srd(a.w1_hi), srd1_lo = make_srd(a.w1_lo ? a.w1_lo : a.w1_hi);
    const __amdgpu_buffer_rsrc_t srd2_hi = make_srd(a.w2_hi), srd2_lo = make_srd(a.w2_lo ? a.w2_lo : a.w2_hi);
    const __amdgpu_buffer_rsrc_t srd3_hi = make_srd(PROJ ? a.w3_hi : a.w2_hi), srd3_lo = make_srd(PROJ ? (a.w3_lo ? a.w3_lo : a.w3_hi) : a.w2_hi);
    const uint32_t lane16 = lane * 16;
    const int nct1 = (2 * a.inner) >> 4;          // column tiles of W1 (x / gate tiles alternate: wd_ff_fused's packing)

    // ---- the weight stream: group G of chunk j -> (descriptor, byte offset) of its two kilobytes (hi, lo)
    //   g < 20: first product, k-step g >> 1, tile (g & 1: x | gate) of this wave's 16 hidden units
    //   g >= 20: second product, k-step 4 j + 2 kh + (g - 20) / 5 of W2, column tile 5 cg + (g - 20) % 5
    bf16x8 ring[FRING][NPL];
    auto issue = [&](const int slot, const int g, const int j) {  // slot, g compile-time; j run-time (uniform)
        const bool live = j < nchunk;
        const uint32_t vo = live ? lane16 : F_OOB;
        if (PROJ && g < FRING) {
            // the first groups of "chunk nchunk" are the first groups of the proj_out product (k-step 2 (g / 5) + kh, tile g % 5): ONE
            // pair of loads with the descriptor and offset selected - a load inside a run-time branch makes hipcc's wait-count pass
            // assume it may not have been issued, every such branch lowers the vmcnt it dares to wait for by two, and the six of
            // them at the end of a chunk drained the ring once per chunk
            const uint32_t so3 = (uint32_t)(((2 * (g / 5) + kh) * (FC / 16) + 5 * cg + g % 5) * 1024);
            const uint32_t so1 = (uint32_t)(((g >> 1) * nct1 + 2 * (j * 8 + wave) + (g & 1)) * 1024);
            const uint32_t so = live ? so1 : so3;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                const __amdgpu_buffer_rsrc_t srd = live ? (p ? srd1_lo : srd1_hi) : (p ? srd3_lo : srd3_hi);
                ring[slot][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd, lane16, so, 0));
            }
        } else if (g < 20) {
            const uint32_t so = (uint32_t)(((g >> 1) * nct1 + 2 * (j * 8 + wave) + (g & 1)) * 1024);
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                ring[slot][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(p ? srd1_lo : srd1_hi, vo, live ? so : 0u, 0));
        } else {
            const int q = (g - 20) / 5, t = (g - 20) % 5;
            const uint32_t so = (uint32_t)(((4 * j + 2 * kh + q) * (FC / 16) + 5 * cg + t) * 1024);
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                ring[slot][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(p ? srd2_lo : srd2_hi, vo, live ? so : 0u, 0));
        }
    };
    // the first FRING groups go out before anything else (they have the whole prologue to land)
#pragma unroll
    for (int g = 0; g < FRING; ++g) issue(g, g, 0);

    // ---- the token rows: global planes -> LDS (5 slabs x 2 planes, 16-byte chunks, the GEMM stage image)
    {
        const __amdgpu_buffer_rsrc_t sx_hi = make_srd(a.x_hi), sx_lo = make_srd(a.x_lo ? a.x_lo : a.x_hi);
        const int arow = tid >> 3, ach = tid & 7;
        const uint32_t vo = (m0 + arow < a.m) ? (uint32_t)(m0 + arow) * (uint32_t)(a.x_ld * 2) + (uint32_t)(ach * 16) : F_OOB;
        f_u32x4 v[5][NPL];
#pragma unroll
        for (int sl = 0; sl < 5; ++sl)
#pragma unroll
            for (int p = 0; p < NPL; ++p) v[sl][p] = __builtin_amdgcn_raw_buffer_load_b128(p ? sx_lo : sx_hi, vo, sl * 128, 0);
        const int dst = f_lds_off(arow, ach);
#pragma unroll
        for (int sl = 0; sl < 5; ++sl)
#pragma unroll
            for (int p = 0; p < NPL; ++p) *reinterpret_cast<f_u32x4*>(s_a + sl * SLAB + p * (FBM * 128) + dst) = v[sl][p];
        for (int i = tid; i < 2 * a.inner; i += FNT) s_b1[i] = a.b1 ? a.b1[i] : 0.0f;
    }
    __syncthreads();

    f32x4 acc2[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[i][t][r] = 0.0f;

    bf16x8 xa[4][NPL];
    auto read_frags = [&](const char* slab_base, const int half) {  // the four row tiles of k-step `half` (0 / 1) of a 64-deep slab
        const int ch = half * 4 + lq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ao = f_lds_off(i * 16 + l15, ch);
#pragma unroll
            for (int p = 0; p < NPL; ++p) xa[i][p] = *reinterpret_cast<const bf16x8*>(slab_base + p * (FBM * 128) + ao);
        }
    };
    auto mfma12 = [&](f32x4 (&acc)[4], const bf16x8 (&w)[NPL]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (NPL == 2) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][NPL - 1], w[0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], w[NPL - 1], acc[i], 0, 0, 0);
            }
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], w[0], acc[i], 0, 0, 0);
        }
    };

    // bias of this wave's hidden units (x and gate) changes per chunk: read inside the loop
    for (int j = 0; j < nchunk; ++j) {
        char* hbuf = s_h + (j & 1) * H_BYTES;
        // ---- first product: 16 hidden units (x tile, gate tile) x 64 tokens, K = 320
        f32x4 ax[4], ag[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ax[i][r] = 0.0f;
                ag[i][r] = 0.0f;
            }
#pragma unroll
        for (int g = 0; g < 20; ++g) {
            if ((g & 1) == 0) read_frags(s_a + (g >> 2) * SLAB, (g >> 1) & 1);
            if (g & 1) mfma12(ag, ring[g % FRING]);
            else mfma12(ax, ring[g % FRING]);
            // the slot is free: group g + FRING of this chunk, or of the next
            if (g + FRING < FGRP) issue(g % FRING, g + FRING, j);
            else issue(g % FRING, g + FRING - FGRP, j + 1);
            __builtin_amdgcn_sched_barrier(0);  // (else hipcc sinks the loads to just before their use: no ring left)
        }
        // ---- GEGLU on the accumulators (x and gate of an element share lane and register), h -> LDS as split-bf16 planes
        {
            const int hcol = (j * 8 + wave) * 32 + l15;            // bias index of the x unit; its gate: + 16
            const float bx = s_b1[hcol], bg = s_b1[hcol + 16];
            const int slab = wave >> 2, ch = (wave & 3) * 2 + (l15 >> 3), e = (l15 & 7) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = (ax[i][r] + bx) * wd_gelu_erf(ag[i][r] + bg);
                    uint32_t hb, lb;
                    wd_split1(h, hb, lb);
                    char* p = hbuf + slab * SLAB + f_lds_off(i * 16 + lq * 4 + r, ch) + e;
                    *reinterpret_cast<unsigned short*>(p) = (unsigned short)hb;
                    if (NPL == 2) *reinterpret_cast<unsigned short*>(p + FBM * 128) = (unsigned short)lb;
                }
        }
        __syncthreads();  // h(j) complete; (the other h image was last read before the previous barrier)
        // ---- second product: this wave's K-half (two k-steps of the chunk) x its 80 output columns
#pragma unroll
        for (int g = 20; g < FGRP; ++g) {
            const int q = (g - 20) / 5, t = (g - 20) % 5;
            if (t == 0) read_frags(hbuf + kh * SLAB, q);
            {
                f32x4 col[4] = {acc2[0][t], acc2[1][t], acc2[2][t], acc2[3][t]};
                mfma12(col, ring[g % FRING]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[i][t] = col[i];
            }
            if (g + FRING < FGRP) issue(g % FRING, g + FRING, j);
            else issue(g % FRING, g + FRING - FGRP, j + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the two K-halves summed in a fixed order through the fp32 image, then the shared GEMM epilogue (+ b2, + x, planes / fp32)
    constexpr int LDE = FC + 4;
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
                        float* pe = ep + (i * 16 + 4 * lq + r) * LDE + cg * 80 + t * 16 + l15;
                        *pe = (hh == 0) ? acc2[i][t][r] : *pe + acc2[i][t][r];
                    }
        }
        __syncthreads();
    }
    wd_gemm_args e = {};
    e.m = a.m;
    e.n = FC;
    e.hw_out = a.hw_out > 0 ? a.hw_out : 1;
    e.act = WD_ACT_NONE;
    e.out_f32 = a.out_f32;
    e.out_ld = a.out_ld;
    e.out_hi = a.out_hi;
    e.out_lo = a.out_lo;
    e.out_pl_ld = a.out_pl_ld;
    e.ksplit = 1;
    e.stat_part = a.stat_part;
    e.stat_cpg = a.stat_cpg;
    if constexpr (!PROJ) {
        e.bias = a.b2;
        e.resid = a.resid;
        e.resid_ld = a.resid_ld;
        wd_epilogue_from_image<FBM, FC, FNT>(e, ep, m0, 0, tid);
    } else {
        // ---- x' = image + b2 + x -> split-bf16 planes IN PLACE (a row's 320 floats = 1296 bytes with the pitch; its planes take
        // 2 x 640): wave w converts rows 8 w .. 8 w + 7, every read of those rows before the first write (same wave, in order)
        constexpr int ROWB = LDE * 4;
        {
            float4 v[10];
#pragma unroll
            for (int it = 0; it < 10; ++it) {
                const int idx = it * 64 + lane, row = wave * 8 + idx / 80, c = (idx % 80) * 4;
                const int m = m0 + row;
                float4 x = *reinterpret_cast<const float4*>(ep + row * LDE + c);
                if (a.b2) {
                    const float4 q = *reinterpret_cast<const float4*>(a.b2 + c);
                    x.x += q.x; x.y += q.y; x.z += q.z; x.w += q.w;
                }
                if (a.resid && m < a.m) {
                    const float4 q = *reinterpret_cast<const float4*>(a.resid + (long)m * a.resid_ld + c);
                    x.x += q.x; x.y += q.y; x.z += q.z; x.w += q.w;
                }
                v[it] = x;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < 10; ++it) {
                const int idx = it * 64 + lane, row = wave * 8 + idx / 80, c = (idx % 80) * 4;
                uint2 hh, ll;
                wd_split4(v[it], hh, ll);
                char* rowp = reinterpret_cast<char*>(ep) + row * ROWB;
                *reinterpret_cast<uint2*>(rowp + c * 2) = hh;
                if (NPL == 2) *reinterpret_cast<uint2*>(rowp + 640 + c * 2) = ll;
            }
        }
        __syncthreads();
        // ---- out = x' Wo^T: K = 320 as ten k-steps, K-half kh takes the k-steps 2 q + kh; 25 groups through the same ring
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc2[i][t][r] = 0.0f;
#pragma unroll
        for (int g = 0; g < 25; ++g) {
            const int q = g / 5, t = g % 5;
            if (t == 0) {
                const int ks = 2 * q + kh;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const char* rp = reinterpret_cast<const char*>(ep) + (i * 16 + l15) * ROWB + (ks * 4 + lq) * 16;
#pragma unroll
                    for (int p = 0; p < NPL; ++p) xa[i][p] = *reinterpret_cast<const bf16x8*>(rp + p * 640);
                }
            }
            {
                f32x4 col[4] = {acc2[0][t], acc2[1][t], acc2[2][t], acc2[3][t]};
                mfma12(col, ring[g % FRING]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[i][t] = col[i];
            }
            if (g + FRING < 25) {
                const int g2 = g + FRING;
                const uint32_t so3 = (uint32_t)(((2 * (g2 / 5) + kh) * (FC / 16) + 5 * cg + g2 % 5) * 1024);
#pragma unroll
                for (int p = 0; p < NPL; ++p)
                    ring[g % FRING][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(p ? srd3_lo : srd3_hi, lane16, so3, 0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // every fragment read of the planes is done: the image goes over them
        for (int hh = 0; hh < 2; ++hh) {
            if (kh == hh) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 5; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* pe = ep + (i * 16 + 4 * lq + r) * LDE + cg * 80 + t * 16 + l15;
                            *pe = (hh == 0) ? acc2[i][t][r] : *pe + acc2[i][t][r];
                        }
            }
            __syncthreads();
        }
        e.bias = a.b3;
        e.resid = a.resid3;
        e.resid_ld = a.resid3_ld;
        wd_epilogue_from_image<FBM, FC, FNT>(e, ep, m0, 0, tid);
    }
#endif
}

template <int NPASS, bool PROJ>
int launch_ff(const wd_ff_args& a, hipStream_t st) {
    constexpr int max_smem = 5 * 2 * FBM * 128 + 2 * 2 * 2 * FBM * 128 + 2 * F_MAX_INNER * 4;  // token rows + two h images + b1
    constexpr int red_smem = FBM * (FC + 4) * 4 + WD_STAT_SCRATCH;
    static_assert(max_smem <= 160 * 1024 && red_smem <= max_smem, "LDS budget");
    const int loop_smem = max_smem - 2 * (F_MAX_INNER - a.inner) * 4;
    const int smem = loop_smem > red_smem ? loop_smem : red_smem;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wd_ff_kernel<NPASS, PROJ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                max_smem) != hipSuccess)
            return WD_ELAUNCH;
        attr_done = true;
    }
    const int nb = (a.m + FBM - 1) / FBM;
    const double fl = 2.0 * (double)a.m * (3.0 * (double)a.inner * FC + (PROJ ? (double)FC * FC : 0.0));
    WdLaunchScope scope(WD_CLS_FF, st, fl);
    hipLaunchKernelGGL((wd_ff_kernel<NPASS, PROJ>), dim3(nb), dim3(FNT), smem, st, a);
    return wd_check_launch();
}

}  // namespace

extern "C" int wd_ff_args_bytes(void) { return (int)sizeof(wd_ff_args); }

extern "C" int wd_ff_supported(int c, int inner) { return (c == FC && inner > 0 && inner % FCH == 0 && inner <= F_MAX_INNER) ? 1 : 0; }

extern "C" int wd_ff_fused(const wd_ff_args* pa, void* stream) {
    if (!pa) return WD_EINVAL;
    const wd_ff_args& a = *pa;
    if (!wd_ff_supported(a.c, a.inner) || a.m <= 0 || (a.npass != 1 && a.npass != 3)) return WD_EINVAL;
    if (!a.x_hi || !a.w1_hi || !a.w2_hi || (a.npass == 3 && (!a.x_lo || !a.w1_lo || !a.w2_lo))) return WD_EINVAL;
    if (!a.out_f32 && !a.out_hi) return WD_EINVAL;
    if (a.x_ld % 8 || a.x_ld < a.c) return WD_EINVAL;
    if ((a.out_ld | a.resid_ld | a.out_pl_ld) & 3) return WD_EINVAL;  // the vector epilogue
    if (a.resid && a.resid_ld <= 0) return WD_EINVAL;
    if ((long)a.m * a.x_ld * 2 >= 0x7FFFFFF0L || (long)2 * a.inner * a.c * 2 >= 0x7FFFFFF0L) return WD_EINVAL;
    const bool proj = a.w3_hi != nullptr;
    if (proj && ((a.npass == 3 && !a.w3_lo) || (a.resid3 && (a.resid3_ld <= 0 || (a.resid3_ld & 3))))) return WD_EINVAL;
    if (a.stat_part && (a.stat_cpg <= 0 || FC % a.stat_cpg || a.hw_out <= 0 || !(a.hw_out % FBM == 0 || FBM % a.hw_out == 0) ||
                        (FBM > a.hw_out && FBM / a.hw_out > WD_STAT_MAXNS)))
        return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (proj) return a.npass == 3 ? launch_ff<3, true>(a, st) : launch_ff<1, true>(a, st);
    return a.npass == 3 ? launch_ff<3, false>(a, st) : launch_ff<1, false>(a, st);
}
