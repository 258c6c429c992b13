// Weight repack on the device: reference-layout fp32 parameters (OIHW convolutions, [out, in] linears, vectors) -> the
// operands the kernels consume (split-bf16 planes [rows][K], K = tap-major / channel-minor; the transposed
// [C_in][tap][C_out] planes of the data-gradient GEMMs; concatenated / summed fp32 vectors).  One table-driven launch
// refreshes every operand after an optimiser step (the reference re-reads its parameters in place each forward; here the
// packed copies are part of the step), instead of ~10 small ATen launches per tensor.
#include "wd_common.h"

namespace {

struct PackRef {        // one piece of one packed operand; built once by the host (engine.py)
    const float* src0;  // [N][C][T] (T = 9 for 3x3 kernels, 1 for linears) or a vector of N floats
    const float* src1;  // optional second addend (vectors only: conv bias + skip bias)
    void* dst_hi;       // planes: bf16 hi plane, already offset to (row_off, col_off); vectors: fp32 destination
    void* dst_lo;
    int32_t N, C, T, mode;     // mode 0: forward planes, 1: data-gradient planes, 2: fp32 vector, 3: forward planes, fragment-major
    int32_t npad, ld, g, ntile_c;
    int64_t chunk0;     // index of this piece's first chunk (tile)
    int32_t row_off, col_off;  // mode 3 only (dst_hi / dst_lo are the image's base there, ld = the matrix's row count)
};

constexpr int PT = 32;          // tile: 32 output channels x 32 input channels x T taps
constexpr int VCHUNK = 8192;    // vector elements per workgroup

__device__ __forceinline__ int geglu_perm(int n, int N, int g) {
    // rows [x | gate] -> blocks of g x-rows followed by their g gate rows (engine.geglu_interleave)
    if (g <= 0) return n;
    const int inner = N >> 1;
    const int half = n >= inner;
    const int m = half ? n - inner : n;
    return (m / g) * 2 * g + half * g + (m % g);
}

template <int T>
__device__ __forceinline__ void repack_full_tile(const PackRef& e, float* s, const int n0, const int c0, const int tid) {
    constexpr int ROW = PT * T + 1;
    const int C = e.C, N = e.N;
    // load: for each output channel the (c, t) block of PT * T floats is contiguous in the OIHW source
    for (int idx = tid; idx < PT * PT * T; idx += 256) {
        const int nl = idx / (PT * T), rem = idx - nl * (PT * T);
        const int n = n0 + nl;
        s[nl * ROW + rem] = n < N ? e.src0[((int64_t)n * C + c0) * T + rem] : 0.f;
    }
    __syncthreads();
    uint32_t* dh = reinterpret_cast<uint32_t*>(e.dst_hi);
    uint32_t* dl = reinterpret_cast<uint32_t*>(e.dst_lo);
    const int pr = tid & 15, r0 = tid >> 4;  // pr: element pair inside the 32-wide run, r0: one of 16 run lanes
    if (e.mode == 3) {
        // the fragment-major image of mode 0's matrix (wd_gemm_pack_w): 16-byte chunk ((k / 32) * rows / 16 + row / 16) * 64 +
        // (row & 15) + 16 * ((k / 8) & 3) holds elements k & ~7 .. of the row; 16 consecutive rows x 16 bytes are contiguous
        const int nct = e.ld >> 4;
        for (int r = r0; r < PT * T; r += 16) {
            const int nl = r / T, t = r - nl * T;
            const int n = n0 + nl;
            if (n >= N) continue;
            uint32_t h0, l0, h1, l1;
            wd_split1(s[nl * ROW + (2 * pr) * T + t], h0, l0);
            wd_split1(s[nl * ROW + (2 * pr + 1) * T + t], h1, l1);
            const int row = geglu_perm(n, N, e.g) + e.row_off;
            const int k = e.col_off + t * C + c0 + 2 * pr;
            const int64_t o = ((((int64_t)(k >> 5) * nct + (row >> 4)) * 64 + (row & 15) + 16 * ((k >> 3) & 3)) * 8 + (k & 7)) >> 1;
            dh[o] = h0 | (h1 << 16);
            if (dl) dl[o] = l0 | (l1 << 16);
        }
    } else if (e.mode == 0) {
        // dst[perm(n)][t * C + c]: c fastest -> runs of 32 channels per (n, t)
        for (int r = r0; r < PT * T; r += 16) {
            const int nl = r / T, t = r - nl * T;
            const int n = n0 + nl;
            if (n >= N) continue;
            uint32_t h0, l0, h1, l1;
            wd_split1(s[nl * ROW + (2 * pr) * T + t], h0, l0);
            wd_split1(s[nl * ROW + (2 * pr + 1) * T + t], h1, l1);
            const int64_t o = ((int64_t)geglu_perm(n, N, e.g) * e.ld + (int64_t)t * C + c0 + 2 * pr) >> 1;
            dh[o] = h0 | (h1 << 16);
            if (dl) dl[o] = l0 | (l1 << 16);
        }
    } else {
        // dst[c][t * npad + n]: n fastest -> runs of 32 output channels per (c, t); columns n >= N are written as zeros
        for (int r = r0; r < PT * T; r += 16) {
            const int cl = r / T, t = r - cl * T;
            uint32_t h0, l0, h1, l1;
            wd_split1(s[(2 * pr) * ROW + cl * T + t], h0, l0);
            wd_split1(s[(2 * pr + 1) * ROW + cl * T + t], h1, l1);
            const int64_t o = ((int64_t)(c0 + cl) * e.ld + (int64_t)t * e.npad + n0 + 2 * pr) >> 1;
            dh[o] = h0 | (h1 << 16);
            if (dl) dl[o] = l0 | (l1 << 16);
        }
    }
}

__global__ void __launch_bounds__(256) repack_multi_kernel(const PackRef* __restrict__ tab, int nentries) {
    __shared__ float s[PT * (PT * 9 + 1)];
    int lo = 0, hi = nentries - 1;
    const int64_t ch = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].chunk0 <= ch) lo = mid;
        else hi = mid - 1;
    }
    const PackRef e = tab[lo];
    const int64_t lc = ch - e.chunk0;
    const int tid = threadIdx.x;
    if (e.mode == 2) {
        float* dst = reinterpret_cast<float*>(e.dst_hi);
        const int64_t base = lc * VCHUNK, end = base + VCHUNK < e.N ? base + VCHUNK : e.N;
        for (int64_t i = base + tid; i < end; i += 256) {
            float v = e.src0[i];
            if (e.src1) v += e.src1[i];
            dst[geglu_perm((int)i, e.N, e.g)] = v;
        }
        return;
    }
    const int T = e.T, C = e.C, N = e.N;
    const int n0 = (int)(lc / e.ntile_c) * PT, c0 = (int)(lc % e.ntile_c) * PT;
    const int nc = min(PT, C - c0);           // valid input channels in this tile
    const int row = PT * T + 1;
    if (nc == PT && (T == 9 || T == 1) && (e.mode != 1 || e.npad - n0 >= PT) && (C & 1) == 0 && (e.ld & 1) == 0 &&
        (e.npad & 1) == 0 && ((reinterpret_cast<uintptr_t>(e.dst_hi) | reinterpret_cast<uintptr_t>(e.dst_lo)) & 3) == 0) {
        // full tile: every index below is a shift / a division by a compile-time constant (the generic path spends ~100
        // instructions per element on run-time div / mod), and the planes are written two elements (4 bytes) at a time
        if (T == 9) repack_full_tile<9>(e, s, n0, c0, tid);
        else repack_full_tile<1>(e, s, n0, c0, tid);
        return;
    }
    if (e.mode == 3) return;  // (wd_repack_multi's caller only builds mode-3 pieces of full tiles: C % 32 == 0, T in {1, 9})
    // load: for each output channel the (c, t) block is contiguous in the OIHW source
    for (int idx = tid; idx < PT * nc * T; idx += 256) {
        const int nl = idx / (nc * T), rem = idx - nl * (nc * T);
        const int n = n0 + nl;
        s[nl * row + rem] = n < N ? e.src0[((int64_t)n * C + c0) * T + rem] : 0.f;
    }
    __syncthreads();
    wd_bf16* dh = reinterpret_cast<wd_bf16*>(e.dst_hi);
    wd_bf16* dl = reinterpret_cast<wd_bf16*>(e.dst_lo);
    if (e.mode == 0) {
        // dst[perm(n)][t * C + c]: c fastest
        for (int idx = tid; idx < PT * T * nc; idx += 256) {
            const int cl = idx % nc, t = (idx / nc) % T, nl = idx / (nc * T);
            const int n = n0 + nl;
            if (n >= N) continue;
            uint32_t h, l;
            wd_split1(s[nl * row + cl * T + t], h, l);
            const int64_t o = (int64_t)geglu_perm(n, N, e.g) * e.ld + (int64_t)t * C + c0 + cl;
            dh[o] = (wd_bf16)h;
            if (dl) dl[o] = (wd_bf16)l;
        }
    } else {
        // dst[c][t * npad + n]: n fastest (columns n >= N of the padded block are written as zeros)
        const int nn = min(PT, e.npad - n0);
        for (int idx = tid; idx < nc * T * nn; idx += 256) {
            const int nl = idx % nn, t = (idx / nn) % T, cl = idx / (nn * T);
            uint32_t h, l;
            wd_split1(s[nl * row + cl * T + t], h, l);
            const int64_t o = (int64_t)(c0 + cl) * e.ld + (int64_t)t * e.npad + n0 + nl;
            dh[o] = (wd_bf16)h;
            if (dl) dl[o] = (wd_bf16)l;
        }
    }
}

}  // namespace

extern "C" int wd_repack_entry_bytes(void) { return (int)sizeof(PackRef); }
extern "C" int wd_repack_tile(void) { return PT; }
extern "C" int wd_repack_vchunk(void) { return VCHUNK; }

extern "C" int wd_repack_multi(const void* table, int nentries, int64_t total_chunks, void* stream) {
    if (!table || nentries <= 0 || total_chunks <= 0 || total_chunks > 0x7fffffffLL) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(repack_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, st,
                       reinterpret_cast<const PackRef*>(table), nentries);
    return wd_check_launch();
}
