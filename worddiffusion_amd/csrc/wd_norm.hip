// GroupNorm statistics / apply(+SiLU) -> split-bf16 planes, LayerNorm -> planes, plain split.
// HBM-bound streaming kernels: 16-byte loads, 8-byte plane stores, deterministic reductions (no atomics).
// Reference: GroupNorm32 unet.py:427-431 (eps 1e-5), Normalize unet.py:161-162 (eps 1e-6),
// nn.LayerNorm unet.py:314-316, SiLU unet.py:594,618.
#include "wd_common.h"

namespace {

constexpr int GN_TOK = 32;  // tokens per statistics chunk

// grid (nchunk, batch); block (64 * ceil(c/4/64), 2).  thread x owns channels 4x..4x+3, y splits tokens.
__global__ void gn_stats_kernel(const float* __restrict__ x, int ld, int hw, int c, int cpg, int nchunk,
                                double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_sum = reinterpret_cast<float*>(smem);  // [2][c]
    float* s_sq = s_sum + 2 * c;                    // [2][c]
    const int b = blockIdx.y, j = blockIdx.x;
    const int cx = threadIdx.x * 4;
    const int t0 = j * GN_TOK, t1 = min(hw, t0 + GN_TOK);
    if (cx < c) {
        float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
        const float* base = x + ((long)b * hw) * ld + cx;
        for (int t = t0 + threadIdx.y; t < t1; t += 2) {
            const float4 v = *reinterpret_cast<const float4*>(base + (long)t * ld);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
        }
        *reinterpret_cast<float4*>(s_sum + threadIdx.y * c + cx) = s;
        *reinterpret_cast<float4*>(s_sq + threadIdx.y * c + cx) = q;
    }
    __syncthreads();
    const int ng = c / cpg;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (tid < ng) {
        double ds = 0.0, dq = 0.0;
        for (int k = 0; k < cpg; ++k) {
            const int ch = tid * cpg + k;
            ds += (double)s_sum[ch] + (double)s_sum[c + ch];
            dq += (double)s_sq[ch] + (double)s_sq[c + ch];
        }
        double* o = part + (((long)b * nchunk + j) * ng + tid) * 2;
        o[0] = ds;
        o[1] = dq;
    }
}

// grid (ceil(hw / TOK_PER_WG), batch), block 256.  One float4 (4 channels) per thread per step.
constexpr int AP_TOK = 8;
struct GnSrc {  // one source tensor of a GroupNorm over a channel concat: its rows, its statistics, its channel offset in the concat
    const float* x;
    const double* part;
    int ld, c, nchunk, part_cpg, c_off;
    const int32_t* perm;  // NULL, or [hw]: the row (inside its sample) that holds position t - a producer that wrote its rows in another order
};

// grid (token tiles, batch, sources): blockIdx.z picks the source
__global__ void gn_apply_kernel(const GnSrc s0, const GnSrc s1, int hw, int cpg, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int silu, wd_bf16* __restrict__ out_hi,
                                wd_bf16* __restrict__ out_lo, int out_ld, wd_bf16* __restrict__ raw_hi,
                                wd_bf16* __restrict__ raw_lo) {
    const GnSrc& sr = blockIdx.z ? s1 : s0;
    const float* __restrict__ x = sr.x;
    const double* __restrict__ part = sr.part;
    const int ld = sr.ld, c = sr.c, nchunk = sr.nchunk, part_cpg = sr.part_cpg, c_off = sr.c_off;
    __shared__ float s_mean[32], s_rstd[32];
    const int b = blockIdx.y;
    const int ng = c / cpg;
    if (threadIdx.x < ng) {
        // the statistics array holds c / part_cpg groups per (sample, chunk); this norm's group = `ratio` of them
        const int ratio = cpg / part_cpg, ngs = c / part_cpg;
        double ds = 0.0, dq = 0.0;
        for (int j = 0; j < nchunk; ++j) {
            const double* p = part + (((long)b * nchunk + j) * ngs + threadIdx.x * ratio) * 2;
            for (int k = 0; k < ratio; ++k) {
                ds += p[2 * k];
                dq += p[2 * k + 1];
            }
        }
        const double n = (double)hw * cpg;
        const double mean = ds / n;
        double var = dq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[threadIdx.x] = (float)mean;
        s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int c4 = c >> 2;
    const int t0 = blockIdx.x * AP_TOK, nt = min(AP_TOK, hw - t0);
    if (c4 <= 256) {
        // thread = (token lane, channel quad): the quad is fixed, so y = v * scale + shift with four per-thread constants each
        // (the group lookups - integer divisions by a run-time cpg - happen once, not per element)
        const int rpp = 256 / c4;  // tokens per pass
        const int tl = threadIdx.x / c4, cx = (threadIdx.x - tl * c4) * 4;
        if (tl >= rpp) return;
        float sc[4], sh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (cx + j) / cpg;  // the four channels may straddle two groups
            sc[j] = s_rstd[g] * gamma[c_off + cx + j];
            sh[j] = beta[c_off + cx + j] - s_mean[g] * sc[j];
        }
        for (int t = tl; t < nt; t += rpp) {
            const long row = (long)b * hw + t0 + t;
            const long srow = sr.perm ? (long)b * hw + sr.perm[t0 + t] : row;
            const float4 v = *reinterpret_cast<const float4*>(x + srow * ld + cx);
            float4 y = make_float4(v.x * sc[0] + sh[0], v.y * sc[1] + sh[1], v.z * sc[2] + sh[2], v.w * sc[3] + sh[3]);
            if (silu) {
                y.x = wd_silu(y.x); y.y = wd_silu(y.y); y.z = wd_silu(y.z); y.w = wd_silu(y.w);
            }
            uint2 h, l;
            wd_split4(y, h, l);
            const long o = row * out_ld + c_off + cx;
            *reinterpret_cast<uint2*>(out_hi + o) = h;
            if (out_lo) *reinterpret_cast<uint2*>(out_lo + o) = l;
            if (raw_hi) {
                wd_split4(v, h, l);
                *reinterpret_cast<uint2*>(raw_hi + o) = h;
                if (raw_lo) *reinterpret_cast<uint2*>(raw_lo + o) = l;
            }
        }
        return;
    }
    const int total = nt * c4;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int t = i / c4, cx = (i - t * c4) * 4;
        const long row = (long)b * hw + t0 + t;
        const long srow = sr.perm ? (long)b * hw + sr.perm[t0 + t] : row;
        const float4 v = *reinterpret_cast<const float4*>(x + srow * ld + cx);
        float4 y;
        {
            const int g = cx / cpg, g1 = (cx + 1) / cpg, g2 = (cx + 2) / cpg, g3 = (cx + 3) / cpg;
            const float4 ga = *reinterpret_cast<const float4*>(gamma + c_off + cx);
            const float4 be = *reinterpret_cast<const float4*>(beta + c_off + cx);
            const float s0 = s_rstd[g] * ga.x, s1 = s_rstd[g1] * ga.y, s2 = s_rstd[g2] * ga.z, s3 = s_rstd[g3] * ga.w;
            y.x = v.x * s0 + (be.x - s_mean[g] * s0);
            y.y = v.y * s1 + (be.y - s_mean[g1] * s1);
            y.z = v.z * s2 + (be.z - s_mean[g2] * s2);
            y.w = v.w * s3 + (be.w - s_mean[g3] * s3);
        }
        if (silu) {
            y.x = wd_silu(y.x); y.y = wd_silu(y.y); y.z = wd_silu(y.z); y.w = wd_silu(y.w);
        }
        uint2 h, l;
        wd_split4(y, h, l);
        const long o = row * out_ld + c_off + cx;
        *reinterpret_cast<uint2*>(out_hi + o) = h;
        if (out_lo) *reinterpret_cast<uint2*>(out_lo + o) = l;
        if (raw_hi) {
            wd_split4(v, h, l);
            *reinterpret_cast<uint2*>(raw_hi + o) = h;
            if (raw_lo) *reinterpret_cast<uint2*>(raw_lo + o) = l;
        }
    }
}

// one wave per row; c <= 64 * 4 * LN_MAX4
constexpr int LN_MAX4 = 8;
__global__ void layernorm_kernel(const float* __restrict__ x, int ld, int rows, int c, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, wd_bf16* __restrict__ out_hi,
                                 wd_bf16* __restrict__ out_lo, int out_ld) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int c4 = c >> 2;
    float4 v[LN_MAX4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX4; ++i) {
        const int f = lane + 64 * i;
        if (f < c4) {
            v[i] = *reinterpret_cast<const float4*>(x + (long)row * ld + f * 4);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wd_wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX4; ++i) {
        const int f = lane + 64 * i;
        if (f < c4) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wd_wave_sum(q) / (float)c + eps);
#pragma unroll
    for (int i = 0; i < LN_MAX4; ++i) {
        const int f = lane + 64 * i;
        if (f < c4) {
            const float4 ga = *reinterpret_cast<const float4*>(gamma + f * 4);
            const float4 be = *reinterpret_cast<const float4*>(beta + f * 4);
            float4 y;
            y.x = (v[i].x - mean) * rstd * ga.x + be.x;
            y.y = (v[i].y - mean) * rstd * ga.y + be.y;
            y.z = (v[i].z - mean) * rstd * ga.z + be.z;
            y.w = (v[i].w - mean) * rstd * ga.w + be.w;
            uint2 h, l;
            wd_split4(y, h, l);
            const long o = (long)row * out_ld + f * 4;
            *reinterpret_cast<uint2*>(out_hi + o) = h;
            if (out_lo) *reinterpret_cast<uint2*>(out_lo + o) = l;
        }
    }
}

__global__ void split_kernel(const float* __restrict__ x, int ld, int rows, int c, int silu,
                             wd_bf16* __restrict__ out_hi, wd_bf16* __restrict__ out_lo, int out_ld) {
    const int c4 = c >> 2;
    const long total = (long)rows * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4;
        const int cx = (int)(i - r * c4) * 4;
        float4 v = *reinterpret_cast<const float4*>(x + r * ld + cx);
        if (silu) {
            v.x = wd_silu(v.x); v.y = wd_silu(v.y); v.z = wd_silu(v.z); v.w = wd_silu(v.w);
        }
        uint2 h, l;
        wd_split4(v, h, l);
        *reinterpret_cast<uint2*>(out_hi + r * out_ld + cx) = h;
        if (out_lo) *reinterpret_cast<uint2*>(out_lo + r * out_ld + cx) = l;
    }
}

// ---- out[b][o][y][x] = Conv3x3(SiLU(GroupNorm(x)))[o] + bias[o] for a convolution with FEW output channels (<= 4): the UNet's
// last layer (unet.py:1453-1458: GroupNorm32, SiLU, conv 320 -> 4).  As a GEMM it fills 4 of a tile's 64 columns; here one
// workgroup owns one image row of one sample and walks the channels in chunks of 64: the chunk's 3 x (W + 2) pixel tile is
// normalised + SiLU'd into LDS (zero halo) next to the chunk's weights as [channel][tap][4], and multiplied in fp32 on the VALU
// (lanes = pixels, so the weight reads are broadcasts).  The rows of the NEXT chunk are requested before the current one is
// multiplied, so the HBM latency of a chunk hides under the previous chunk's arithmetic.
// grid (H, batch), block 256 = PXL pixel lanes x (256 / PXL) channel groups; W <= 64, c % 64 == 0, weights = the parameter
// [oc][c][3][3].
constexpr int GC_CH = 64;            // channels per chunk
constexpr int GC_PITCH = GC_CH + 4;  // floats per pixel in the tile (shifts the banks from pixel to pixel)
template <int PXL>
__global__ void __launch_bounds__(256) gn_conv_few_kernel(const float* __restrict__ x, int ld, int H, int W, int c, int cpg, int nchunk,
                                                         int part_cpg, const double* __restrict__ part,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         int silu, const float* __restrict__ w, const float* __restrict__ bias, int oc,
                                                         float* __restrict__ out) {
    constexpr int CG = 256 / PXL, CPT = GC_CH / CG;  // channel groups, channels per thread and chunk
    constexpr int NLD = 3 * PXL * (GC_CH / 4) / 256;  // float4 of a chunk's rows per thread (6 at 32 pixel lanes, 12 at 64)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_w = reinterpret_cast<float*>(smem);   // [GC_CH][9][4]: this chunk's weights
    float* s_t = s_w + 9 * GC_CH * 4;               // [3][W + 2][GC_PITCH]
    __shared__ float s_mean[32], s_rstd[32];
    // grid (batch, H): consecutive workgroup ids = consecutive samples, so the rows of one sample - each of which is read by the
    // workgroups of the rows above and below too - are dealt to the same XCD and re-read from its L2 (with (H, batch) the eight rows
    // of a sample went to eight XCDs: 58.8 MB fetched for a 21 MB input)
    const int b = blockIdx.x, y = blockIdx.y, tid = threadIdx.x;
    const int hw = H * W, ng = c / cpg;
    // this thread's share of a chunk's three rows: item i = tid + 256 k -> (row r, pixel px, channel quad c4)
    float4 pv[NLD];
    auto request = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            const int r = i / (PXL * (GC_CH / 4)), rem = i - r * (PXL * (GC_CH / 4));
            const int px = rem / (GC_CH / 4), c4 = rem - px * (GC_CH / 4);
            const int yy = y + r - 1;
            pv[k] = (yy >= 0 && yy < H && px < W)
                        ? *reinterpret_cast<const float4*>(x + ((long)b * hw + (long)yy * W + px) * ld + c0 + c4 * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    request(0);
    if (tid < ng) {
        const int ratio = cpg / part_cpg, ngs = c / part_cpg;
        double ds = 0.0, dq = 0.0;
        for (int j = 0; j < nchunk; ++j) {
            const double* p = part + (((long)b * nchunk + j) * ngs + tid * ratio) * 2;
            for (int k = 0; k < ratio; ++k) {
                ds += p[2 * k];
                dq += p[2 * k + 1];
            }
        }
        const double n = (double)hw * cpg;
        const double mean = ds / n;
        double var = dq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    // the tile's halo columns and the weight slots of o >= oc stay zero for the whole kernel
    for (int i = tid; i < 3 * 2 * (GC_CH / 4); i += 256) {
        const int r = i / (2 * (GC_CH / 4)), rem = i - r * (2 * (GC_CH / 4));
        const int side = rem / (GC_CH / 4), c4 = rem - side * (GC_CH / 4);
        *reinterpret_cast<float4*>(s_t + ((long)r * (W + 2) + (side ? W + 1 : 0)) * GC_PITCH + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = tid; i < 9 * GC_CH; i += 256) *reinterpret_cast<float4*>(s_w + (long)i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int pl = tid % PXL, cg = tid / PXL;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = 0; c0 < c; c0 += GC_CH) {
        // ---- this chunk: weights [o][c0 .. c0+63][9] (contiguous per o) -> [channel][tap][4]; rows normalised (+ SiLU) -> tile
        for (int i = tid; i < oc * GC_CH * 9; i += 256) {
            const int o = i / (GC_CH * 9), rem = i - o * (GC_CH * 9);  // rem = channel * 9 + tap: the parameter's own order
            s_w[rem * 4 + o] = w[((long)o * c + c0) * 9 + rem];
        }
        // (the channel quad of a thread's items is the same for all of them: 256 is a multiple of the 16 quads of a chunk)
        float sc[4], sh[4];
        {
            const int ch = c0 + (tid & (GC_CH / 4 - 1)) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = (ch + j) / cpg;
                sc[j] = s_rstd[g] * gamma[ch + j];
                sh[j] = beta[ch + j] - s_mean[g] * sc[j];
            }
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            const int r = i / (PXL * (GC_CH / 4)), rem = i - r * (PXL * (GC_CH / 4));
            const int px = rem / (GC_CH / 4), c4 = rem - px * (GC_CH / 4);
            const int yy = y + r - 1;
            if (px >= W) continue;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < H) {
                const float4 v = pv[k];
                o = make_float4(v.x * sc[0] + sh[0], v.y * sc[1] + sh[1], v.z * sc[2] + sh[2], v.w * sc[3] + sh[3]);
                if (silu) {
                    o.x = wd_silu(o.x); o.y = wd_silu(o.y); o.z = wd_silu(o.z); o.w = wd_silu(o.w);
                }
            }
            *reinterpret_cast<float4*>(s_t + ((long)r * (W + 2) + px + 1) * GC_PITCH + c4 * 4) = o;
        }
        if (c0 + GC_CH < c) request(c0 + GC_CH);  // in flight while this chunk is multiplied
        __syncthreads();
        if (pl < W) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* at = s_t + ((long)dy * (W + 2) + pl + dx) * GC_PITCH + cg * CPT;
                    const float* wt = s_w + ((long)(cg * CPT) * 9 + dy * 3 + dx) * 4;
#pragma unroll
                    for (int k = 0; k < CPT / 4; ++k) {
                        const float4 a = *reinterpret_cast<const float4*>(at + 4 * k);
                        const float4 w0 = *reinterpret_cast<const float4*>(wt + (4 * k + 0) * 36);
                        const float4 w1 = *reinterpret_cast<const float4*>(wt + (4 * k + 1) * 36);
                        const float4 w2 = *reinterpret_cast<const float4*>(wt + (4 * k + 2) * 36);
                        const float4 w3 = *reinterpret_cast<const float4*>(wt + (4 * k + 3) * 36);
                        acc.x += a.x * w0.x; acc.y += a.x * w0.y; acc.z += a.x * w0.z; acc.w += a.x * w0.w;
                        acc.x += a.y * w1.x; acc.y += a.y * w1.y; acc.z += a.y * w1.z; acc.w += a.y * w1.w;
                        acc.x += a.z * w2.x; acc.y += a.z * w2.y; acc.z += a.z * w2.z; acc.w += a.z * w2.w;
                        acc.x += a.w * w3.x; acc.y += a.w * w3.y; acc.z += a.w * w3.z; acc.w += a.w * w3.w;
                    }
                }
        }
        __syncthreads();
    }
    // ---- sum the channel groups in a fixed order, add the bias, write NCHW
    float* s_red = s_t;  // [CG][PXL][4]
    *reinterpret_cast<float4*>(s_red + ((long)cg * PXL + pl) * 4) = acc;
    __syncthreads();
    for (int i = tid; i < W * oc; i += 256) {
        const int o = i / W, px = i - o * W;
        float v = bias ? bias[o] : 0.f;
        for (int g = 0; g < CG; ++g) v += s_red[((long)g * PXL + px) * 4 + o];
        out[(((long)b * oc + o) * H + y) * W + px] = v;
    }
}

// part[b][nchunk][ng][2] -> out[b][1][ng][2]: the chunk sums of a sample folded once (fixed order) instead of by every workgroup
// of wd_gn_apply - at 16384 positions per sample (the VAE decoder's last level) that loop is 128 chunks long.
// grid (batch), block 256 = 4 chunk lanes x 64 (group, sum / sum of squares) columns; ng * 2 <= 64.
__global__ void __launch_bounds__(256) gn_fold_chunks_kernel(const double* __restrict__ part, int nchunk, int ng2,
                                                            double* __restrict__ out) {
    __shared__ double red[4][64];
    const int col = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const long b = blockIdx.x;
    double acc = 0.0;
    if (col < ng2)
        for (int j = lane; j < nchunk; j += 4) acc += part[(b * nchunk + j) * ng2 + col];
    red[lane][col] = acc;
    __syncthreads();
    if (lane == 0 && col < ng2) out[b * ng2 + col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

}  // namespace

extern "C" int wd_gn_nchunk(int hw) { return (hw + GN_TOK - 1) / GN_TOK; }

extern "C" int wd_gn_fold_chunks(const double* part, int batch, int nchunk, int ngroups, double* out, void* stream) {
    if (!part || !out || batch <= 0 || nchunk <= 0 || ngroups <= 0 || ngroups > 32) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_GNSTATS, st);
    hipLaunchKernelGGL(gn_fold_chunks_kernel, dim3(batch), dim3(256), 0, st, part, nchunk, 2 * ngroups, out);
    return wd_check_launch();
}

extern "C" int wd_gn_stats(const float* x, int ld, int batch, int hw, int c, int cpg, double* part, void* stream) {
    if (!x || !part || batch <= 0 || hw <= 0 || c <= 0 || cpg <= 0) return WD_EINVAL;
    if (c % 4 || ld % 4 || c % cpg || c / cpg > 256 || c > 4096) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nchunk = wd_gn_nchunk(hw);
    const int bx = 64 * ((c / 4 + 63) / 64);
    if (bx * 2 > 1024) return WD_EINVAL;
    WdLaunchScope scope(WD_CLS_GNSTATS, st);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, batch), dim3(bx, 2), 4 * c * sizeof(float), st, x, ld, hw, c, cpg,
                       nchunk, part);
    return wd_check_launch();
}

extern "C" int wd_gn_apply(const float* x, int ld, int batch, int hw, int c, int cpg, const double* part, int nchunk,
                           int part_cpg, const float* gamma, const float* beta, float eps, int silu, wd_bf16* out_hi,
                           wd_bf16* out_lo, int out_ld, int c_off, wd_bf16* raw_hi, wd_bf16* raw_lo, void* stream) {
    if (!x || !part || !gamma || !beta || !out_hi || batch <= 0 || hw <= 0 || nchunk <= 0 || part_cpg <= 0) return WD_EINVAL;
    if (c % 4 || ld % 4 || out_ld % 4 || c_off % 4 || c % cpg || c / cpg > 32 || cpg % part_cpg) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_GNAPPLY, st);
    const GnSrc s0 = {x, part, ld, c, nchunk, part_cpg, c_off, nullptr};
    hipLaunchKernelGGL(gn_apply_kernel, dim3((hw + AP_TOK - 1) / AP_TOK, batch, 1), dim3(256), 0, st, s0, s0, hw, cpg, gamma, beta, eps,
                       silu, out_hi, out_lo, out_ld, raw_hi, raw_lo);
    return wd_check_launch();
}

extern "C" int wd_gn_apply2(const float* xa, int lda, int ca, const double* part_a, int nchunk_a, int part_cpg_a, int c_off_a,
                            const float* xb, int ldb, int cb, const double* part_b, int nchunk_b, int part_cpg_b, int c_off_b,
                            int batch, int hw, int cpg, const float* gamma, const float* beta, float eps, int silu,
                            wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, wd_bf16* raw_hi, wd_bf16* raw_lo, const int32_t* perm_a,
                            void* stream) {
    if (!xa || !xb || !part_a || !part_b || !gamma || !beta || !out_hi || batch <= 0 || hw <= 0 || nchunk_a <= 0 || nchunk_b <= 0 ||
        part_cpg_a <= 0 || part_cpg_b <= 0)
        return WD_EINVAL;
    if (ca % 4 || cb % 4 || lda % 4 || ldb % 4 || out_ld % 4 || c_off_a % 4 || c_off_b % 4 || ca % cpg || cb % cpg ||
        ca / cpg > 32 || cb / cpg > 32 || cpg % part_cpg_a || cpg % part_cpg_b)
        return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_GNAPPLY, st);
    const GnSrc s0 = {xa, part_a, lda, ca, nchunk_a, part_cpg_a, c_off_a, perm_a}, s1 = {xb, part_b, ldb, cb, nchunk_b, part_cpg_b, c_off_b, nullptr};
    hipLaunchKernelGGL(gn_apply_kernel, dim3((hw + AP_TOK - 1) / AP_TOK, batch, 2), dim3(256), 0, st, s0, s1, hw, cpg, gamma, beta, eps,
                       silu, out_hi, out_lo, out_ld, raw_hi, raw_lo);
    return wd_check_launch();
}

// the tile region doubles as the [row][channel group][pixel lane][4] reduction image at the end
static int gn_conv_tile_floats(int w) { return 3 * (w + 2) * GC_PITCH > 1024 ? 3 * (w + 2) * GC_PITCH : 1024; }

extern "C" int wd_gn_conv3x3_few_supported(int c, int w, int oc) {
    return c > 0 && c % GC_CH == 0 && w > 0 && w <= 64 && oc >= 1 && oc <= 4 &&
           (size_t)(9 * GC_CH * 4 + gn_conv_tile_floats(w)) * sizeof(float) <= 64 * 1024;
}

extern "C" int wd_gn_conv3x3_few(const float* x, int ld, int batch, int h, int w, int c, int cpg, const double* part, int nchunk,
                                 int part_cpg, const float* gamma, const float* beta, float eps, int silu, const float* weight,
                                 const float* bias, int oc, float* out, void* stream) {
    if (!x || !part || !gamma || !beta || !weight || !out || batch <= 0 || h <= 0 || nchunk <= 0 || part_cpg <= 0) return WD_EINVAL;
    if (!wd_gn_conv3x3_few_supported(c, w, oc) || ld % 4 || cpg <= 0 || c % cpg || c / cpg > 32 || cpg % part_cpg) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t smem = (size_t)(9 * GC_CH * 4 + gn_conv_tile_floats(w)) * sizeof(float);  // <= 64 KB: no attribute needed
    WdLaunchScope scope(WD_CLS_OTHER, st);
    if (w <= 32)
        hipLaunchKernelGGL(gn_conv_few_kernel<32>, dim3(batch, h), dim3(256), smem, st, x, ld, h, w, c, cpg, nchunk, part_cpg, part,
                           gamma, beta, eps, silu, weight, bias, oc, out);
    else
        hipLaunchKernelGGL(gn_conv_few_kernel<64>, dim3(batch, h), dim3(256), smem, st, x, ld, h, w, c, cpg, nchunk, part_cpg, part,
                           gamma, beta, eps, silu, weight, bias, oc, out);
    return wd_check_launch();
}

extern "C" int wd_layernorm(const float* x, int ld, int rows, int c, const float* gamma, const float* beta, float eps,
                            wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, void* stream) {
    if (!x || !gamma || !beta || !out_hi || rows <= 0 || c <= 0) return WD_EINVAL;
    if (c % 4 || ld % 4 || out_ld % 4 || c > 64 * 4 * LN_MAX4) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_LN, st);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ld, rows, c, gamma, beta, eps,
                       out_hi, out_lo, out_ld);
    return wd_check_launch();
}

extern "C" int wd_split(const float* x, int ld, int rows, int c, int silu, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld,
                        void* stream) {
    if (!x || !out_hi || rows <= 0 || c <= 0 || c % 4 || ld % 4 || out_ld % 4) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long total = (long)rows * (c / 4);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(256), 0, st, x, ld, rows, c, silu, out_hi, out_lo, out_ld);
    return wd_check_launch();
}
