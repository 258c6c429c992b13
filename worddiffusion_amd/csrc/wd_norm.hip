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
constexpr int AP_TOK = 16;
__global__ void gn_apply_kernel(const float* __restrict__ x, int ld, int hw, int c, int cpg, int nchunk, int part_cpg,
                                const double* __restrict__ part, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int silu, wd_bf16* __restrict__ out_hi,
                                wd_bf16* __restrict__ out_lo, int out_ld, int c_off, wd_bf16* __restrict__ raw_hi,
                                wd_bf16* __restrict__ raw_lo) {
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
    const int total = nt * c4;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int t = i / c4, cx = (i - t * c4) * 4;
        const long row = (long)b * hw + t0 + t;
        const float4 v = *reinterpret_cast<const float4*>(x + row * ld + cx);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c_off + cx);
        const float4 be = *reinterpret_cast<const float4*>(beta + c_off + cx);
        float4 y;
        {
            const int g = cx / cpg;  // the four channels may straddle two groups
            const int g1 = (cx + 1) / cpg, g2 = (cx + 2) / cpg, g3 = (cx + 3) / cpg;
            y.x = (v.x - s_mean[g]) * s_rstd[g] * ga.x + be.x;
            y.y = (v.y - s_mean[g1]) * s_rstd[g1] * ga.y + be.y;
            y.z = (v.z - s_mean[g2]) * s_rstd[g2] * ga.z + be.z;
            y.w = (v.w - s_mean[g3]) * s_rstd[g3] * ga.w + be.w;
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
    hipLaunchKernelGGL(gn_apply_kernel, dim3((hw + AP_TOK - 1) / AP_TOK, batch), dim3(256), 0, st, x, ld, hw, c, cpg,
                       nchunk, part_cpg, part, gamma, beta, eps, silu, out_hi, out_lo, out_ld, c_off, raw_hi, raw_lo);
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
