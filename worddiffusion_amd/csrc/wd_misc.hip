// Small kernels at the edges of the UNet forward and the DDPM loop around it.
// Reference: timestep_embedding unet.py:96-116; CharacterEncoder embedding + PE unet.py:860-872;
// first conv unet.py:1251; Diffusion.sampling update train.py:229-236; noise_images train.py:190-194;
// EMA train.py:151-159.
#include "wd_common.h"

// The DDPM update, noise_images and the EMA are compared bit-for-bit with the reference's unfused fp32 torch ops:
// no mul+add contraction anywhere in this translation unit.  (Plain operators are used on purpose: the header
// intrinsics __fmul_rn/__fadd_rn are inline functions parsed with contraction allowed and fuse after inlining.)
#pragma clang fp contract(off)

namespace {

__global__ void temb_kernel(const int64_t* __restrict__ t, int batch, const float* __restrict__ freqs, int half,
                            wd_bf16* __restrict__ out_hi, wd_bf16* __restrict__ out_lo, int out_ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * half) return;
    const int b = i / half, kx = i - b * half;
    const float arg = (float)t[b] * freqs[kx];
    const float c = cosf(arg), s = sinf(arg);
    uint32_t h, l;
    wd_split1(c, h, l);
    out_hi[(long)b * out_ld + kx] = (wd_bf16)h;
    if (out_lo) out_lo[(long)b * out_ld + kx] = (wd_bf16)l;
    wd_split1(s, h, l);
    out_hi[(long)b * out_ld + half + kx] = (wd_bf16)h;
    if (out_lo) out_lo[(long)b * out_ld + half + kx] = (wd_bf16)l;
}

__global__ void embed_kernel(const void* __restrict__ ids, int i64, int rows, int seq_len,
                             const float* __restrict__ table, int vocab, int c, const float* __restrict__ pe,
                             wd_bf16* __restrict__ out_hi, wd_bf16* __restrict__ out_lo, int out_ld) {
    const int c4 = c >> 2;
    const long total = (long)rows * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4), cx = (int)(i - (long)r * c4) * 4;
        long id = i64 ? (long)reinterpret_cast<const int64_t*>(ids)[r] : (long)reinterpret_cast<const int32_t*>(ids)[r];
        if (id < 0) id = 0;
        if (id >= vocab) id = vocab - 1;  // the reference would raise; never hit with valid ids
        float4 v = *reinterpret_cast<const float4*>(table + id * c + cx);
        if (pe) {
            const float4 p = *reinterpret_cast<const float4*>(pe + (long)(r % seq_len) * c + cx);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        uint2 h, l;
        wd_split4(v, h, l);
        *reinterpret_cast<uint2*>(out_hi + (long)r * out_ld + cx) = h;
        if (out_lo) *reinterpret_cast<uint2*>(out_lo + (long)r * out_ld + cx) = l;
    }
}

__global__ void im2col_kernel(const float* __restrict__ x, int batch, int cin, int h, int w,
                              wd_bf16* __restrict__ out_hi, wd_bf16* __restrict__ out_lo, int kpad) {
    const long total = (long)batch * h * w * kpad;
    const int hw = h * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / kpad;
        const int col = (int)(i - row * kpad);
        float v = 0.0f;
        if (col < 9 * cin) {
            const int tap = col / cin, ci = col - tap * cin;
            const int b = (int)(row / hw), p = (int)(row - (long)b * hw);
            const int yy = p / w + tap / 3 - 1, xx = p % w + tap % 3 - 1;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = x[(((long)b * cin + ci) * h + yy) * w + xx];
        }
        uint32_t hb, lb;
        wd_split1(v, hb, lb);
        out_hi[i] = (wd_bf16)hb;
        if (out_lo) out_lo[i] = (wd_bf16)lb;
    }
}

__global__ void nchw_to_tok_kernel(const float* __restrict__ x, int batch, int c, int hw, float* __restrict__ out,
                                   int ld) {
    const long total = (long)batch * c * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c;  // iterate output-major: row = b*hw + p, channel fastest
        const int ch = (int)(i - row * c);
        const int b = (int)(row / hw), p = (int)(row - (long)b * hw);
        out[row * ld + ch] = x[((long)b * c + ch) * hw + p];
    }
}

__global__ void tok_to_nchw_kernel(const float* __restrict__ x, int ld, int batch, int c, int hw,
                                   float* __restrict__ out) {
    const long total = (long)batch * c * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % hw);
        const long bc = i / hw;
        const int ch = (int)(bc % c), b = (int)(bc / c);
        out[i] = x[((long)b * hw + p) * ld + ch];
    }
}

// ---- Philox4x32-10 (Salmon et al. 2011) --------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float4 philox_normal4(uint64_t seed, uint64_t sample, uint32_t tag, uint32_t e4) {
    uint32_t c[4] = {e4, tag, (uint32_t)sample, (uint32_t)(sample >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    // Box-Muller on (0,1] uniforms
    const float u0 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(c[1] >> 8)) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(c[3] >> 8)) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0);
    sincosf(6.283185307179586f * u3, &s1, &c1);
    return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

// x = ca[t] * (x - cb[t] * eps) + cs[t] * z  with the reference's rounding order (no contraction)
__global__ void ddpm_step_kernel(float* __restrict__ x, const float* __restrict__ eps, int batch, int n4,
                                 const float* __restrict__ ca, const float* __restrict__ cb,
                                 const float* __restrict__ cs, const int32_t* __restrict__ t_dev,
                                 const float* __restrict__ noise, uint64_t seed, uint64_t sample_offset) {
    const int t = *t_dev;
    const float a = ca[t], bb = cb[t], s = cs[t];
    const long total = (long)batch * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n4);
        const uint32_t e4 = (uint32_t)(i - (long)b * n4);
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t > 1) {
            if (noise) z = reinterpret_cast<const float4*>(noise)[i];
            else z = philox_normal4(seed, sample_offset + (uint64_t)b, (uint32_t)t, e4);
        }
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float4 ev = reinterpret_cast<const float4*>(eps)[i];
        float4 o;
        o.x = a * (xv.x - bb * ev.x) + s * z.x;
        o.y = a * (xv.y - bb * ev.y) + s * z.y;
        o.z = a * (xv.z - bb * ev.z) + s * z.z;
        o.w = a * (xv.w - bb * ev.w) + s * z.w;
        reinterpret_cast<float4*>(x)[i] = o;
    }
}

__global__ void advance_kernel(int32_t* t_dev, int delta, int64_t* t64, int batch) {
    __shared__ int tn;
    if (threadIdx.x == 0) {
        tn = *t_dev + delta;
        if (tn < 0) tn = 0;  // index 0 of the schedule is never used (train.py:221); never step below it
    }
    __syncthreads();
    const int v = tn;
    for (int b = threadIdx.x; b < batch; b += blockDim.x) t64[b] = (int64_t)v;
    __syncthreads();
    if (threadIdx.x == 0) *t_dev = v;
}

__global__ void randn_kernel(float* __restrict__ out, int batch, int n4, uint64_t seed, uint64_t sample_offset,
                             uint32_t tag) {
    const long total = (long)batch * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n4);
        reinterpret_cast<float4*>(out)[i] =
            philox_normal4(seed, sample_offset + (uint64_t)b, tag, (uint32_t)(i - (long)b * n4));
    }
}

__global__ void noise_images_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                    const int64_t* __restrict__ t, const float* __restrict__ sa_tab,
                                    const float* __restrict__ sb_tab, int batch, int n, float* __restrict__ out) {
    const long total = (long)batch * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n);
        const float sa = sa_tab[t[b]], sb = sb_tab[t[b]];
        out[i] = sa * x[i] + sb * eps[i];
    }
}

// python evaluates (1 - self.beta) in double; torch then multiplies the fp32 tensor by the fp32-rounded scalar
__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, int64_t n, float beta, float omb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        ema[i] = ema[i] * beta + omb * p[i];
}

inline int grid_for(long total, int block = 256, int cap = 2048) {
    long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

// emb rows for every (timestep, sample): SiLU(time[t] + label[y_b]) as operand planes (unet.py:1550-1581 followed by the
// SiLU that opens every emb_layers, unet.py:609) - lets the sampler tabulate the FiLM vectors of all steps in one GEMM
__global__ void emb_combine_kernel(const float* __restrict__ time, const float* __restrict__ label, const int64_t* __restrict__ y,
                                   int num_classes, int T, int B, int ted, wd_bf16* __restrict__ out_hi,
                                   wd_bf16* __restrict__ out_lo, int out_ld) {
    const int t4 = ted >> 2;
    const long total = (long)T * B * t4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / t4;
        const int cx = (int)(i - row * t4) * 4;
        const int t = (int)(row / B), b = (int)(row - (long)t * B);
        float4 v = *reinterpret_cast<const float4*>(time + (long)t * ted + cx);
        if (label) {
            long yb = y[b];  // range-checked on the host (engine.check_ids); clamped here so that a stale id can never read
            yb = yb < 0 ? 0 : (yb >= num_classes ? num_classes - 1 : yb);  // past the table
            const float4 l = *reinterpret_cast<const float4*>(label + yb * ted + cx);
            v.x += l.x; v.y += l.y; v.z += l.z; v.w += l.w;
        }
        v.x = wd_silu(v.x); v.y = wd_silu(v.y); v.z = wd_silu(v.z); v.w = wd_silu(v.w);
        uint2 hi, lo;
        wd_split4(v, hi, lo);
        *reinterpret_cast<uint2*>(out_hi + row * out_ld + cx) = hi;
        if (out_lo) *reinterpret_cast<uint2*>(out_lo + row * out_ld + cx) = lo;
    }
}
// out[b][:] = table[((*t_dev) % chunk) * B + b][:]  (the rows of the current timestep inside the resident chunk of `chunk`
// timesteps; t_dev lives on the device so the launch replays)
__global__ void select_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ t_dev, int B, long row4,
                                   int chunk, float* __restrict__ out) {
    const long base = (long)((*t_dev) % chunk) * B * row4;
    const long total = (long)B * row4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(table)[base + i];
}

}  // namespace

extern "C" int wd_timestep_embedding(const int64_t* t, int batch, const float* freqs, int half, wd_bf16* out_hi,
                                     wd_bf16* out_lo, int out_ld, void* stream) {
    if (!t || !freqs || !out_hi || batch <= 0 || half <= 0 || out_ld < 2 * half) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(temb_kernel, dim3((batch * half + 255) / 256), dim3(256), 0, st, t, batch, freqs, half, out_hi,
                       out_lo, out_ld);
    return wd_check_launch();
}

extern "C" int wd_embed_tokens(const void* ids, int ids_are_i64, int rows, int seq_len, const float* table, int vocab,
                               int c, const float* pe, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, void* stream) {
    if (!ids || !table || !out_hi || rows <= 0 || seq_len <= 0 || vocab <= 0 || c <= 0 || c % 4 || out_ld % 4)
        return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(embed_kernel, dim3(grid_for((long)rows * (c / 4))), dim3(256), 0, st, ids, ids_are_i64, rows,
                       seq_len, table, vocab, c, pe, out_hi, out_lo, out_ld);
    return wd_check_launch();
}

extern "C" int wd_im2col3x3(const float* x, int batch, int cin, int h, int w, wd_bf16* out_hi, wd_bf16* out_lo,
                            int kpad, void* stream) {
    if (!x || !out_hi || batch <= 0 || cin <= 0 || h <= 0 || w <= 0 || kpad < 9 * cin || kpad % 32) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((long)batch * h * w * kpad)), dim3(256), 0, st, x, batch, cin, h, w,
                       out_hi, out_lo, kpad);
    return wd_check_launch();
}

extern "C" int wd_nchw_to_tokens(const float* x, int batch, int c, int hw, float* out, int ld, void* stream) {
    if (!x || !out || batch <= 0 || c <= 0 || hw <= 0 || ld < c) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(nchw_to_tok_kernel, dim3(grid_for((long)batch * c * hw)), dim3(256), 0, st, x, batch, c, hw, out,
                       ld);
    return wd_check_launch();
}

extern "C" int wd_tokens_to_nchw(const float* x, int ld, int batch, int c, int hw, float* out, void* stream) {
    if (!x || !out || batch <= 0 || c <= 0 || hw <= 0 || ld < c) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(tok_to_nchw_kernel, dim3(grid_for((long)batch * c * hw)), dim3(256), 0, st, x, ld, batch, c, hw,
                       out);
    return wd_check_launch();
}

extern "C" int wd_ddpm_step(float* x, const float* eps, int batch, int n_per_sample, const float* ca, const float* cb,
                            const float* cs, const int32_t* t_dev, const float* noise, uint64_t seed,
                            uint64_t sample_offset, void* stream) {
    if (!x || !eps || !ca || !cb || !cs || !t_dev || batch <= 0 || n_per_sample <= 0 || n_per_sample % 4)
        return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for((long)batch * (n_per_sample / 4))), dim3(256), 0, st, x, eps,
                       batch, n_per_sample / 4, ca, cb, cs, t_dev, noise, seed, sample_offset);
    return wd_check_launch();
}

extern "C" int wd_advance_timestep(int32_t* t_dev, int delta, int64_t* t64, int batch, void* stream) {
    if (!t_dev || !t64 || batch <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(256), 0, st, t_dev, delta, t64, batch);
    return wd_check_launch();
}

extern "C" int wd_randn(float* out, int batch, int n_per_sample, uint64_t seed, uint64_t sample_offset,
                        uint32_t stream_id, void* stream) {
    if (!out || batch <= 0 || n_per_sample <= 0 || n_per_sample % 4) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for((long)batch * (n_per_sample / 4))), dim3(256), 0, st, out, batch,
                       n_per_sample / 4, seed, sample_offset, 0x80000000u | stream_id);
    return wd_check_launch();
}

extern "C" int wd_noise_images(const float* x, const float* eps, const int64_t* t, const float* sqrt_ah,
                               const float* sqrt_1m_ah, int batch, int n_per_sample, float* out, void* stream) {
    if (!x || !eps || !t || !sqrt_ah || !sqrt_1m_ah || !out || batch <= 0 || n_per_sample <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(noise_images_kernel, dim3(grid_for((long)batch * n_per_sample)), dim3(256), 0, st, x, eps, t,
                       sqrt_ah, sqrt_1m_ah, batch, n_per_sample, out);
    return wd_check_launch();
}

extern "C" int wd_copy2d(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width_bytes,
                         int64_t rows, void* stream) {
    if (!dst || !src || width_bytes <= 0 || rows <= 0 || dst_pitch < width_bytes || src_pitch < width_bytes)
        return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    return hipMemcpy2DAsync(dst, (size_t)dst_pitch, src, (size_t)src_pitch, (size_t)width_bytes, (size_t)rows,
                            hipMemcpyDeviceToDevice, st) == hipSuccess
               ? WD_OK
               : WD_ELAUNCH;
}

extern "C" int wd_ema_update(float* ema, const float* p, int64_t n, double beta, void* stream) {
    if (!ema || !p || n <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, st, ema, p, n, (float)beta,
                       (float)(1.0 - beta));
    return wd_check_launch();
}

extern "C" int wd_emb_combine(const float* time, const float* label, const int64_t* y, int num_classes, int T, int B, int ted,
                              wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, void* stream) {
    if (!time || !out_hi || T <= 0 || B <= 0 || ted <= 0 || ted % 4 || out_ld % 4 || (label && (!y || num_classes <= 0)))
        return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(emb_combine_kernel, dim3(grid_for((long)T * B * (ted / 4))), dim3(256), 0, st, time, label, y, num_classes, T,
                       B, ted, out_hi, out_lo, out_ld);
    return wd_check_launch();
}

extern "C" int wd_select_rows(const float* table, const int32_t* t_dev, int batch, int64_t row_floats, int chunk, float* out,
                              void* stream) {
    if (!table || !t_dev || !out || batch <= 0 || row_floats <= 0 || row_floats % 4 || chunk <= 0) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(select_rows_kernel, dim3(grid_for((long)batch * (row_floats / 4))), dim3(256), 0, st, table, t_dev, batch,
                       (long)(row_floats / 4), chunk, out);
    return wd_check_launch();
}
