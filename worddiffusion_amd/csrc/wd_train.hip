// Optimiser side of the training step (SURVEY.md row a-T): MSE loss + its gradient, multi-tensor fused AdamW with the
// EMA of the weights folded in.  Reference: nn.MSELoss()(noise, predicted_noise) train.py:289; optim.AdamW(lr=1e-4)
// train.py:405 (torch defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0.01); EMA.step_ema train.py:146-170, which
// the reference runs as ~260 x 3 separate elementwise launches per step.  Arithmetic follows torch's single-tensor
// AdamW op order in fp32 (no contraction) so that updates are comparable bit-for-bit up to the division rounding.
#include "wd_common.h"

#pragma clang fp contract(off)

namespace {

struct TensorRef {  // one entry of the device-side parameter table (built once by the host)
    float* p;
    const float* g;
    float* m;
    float* v;
    float* ema;        // may be NULL
    int64_t n;
    int64_t chunk0;    // index of this tensor's first chunk
};

constexpr int CHUNK = 8192;  // elements per workgroup

__global__ void adamw_multi_kernel(const TensorRef* __restrict__ tab, int ntensor, float decay, float omb1, float b2,
                                   float omb2, float step_size, float bc2_sqrt, float eps, int ema_mode, float ema_b,
                                   float ema_omb) {
    // binary search: which tensor owns chunk blockIdx.x
    int lo = 0, hi = ntensor - 1;
    const int64_t c = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].chunk0 <= c) lo = mid;
        else hi = mid - 1;
    }
    const TensorRef t = tab[lo];
    const int64_t base = (c - t.chunk0) * CHUNK;
    const int64_t end = base + CHUNK < t.n ? base + CHUNK : t.n;
    if (!t.g) {
        // a parameter the loss does not reach (grad None in the reference: AdamW skips it, train.py:293); EMA still runs
        if (t.ema && ema_mode)
            for (int64_t i = base + threadIdx.x; i < end; i += blockDim.x)
                t.ema[i] = ema_mode == 1 ? t.p[i] : t.ema[i] * ema_b + ema_omb * t.p[i];
        return;
    }
    for (int64_t i = base + threadIdx.x; i < end; i += blockDim.x) {
        const float g = t.g[i];
        float p = t.p[i] * decay;                 // param.mul_(1 - lr * weight_decay)
        float m = t.m[i];
        m = m + omb1 * (g - m);                   // exp_avg.lerp_(grad, 1 - beta1)
        float v = t.v[i] * b2;                    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        v = v + (omb2 * g) * g;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        p = p + (-step_size * m) / denom;         // param.addcdiv_(exp_avg, denom, value=-step_size)
        t.p[i] = p;
        t.m[i] = m;
        t.v[i] = v;
        if (t.ema) {
            if (ema_mode == 1) t.ema[i] = p;                                  // warm-up: plain copy (train.py:161-170)
            else if (ema_mode == 2) t.ema[i] = t.ema[i] * ema_b + ema_omb * p;  // old * beta + (1 - beta) * new
        }
    }
}

// deterministic two-stage sum of (a - b)^2; stage 1 also writes d loss / d pred = 2 (pred - target) / n
__global__ void mse_stage1(const float* __restrict__ pred, const float* __restrict__ target, int64_t n, float scale,
                           float* __restrict__ grad, double* __restrict__ partial) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        acc += (double)(d * d);
        if (grad) grad[i] = scale * d;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ void mse_stage2(const double* __restrict__ partial, int nb, int64_t n, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nb; ++i) s += partial[i];
        *loss = (float)(s / (double)n);
    }
}

}  // namespace

extern "C" int wd_adamw_table_entry_bytes(void) { return (int)sizeof(TensorRef); }
extern "C" int wd_adamw_chunk(void) { return CHUNK; }

extern "C" int wd_adamw_multi(const void* table, int ntensor, int64_t total_chunks, double lr, double beta1, double beta2,
                              double eps, double weight_decay, int64_t step, int ema_mode, double ema_beta, void* stream) {
    if (!table || ntensor <= 0 || total_chunks <= 0 || step <= 0 || ema_mode < 0 || ema_mode > 2) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // the scalars exactly as torch forms them (python doubles), then rounded once to fp32
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float decay = (float)(1.0 - lr * weight_decay), omb1 = (float)(1.0 - beta1), b2 = (float)beta2,
                omb2 = (float)(1.0 - beta2), step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, st,
                       reinterpret_cast<const TensorRef*>(table), ntensor, decay, omb1, b2, omb2, step_size, bc2_sqrt,
                       (float)eps, ema_mode, (float)ema_beta, (float)(1.0 - ema_beta));
    return wd_check_launch();
}

extern "C" int wd_mse_loss(const float* pred, const float* target, int64_t n, float* grad, float* loss, double* scratch,
                           int scratch_len, void* stream) {
    if (!pred || !target || !loss || !scratch || n <= 0 || scratch_len < 1) return WD_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int nb = (int)((n + 255) / 256 < scratch_len ? (n + 255) / 256 : scratch_len);
    if (nb > 1024) nb = 1024;
    WdLaunchScope scope(WD_CLS_OTHER, st);
    hipLaunchKernelGGL(mse_stage1, dim3(nb), dim3(256), 0, st, pred, target, n, (float)(2.0 / (double)n), grad, scratch);
    hipLaunchKernelGGL(mse_stage2, dim3(1), dim3(64), 0, st, scratch, nb, n, loss);
    return wd_check_launch();
}
