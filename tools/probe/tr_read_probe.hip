// Prints what ds_read_b64_tr_b16 hands every lane of a wave for a [row][64 x 16-bit] LDS image (the operand layout csrc/wd_dw.hip relies on):
//   hipcc -O3 --offload-arch=gfx950 tools/probe/tr_read_probe.hip -o tools/probe/tr_read_probe && ./tools/probe/tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
__global__ void k(short* out) {
    __shared__ short sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = i;   // element value = row*64 + col  (rows of 64 shorts)
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + (4 * g + q) * 64 + 4 * p));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d; (void)hipMalloc(&d, 64 * 4 * 2);
    k<<<1, 64>>>(d);
    short h[256]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l*4+e] / 64, h[l*4+e] % 64); printf("\n"); }
    return 0;
}
