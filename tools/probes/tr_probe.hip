// Probe of ds_read_b64_tr_b16 semantics on gfx950 (run on the GPU box): which element lands where.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(short* out, int mode) {
    __shared__ short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;   // value = row*64 + col
    __syncthreads();
    int l = threadIdx.x, g = l >> 4, t = l & 15;
    int q = t >> 2, p = t & 3;
    const short* addr;
    if (mode == 0) addr = lds + q * 64 + g * 16 + 4 * p;            // guide: lane 4q+p -> row q, cols 4p..4p+3
    else addr = lds + (4 * g + q) * 64 + 4 * p;                     // groups take different row blocks
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)addr);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * sizeof(short));
    short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
            printf("\n");
        }
    }
    return 0;
}
