// How many independent accumulators does v_mfma_f32_32x32x16_f16 need to run at full rate, pass-major as in the split-f16
// kernels (NACC accumulators, each updated 3 times per trip)?   hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int q = 0; q < 16; ++q)
            acc[t][q] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)(threadIdx.x * 0.001f + e);
        b[e] = (_Float16)(threadIdx.x * 0.002f - e);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
            for (int t = 0; t < NACC; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int q = 0; q < 16; ++q)
            s += acc[t][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(float *out, int wgs_per_cu)
{
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 3 * NACC * 32768.0;
    printf("%2d accumulators, %d waves / SIMD: %.1f TFLOP/s f16 (%.2f ms)\n", NACC, wgs_per_cu, flops / ms / 1e9, ms);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 4; w *= 2) {
        run<1>(out, w); run<2>(out, w); run<3>(out, w); run<4>(out, w); run<6>(out, w); run<12>(out, w);
    }
    return 0;
}
