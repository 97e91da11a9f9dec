// Does a second pass over a channel's planes hit the XCD's L2?  Teams of 32 persistent workgroups (one per CU of an XCD:
// blockIdx % 8 = XCD) walk over "channels" of `per` floats in two tensors a, b:  pass 1 reads a and b (sums), [team barrier],
// pass 2 re-reads a and b and writes c.   mode 0: pass 1 only (151 MB-type traffic);  1: pass 1 + pass 2 without barrier;
// 2: with a team barrier on an agent-scope counter;  3: pass 2 only (read a, b, write c).
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(256) void k_l2(const float *a, const float *b, float *c, float *sums, int *ctr,
                                                       int channels, long per, int mode)
{
    const int xcd = blockIdx.x & 7, member = blockIdx.x >> 3;        // 32 members per XCD team (grid = 256)
    const int T = gridDim.x >> 3;
    for (int ch = xcd; ch < channels; ch += 8) {
        const float4 *pa = (const float4 *)(a + (long)ch * per), *pb = (const float4 *)(b + (long)ch * per);
        float4 *pc = (float4 *)(c + (long)ch * per);
        const long n4 = per / 4, chunk = (n4 + T - 1) / T;
        const long lo = member * chunk, hi = min(lo + chunk, n4);
        float s = 0.f;
        if (mode != 3) {
            for (long i = lo + threadIdx.x; i < hi; i += 256 * 4) {
                float4 x[4], y[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long j = min(i + u * 256, hi - 1);
                    x[u] = pa[j];
                    y[u] = pb[j];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    s += x[u].x * y[u].x + x[u].y * y[u].y + x[u].z * y[u].z + x[u].w * y[u].w;
            }
        }
        float coef = 1.0f;
        if (mode == 2) {
            // team barrier: one arrival per workgroup, spin on the counter (relaxed agent-scope accesses)
            __shared__ float red[4];
            for (int o = 32; o > 0; o >>= 1)
                s += __shfl_xor(s, o, 64);
            if ((threadIdx.x & 63) == 0)
                red[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) {
                sums[ch * T + member] = red[0] + red[1] + red[2] + red[3];
                __atomic_fetch_add(&ctr[ch], 1, __ATOMIC_RELEASE);
                while (__atomic_load_n(&ctr[ch], __ATOMIC_ACQUIRE) < T) {
                }
                float t = 0.f;
                for (int m = 0; m < T; ++m)
                    t += __builtin_nontemporal_load(&sums[ch * T + m]);
                red[0] = t;
            }
            __syncthreads();
            coef = red[0] * 1e-30f + 1.0f;
        } else if (mode != 3) {
            coef = s * 1e-30f + 1.0f;
        }
        if (mode >= 1) {
            for (long i = lo + threadIdx.x; i < hi; i += 256 * 4) {
                float4 x[4], y[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long j = min(i + u * 256, hi - 1);
                    x[u] = pa[j];
                    y[u] = pb[j];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long j = i + u * 256;
                    if (j < hi)
                        pc[j] = float4{x[u].x * coef + y[u].x, x[u].y * coef + y[u].y, x[u].z * coef + y[u].z, x[u].w * coef + y[u].w};
                }
            }
        } else if (threadIdx.x == 0) {
            sums[ch * T + member] = s;
        }
    }
}

extern "C" int l2_launch(const float *a, const float *b, float *c, float *sums, int *ctr, int channels, long per, int mode,
                         void *stream)
{
    hipLaunchKernelGGL(k_l2, dim3(256), dim3(256), 0, (hipStream_t)stream, a, b, c, sums, ctr, channels, per, mode);
    return (int)hipGetLastError();
}
