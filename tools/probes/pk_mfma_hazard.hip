// pk_mfma_hazard.hip -- do packed-FP32 instructions (v_pk_fma_f32 ...) of one wave return wrong bits while ANOTHER wave on the same
// SIMD issues matrix instructions?  Stand-alone (no library, no torch):
//
//     hipcc -O3 --offload-arch=gfx950 tools/probes/pk_mfma_hazard.hip -o tools/probes/variants/pk_mfma_hazard && gpurun -- ...
//
// Stream A: k_pk -- every lane runs a chain of packed operations AND the same chain on scalar v_fma_f32 / v_mul_f32 / v_add_f32
// (inline asm, so the compiler cannot merge or split them), compares the two bit patterns after every link and counts mismatches
// per lane.  Stream B, concurrently: k_mfma<KIND> -- a chain of matrix instructions of one shape (or of plain VALU FMAs), few
// registers, so that its waves share SIMDs with k_pk's.  Found with it: see DESIGN.md section 7, "Packed FP32 beside MFMA".
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                      \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                 \
            exit(1);                                                                                  \
        }                                                                                             \
    } while (0)

// OP: 0 = v_pk_fma_f32, 1 = v_pk_mul_f32, 2 = v_pk_add_f32
template <int OP>
__global__ __launch_bounds__(256) void k_pk(const float *__restrict__ in, unsigned *__restrict__ bad, int iters)
{
    const int lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x2 a = {in[4 * g], in[4 * g + 1]}, b = {in[4 * g + 2], in[4 * g + 3]};
    f32x2 c = {0.25f, -0.5f};
    float s0 = c.x, s1 = c.y;
    unsigned nbad_lo = 0, nbad_hi = 0;
    for (int i = 0; i < iters; ++i) {
        f32x2 p;
        float q0, q1;
        if (OP == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(a.x), "v"(b.x), "v"(s0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(a.y), "v"(b.y), "v"(s1));
        } else if (OP == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(c));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(a.x), "v"(s0));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(a.y), "v"(s1));
        } else {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(b), "v"(c));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q0) : "v"(b.x), "v"(s0));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q1) : "v"(b.y), "v"(s1));
        }
        nbad_lo += __float_as_uint(p.x) != __float_as_uint(q0);
        nbad_hi += __float_as_uint(p.y) != __float_as_uint(q1);
        // keep the chain bounded: fold back into (-1, 1), the scalar chain continues from the SCALAR results
        c.x = q0 - truncf(q0);
        c.y = q1 - truncf(q1);
        s0 = c.x;
        s1 = c.y;
    }
    if (nbad_lo)
        atomicAdd(bad + lane, nbad_lo);
    if (nbad_hi)
        atomicAdd(bad + 64 + lane, nbad_hi);
}

// KIND: 0 = VALU FMAs, 1 = v_mfma_f32_16x16x32_f16, 2 = v_mfma_f32_32x32x16_f16, 3 = v_mfma_f32_16x16x16_f16, 4 = v_mfma_f32_32x32x8_f16
template <int KIND>
__global__ __launch_bounds__(256) void k_mfma(float *__restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63;
    half8 a8, b8;
    half4 a4, b4;
    for (int i = 0; i < 8; ++i) {
        a8[i] = (_Float16)(0.001f * (lane + i));
        b8[i] = (_Float16)(0.002f * (lane - i));
    }
    for (int i = 0; i < 4; ++i) {
        a4[i] = a8[i];
        b4[i] = b8[i];
    }
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 c16;
    for (int i = 0; i < 16; ++i)
        c16[i] = 0.f;
    float f = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(0.999f), "v"(0.001f));
        } else if (KIND == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c4, 0, 0, 0);
        } else if (KIND == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c16, 0, 0, 0);
        } else if (KIND == 3) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                c4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                c16 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c16, 0, 0, 0);
        }
    }
    float r = f + c4[0] + c4[1] + c4[2] + c4[3];
    for (int i = 0; i < 16; ++i)
        r += c16[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
static void launch_mfma(float *out, int blocks, int iters, hipStream_t st)
{
    hipLaunchKernelGGL(k_mfma<KIND>, dim3(blocks), dim3(256), 0, st, out, iters);
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    const int pk_blocks = 512, mf_blocks = 512;
    float *in, *out;
    unsigned *bad;
    std::vector<float> h((size_t)pk_blocks * 256 * 4);
    unsigned seed = 12345u;
    for (auto &v : h) {
        seed = seed * 1664525u + 1013904223u;
        v = ((seed >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f;
    }
    CHECK(hipMalloc(&in, h.size() * 4));
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, (size_t)mf_blocks * 256 * 4));
    CHECK(hipMalloc(&bad, 128 * 4));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const char *kinds[] = {"VALU v_fma_f32", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x16_f16",
                           "v_mfma_f32_32x32x8_f16", "nothing"};
    const char *ops[] = {"v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"};
    for (int kind = 0; kind < 6; ++kind)
        for (int op = 0; op < 3; ++op) {
            CHECK(hipMemset(bad, 0, 128 * 4));
            CHECK(hipDeviceSynchronize());
            for (int r = 0; r < rounds; ++r) {
                if (kind == 0) launch_mfma<0>(out, mf_blocks, 4000, sb);
                if (kind == 1) launch_mfma<1>(out, mf_blocks, 4000, sb);
                if (kind == 2) launch_mfma<2>(out, mf_blocks, 2000, sb);
                if (kind == 3) launch_mfma<3>(out, mf_blocks, 4000, sb);
                if (kind == 4) launch_mfma<4>(out, mf_blocks, 2000, sb);
                for (int k = 0; k < 4; ++k) {
                    if (op == 0) hipLaunchKernelGGL(k_pk<0>, dim3(pk_blocks), dim3(256), 0, sa, in, bad, 2000);
                    if (op == 1) hipLaunchKernelGGL(k_pk<1>, dim3(pk_blocks), dim3(256), 0, sa, in, bad, 2000);
                    if (op == 2) hipLaunchKernelGGL(k_pk<2>, dim3(pk_blocks), dim3(256), 0, sa, in, bad, 2000);
                }
                CHECK(hipDeviceSynchronize());
            }
            unsigned hb[128];
            CHECK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
            unsigned long long lo = 0, hi = 0, q[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) {
                lo += hb[l];
                hi += hb[64 + l];
                q[l / 16] += hb[l] + hb[64 + l];
            }
            const double total = (double)rounds * 4 * pk_blocks * 256 * 2000.0 * 2;
            printf("beside %-26s %-13s: %llu mismatching results of %.3g (low half %llu, high half %llu; lanes 0-15 %llu, 16-31 %llu, "
                   "32-47 %llu, 48-63 %llu)\n", kinds[kind], ops[op], lo + hi, total, lo, hi, q[0], q[1], q[2], q[3]);
            fflush(stdout);
        }
    return 0;
}
