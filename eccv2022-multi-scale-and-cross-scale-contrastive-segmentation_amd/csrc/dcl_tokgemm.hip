// dcl_tokgemm.hip -- y[M, N] = x[M, K] W^T (+ bias) on token-major rows, fp32-equivalent on the f16 matrix pipe (gfx950).
//
// The Linear layers of the Swin port (reference models/Swin.py:62-76 Mlp.fc1 / fc2, :198-230 WindowAttention.qkv / proj,
// :357-362 PatchMerging.reduction) and their data gradient (the same kernel on the transposed fragments, x = dy).  Same
// arithmetic as the direct convolutions: operands scaled by a power of two derived on the device from their absmax,
// split into f16 hi / lo, products hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with f32 accumulation.
//
// A wave owns 32 rows x R column tiles of 32: the x fragment of a 16-wide k chunk is 8 consecutive floats of the lane's
// row (row = lane % 32, k half = lane / 32), loaded straight from global memory one chunk ahead and split in registers
// (2 VALU per value); the weight fragments come pre-packed by dcl_conv3x3_pack's one-tap mode (lane = (column, k half),
// the layout of the MFMA's other operand -- the two are symmetric) and stream from L2.  No LDS, no barriers.  The result
// tile has the 32 columns across the lanes, so every store is two 128-byte row segments.  The column groups of a row
// block are adjacent workgroups (they share the x rows through L2).
#include "dcl_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr float F16_TARGET = 16384.0f;      // as in dcl_conv3x3.hip: the packer and the consumer must agree

__device__ __forceinline__ float pow2_scale(float amax)
{
    return amax == 0.f ? 1.f : exp2f(fminf(fmaxf(floorf(log2f(F16_TARGET / amax)), -100.f), 100.f));
}

__device__ __forceinline__ void split2(float v0, float v1, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(s), "v"(hi));
}

struct TgArgs {
    const float *x;             // [M, K]
    const uint4 *wp;            // [N / 32][K / 16][hi | lo][64] fragments (dcl_conv3x3_pack, one tap)
    const float *bias;          // [N] or null
    float *y;                   // [M, N]
    const float *xamax, *wamax; // max|x| as xcount partial maxima, max|w| (1 value)
    float *yamax;               // optional: DCL_AMAX_SLOTS partial maxima of |y| (integer atomicMax on the float bits)
    long long M;
    int K, N, xcount, nchunk, ngroups;
    int dbg;                    // timing experiments only: 1 = x always from chunk 0, 2 = weights always from chunk 0
};

template <int R, int P>
__global__ __launch_bounds__(256) void k_tok_gemm(TgArgs a)
{
    constexpr int RW = 32 * P;                               // rows per wave: P row tiles share every weight fragment
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x % a.ngroups;
    const long long row0 = (long long)(blockIdx.x / a.ngroups) * (4 * RW) + wave * RW;
    if (row0 >= a.M)
        return;
    float m = 0.f;
    for (int i = lane; i < a.xcount; i += 64)
        m = fmaxf(m, a.xamax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = fmaxf(m, __shfl_xor(m, o, 64));
    const float xs = pow2_scale(m);
    const int li = lane & 31, h = lane >> 5;
    const float *xp[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const long long r = row0 + 32 * p + li < a.M ? row0 + 32 * p + li : a.M - 1;
        xp[p] = a.x + r * a.K + 8 * h;
    }
    const uint4 *wp = a.wp + (size_t)(g * R) * a.nchunk * 128 + lane;
    const size_t tstride = (size_t)a.nchunk * 128;          // uint4 per column tile

    f32x16 acc[P][R];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int t = 0; t < R; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                acc[p][t][q] = 0.f;

    f32x4 xa[P], xb[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        xa[p] = *(const f32x4 *)xp[p];
        xb[p] = *(const f32x4 *)(xp[p] + 4);
    }
    uint4 bh[R], bl[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        bh[t] = wp[t * tstride];
        bl[t] = wp[t * tstride + 64];
    }
    for (int c = 0; c < a.nchunk; ++c) {
        // the next chunk's loads are pinned above the MFMA block (left alone the scheduler sinks them to their use and
        // every trip waits for memory); the last trip re-loads its own chunk instead of branching
        const int cn = c + 1 < a.nchunk ? c + 1 : c;
        const int cx = (a.dbg & 1) ? 0 : cn, cw = (a.dbg & 2) ? 0 : cn;
        f32x4 nxa[P], nxb[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            nxa[p] = *(const f32x4 *)(xp[p] + 16 * cx);
            nxb[p] = *(const f32x4 *)(xp[p] + 16 * cx + 4);
        }
        uint4 nbh[R], nbl[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            nbh[t] = wp[t * tstride + (size_t)cw * 128];
            nbl[t] = wp[t * tstride + (size_t)cw * 128 + 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        half8 ah[P], al[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            uint4 uh, ul;
            split2(xa[p].x, xa[p].y, xs, uh.x, ul.x);
            split2(xa[p].z, xa[p].w, xs, uh.y, ul.y);
            split2(xb[p].x, xb[p].y, xs, uh.z, ul.z);
            split2(xb[p].z, xb[p].w, xs, uh.w, ul.w);
            ah[p] = __builtin_bit_cast(half8, uh);
            al[p] = __builtin_bit_cast(half8, ul);
        }
        // pass-major, nothing in between: the three MFMAs of an accumulator are P R instructions apart.  (With the
        // split's inline asm scheduled BETWEEN two dependent MFMAs the first column tile came out wrong by 1e-4 -- the
        // hazard recognizer does not see through the asm.)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int t = 0; t < R; ++t)
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[p], __builtin_bit_cast(half8, bh[t]), acc[p][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int t = 0; t < R; ++t)
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[p], __builtin_bit_cast(half8, bl[t]), acc[p][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int t = 0; t < R; ++t)
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[p], __builtin_bit_cast(half8, bh[t]), acc[p][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            xa[p] = nxa[p];
            xb[p] = nxb[p];
        }
#pragma unroll
        for (int t = 0; t < R; ++t) {
            bh[t] = nbh[t];
            bl[t] = nbl[t];
        }
    }

    const float inv = 1.0f / (xs * pow2_scale(a.wamax[0]));
    float ymax = 0.f;
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int col = (g * R + t) * 32 + li;
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const long long row = row0 + 32 * p + (q & 3) + 8 * (q >> 2) + 4 * h;
                const float v = acc[p][t][q] * inv + bv;
                if (row < a.M) {
                    a.y[row * a.N + col] = v;
                    ymax = fmaxf(ymax, fabsf(v));
                }
            }
    }
    if (a.yamax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
        if (lane == 0)
            atomicMax((int *)a.yamax + (blockIdx.x & (DCL_AMAX_SLOTS - 1)), __float_as_int(ymax));
    }
}

}  // namespace

static int g_tok_dbg = 0;
static int g_tok_p = 0;        // tuning override: row tiles per wave (0 = automatic)

extern "C" int dcl_tok_gemm_set_rows(int p)
{
    g_tok_dbg = p >= 16 ? (p >> 4) : 0;          // timing experiments (results are wrong): 16 | 32 added to p
    p &= 15;
    g_tok_p = p == 1 || p == 2 ? p : 0;
    return 0;
}

extern "C" int dcl_tok_gemm_supported(int K, int N)
{
    return K > 0 && N > 0 && K % 16 == 0 && N % 32 == 0;
}

extern "C" int dcl_tok_gemm_f16x3(const float *x, long long M, int K, const void *wp, int N, const float *xamax,
                                  int xcount, const float *wamax, const float *bias, float *y, float *yamax,
                                  void *stream)
{
    DCL_CHECK_ARG(x && wp && xamax && wamax && y, "null pointer");
    DCL_CHECK_ARG(M > 0 && xcount > 0 && dcl_tok_gemm_supported(K, N), "bad shape (K % 16 == 0, N % 32 == 0)");
    DCL_CHECK_ARG((((uintptr_t)x) & 15) == 0, "16-byte alignment");
    TgArgs a;
    a.x = x; a.wp = (const uint4 *)wp; a.bias = bias; a.y = y; a.xamax = xamax; a.wamax = wamax; a.yamax = yamax;
    a.M = M; a.K = K; a.N = N; a.xcount = xcount; a.nchunk = K / 16; a.dbg = g_tok_dbg;
    const int ntile = N / 32;
    const int R = ntile % 3 == 0 ? 3 : (ntile % 2 == 0 ? 2 : 1);
    a.ngroups = ntile / R;
    // two row tiles per wave (every weight fragment feeds two MFMAs: the kernel is bound by the L1 path of the
    // fragments) when that still leaves two workgroups per CU
    int P = g_tok_p;
    if (P <= 0)
        P = ((M + 255) / 256) * a.ngroups >= 512 ? 2 : 1;
    const long long rb = (M + 128 * P - 1) / (128 * P);
    DCL_CHECK_ARG(rb * a.ngroups < ((long long)1 << 31), "grid too large");
    const dim3 grid((unsigned)(rb * a.ngroups));
    hipStream_t s = (hipStream_t)stream;
#define DCL_TG_CASE(r, p)                                                   \
    if (R == r && P == p)                                                   \
        hipLaunchKernelGGL((k_tok_gemm<r, p>), grid, dim3(256), 0, s, a);
    DCL_TG_CASE(3, 2) DCL_TG_CASE(3, 1) DCL_TG_CASE(2, 2) DCL_TG_CASE(2, 1) DCL_TG_CASE(1, 2) DCL_TG_CASE(1, 1)
#undef DCL_TG_CASE
    DCL_LAUNCH_CHECK();
    return 0;
}
