// dcl_bank.hip -- K3 (gather + L2-normalise into a class-sorted bank) and K6 (slab reduce +
// normalise-backward + scatter into the dense feature gradient).  HBM-bound: one wave per bank
// row, 1 KiB coalesced row stores / loads on the bank side.
#include "dcl_common.h"

// One wave per bank row; 4 rows per workgroup.  Lane l owns channels l, l+64, l+128, l+192 so that
// for channel-last features (stride_c == 1) the gather reads are 256-B coalesced.
// Replaces DenseContrastiveLossV2.py:123 (gather), :138 (F.normalize, eps 1e-12), :139-149 (layout).
__global__ __launch_bounds__(256) void k_gather_normalize(
    const float *__restrict__ feat, int64_t sn, int64_t sc, int64_t sp, int C,
    const int32_t *__restrict__ pix, const int32_t *__restrict__ pair_b,
    const int32_t *__restrict__ slot_pair, int N, int V, int Npad, float *__restrict__ bank,
    float *__restrict__ nrm, _Float16 *__restrict__ bank_h)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Npad)
        return;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    if (row < N) {
        const int u = row / V, v = row - u * V;
        const int t = slot_pair[u];
        const int64_t base = (int64_t)pair_b[t] * sn + (int64_t)pix[(int64_t)t * V + v] * sp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane + 64 * q;
            if (c < C)
                x[q] = feat[base + (int64_t)c * sc];
            ss += x[q] * x[q];
        }
        ss = wave_sum(ss);
    }
    const float norm = sqrtf(ss);
    const float inv = 1.0f / fmaxf(norm, 1e-12f);
    float *out = bank + (int64_t)row * DCL_CP;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        out[lane + 64 * q] = x[q] * inv;
    if (bank_h) {
        // f16x3 format of the same row: [256 hi | 256 lo] halves, hi = f16(f * 2^10), lo = f16(f * 2^10 - hi)
        _Float16 *oh = bank_h + (int64_t)row * (2 * DCL_CP);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = x[q] * inv * 1024.0f;
            const _Float16 hi = (_Float16)v;
            oh[lane + 64 * q] = hi;
            oh[DCL_CP + lane + 64 * q] = (_Float16)(v - (float)hi);
        }
    }
    if (lane == 0)
        nrm[row] = norm;
}

extern "C" int dcl_gather_normalize(const float *feat, int64_t stride_n, int64_t stride_c,
                                    int64_t stride_p, int C, const int32_t *pix,
                                    const int32_t *pair_b, const int32_t *slot_pair, int T, int V,
                                    float *bank, float *nrm, void *bank_h, void *stream)
{
    DCL_CHECK_ARG(feat && pix && pair_b && slot_pair && bank && nrm, "null pointer");
    DCL_CHECK_ARG(C > 0 && C <= DCL_CP, "embedding width must be in [1, 256]");
    DCL_CHECK_ARG(T > 0 && V > 0, "empty bank");
    const int N = T * V, Npad = dcl_round_up(N, DCL_ROW_TILE);
    hipLaunchKernelGGL(k_gather_normalize, dim3(Npad / 4), dim3(256), 0, (hipStream_t)stream, feat,
                       stride_n, stride_c, stride_p, C, pix, pair_b, slot_pair, N, V, Npad, bank, nrm,
                       (_Float16 *)bank_h);
    DCL_LAUNCH_CHECK();
    return 0;
}

// X[t, c, v] = feat[b_t, c, pix[t, v]] in the reference's own [T, C, V] layout.
__global__ __launch_bounds__(256) void k_gather_raw(const float *__restrict__ feat, int64_t sn,
                                                   int64_t sc, int64_t sp, int C,
                                                   const int32_t *__restrict__ pix,
                                                   const int32_t *__restrict__ pair_b, int V,
                                                   int64_t total, float *__restrict__ X)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total)
        return;
    const int v = (int)(e % V);
    const int64_t tc = e / V;
    const int c = (int)(tc % C);
    const int t = (int)(tc / C);
    X[e] = feat[(int64_t)pair_b[t] * sn + (int64_t)c * sc + (int64_t)pix[(int64_t)t * V + v] * sp];
}

extern "C" int dcl_gather_raw(const float *feat, int64_t stride_n, int64_t stride_c,
                              int64_t stride_p, int C, const int32_t *pix, const int32_t *pair_b,
                              int T, int V, float *X, void *stream)
{
    DCL_CHECK_ARG(feat && pix && pair_b && X, "null pointer");
    DCL_CHECK_ARG(C > 0 && T > 0 && V > 0, "bad sizes");
    const int64_t total = (int64_t)T * C * V;
    hipLaunchKernelGGL(k_gather_raw, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, stride_n, stride_c, stride_p, C, pix, pair_b, V,
                       total, X);
    DCL_LAUNCH_CHECK();
    return 0;
}

// Adjoint of k_gather_raw: dfeat[b_t, c, pix[t, v]] = dX[t, c, v].  The sampled pixels of a scale are unique per image
// and an (image, class) pair owns its pixels, so every address is written once: plain stores into the zero-filled map,
// deterministic (autograd's index_put_(accumulate=True) of the [T, C, V] bank in the reference).
__global__ __launch_bounds__(256) void k_scatter_raw(const float *__restrict__ dX, int64_t sn, int64_t sc, int64_t sp,
                                                    int C, const int32_t *__restrict__ pix,
                                                    const int32_t *__restrict__ pair_b, int V, int64_t total,
                                                    float *__restrict__ dfeat)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total)
        return;
    const int v = (int)(e % V);
    const int64_t tc = e / V;
    const int c = (int)(tc % C);
    const int t = (int)(tc / C);
    dfeat[(int64_t)pair_b[t] * sn + (int64_t)c * sc + (int64_t)pix[(int64_t)t * V + v] * sp] = dX[e];
}

extern "C" int dcl_scatter_raw(const float *dX, int64_t stride_n, int64_t stride_c, int64_t stride_p, int C,
                               const int32_t *pix, const int32_t *pair_b, int T, int V, float *dfeat, void *stream)
{
    DCL_CHECK_ARG(dX && pix && pair_b && dfeat, "null pointer");
    DCL_CHECK_ARG(C > 0 && T > 0 && V > 0, "bad sizes");
    const int64_t total = (int64_t)T * C * V;
    hipLaunchKernelGGL(k_scatter_raw, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dX,
                       stride_n, stride_c, stride_p, C, pix, pair_b, V, total, dfeat);
    DCL_LAUNCH_CHECK();
    return 0;
}

#define DCL_MAX_SLABS 64
struct SlabList {
    const float *p[DCL_MAX_SLABS];
};

// One wave per bank row: dF = sum of slabs (fixed order -> bitwise reproducible), then
// dx = (dF - f (f . dF)) / max(|x|, eps)   [|x| <= eps: dF / eps], scattered to dfeat[b, :, pix].
__global__ __launch_bounds__(256) void k_normalize_bwd_scatter(
    SlabList slabs, int nslab, const float *__restrict__ bank, const float *__restrict__ nrm,
    const int32_t *__restrict__ pix, const int32_t *__restrict__ pair_b,
    const int32_t *__restrict__ slot_pair, int N, int V, int C, float *__restrict__ dfeat,
    int64_t sn, int64_t sc, int64_t sp, float *__restrict__ amax)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N)
        return;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nslab; ++s) {
        const float *src = slabs.p[s] + (int64_t)row * DCL_CP;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            d[q] += src[lane + 64 * q];
    }
    float f[4];
    float inner = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f[q] = bank[(int64_t)row * DCL_CP + lane + 64 * q];
        inner += f[q] * d[q];
    }
    inner = wave_sum(inner);
    const float norm = nrm[row];
    const float inv = 1.0f / fmaxf(norm, 1e-12f);
    const bool clamped = !(norm > 1e-12f);
    const int u = row / V, v = row - u * V;
    const int t = slot_pair[u];
    const int64_t base = (int64_t)pair_b[t] * sn + (int64_t)pix[(int64_t)t * V + v] * sp;
    float am = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        if (c < C) {
            const float dx = clamped ? d[q] * inv : (d[q] - f[q] * inner) * inv;
            dfeat[base + (int64_t)c * sc] = dx;
            am = fmaxf(am, fabsf(dx));
        }
    }
    // absmax side channel of the (otherwise zero) feature gradient, for the f16x3 convolutions of the projector's
    // backward: DCL_AMAX_SLOTS partial maxima, integer atomicMax on the float bits (values >= 0: order independent)
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            am = fmaxf(am, __shfl_xor(am, o, 64));
        // (a plain load first: atomics on the two cache lines of a tag retire one after the other, one per sampled row adds up;
        // a stale read only costs an atomic that changes nothing)
        if (lane == 0)
            atomicMax((unsigned int *)(amax + (row & (DCL_AMAX_SLOTS - 1))), __float_as_uint(am));
    }
}

extern "C" int dcl_normalize_bwd_scatter(const float *const *slabs_host, int nslab,
                                         const float *bank, const float *nrm, const int32_t *pix,
                                         const int32_t *pair_b, const int32_t *slot_pair, int T,
                                         int V, int C, float *dfeat, int64_t stride_n,
                                         int64_t stride_c, int64_t stride_p, float *amax, void *stream)
{
    DCL_CHECK_ARG(slabs_host && bank && nrm && pix && pair_b && slot_pair && dfeat, "null pointer");
    DCL_CHECK_ARG(nslab >= 0 && nslab <= DCL_MAX_SLABS, "too many slabs (max 64)");
    DCL_CHECK_ARG(C > 0 && C <= DCL_CP && T > 0 && V > 0, "bad sizes");
    SlabList sl;
    for (int i = 0; i < DCL_MAX_SLABS; ++i)
        sl.p[i] = i < nslab ? slabs_host[i] : nullptr;
    const int N = T * V;
    hipLaunchKernelGGL(k_normalize_bwd_scatter, dim3((N + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, sl, nslab, bank, nrm, pix, pair_b, slot_pair, N, V, C,
                       dfeat, stride_n, stride_c, stride_p, amax);
    DCL_LAUNCH_CHECK();
    return 0;
}
