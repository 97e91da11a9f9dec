// dcl_layernorm.hip -- LayerNorm over the channel axis of token-major rows [M, C] (gfx950).
//
// Replaces nn.LayerNorm in the Swin port (reference models/Swin.py:251-332 norm1 / norm2, :357-362 PatchMerging.norm,
// :452-455 PatchEmbed.norm, :560-565 the per-stage output norms) and its autograd.  HBM-bound: the forward reads x and
// writes y once, the backward reads gy and x and writes gx once; a row lives in the registers of G = C / (4 V) lanes
// (V float4 vectors per lane), so rows of 96 channels share a wave eight at a time and every load / store is a 16-byte
// vector of a contiguous row.  Statistics are two-pass in registers (mean, then the centred sum of squares).
//
// Backward: gx = rstd * (a - mean_c(a) - xhat * mean_c(a * xhat)), a = gy * gamma.  dgamma = sum_rows gy * xhat and
// dbeta = sum_rows gy accumulate per lane over the rows a workgroup walks, are reduced over the lanes that own the same
// columns and over the four waves through LDS, and leave as one partial row per workgroup; a second launch adds the
// partial rows in fixed order: deterministic, no float atomics.
#include <type_traits>

#include "dcl_common.h"

namespace {

constexpr int LN_BLOCKS = 1024;      // backward partial rows (4 workgroups per CU)

template <int G>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the lanes of a wave that hold the same columns (lane % G equal)
template <int G>
__device__ __forceinline__ float cross_group_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= G; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// Workgroup maximum -> at most ONE integer atomic max (values >= 0: the order of the float bits), and none once the slot already
// holds a value at least as large (a plain load first; a stale read only costs an atomic that changes nothing).  The 64 slots of
// a tag share two cache lines, and atomics on one line retire one after the other (~4 ns each): with one atomic per WAVE the
// forward on 25 600 x 768 took 125 us against 24 us without the side channel (tools/probes/ln_time.py) -- 38 launches of a Swin-L
// step.
__device__ __forceinline__ void block_amax_ln(float m, float *dst)
{
    __shared__ float wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0)
        wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > __builtin_nontemporal_load(dst))
            atomicMax((int *)dst, __float_as_int(m));
    }
}

template <int V, int G>
__global__ __launch_bounds__(256) void k_ln_fwd(const float *__restrict__ x, const float *__restrict__ gamma,
                                                const float *__restrict__ beta, long long M, float eps,
                                                float *__restrict__ y, float *__restrict__ mean,
                                                float *__restrict__ rstd, float *__restrict__ yamax)
{
    constexpr int C = 4 * V * G, RW = 64 / G;                 // rows per wave and trip
    float ymax = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G, rl = lane / G;
    f32x4 gm[V], bt[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        gm[v] = *(const f32x4 *)(gamma + 4 * (sub + v * G));
        bt[v] = *(const f32x4 *)(beta + 4 * (sub + v * G));
    }
    const long long stride = (long long)gridDim.x * 4 * RW;
    for (long long r0 = ((long long)blockIdx.x * 4 + wave) * RW; r0 < M; r0 += stride) {
        const long long row = r0 + rl;
        const bool ok = row < M;
        const float *xp = x + (ok ? row : M - 1) * C;
        f32x4 xv[V];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            xv[v] = *(const f32x4 *)(xp + 4 * (sub + v * G));
            s += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
        }
        const float mu = group_sum<G>(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            xv[v].x -= mu; xv[v].y -= mu; xv[v].z -= mu; xv[v].w -= mu;
            q += (xv[v].x * xv[v].x + xv[v].y * xv[v].y) + (xv[v].z * xv[v].z + xv[v].w * xv[v].w);
        }
        const float rs = 1.0f / sqrtf(group_sum<G>(q) * (1.0f / C) + eps);
        if (ok) {
            float *yp = y + row * C;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                f32x4 o;
                o.x = xv[v].x * rs * gm[v].x + bt[v].x;
                o.y = xv[v].y * rs * gm[v].y + bt[v].y;
                o.z = xv[v].z * rs * gm[v].z + bt[v].z;
                o.w = xv[v].w * rs * gm[v].w + bt[v].w;
                *(f32x4 *)(yp + 4 * (sub + v * G)) = o;
                ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
            if (sub == 0) {
                mean[row] = mu;
                rstd[row] = rs;
            }
        }
    }
    if (yamax)                                               // absmax side channel for an f16x3 consumer (dcl_gemm.hip)
        block_amax_ln(ymax, yamax + (blockIdx.x & (DCL_AMAX_SLOTS - 1)));
}

template <int V, int G>
__global__ __launch_bounds__(256) void k_ln_bwd(const float *__restrict__ gy, const float *__restrict__ x,
                                                const float *__restrict__ gamma, const float *__restrict__ mean,
                                                const float *__restrict__ rstd, long long M, float *__restrict__ gx,
                                                float *__restrict__ parts, const float *__restrict__ addend,
                                                float *__restrict__ gxamax)
{
    constexpr int C = 4 * V * G, RW = 64 / G;
    __shared__ float red[4][2][C];
    float gmax = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G, rl = lane / G;
    f32x4 gm[V], dg[V], db[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        gm[v] = *(const f32x4 *)(gamma + 4 * (sub + v * G));
        dg[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const long long stride = (long long)gridDim.x * 4 * RW;
    for (long long r0 = ((long long)blockIdx.x * 4 + wave) * RW; r0 < M; r0 += stride) {
        const long long row = r0 + rl;
        const bool ok = row < M;
        const long long rc = ok ? row : M - 1;
        const float keep = ok ? 1.f : 0.f;
        const float mu = mean[rc], rs = rstd[rc];
        const float *xp = x + rc * C, *gp = gy + rc * C;
        f32x4 xh[V], a[V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const f32x4 xv = *(const f32x4 *)(xp + 4 * (sub + v * G));
            f32x4 g = *(const f32x4 *)(gp + 4 * (sub + v * G));
            g.x *= keep; g.y *= keep; g.z *= keep; g.w *= keep;
            xh[v].x = (xv.x - mu) * rs; xh[v].y = (xv.y - mu) * rs;
            xh[v].z = (xv.z - mu) * rs; xh[v].w = (xv.w - mu) * rs;
            db[v].x += g.x; db[v].y += g.y; db[v].z += g.z; db[v].w += g.w;
            dg[v].x += g.x * xh[v].x; dg[v].y += g.y * xh[v].y; dg[v].z += g.z * xh[v].z; dg[v].w += g.w * xh[v].w;
            a[v].x = g.x * gm[v].x; a[v].y = g.y * gm[v].y; a[v].z = g.z * gm[v].z; a[v].w = g.w * gm[v].w;
            s1 += (a[v].x + a[v].y) + (a[v].z + a[v].w);
            s2 += (a[v].x * xh[v].x + a[v].y * xh[v].y) + (a[v].z * xh[v].z + a[v].w * xh[v].w);
        }
        const float c1 = group_sum<G>(s1) * (1.0f / C), c2 = group_sum<G>(s2) * (1.0f / C);
        if (ok) {
            float *op = gx + row * C;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                f32x4 o;
                o.x = rs * (a[v].x - c1 - xh[v].x * c2);
                o.y = rs * (a[v].y - c1 - xh[v].y * c2);
                o.z = rs * (a[v].z - c1 - xh[v].z * c2);
                o.w = rs * (a[v].w - c1 - xh[v].w * c2);
                if (addend) {           // the gradient of the residual connection around this norm: one pass instead of
                                        // autograd's separate add (x feeds the norm AND the shortcut)
                    const f32x4 r = *(const f32x4 *)(addend + row * C + 4 * (sub + v * G));
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                *(f32x4 *)(op + 4 * (sub + v * G)) = o;
                gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
    }
    if (gxamax)                         // absmax of the result: the gradient the previous block's Linears receive
        block_amax_ln(gmax, gxamax + (blockIdx.x & (DCL_AMAX_SLOTS - 1)));
    // columns: over the row groups of the wave, then over the waves
#pragma unroll
    for (int v = 0; v < V; ++v) {
        dg[v].x = cross_group_sum<G>(dg[v].x); dg[v].y = cross_group_sum<G>(dg[v].y);
        dg[v].z = cross_group_sum<G>(dg[v].z); dg[v].w = cross_group_sum<G>(dg[v].w);
        db[v].x = cross_group_sum<G>(db[v].x); db[v].y = cross_group_sum<G>(db[v].y);
        db[v].z = cross_group_sum<G>(db[v].z); db[v].w = cross_group_sum<G>(db[v].w);
        if (rl == 0) {
            *(f32x4 *)(&red[wave][0][4 * (sub + v * G)]) = dg[v];
            *(f32x4 *)(&red[wave][1][4 * (sub + v * G)]) = db[v];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * C; idx += 256) {
        const int which = idx / C, c = idx - which * C;
        parts[((size_t)blockIdx.x * 2 + which) * C + c] =
            (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// dgamma | dbeta [2][C] = fixed-order sum of the partial rows: 16 columns x 16 row groups per workgroup (a thread adds
// every 16th partial row, four independent chains; the groups meet in LDS) -- 3 workgroups walking 256 rows each took
// 63 us per call
__global__ __launch_bounds__(256) void k_ln_parts_sum(const float *__restrict__ parts, int nparts, int C2,
                                                      float *__restrict__ out)
{
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < C2) {
        int p = rg;
        for (; p + 48 < nparts; p += 64) {
            s0 += parts[(size_t)p * C2 + col];
            s1 += parts[(size_t)(p + 16) * C2 + col];
            s2 += parts[(size_t)(p + 32) * C2 + col];
            s3 += parts[(size_t)(p + 48) * C2 + col];
        }
        for (; p < nparts; p += 16)
            s0 += parts[(size_t)p * C2 + col];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && col < C2) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g)
            s += red[g][cl];
        out[col] = s;
    }
}

// C = 4 V G with G a power of two in 8 .. 64: the smallest V in {1, 2, 3, 4, 6, 8} that fits
bool ln_plan(int C, int &V, int &G)
{
    if (C <= 0 || C % 4)
        return false;
    const int vecs = C / 4;
    const int vs[] = {1, 2, 3, 4, 6, 8};
    for (int v : vs) {
        if (vecs % v)
            continue;
        const int g = vecs / v;
        if (g >= 8 && g <= 64 && (g & (g - 1)) == 0) {
            V = v;
            G = g;
            return true;
        }
    }
    return false;
}

int ln_blocks(long long M, int G)
{
    const long long rows_per_block = 4 * (64 / G);
    long long b = (M + rows_per_block - 1) / rows_per_block;
    return (int)(b < LN_BLOCKS ? (b < 1 ? 1 : b) : LN_BLOCKS);
}

}  // namespace

extern "C" int dcl_layernorm_supported(int C)
{
    int V, G;
    return ln_plan(C, V, G) ? 1 : 0;
}

extern "C" int dcl_layernorm_bwd_parts(long long M, int C)
{
    int V, G;
    if (M <= 0 || !ln_plan(C, V, G))
        return 0;
    return ln_blocks(M, G);
}

// the (V, G) pairs ln_plan can return (it takes the smallest V that fits) with G >= 8
#define DCL_LN_DISPATCH(KERNEL, ...)                                                          \
    do {                                                                                      \
        bool done = false;                                                                    \
        auto go = [&](auto vtag, auto gtag) {                                                 \
            constexpr int VV = decltype(vtag)::value, GG = decltype(gtag)::value;             \
            if (!done && V == VV && G == GG) {                                                \
                hipLaunchKernelGGL((KERNEL<VV, GG>), dim3(blocks), dim3(256), 0, s, __VA_ARGS__); \
                done = true;                                                                  \
            }                                                                                 \
        };                                                                                    \
        using std::integral_constant;                                                         \
        go(integral_constant<int, 1>{}, integral_constant<int, 8>{});                         \
        go(integral_constant<int, 1>{}, integral_constant<int, 16>{});                        \
        go(integral_constant<int, 1>{}, integral_constant<int, 32>{});                        \
        go(integral_constant<int, 1>{}, integral_constant<int, 64>{});                        \
        go(integral_constant<int, 3>{}, integral_constant<int, 8>{});                         \
        go(integral_constant<int, 3>{}, integral_constant<int, 16>{});                        \
        go(integral_constant<int, 3>{}, integral_constant<int, 32>{});                        \
        go(integral_constant<int, 3>{}, integral_constant<int, 64>{});                        \
        go(integral_constant<int, 2>{}, integral_constant<int, 64>{});                        \
        go(integral_constant<int, 4>{}, integral_constant<int, 64>{});                        \
        go(integral_constant<int, 6>{}, integral_constant<int, 64>{});                        \
        go(integral_constant<int, 8>{}, integral_constant<int, 64>{});                        \
    } while (0)

extern "C" int dcl_layernorm_fwd(const float *x, const float *gamma, const float *beta, long long M, int C, float eps,
                                 float *y, float *mean, float *rstd, float *yamax, void *stream)
{
    DCL_CHECK_ARG(x && gamma && beta && y && mean && rstd, "null pointer");
    int V = 0, G = 0;
    DCL_CHECK_ARG(M > 0 && ln_plan(C, V, G), "unsupported row length (C = 4 V G, G a power of two in 8..64, V in 1,2,3,4,6,8)");
    DCL_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0,
                  "16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const long long rows_per_block = 4 * (64 / G);
    long long nb = (M + rows_per_block - 1) / rows_per_block;
    const unsigned blocks = (unsigned)(nb < 2048 ? nb : 2048);      // 8 resident workgroups per CU, grid-stride over the rows
    DCL_LN_DISPATCH(k_ln_fwd, x, gamma, beta, M, eps, y, mean, rstd, yamax);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_layernorm_bwd(const float *gy, const float *x, const float *gamma, const float *mean,
                                 const float *rstd, long long M, int C, float *gx, float *parts, float *dgamma_dbeta,
                                 const float *addend, float *gxamax, void *stream)
{
    DCL_CHECK_ARG(gy && x && gamma && mean && rstd && gx && parts && dgamma_dbeta, "null pointer");
    DCL_CHECK_ARG(!addend || (((uintptr_t)addend) & 15) == 0, "16-byte alignment");
    int V = 0, G = 0;
    DCL_CHECK_ARG(M > 0 && ln_plan(C, V, G), "unsupported row length (C = 4 V G, G a power of two in 8..64, V in 1,2,3,4,6,8)");
    DCL_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)gx) | ((uintptr_t)gamma)) & 15) == 0,
                  "16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)ln_blocks(M, G);
    DCL_LN_DISPATCH(k_ln_bwd, gy, x, gamma, mean, rstd, M, gx, parts, addend, gxamax);
    DCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ln_parts_sum, dim3((2 * C + 15) / 16), dim3(256), 0, s, parts, (int)blocks, 2 * C,
                       dgamma_dbeta);
    DCL_LAUNCH_CHECK();
    return 0;
}
