// dcl_bn.hip -- fused training-mode BatchNorm2d (+ residual add) (+ ReLU) for NCHW f32 tensors.
//
// HRNet-W48 has 310 BatchNorm layers, almost all followed by ReLU and a third of them by a residual
// add (reference models/HRNet.py:77-93, 118-137, 270-285).  In the reference each of these is 2-4
// separate HBM-bound kernels (batch_norm, add_, relu_ and their backwards).  Here one statistics pass
// and one apply pass per direction do all of it:
//   forward : stats (sum, sumsq per channel) -> finalize (mean, invstd, running stats) ->
//             y = relu(gamma * (x - mean) * invstd + beta + residual)
//   backward: g = dy * (y > 0);  reduce (sum g, sum g*xhat) -> dx = gamma*invstd*(g - mean_g - xhat*mean_gx),
//             d_residual = g, dgamma, dbeta
// All kernels are HBM-bound streaming kernels (16-B accesses, one (n, c) plane per workgroup row).
// The statistics are exchanged between ranks by the caller (SyncBatchNorm semantics) between the
// stats/reduce kernel and the finalize/apply kernel.
#include "dcl_common.h"

namespace {

constexpr int BN_THREADS = 256;
#ifndef DCL_BN_UNROLL
#define DCL_BN_UNROLL 4
#endif
constexpr int BN_UNROLL = DCL_BN_UNROLL;       // 16-byte vectors per thread in the element-wise kernels

// Probe builds (tools/probes/bn_coherence.sh, -DDCL_BN_PROBE=<bits>; 0 in the product): 1 = the per-slice partial sums are read
// with agent-scope VECTOR loads instead of the scalar loads the compiler picks for uniform addresses, 2 = the per-channel mean /
// invstd / gamma / beta of the backward kernels as well.
#ifndef DCL_BN_PROBE
#define DCL_BN_PROBE 0
#endif
__device__ __forceinline__ float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ldc(const float *p) { return (DCL_BN_PROBE & 2) ? ld_agent(p) : *p; }

__device__ inline void block_reduce2(float &a, float &b, float *sh)
{
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sh[w] = a;
        sh[4 + w] = b;
    }
    __syncthreads();
    a = sh[0] + sh[1] + sh[2] + sh[3];
    b = sh[4] + sh[5] + sh[6] + sh[7];
}

// grid (C, nslice): workgroup (c, s) reduces planes n = s, s + nslice, ... of channel c.
// part[(c * nslice + s) * 2 + {0,1}] = {sum x, sum x^2}
// Shifted sums: with a pivot p[c] (the layer's running mean: the same value on every rank, and close to the batch
// mean after the first steps) the kernel accumulates sum (x - p) and sum (x - p)^2, so that var = E[(x-p)^2] -
// E[x-p]^2 does not cancel catastrophically when |mean| >> std (plain sums: 25 % error in var at mean 50, std 0.1
// over 393k elements).  The pivot actually used is copied to pivot_out for the apply kernel (the running mean itself
// is updated there, concurrently with other workgroups' prologues).
// MM: the slice's extrema of the RAW values as well, mm[(c * nslice + s) * 2 + {0,1}] = {min x, max x} -- for consumers that
// apply the norm's map on the fly (dcl_conv3x3_pre.hip) and need max|relu(sc x + sh)| without the mapped tensor: the map is
// monotone per channel, so the channel's extrema give it exactly (k_bn_finalize_pre).
// TAIL (with MM): the workgroup's four results leave as agent-scope stores and the workgroup reports to the caller (thread 0:
// {sum, sum of squares, min, max} in res) instead of writing them plainly -- k_bn_stats_pre, below.
template <bool MM = false, bool TAIL = false>
__device__ __forceinline__ void bn_stats_body(const float *__restrict__ x, int N, int C,
                                              int HW, int nslice, float *__restrict__ part,
                                              const float *__restrict__ pivot_src,
                                              float *__restrict__ pivot_out, const int c, const int s,
                                              float *__restrict__ mm = nullptr, float *pv_out = nullptr)
{
    static_assert(!TAIL || MM, "tail form: with extrema");
    __shared__ float sh[8];
    const float pv = pivot_src ? pivot_src[c] : 0.f;
    if (pivot_out && s == 0 && threadIdx.x == 0)
        pivot_out[c] = pv;
    if (TAIL)
        *pv_out = pv;
    float a = 0.f, b = 0.f;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    const int hw4 = (HW & 3) ? 0 : (HW >> 2);      // 16-B loads only when every plane base is 16-B aligned
    for (int n = s; n < N; n += nslice) {
        const float *p = x + ((size_t)n * C + c) * HW;
        const f32x4 *p4 = (const f32x4 *)p;
        int i = threadIdx.x;
        for (; i + 3 * BN_THREADS < hw4; i += 4 * BN_THREADS) {       // four independent 16-byte loads in flight
            f32x4 v0 = p4[i], v1 = p4[i + BN_THREADS], v2 = p4[i + 2 * BN_THREADS], v3 = p4[i + 3 * BN_THREADS];
            if (MM) {
                lo = fminf(fminf(fminf(lo, fminf(v0.x, v0.y)), fminf(fminf(v0.z, v0.w), fminf(v1.x, v1.y))),
                           fminf(fminf(fminf(v1.z, v1.w), fminf(v2.x, v2.y)), fminf(fminf(v2.z, v2.w), fminf(fminf(v3.x, v3.y), fminf(v3.z, v3.w)))));
                hi = fmaxf(fmaxf(fmaxf(hi, fmaxf(v0.x, v0.y)), fmaxf(fmaxf(v0.z, v0.w), fmaxf(v1.x, v1.y))),
                           fmaxf(fmaxf(fmaxf(v1.z, v1.w), fmaxf(v2.x, v2.y)), fmaxf(fmaxf(v2.z, v2.w), fmaxf(fmaxf(v3.x, v3.y), fmaxf(v3.z, v3.w)))));
            }
            v0.x -= pv; v0.y -= pv; v0.z -= pv; v0.w -= pv;
            v1.x -= pv; v1.y -= pv; v1.z -= pv; v1.w -= pv;
            v2.x -= pv; v2.y -= pv; v2.z -= pv; v2.w -= pv;
            v3.x -= pv; v3.y -= pv; v3.z -= pv; v3.w -= pv;
            a += ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w)) +
                 (((v2.x + v2.y) + (v2.z + v2.w)) + ((v3.x + v3.y) + (v3.z + v3.w)));
            b += ((v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w)) +
                 ((v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w)) +
                 (((v2.x * v2.x + v2.y * v2.y) + (v2.z * v2.z + v2.w * v2.w)) +
                  ((v3.x * v3.x + v3.y * v3.y) + (v3.z * v3.z + v3.w * v3.w)));
        }
        for (; i < hw4; i += BN_THREADS) {
            f32x4 v = p4[i];
            if (MM) {
                lo = fminf(lo, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
                hi = fmaxf(hi, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            }
            v.x -= pv; v.y -= pv; v.z -= pv; v.w -= pv;
            a += (v.x + v.y) + (v.z + v.w);
            b += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        for (int i = (hw4 << 2) + threadIdx.x; i < HW; i += BN_THREADS) {
            if (MM) {
                lo = fminf(lo, p[i]);
                hi = fmaxf(hi, p[i]);
            }
            const float v = p[i] - pv;
            a += v;
            b += v * v;
        }
    }
    block_reduce2(a, b, sh);
    if (threadIdx.x == 0) {
        if (TAIL) {
            __hip_atomic_store(part + ((size_t)c * nslice + s) * 2 + 0, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part + ((size_t)c * nslice + s) * 2 + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            part[((size_t)c * nslice + s) * 2 + 0] = a;
            part[((size_t)c * nslice + s) * 2 + 1] = b;
        }
    }
    if (MM) {
        __shared__ float shm[8];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, 64));
            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            shm[threadIdx.x >> 6] = lo;
            shm[4 + (threadIdx.x >> 6)] = hi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float l = fminf(fminf(shm[0], shm[1]), fminf(shm[2], shm[3])), h = fmaxf(fmaxf(shm[4], shm[5]), fmaxf(shm[6], shm[7]));
            if (TAIL) {
                __hip_atomic_store(mm + ((size_t)c * nslice + s) * 2 + 0, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mm + ((size_t)c * nslice + s) * 2 + 1, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                mm[((size_t)c * nslice + s) * 2 + 0] = l;
                mm[((size_t)c * nslice + s) * 2 + 1] = h;
            }
        }
    }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_stats_mm(const float *__restrict__ x, int N, int C, int HW, int nslice,
                                                           float *__restrict__ part, const float *__restrict__ pivot_src,
                                                           float *__restrict__ pivot_out, float *__restrict__ mm)
{
    bn_stats_body<true>(x, N, C, HW, nslice, part, pivot_src, pivot_out, blockIdx.x, blockIdx.y, mm);
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_stats(const float *__restrict__ x, int N, int C,
                                                        int HW, int nslice, float *__restrict__ part,
                                                        const float *__restrict__ pivot_src = nullptr,
                                                        float *__restrict__ pivot_out = nullptr)
{
    bn_stats_body(x, N, C, HW, nslice, part, pivot_src, pivot_out, blockIdx.x, blockIdx.y);
}

// sums[c*2 + {0,1}] = fixed-order sum of the slices (double accumulation).  Optional extras, so that the
// single-rank path needs no further tiny launches: o0/o1 receive the two sums as separate [C] vectors
// (dbeta / dgamma of the backward), and with `finalize` the forward statistics are turned into
// mean / invstd (+ running-stat update, PyTorch's convention: unbiased variance, r = (1-m) r + m s).
__global__ void k_bn_combine(const float *__restrict__ part, int C, int nslice, float *__restrict__ sums,
                             float *__restrict__ o0, float *__restrict__ o1, int finalize, double count,
                             float eps, float momentum, float *__restrict__ mean,
                             float *__restrict__ invstd, float *__restrict__ running_mean,
                             float *__restrict__ running_var, long long *__restrict__ batches_tracked)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && batches_tracked)
        batches_tracked[0] += 1;                 // nn.BatchNorm2d.num_batches_tracked, without its own launch
    if (c >= C)
        return;
    double a = 0.0, b = 0.0;
    for (int s = 0; s < nslice; ++s) {
        a += part[((size_t)c * nslice + s) * 2 + 0];
        b += part[((size_t)c * nslice + s) * 2 + 1];
    }
    sums[c * 2 + 0] = (float)a;
    sums[c * 2 + 1] = (float)b;
    if (o0)
        o0[c] = (float)a;
    if (o1)
        o1[c] = (float)b;
    if (finalize) {
        const double m = (double)(float)a / count;
        double var = (double)(float)b / count - m * m;
        var = var > 0.0 ? var : 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
}

// mean / invstd from all-reduced sums over `count` elements (multi-rank path).
__global__ void k_bn_finalize(const float *__restrict__ sums, int C, double count, float eps,
                              float momentum, float *__restrict__ mean, float *__restrict__ invstd,
                              float *__restrict__ running_mean, float *__restrict__ running_var)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C)
        return;
    const double m = (double)sums[c * 2] / count;
    double var = (double)sums[c * 2 + 1] / count - m * m;
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// grid: ceil(row vectors / 1024) chunks x rows, linearised (see k_bn_apply)
// per-plane (image, channel) absmax side output (consumed by the f16x3 convolution to pick its power-of-two
// operand scale; one slot per plane keeps the atomics per address at HW / 1024):
// block maximum -> one integer atomicMax on the float bits (values are >= 0, so uint order == float order and
// the result does not depend on arrival order).  All threads of a block work on the same channel.
__device__ __forceinline__ void block_amax(float m, float *dst)
{
    __shared__ float wmax[BN_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0)
        wmax[threadIdx.x >> 6] = m;
    // raw barrier behind an LDS-only wait: __syncthreads() also waits for the wave's outstanding STORES (they count in
    // vmcnt on this target), i.e. every workgroup would sit on its CU slot until its output has been acknowledged
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < BN_THREADS / 64; ++w)
            m = fmaxf(m, wmax[w]);
        atomicMax((unsigned int *)dst, __float_as_uint(m));
    }
}

// y = x * sc + sh: ONE definition of the affine form, so that the backward's recomputed ReLU mask (y > 0 without
// reading y) sees bit-identical values
__device__ __forceinline__ float bn_shift(float beta, float mean, float sc) { return __builtin_fmaf(-mean, sc, beta); }
__device__ __forceinline__ void bn_affine(const float *invstd, const float *gamma, const float *beta,
                                          const float *mean, int c, float &sc, float &sh)
{
    sc = ldc(invstd + c) * (gamma ? ldc(gamma + c) : 1.f);
    sh = bn_shift(beta ? ldc(beta + c) : 0.f, ldc(mean + c), sc);
}
// (explicit fused operations: written as x * sc + sh the compiler is free to contract -- or not -- per call site, and it did decide
// differently for the forward and the backward kernels once packed FP32 was switched off: masks then disagree wherever y rounds
// to the other side of zero, 6e-3 of max in the input gradient of the layer-1 fixture)
__device__ __forceinline__ float bn_eval(float x, float sc, float sh) { return __builtin_fmaf(x, sc, sh); }

// "Fused" mode of the apply kernels: the per-slice partial sums `part` (all-reduced across ranks by the caller for
// SyncBatchNorm) are combined in every workgroup's prologue instead of by a k_bn_combine launch -- same fixed
// order, same double accumulation, two tiny launches less per norm layer and direction.
struct BnFused {
    const float *part;          // [C][ns][2]; NULL = not fused (mean / invstd / sums are inputs)
    const float *part_local;    // backward only: this rank's partial sums (dgamma / dbeta), may equal part
    int ns;
    double count;               // elements per channel over all ranks
    float eps, momentum;
    float *mean, *invstd;       // forward outputs (saved for the backward)
    float *running_mean, *running_var;
    long long *batches_tracked;
    float *dbeta, *dgamma;      // backward outputs (may be NULL)
    const float *pivot;         // forward: [C] shift of the partial sums (k_bn_stats), NULL = 0
    unsigned long long *mask_out;   // forward, RELU: packed y > 0 bits for the backward (see relu_mask_words), or NULL
    int xmask;                      // backward apply: the norm's INPUT is a ReLU output (conv -> ReLU -> norm, the projection heads:
                                    // reference models/Projector.py:46-51) -- dx is zeroed where x <= 0, i.e. the ReLU's backward
                                    // rides in this kernel (which reads x anyway) instead of a pass of its own
};

// Packed ReLU mask of an [N*C, HW] tensor with HW % 256 == 0: the 64 consecutive 16-byte vectors a wave handles give
// four 64-bit words, word k = the ballot of component k (bit = lane).  1/32 of the tensor's size: the backward of a
// norm + residual + ReLU reads it instead of y (whose values it needs only for the sign).
__device__ __forceinline__ size_t relu_mask_word(size_t plane, int hw4, int iv) { return (plane * (hw4 >> 6) + (iv >> 6)) * 4; }

template <bool AGENT = false>
__device__ __forceinline__ void part_sums(const float *part, int c, int ns, float &a, float &b)
{
    // eight slices per trip, their loads issued together (clamped: no branch around a load), added in slice order -- the same
    // double-precision sum as one load and one add per trip, without eight dependent memory round trips in the one thread every
    // workgroup waits for (on the small maps -- 12 x 192 x 32 x 64: one round of workgroups -- this prologue was a third of the kernel)
    double da = 0.0, db = 0.0;
    const float2 *p = (const float2 *)(part + (size_t)c * ns * 2);
    for (int s = 0; s < ns; s += 8) {
        float2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (AGENT || (DCL_BN_PROBE & 1)) {
                const float *q = (const float *)(p + min(s + i, ns - 1));
                v[i] = float2{ld_agent(q), ld_agent(q + 1)};
            } else {
                v[i] = p[min(s + i, ns - 1)];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (s + i < ns) {
                da += v[i].x;
                db += v[i].y;
            }
    }
    a = (float)da;
    b = (float)db;
}

template <bool RELU, bool RES>
__device__ __forceinline__ void bn_apply_body(const float *__restrict__ x,
                                              const float *__restrict__ res,
                                              const float *__restrict__ mean,
                                              const float *__restrict__ invstd,
                                              const float *__restrict__ gamma,
                                              const float *__restrict__ beta, int C, int HW,
                                              float *__restrict__ y, float *__restrict__ amax,
                                              const BnFused &f, int N, unsigned nchunk, const unsigned bid)
{
    // One grid row per CHANNEL; its N * HW elements (N segments of HW, C * HW apart) are one index space cut into
    // workgroup-sized pieces -- with one grid row per (image, channel) plane the small maps (384 channels, 16 x 32)
    // were 4 608 workgroups of half a vector per thread, bound by the workgroup dispatch rate (25 us for 19 MB).
    // That layout is used when a plane is smaller than a workgroup's share (`flat`); large planes keep one grid row per
    // plane (n0 fixed), which streams them in address order (the channel-major order cost the 48-channel, 128 x 256
    // maps 30 %).
    // 1-D grid: (chunk of the row, row) linearised with the chunk fastest -- a 2-D grid would cap the rows (N * C
    // planes) at 65 535
    const unsigned bx = bid % nchunk, by = bid / nchunk;
    const bool flat = N > 0;
    const int c = flat ? (int)by : (int)(by % (unsigned)C), n0 = flat ? 0 : (int)(by / (unsigned)C);
    const int hw4 = HW >> 2;
    const unsigned nv = flat ? (unsigned)N * (unsigned)hw4 : (unsigned)hw4;      // vectors in this grid row
    const unsigned j0 = bx * (BN_UNROLL * BN_THREADS) + threadIdx.x;
    float am = 0.f;
    float sc = 0.f, sh = 0.f;
    // The per-channel constants are computed by ONE thread and handed to the others through LDS after the tensor loads
    // have been issued: on the small planes (384 channels, 16 x 32: a single vector per thread) every thread running the
    // slice loop, the double-precision square root and the division by itself made this prologue the whole kernel
    // (25.7 us for 19 MB).  Same arithmetic, same order: bitwise the same statistics.
    __shared__ float bc[2];
    if (threadIdx.x == 0) {
    if (f.part) {
        // statistics of channel c from the partial sums (k_bn_combine's arithmetic), written out once per channel
        float a, b;
        part_sums(f.part, c, f.ns, a, b);
        const double ms = (double)a / f.count;                 // mean of the shifted values
        double var = (double)b / f.count - ms * ms;
        var = var > 0.0 ? var : 0.0;
        const double m = ms + (f.pivot ? (double)f.pivot[c] : 0.0);
        const float mean_f = (float)m, invstd_f = (float)(1.0 / sqrt(var + (double)f.eps));
        if (bx == 0 && n0 == 0) {
            f.mean[c] = mean_f;
            f.invstd[c] = invstd_f;
            if (f.running_mean) {
                const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
                f.running_mean[c] = (float)((1.0 - f.momentum) * f.running_mean[c] + f.momentum * m);
                f.running_var[c] = (float)((1.0 - f.momentum) * f.running_var[c] + f.momentum * unbiased);
            }
            if (c == 0 && f.batches_tracked)
                f.batches_tracked[0] += 1;
        }
        sc = invstd_f * (gamma ? gamma[c] : 1.f);
        sh = bn_shift(beta ? beta[c] : 0.f, mean_f, sc);
    } else {
        bn_affine(invstd, gamma, beta, mean, c, sc, sh);
    }
    bc[0] = sc;
    bc[1] = sh;
    }
    // BN_UNROLL 16-byte vectors per thread, BN_THREADS apart (coalesced), all loads issued before the first use
    if ((HW & 3) == 0) {
        f32x4 v[BN_UNROLL], r[BN_UNROLL];
        size_t off[BN_UNROLL];                                  // element offset of the vector, per u
        int pl[BN_UNROLL], ivv[BN_UNROLL];                      // its plane and vector index inside the plane
        // Loads are UNCONDITIONAL from clamped (always valid) vector indices and only the stores are predicated: with the
        // loads inside `if (j < nv)` branches the compiler lost count of the outstanding memory operations and put
        // s_waitcnt vmcnt(0) in front of EVERY store -- on this target stores count in vmcnt, so each of a thread's four
        // stores waited for the previous one to be acknowledged.
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            const unsigned j = min(j0 + u * BN_THREADS, nv - 1);
            const int n = flat ? (int)(j / (unsigned)hw4) : n0;
            ivv[u] = flat ? (int)(j - (unsigned)n * (unsigned)hw4) : (int)j;
            pl[u] = n * C + c;
            off[u] = (size_t)pl[u] * HW + 4 * (size_t)ivv[u];
            v[u] = *(const f32x4 *)(x + off[u]);
            if (RES)
                r[u] = *(const f32x4 *)(res + off[u]);
        }
        __syncthreads();
        sc = bc[0];
        sh = bc[1];
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            {
                f32x4 w = v[u];
                w.x = bn_eval(w.x, sc, sh); w.y = bn_eval(w.y, sc, sh); w.z = bn_eval(w.z, sc, sh); w.w = bn_eval(w.w, sc, sh);
                if (RES) {
                    w.x += r[u].x; w.y += r[u].y; w.z += r[u].z; w.w += r[u].w;
                }
                if (RELU) {
                    w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f);
                    if (f.mask_out && j0 + u * BN_THREADS < nv) {   // (wave-uniform: HW % 256 == 0)
                        const unsigned long long b0 = __ballot(w.x > 0.f), b1 = __ballot(w.y > 0.f),
                                                 b2 = __ballot(w.z > 0.f), b3 = __ballot(w.w > 0.f);
                        const int ln = threadIdx.x & 63;
                        if (ln < 4)
                            f.mask_out[relu_mask_word(pl[u], hw4, ivv[u]) + ln] = ln == 0 ? b0 : ln == 1 ? b1 : ln == 2 ? b2 : b3;
                    }
                }
                v[u] = w;
                if (j0 + u * BN_THREADS < nv)
                    am = fmaxf(am, fmaxf(fmaxf(fabsf(w.x), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w))));
            }
        }
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u)
            if (j0 + u * BN_THREADS < nv)
                *(f32x4 *)(y + off[u]) = v[u];
        // the absmax exchange behind the stores (its barrier waits for LDS only, see block_amax)
        if (amax)
            block_amax(am, amax + ((n0 * C + c) & (DCL_AMAX_SLOTS - 1)));
        return;
    } else {
        __syncthreads();
        sc = bc[0];
        sh = bc[1];
        const unsigned ne = flat ? (unsigned)N * (unsigned)HW : (unsigned)HW;          // elements in this grid row
        for (int u = 0; u < BN_UNROLL; ++u)
            for (unsigned e = (j0 + u * BN_THREADS) * 4; e < ne && e < (j0 + u * BN_THREADS) * 4 + 4; ++e) {
                const int n = flat ? (int)(e / (unsigned)HW) : n0;
                const size_t o = ((size_t)n * C + c) * HW + (size_t)(e - (flat ? (unsigned)n * (unsigned)HW : 0u));
                float v = bn_eval(x[o], sc, sh);
                if (RES)
                    v += res[o];
                if (RELU)
                    v = fmaxf(v, 0.f);
                y[o] = v;
                am = fmaxf(am, fabsf(v));
            }
    }
    if (amax)
        block_amax(am, amax + ((n0 * C + c) & (DCL_AMAX_SLOTS - 1)));
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(BN_THREADS) void k_bn_apply(const float *__restrict__ x,
                                                        const float *__restrict__ res,
                                                        const float *__restrict__ mean,
                                                        const float *__restrict__ invstd,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, int C, int HW,
                                                        float *__restrict__ y, float *__restrict__ amax,
                                                        BnFused f, int N, unsigned nchunk)
{
    bn_apply_body<RELU, RES>(x, res, mean, invstd, gamma, beta, C, HW, y, amax, f, N, nchunk, blockIdx.x);
}

// The statistics' finalisation WITHOUT the apply pass, for a consumer that forms relu(x sc + sh) itself while it stages x
// (dcl_conv3x3_pre.hip, the weight-gradient kernels' PRE forms): thread = channel.  mean / invstd / running statistics / sc / sh are
// the apply kernel's prologue, operation for operation (part_sums, the double-precision mean and variance, bn_shift), so that the
// consumer's fma reproduces the values k_bn_apply<true,false> would have written bit for bit, and so does the backward kernels'
// recomputed ReLU mask.  amax: max over this rank's values of relu(x sc + sh) per channel -- the map is monotone in x, so it is
// the map of one of the channel's two extrema -- max-ed into slot c % DCL_AMAX_SLOTS (integer atomicMax on non-negative floats:
// order-independent), the same side channel the apply kernel feeds.
struct BnPre {
    const float *gamma, *beta;
    double count;
    float eps, momentum;
    float *mean, *invstd, *running_mean, *running_var;
    long long *batches_tracked;
    float *sc, *sh, *amax;
};

// AGENT: the partial results were written by OTHER workgroups of the running launch (k_bn_stats_pre): agent-scope loads
template <bool AGENT>
__device__ __forceinline__ void bn_finalize_pre_channel(const float *part, const float *mm, int ns, int c, float pv, const BnPre &f)
{
    const double count = f.count;
    const float eps = f.eps, momentum = f.momentum;
    const float *gamma = f.gamma, *beta = f.beta;
    float *mean = f.mean, *invstd = f.invstd, *running_mean = f.running_mean, *running_var = f.running_var;
    long long *batches_tracked = f.batches_tracked;
    float *sc_out = f.sc, *sh_out = f.sh, *amax = f.amax;
    float a, b;
    part_sums<AGENT>(part, c, ns, a, b);
    const double ms = (double)a / count;
    double var = (double)b / count - ms * ms;
    var = var > 0.0 ? var : 0.0;
    const double m = ms + (double)pv;
    const float mean_f = (float)m, invstd_f = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = mean_f;
    invstd[c] = invstd_f;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
    if (c == 0 && batches_tracked)
        batches_tracked[0] += 1;
    const float sc = invstd_f * (gamma ? gamma[c] : 1.f);
    const float sh = bn_shift(beta ? beta[c] : 0.f, mean_f, sc);
    sc_out[c] = sc;
    sh_out[c] = sh;
    if (amax) {
        float lo = __builtin_inff(), hi = -__builtin_inff();
        for (int i0 = 0; i0 < ns; i0 += 8) {
            float l[8], h[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float *q = mm + ((size_t)c * ns + min(i0 + i, ns - 1)) * 2;
                l[i] = AGENT ? ld_agent(q) : q[0];
                h[i] = AGENT ? ld_agent(q + 1) : q[1];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                lo = fminf(lo, l[i]);
                hi = fmaxf(hi, h[i]);
            }
        }
        const float y = fmaxf(fmaxf(bn_eval(lo, sc, sh), bn_eval(hi, sc, sh)), 0.f);
        atomicMax((unsigned int *)(amax + (c & (DCL_AMAX_SLOTS - 1))), __float_as_uint(y));
    }
}

__global__ __launch_bounds__(64) void k_bn_finalize_pre(const float *__restrict__ part, const float *__restrict__ mm, int ns, int C,
                                                       const float *__restrict__ pivot, BnPre f)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C)
        return;
    bn_finalize_pre_channel<false>(part, mm, ns, c, pivot ? pivot[c] : 0.f, f);
}

// Statistics AND their finalisation in one launch (one rank: nothing to exchange in between): workgroup (c, s) publishes its four
// partial results with agent-scope stores, waits for them to be acknowledged, and takes a ticket of channel c (an agent-scope
// counter that starts at zero); the workgroup that draws the LAST ticket of the channel finalises it (bn_finalize_pre_channel
// on agent-scope loads: the other workgroups' results come from memory, not from this XCD's L2) -- same sums in the same order
// as k_bn_finalize_pre, bitwise the same outputs, without the ~5 us of a 64-thread launch between the statistics and the
// convolution that waits for the map.  The pivot (the running mean) is read by every workgroup of the channel before it takes
// its ticket, so the last one may update it.
__global__ __launch_bounds__(BN_THREADS) void k_bn_stats_pre(const float *__restrict__ x, int N, int C, int HW, int nslice,
                                                            float *__restrict__ part, float *__restrict__ mm,
                                                            const float *__restrict__ pivot_src, unsigned *__restrict__ tickets,
                                                            BnPre f)
{
    const int c = blockIdx.x, s = blockIdx.y;
    float pv;
    bn_stats_body<true, true>(x, N, C, HW, nslice, part, pivot_src, nullptr, c, s, mm, &pv);
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(tickets + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)nslice - 1u)
            bn_finalize_pre_channel<true>(part, mm, nslice, c, pv, f);
    }
}

// part[(c*nslice + s)*2 + {0,1}] = {sum g, sum g * xhat},  g = dy * (y > 0 if RELU)
template <bool RELU>
__device__ __forceinline__ void bn_bwd_reduce_body(const float *__restrict__ dy,
                                                   const float *__restrict__ x,
                                                   const float *__restrict__ y,
                                                   const float *__restrict__ mean,
                                                   const float *__restrict__ invstd,
                                                   const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, int N,
                                                   int C, int HW, int nslice,
                                                   float *__restrict__ part,
                                                   const unsigned long long *__restrict__ mask, const int c, const int s)
{
    __shared__ float sh[8];
    const float m = ldc(mean + c), is = ldc(invstd + c);
    // y == NULL (ReLU without residual): the mask y > 0 is recomputed from x -- one tensor less to read
    float asc, ash;
    bn_affine(invstd, gamma, beta, mean, c, asc, ash);
    const bool rec = RELU && y == nullptr && mask == nullptr;
    const int ln = threadIdx.x & 63;
    float a = 0.f, b = 0.f;
    const int hw4 = (HW & 3) ? 0 : (HW >> 2);      // see k_bn_stats
    for (int n = s; n < N; n += nslice) {
        const size_t base = ((size_t)n * C + c) * HW;
        const f32x4 *d4 = (const f32x4 *)(dy + base), *x4 = (const f32x4 *)(x + base),
                    *y4 = (const f32x4 *)(y + base);
        // one vector's contribution to (sum g, sum g (x - mean)); g = dy masked by y > 0 (y read, or recomputed)
        // packed mask: the sign bits of vector iv as +1 / 0 "y values"
        auto mask_y = [&](int iv) {
            const unsigned long long *mw = mask + relu_mask_word((size_t)n * C + c, hw4, iv);
            return f32x4{(float)((mw[0] >> ln) & 1), (float)((mw[1] >> ln) & 1), (float)((mw[2] >> ln) & 1),
                         (float)((mw[3] >> ln) & 1)};
        };
        auto accum = [&](f32x4 g, const f32x4 &xv, const f32x4 &ym) {
            if (RELU) {
                f32x4 yv = ym;
                if (rec) {
                    yv.x = bn_eval(xv.x, asc, ash); yv.y = bn_eval(xv.y, asc, ash);
                    yv.z = bn_eval(xv.z, asc, ash); yv.w = bn_eval(xv.w, asc, ash);
                }
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            }
            a += (g.x + g.y) + (g.z + g.w);
            b += (g.x * (xv.x - m) + g.y * (xv.y - m)) + (g.z * (xv.z - m) + g.w * (xv.w - m));
        };
        // two vectors per iteration, all their loads issued before the first is consumed
        int i = threadIdx.x;
        for (; i + BN_THREADS < hw4; i += 2 * BN_THREADS) {
            const f32x4 g0 = d4[i], x0 = x4[i], g1 = d4[i + BN_THREADS], x1 = x4[i + BN_THREADS];
            f32x4 y0 = g0, y1 = g1;
            if (RELU && mask) {
                y0 = mask_y(i);
                y1 = mask_y(i + BN_THREADS);
            } else if (RELU && !rec) {
                y0 = y4[i];
                y1 = y4[i + BN_THREADS];
            }
            accum(g0, x0, y0);
            accum(g1, x1, y1);
        }
        if (i < hw4)
            accum(d4[i], x4[i], (RELU && mask) ? mask_y(i) : (RELU && !rec) ? y4[i] : d4[i]);
        for (int i = (hw4 << 2) + threadIdx.x; i < HW; i += BN_THREADS) {
            float g = dy[base + i];
            if (RELU)
                g = (rec ? bn_eval(x[base + i], asc, ash) : y[base + i]) > 0.f ? g : 0.f;
            a += g;
            b += g * (x[base + i] - m);
        }
    }
    b *= is;
    block_reduce2(a, b, sh);
    if (threadIdx.x == 0) {
        part[((size_t)c * nslice + s) * 2 + 0] = a;
        part[((size_t)c * nslice + s) * 2 + 1] = b;
    }
}

template <bool RELU>
__global__ __launch_bounds__(BN_THREADS) void k_bn_bwd_reduce(const float *__restrict__ dy,
                                                             const float *__restrict__ x,
                                                             const float *__restrict__ y,
                                                             const float *__restrict__ mean,
                                                             const float *__restrict__ invstd,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, int N,
                                                             int C, int HW, int nslice,
                                                             float *__restrict__ part,
                                                             const unsigned long long *__restrict__ mask)
{
    bn_bwd_reduce_body<RELU>(dy, x, y, mean, invstd, gamma, beta, N, C, HW, nslice, part, mask, blockIdx.x, blockIdx.y);
}

// dx = gamma * invstd * (g - sum_g / count - xhat * sum_gx / count);  dres = g (optional)
template <bool RELU>
__device__ __forceinline__ void bn_bwd_apply_body(const float *__restrict__ dy,
                                                  const float *__restrict__ x,
                                                  const float *__restrict__ y,
                                                  const float *__restrict__ mean,
                                                  const float *__restrict__ invstd,
                                                  const float *__restrict__ gamma,
                                                  const float *__restrict__ beta,
                                                  const float *__restrict__ sums,
                                                  float inv_count, int C, int HW,
                                                  float *__restrict__ dx,
                                                  float *__restrict__ dres,
                                                  float *__restrict__ amax, const BnFused &f,
                                                  const unsigned long long *__restrict__ mask, int N,
                                                  unsigned nchunk, const unsigned bid)
{
    // 1-D grid: (chunk of the row, row) linearised with the chunk fastest -- a 2-D grid would cap the rows (N * C
    // planes) at 65 535
    const unsigned bx = bid % nchunk, by = bid / nchunk;
    const bool flat = N > 0;                            // one grid row per channel | per plane (see k_bn_apply)
    const int c = flat ? (int)by : (int)(by % (unsigned)C), n0 = flat ? 0 : (int)(by / (unsigned)C);
    const int hw4 = HW >> 2;
    const unsigned nv = flat ? (unsigned)N * (unsigned)hw4 : (unsigned)hw4;
    const unsigned j0 = bx * (BN_UNROLL * BN_THREADS) + threadIdx.x;
    float am = 0.f;
    const float m = ldc(mean + c), is = ldc(invstd + c);
    float asc, ash;                                   // y == NULL: ReLU mask recomputed from x (see the reduce)
    bn_affine(invstd, gamma, beta, mean, c, asc, ash);
    const bool rec = RELU && y == nullptr && mask == nullptr;
    const float k = is * (gamma ? ldc(gamma + c) : 1.f);
    float mg = 0.f, mgx = 0.f;
    __shared__ float bc[2];                 // (one thread computes, see k_bn_apply)
    if (threadIdx.x == 0) {
    if (f.part) {
        float a, b;
        part_sums(f.part, c, f.ns, a, b);
        mg = a * inv_count;
        mgx = b * inv_count;
        if (bx == 0 && n0 == 0 && (f.dbeta || f.dgamma)) {
            float la = a, lb = b;
            if (f.part_local != f.part)
                part_sums(f.part_local, c, f.ns, la, lb);       // this rank's sums: DDP averages the parameter grads
            if (f.dbeta)
                f.dbeta[c] = la;
            if (f.dgamma)
                f.dgamma[c] = lb;
        }
    } else {
        mg = sums[c * 2] * inv_count;
        mgx = sums[c * 2 + 1] * inv_count;
    }
    bc[0] = mg;
    bc[1] = mgx;
    }
    if ((HW & 3) == 0) {
        f32x4 gv[BN_UNROLL], xv[BN_UNROLL], yv[BN_UNROLL];
        size_t off[BN_UNROLL];
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            const unsigned j = min(j0 + u * BN_THREADS, nv - 1);     // unconditional loads from clamped indices (k_bn_apply)
            const int n = flat ? (int)(j / (unsigned)hw4) : n0;
            const int iv = flat ? (int)(j - (unsigned)n * (unsigned)hw4) : (int)j, plane = n * C + c;
            off[u] = (size_t)plane * HW + 4 * (size_t)iv;
            {
                gv[u] = *(const f32x4 *)(dy + off[u]);
                xv[u] = *(const f32x4 *)(x + off[u]);
                if (RELU && mask) {
                    const unsigned long long *mw = mask + relu_mask_word(plane, hw4, iv);
                    const int ln = threadIdx.x & 63;
                    yv[u] = f32x4{(float)((mw[0] >> ln) & 1), (float)((mw[1] >> ln) & 1), (float)((mw[2] >> ln) & 1),
                                  (float)((mw[3] >> ln) & 1)};
                } else if (RELU && !rec)
                    yv[u] = *(const f32x4 *)(y + off[u]);
            }
        }
        __syncthreads();
        mg = bc[0];
        mgx = bc[1];
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            {
                f32x4 g = gv[u];
                const f32x4 xx = xv[u];
                if (RELU) {
                    f32x4 yy;
                    if (rec) {
                        yy.x = bn_eval(xx.x, asc, ash); yy.y = bn_eval(xx.y, asc, ash);
                        yy.z = bn_eval(xx.z, asc, ash); yy.w = bn_eval(xx.w, asc, ash);
                    } else {
                        yy = yv[u];
                    }
                    g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
                    g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
                }
                f32x4 o;
                o.x = k * (g.x - mg - (xx.x - m) * is * mgx);
                o.y = k * (g.y - mg - (xx.y - m) * is * mgx);
                o.z = k * (g.z - mg - (xx.z - m) * is * mgx);
                o.w = k * (g.w - mg - (xx.w - m) * is * mgx);
                if (f.xmask) {
                    o.x = xx.x > 0.f ? o.x : 0.f;
                    o.y = xx.y > 0.f ? o.y : 0.f;
                    o.z = xx.z > 0.f ? o.z : 0.f;
                    o.w = xx.w > 0.f ? o.w : 0.f;
                }
                gv[u] = g;
                xv[u] = o;
                if (j0 + u * BN_THREADS < nv)
                    am = fmaxf(am, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
            }
        }
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u)
            if (j0 + u * BN_THREADS < nv) {
                if (dres)
                    *(f32x4 *)(dres + off[u]) = gv[u];
                *(f32x4 *)(dx + off[u]) = xv[u];
            }
        if (amax)                                   // behind the stores, LDS-only barrier: see k_bn_apply
            block_amax(am, amax + ((n0 * C + c) & (DCL_AMAX_SLOTS - 1)));
        return;
    } else {
        __syncthreads();
        mg = bc[0];
        mgx = bc[1];
        const unsigned ne = flat ? (unsigned)N * (unsigned)HW : (unsigned)HW;
        for (int u = 0; u < BN_UNROLL; ++u)
            for (unsigned e = (j0 + u * BN_THREADS) * 4; e < ne && e < (j0 + u * BN_THREADS) * 4 + 4; ++e) {
                const int n = flat ? (int)(e / (unsigned)HW) : n0;
                const size_t q = ((size_t)n * C + c) * HW + (size_t)(e - (flat ? (unsigned)n * (unsigned)HW : 0u));
                float g = dy[q];
                if (RELU)
                    g = (rec ? bn_eval(x[q], asc, ash) : y[q]) > 0.f ? g : 0.f;
                if (dres)
                    dres[q] = g;
                float o = k * (g - mg - (x[q] - m) * is * mgx);
                if (f.xmask)
                    o = x[q] > 0.f ? o : 0.f;
                dx[q] = o;
                am = fmaxf(am, fabsf(o));
            }
    }
    if (amax)
        block_amax(am, amax + ((n0 * C + c) & (DCL_AMAX_SLOTS - 1)));
}

template <bool RELU>
__global__ __launch_bounds__(BN_THREADS) void k_bn_bwd_apply(const float *__restrict__ dy,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ y,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            const float *__restrict__ sums,
                                                            float inv_count, int C, int HW,
                                                            float *__restrict__ dx,
                                                            float *__restrict__ dres,
                                                            float *__restrict__ amax, BnFused f,
                                                            const unsigned long long *__restrict__ mask, int N,
                                                            unsigned nchunk)
{
    bn_bwd_apply_body<RELU>(dy, x, y, mean, invstd, gamma, beta, sums, inv_count, C, HW, dx, dres, amax, f, mask, N, nchunk,
                            blockIdx.x);
}

// ---- the head's norm folded into its classifier (reference models/HRNet.py:596-600: conv3x3 -> BatchNorm -> conv1x1, NO activation
// between the norm and the 1x1 convolution) ------------------------------------------------------------------------------------
// logits = W (sc z + sh) = (W diag(sc)) z + W sh: the normalised 720-channel tensor (1.13 GB at batch 12 x 128 x 256) is never formed
// -- the forward is the statistics pass plus the classifier GEMM on z with rescaled weights (models/ops_head.py).  Backward: with
// dy = W^T dl the gradient of the norm's output (never formed either),
//     dz = sc dy - (sc / n) sum dy - (sc / n) xhat sum(dy xhat)  =  W'^T dl + c1 z + c0,     W' = W diag(sc),
// where the two channel sums follow from G = dl z^T (the classifier's weight-gradient product, [K, C]) and s = sum dl: c1 = -sc
// invstd dgamma / n, c0 = -sc dbeta / n - c1 mean.  This kernel forms dz in ONE pass (read z, read the K-channel dl, write dz) instead
// of the classifier's data-gradient GEMM (write dy), the norm's reduce (read dy, z) and its apply (read dy, z, write dz): K <= 32
// products per element from scalar-register weights, HBM-bound.  One thread = four consecutive pixels; a workgroup keeps its dl
// values in registers and walks a group of channels; max|dz| leaves through the absmax slots (one atomic per workgroup).
constexpr int HEAD_KMAX = 32;
template <int K4>       // K rounded up to a multiple of 4, in fours
__global__ __launch_bounds__(256) void k_head_norm_dz(const float *__restrict__ dl, const float *__restrict__ z,
                                                     const float *__restrict__ wt /* [C][4 K4]: W' transposed, zero padded */,
                                                     const float *__restrict__ c0, const float *__restrict__ c1, int K, int C,
                                                     int HW, unsigned total /* N * HW / 4 */, int cper,
                                                     float *__restrict__ dz, float *__restrict__ amax)
{
    constexpr int KP = 4 * K4;
    const unsigned hw4 = (unsigned)HW >> 2;
    const unsigned v = blockIdx.x * 256u + threadIdx.x;             // vector index over (image, pixel quad)
    const bool live = v < total;
    const unsigned vc = live ? v : total - 1;                       // tail threads: clamped loads, no stores
    const unsigned n = vc / hw4, q = vc - n * hw4;
    const size_t pl = (size_t)HW;
    f32x4 d[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k)
        d[k] = *(const f32x4 *)(dl + ((size_t)n * K + min(k, K - 1)) * pl + 4 * (size_t)q);     // (k >= K meets a zero weight)
    const int cg0 = blockIdx.y * cper, cg1 = min(C, cg0 + cper);
    float am = 0.f;
    const float *zp = z + (size_t)n * C * pl + 4 * (size_t)q;
    float *op = dz + (size_t)n * C * pl + 4 * (size_t)q;
    // four channels per trip, their z vectors requested together: one channel per trip exposed a memory round trip per channel
    // (2.1 us x 90 channels per workgroup: 0.47 ms for the benchmark's head against 0.36 at copy speed)
    constexpr int CU4 = 4;
    for (int c0i = cg0; c0i < cg1; c0i += CU4) {
        f32x4 zv[CU4];
#pragma unroll
        for (int u = 0; u < CU4; ++u)
            zv[u] = *(const f32x4 *)(zp + (size_t)min(c0i + u, cg1 - 1) * pl);
#pragma unroll
        for (int u = 0; u < CU4; ++u) {
            const int c = min(c0i + u, cg1 - 1);
            const float a0 = c0[c], a1 = c1[c];
            f32x4 acc{__builtin_fmaf(a1, zv[u].x, a0), __builtin_fmaf(a1, zv[u].y, a0), __builtin_fmaf(a1, zv[u].z, a0),
                      __builtin_fmaf(a1, zv[u].w, a0)};
            const float *w = wt + (size_t)c * KP;                   // wave-uniform: scalar loads
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const float wk = w[k];
                acc.x = __builtin_fmaf(wk, d[k].x, acc.x);
                acc.y = __builtin_fmaf(wk, d[k].y, acc.y);
                acc.z = __builtin_fmaf(wk, d[k].z, acc.z);
                acc.w = __builtin_fmaf(wk, d[k].w, acc.w);
            }
            if (live && c0i + u < cg1) {
                *(f32x4 *)(op + (size_t)c * pl) = acc;
                am = fmaxf(am, fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w))));
            }
        }
    }
    if (amax)
        block_amax(am, amax + (blockIdx.x & (DCL_AMAX_SLOTS - 1)));
}

int pick_slices(int N, int C)
{
    // enough workgroups to fill 256 CUs several times over, at most one slice per image
    int s = (2048 + C - 1) / C;
    if (s > N)
        s = N;
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" int dcl_bn_num_slices(int N, int C) { return pick_slices(N, C); }

// sums f32 [C, 2] = per-channel {sum x, sum x^2} over this rank's N*HW elements; part = workspace
// f32 [C * dcl_bn_num_slices(N, C) * 2].
extern "C" int dcl_bn_stats(const float *x, int N, int C, int HW, float *part, float *sums, void *stream)
{
    DCL_CHECK_ARG(x && part && sums && N > 0 && C > 0 && HW > 0, "bad arguments");
    const int ns = pick_slices(N, C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_stats, dim3(C, ns), dim3(BN_THREADS), 0, st, x, N, C, HW, ns, part);
    DCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 255) / 256), dim3(256), 0, st, part, C, ns, sums,
                       (float *)nullptr, (float *)nullptr, 0, 1.0, 0.f, 0.f, (float *)nullptr,
                       (float *)nullptr, (float *)nullptr, (float *)nullptr, (long long *)nullptr);
    DCL_LAUNCH_CHECK();
    return 0;
}

// Single-rank forward statistics: dcl_bn_stats + dcl_bn_finalize in two launches instead of three.
extern "C" int dcl_bn_stats_finalize(const float *x, int N, int C, int HW, float eps, float momentum,
                                     float *part, float *sums, float *mean, float *invstd,
                                     float *running_mean, float *running_var, int64_t *batches_tracked,
                                     void *stream)
{
    DCL_CHECK_ARG(x && part && sums && mean && invstd && N > 0 && C > 0 && HW > 0, "bad arguments");
    const int ns = pick_slices(N, C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_stats, dim3(C, ns), dim3(BN_THREADS), 0, st, x, N, C, HW, ns, part);
    DCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 255) / 256), dim3(256), 0, st, part, C, ns, sums,
                       (float *)nullptr, (float *)nullptr, 1, (double)N * HW, eps, momentum, mean, invstd,
                       running_mean, running_var, (long long *)batches_tracked);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_bn_finalize(const float *sums, int C, double count, float eps, float momentum,
                               float *mean, float *invstd, float *running_mean, float *running_var,
                               void *stream)
{
    DCL_CHECK_ARG(sums && mean && invstd && C > 0 && count > 0, "bad arguments");
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C,
                       count, eps, momentum, mean, invstd, running_mean, running_var);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_bn_apply(const float *x, const float *res, const float *mean, const float *invstd,
                            const float *gamma, const float *beta, int N, int C, int HW, int relu,
                            float *y, float *amax, void *stream)
{
    DCL_CHECK_ARG(x && mean && invstd && y && N > 0 && C > 0 && HW > 0, "bad arguments");
    DCL_CHECK_ARG((long long)N * HW < (1LL << 31) - 4 * BN_THREADS * BN_UNROLL, "N * H * W must stay below 2^31 per channel");
    // planes smaller than a workgroup's share (4 vectors per thread): one grid row per channel over all N images
    // (kernel argument N > 0); else one grid row per plane (N = 0)
    const bool flat = (HW + 3) / 4 < BN_THREADS * BN_UNROLL;
    const long long nvs = flat ? ((long long)N * HW + 3) / 4 : (HW + 3) / 4;
    const unsigned nchunk = (unsigned)((nvs + BN_THREADS * BN_UNROLL - 1) / (BN_THREADS * BN_UNROLL));
    DCL_CHECK_ARG((long long)nchunk * (flat ? C : (long long)N * C) < (1LL << 31), "tensor too large for one launch");
    dim3 grid(nchunk * (unsigned)(flat ? C : N * C));
    const int Nk = flat ? N : 0;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(R, S) hipLaunchKernelGGL((k_bn_apply<R, S>), grid, dim3(BN_THREADS), 0, st, x, res, mean, invstd, gamma, beta, C, HW, y, amax, BnFused{}, Nk, nchunk)
    if (relu && res) LAUNCH(true, true);
    else if (relu) LAUNCH(true, false);
    else if (res) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_bn_bwd_reduce(const float *dy, const float *x, const float *y, const float *mean,
                                 const float *invstd, const float *gamma, const float *beta, int N, int C,
                                 int HW, int relu, float *part, float *sums, float *dbeta, float *dgamma,
                                 void *stream)
{
    DCL_CHECK_ARG(dy && x && mean && invstd && part && sums, "bad arguments");
    const int ns = pick_slices(N, C);
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long *mask = nullptr;
    if (relu)
        hipLaunchKernelGGL((k_bn_bwd_reduce<true>), dim3(C, ns), dim3(BN_THREADS), 0, st, dy, x, y, mean, invstd, gamma, beta, N, C, HW, ns, part, mask);
    else
        hipLaunchKernelGGL((k_bn_bwd_reduce<false>), dim3(C, ns), dim3(BN_THREADS), 0, st, dy, x, y, mean, invstd, gamma, beta, N, C, HW, ns, part, mask);
    DCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 255) / 256), dim3(256), 0, st, part, C, ns, sums, dbeta,
                       dgamma, 0, 1.0, 0.f, 0.f, (float *)nullptr, (float *)nullptr, (float *)nullptr,
                       (float *)nullptr, (long long *)nullptr);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_bn_bwd_apply(const float *dy, const float *x, const float *y, const float *mean,
                                const float *invstd, const float *gamma, const float *beta, const float *sums,
                                double count, int N, int C, int HW, int relu, float *dx, float *dres,
                                float *amax, void *stream)
{
    DCL_CHECK_ARG(dy && x && mean && invstd && sums && dx && count > 0, "bad arguments");
    DCL_CHECK_ARG((long long)N * HW < (1LL << 31) - 4 * BN_THREADS * BN_UNROLL, "N * H * W must stay below 2^31 per channel");
    // planes smaller than a workgroup's share (4 vectors per thread): one grid row per channel over all N images
    // (kernel argument N > 0); else one grid row per plane (N = 0)
    const bool flat = (HW + 3) / 4 < BN_THREADS * BN_UNROLL;
    const long long nvs = flat ? ((long long)N * HW + 3) / 4 : (HW + 3) / 4;
    const unsigned nchunk = (unsigned)((nvs + BN_THREADS * BN_UNROLL - 1) / (BN_THREADS * BN_UNROLL));
    DCL_CHECK_ARG((long long)nchunk * (flat ? C : (long long)N * C) < (1LL << 31), "tensor too large for one launch");
    dim3 grid(nchunk * (unsigned)(flat ? C : N * C));
    const int Nk = flat ? N : 0;
    hipStream_t st = (hipStream_t)stream;
    const float inv = (float)(1.0 / count);
    if (relu)
        hipLaunchKernelGGL((k_bn_bwd_apply<true>), grid, dim3(BN_THREADS), 0, st, dy, x, y, mean, invstd, gamma, beta, sums, inv, C, HW, dx, dres, amax, BnFused{}, (const unsigned long long *)nullptr, Nk, nchunk);
    else
        hipLaunchKernelGGL((k_bn_bwd_apply<false>), grid, dim3(BN_THREADS), 0, st, dy, x, y, mean, invstd, gamma, beta, sums, inv, C, HW, dx, dres, amax, BnFused{}, (const unsigned long long *)nullptr, Nk, nchunk);
    DCL_LAUNCH_CHECK();
    return 0;
}

// ---- fused forms: no k_bn_combine launches (see BnFused) -------------------------------------------------------------

extern "C" int dcl_bn_stats_part(const float *x, int N, int C, int HW, float *part, const float *pivot_src,
                                 float *pivot_out, void *stream)
{
    DCL_CHECK_ARG(x && part && N > 0 && C > 0 && HW > 0, "bad arguments");
    DCL_CHECK_ARG((pivot_src == nullptr) == (pivot_out == nullptr), "pivot_src and pivot_out go together");
    const int ns = pick_slices(N, C);
    hipLaunchKernelGGL(k_bn_stats, dim3(C, ns), dim3(BN_THREADS), 0, (hipStream_t)stream, x, N, C, HW, ns, part, pivot_src,
                       pivot_out);
    DCL_LAUNCH_CHECK();
    return 0;
}

// Statistics pass for a norm whose normalised output is NOT written (its consumer applies the map while staging the input):
// dcl_bn_stats_part plus the per-slice extrema of x, mm f32 [C * dcl_bn_num_slices(N, C) * 2] = {min, max}.
extern "C" int dcl_bn_stats_minmax_part(const float *x, int N, int C, int HW, float *part, float *mm, const float *pivot_src,
                                        float *pivot_out, void *stream)
{
    DCL_CHECK_ARG(x && part && mm && N > 0 && C > 0 && HW > 0, "bad arguments");
    DCL_CHECK_ARG((pivot_src == nullptr) == (pivot_out == nullptr), "pivot_src and pivot_out go together");
    const int ns = pick_slices(N, C);
    hipLaunchKernelGGL(k_bn_stats_mm, dim3(C, ns), dim3(BN_THREADS), 0, (hipStream_t)stream, x, N, C, HW, ns, part, pivot_src,
                       pivot_out, mm);
    DCL_LAUNCH_CHECK();
    return 0;
}

// ... and its finalisation: what dcl_bn_apply_parts does apart from writing y (see k_bn_finalize_pre).
extern "C" int dcl_bn_finalize_pre(const float *part, const float *mm, int ns, double count, float eps, float momentum,
                                   const float *gamma, const float *beta, int C, float *mean, float *invstd,
                                   float *running_mean, float *running_var, int64_t *batches_tracked, const float *pivot,
                                   float *pre_sc, float *pre_sh, float *amax, void *stream)
{
    DCL_CHECK_ARG(part && mean && invstd && pre_sc && pre_sh && C > 0 && ns >= 1 && ns <= 64 && count > 0.0, "bad arguments");
    DCL_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var go together");
    DCL_CHECK_ARG(amax == nullptr || mm != nullptr, "amax needs the extrema");
    const BnPre f{gamma, beta, count, eps, momentum, mean, invstd, running_mean, running_var, (long long *)batches_tracked,
                  pre_sc, pre_sh, amax};
    hipLaunchKernelGGL(k_bn_finalize_pre, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, part, mm, ns, C, pivot, f);
    DCL_LAUNCH_CHECK();
    return 0;
}

// dcl_bn_stats_minmax_part + dcl_bn_finalize_pre in ONE launch, for one rank (no exchange between the two): see k_bn_stats_pre.
// tickets: C zero-initialised 32-bit words (left at dcl_bn_num_slices(N, C) each); the pivot of the sums is running_mean (NULL: 0).
extern "C" int dcl_bn_stats_pre(const float *x, int N, int C, int HW, float *part, float *mm, void *tickets, double count, float eps,
                                float momentum, const float *gamma, const float *beta, float *mean, float *invstd,
                                float *running_mean, float *running_var, int64_t *batches_tracked, float *pre_sc, float *pre_sh,
                                float *amax, void *stream)
{
    DCL_CHECK_ARG(x && part && mm && tickets && mean && invstd && pre_sc && pre_sh && N > 0 && C > 0 && HW > 0 && count > 0.0,
                  "bad arguments");
    DCL_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var go together");
    const int ns = pick_slices(N, C);
    const BnPre f{gamma, beta, count, eps, momentum, mean, invstd, running_mean, running_var, (long long *)batches_tracked,
                  pre_sc, pre_sh, amax};
    hipLaunchKernelGGL(k_bn_stats_pre, dim3(C, ns), dim3(BN_THREADS), 0, (hipStream_t)stream, x, N, C, HW, ns, part, mm,
                       running_mean, (unsigned *)tickets, f);
    DCL_LAUNCH_CHECK();
    return 0;
}

static int bn_apply_impl(const float *x, const float *res, const float *part, int ns, double count, float eps,
                         float momentum, const float *gamma, const float *beta, int N, int C, int HW,
                         int relu, float *y, float *mean, float *invstd, float *running_mean,
                         float *running_var, int64_t *batches_tracked, float *amax, const float *pivot,
                         void *relu_mask, void *stream);

extern "C" int dcl_bn_apply_fused(const float *x, const float *res, const float *part, double count, float eps,
                                  float momentum, const float *gamma, const float *beta, int N, int C, int HW,
                                  int relu, float *y, float *mean, float *invstd, float *running_mean,
                                  float *running_var, int64_t *batches_tracked, float *amax, const float *pivot,
                                  void *relu_mask, void *stream)
{
    return bn_apply_impl(x, res, part, pick_slices(N, C), count, eps, momentum, gamma, beta, N, C, HW, relu, y, mean, invstd,
                         running_mean, running_var, batches_tracked, amax, pivot, relu_mask, stream);
}

// The same with an explicit number of partial sums per channel: part f32 [C][ns][2] from any producer (ns <= 64: a serial
// fixed-order sum per workgroup).
extern "C" int dcl_bn_apply_parts(const float *x, const float *res, const float *part, int ns, double count, float eps,
                                  float momentum, const float *gamma, const float *beta, int N, int C, int HW,
                                  int relu, float *y, float *mean, float *invstd, float *running_mean,
                                  float *running_var, int64_t *batches_tracked, float *amax, const float *pivot,
                                  void *relu_mask, void *stream)
{
    DCL_CHECK_ARG(ns > 0 && ns <= 64, "1 .. 64 partial sums per channel");
    return bn_apply_impl(x, res, part, ns, count, eps, momentum, gamma, beta, N, C, HW, relu, y, mean, invstd,
                         running_mean, running_var, batches_tracked, amax, pivot, relu_mask, stream);
}

static int bn_apply_impl(const float *x, const float *res, const float *part, int ns, double count, float eps,
                         float momentum, const float *gamma, const float *beta, int N, int C, int HW,
                         int relu, float *y, float *mean, float *invstd, float *running_mean,
                         float *running_var, int64_t *batches_tracked, float *amax, const float *pivot,
                         void *relu_mask, void *stream)
{
    DCL_CHECK_ARG(x && part && y && mean && invstd && N > 0 && C > 0 && HW > 0 && count > 0, "bad arguments");
    DCL_CHECK_ARG(!relu_mask || (relu && HW % 256 == 0), "relu_mask: ReLU and HW % 256 == 0 only");
    BnFused f{};
    f.part = part;
    f.pivot = pivot;
    f.mask_out = (unsigned long long *)relu_mask;
    f.ns = ns;
    f.count = count;
    f.eps = eps;
    f.momentum = momentum;
    f.mean = mean;
    f.invstd = invstd;
    f.running_mean = running_mean;
    f.running_var = running_var;
    f.batches_tracked = (long long *)batches_tracked;
    DCL_CHECK_ARG((long long)N * HW < (1LL << 31) - 4 * BN_THREADS * BN_UNROLL, "N * H * W must stay below 2^31 per channel");
    // planes smaller than a workgroup's share (4 vectors per thread): one grid row per channel over all N images
    // (kernel argument N > 0); else one grid row per plane (N = 0)
    const bool flat = (HW + 3) / 4 < BN_THREADS * BN_UNROLL;
    const long long nvs = flat ? ((long long)N * HW + 3) / 4 : (HW + 3) / 4;
    const unsigned nchunk = (unsigned)((nvs + BN_THREADS * BN_UNROLL - 1) / (BN_THREADS * BN_UNROLL));
    DCL_CHECK_ARG((long long)nchunk * (flat ? C : (long long)N * C) < (1LL << 31), "tensor too large for one launch");
    dim3 grid(nchunk * (unsigned)(flat ? C : N * C));
    const int Nk = flat ? N : 0;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(R, S) hipLaunchKernelGGL((k_bn_apply<R, S>), grid, dim3(BN_THREADS), 0, st, x, res, (const float *)nullptr, (const float *)nullptr, gamma, beta, C, HW, y, amax, f, Nk, nchunk)
    if (relu && res) LAUNCH(true, true);
    else if (relu) LAUNCH(true, false);
    else if (res) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_bn_bwd_reduce_part(const float *dy, const float *x, const float *y, const float *mean,
                                      const float *invstd, const float *gamma, const float *beta, int N, int C,
                                      int HW, int relu, float *part, void *stream)
{
    DCL_CHECK_ARG(dy && x && mean && invstd && part, "bad arguments");
    DCL_CHECK_ARG(relu != 2 || (y && HW % 256 == 0), "relu = 2: y is the packed mask, HW % 256 == 0");
    const unsigned long long *mask = relu == 2 ? (const unsigned long long *)y : nullptr;
    if (mask)
        y = nullptr;
    const int ns = pick_slices(N, C);
    hipStream_t st = (hipStream_t)stream;
    if (relu)
        hipLaunchKernelGGL((k_bn_bwd_reduce<true>), dim3(C, ns), dim3(BN_THREADS), 0, st, dy, x, y, mean, invstd, gamma, beta, N, C, HW, ns, part, mask);
    else
        hipLaunchKernelGGL((k_bn_bwd_reduce<false>), dim3(C, ns), dim3(BN_THREADS), 0, st, dy, x, y, mean, invstd, gamma, beta, N, C, HW, ns, part, mask);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_bn_bwd_apply_fused(const float *dy, const float *x, const float *y, const float *mean,
                                      const float *invstd, const float *gamma, const float *beta,
                                      const float *part, const float *part_local, double count, int N, int C,
                                      int HW, int relu, float *dx, float *dres, float *dbeta, float *dgamma,
                                      float *amax, void *stream)
{
    DCL_CHECK_ARG(dy && x && mean && invstd && part && part_local && dx && count > 0, "bad arguments");
    const int xmask = (relu >> 2) & 1;          // relu + 4: the norm's input is a ReLU output, dx = 0 where x <= 0 (see BnFused)
    relu &= 3;
    DCL_CHECK_ARG(relu != 2 || (y && HW % 256 == 0), "relu = 2: y is the packed mask, HW % 256 == 0");
    const unsigned long long *mask = relu == 2 ? (const unsigned long long *)y : nullptr;
    if (mask)
        y = nullptr;
    BnFused f{};
    f.xmask = xmask;
    f.part = part;
    f.part_local = part_local;
    f.ns = pick_slices(N, C);
    f.count = count;
    f.dbeta = dbeta;
    f.dgamma = dgamma;
    DCL_CHECK_ARG((long long)N * HW < (1LL << 31) - 4 * BN_THREADS * BN_UNROLL, "N * H * W must stay below 2^31 per channel");
    // planes smaller than a workgroup's share (4 vectors per thread): one grid row per channel over all N images
    // (kernel argument N > 0); else one grid row per plane (N = 0)
    const bool flat = (HW + 3) / 4 < BN_THREADS * BN_UNROLL;
    const long long nvs = flat ? ((long long)N * HW + 3) / 4 : (HW + 3) / 4;
    const unsigned nchunk = (unsigned)((nvs + BN_THREADS * BN_UNROLL - 1) / (BN_THREADS * BN_UNROLL));
    DCL_CHECK_ARG((long long)nchunk * (flat ? C : (long long)N * C) < (1LL << 31), "tensor too large for one launch");
    dim3 grid(nchunk * (unsigned)(flat ? C : N * C));
    const int Nk = flat ? N : 0;
    hipStream_t st = (hipStream_t)stream;
    const float inv = (float)(1.0 / count);
    if (relu)
        hipLaunchKernelGGL((k_bn_bwd_apply<true>), grid, dim3(BN_THREADS), 0, st, dy, x, y, mean, invstd, gamma, beta, (const float *)nullptr, inv, C, HW, dx, dres, amax, f, mask, Nk, nchunk);
    else
        hipLaunchKernelGGL((k_bn_bwd_apply<false>), grid, dim3(BN_THREADS), 0, st, dy, x, y, mean, invstd, gamma, beta, (const float *)nullptr, inv, C, HW, dx, dres, amax, f, mask, Nk, nchunk);
    DCL_LAUNCH_CHECK();
    return 0;
}

// dz = W'^T dl + c1 z + c0 (see k_head_norm_dz): dl [N, K, HW] (K <= 32), z / dz [N, C, HW], wt [C][Kp] = W' transposed with the K
// index padded with zeros to Kp = 4 ceil(K / 4), c0 / c1 [C]; HW % 4 == 0, 16-byte aligned tensors.  amax: DCL_AMAX_SLOTS
// zero-initialised slots (max|dz|) or NULL.
extern "C" int dcl_head_norm_dz(const float *dl, const float *z, const float *wt, const float *c0, const float *c1, int N, int K,
                                int C, int HW, float *dz, float *amax, void *stream)
{
    DCL_CHECK_ARG(dl && z && wt && c0 && c1 && dz, "null pointer");
    DCL_CHECK_ARG(N > 0 && C > 0 && HW > 0 && K >= 1 && K <= HEAD_KMAX, "bad shape (1 <= K <= 32)");
    DCL_CHECK_ARG((HW & 3) == 0, "HW must be a multiple of 4");
    DCL_CHECK_ARG((((uintptr_t)dl | (uintptr_t)z | (uintptr_t)dz) & 15) == 0, "tensors must be 16-byte aligned");
    DCL_CHECK_ARG((size_t)N * (HW / 4) < ((size_t)1 << 31), "too many pixels");
    const unsigned total = (unsigned)((size_t)N * (HW / 4));
    const unsigned gx = (total + 255u) / 256u;
    // enough workgroups for a few rounds of the chip: channel groups when the pixel blocks alone are few
    int groups = 1;
    while (groups < 16 && (size_t)gx * groups < 2048 && C / (groups * 2) >= 8)
        groups *= 2;
    const int cper = (C + groups - 1) / groups;
    const dim3 grid(gx, (unsigned)((C + cper - 1) / cper));
    const int K4 = (K + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
#define DCL_HEAD_CASE(k4)                                                                                                   \
    if (K4 == k4)                                                                                                           \
        hipLaunchKernelGGL((k_head_norm_dz<k4>), grid, dim3(256), 0, st, dl, z, wt, c0, c1, K, C, HW, total, cper, dz, amax);
    DCL_HEAD_CASE(1) DCL_HEAD_CASE(2) DCL_HEAD_CASE(3) DCL_HEAD_CASE(4) DCL_HEAD_CASE(5) DCL_HEAD_CASE(6) DCL_HEAD_CASE(7) DCL_HEAD_CASE(8)
#undef DCL_HEAD_CASE
    dcl_note_kernel("k_head_norm_dz<%d>", K4);
    DCL_LAUNCH_CHECK();
    return 0;
}
