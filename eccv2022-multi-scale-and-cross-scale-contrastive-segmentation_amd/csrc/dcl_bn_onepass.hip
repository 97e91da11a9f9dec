// dcl_bn_onepass.hip -- batch-norm backward (and forward) that reads its inputs ONCE.
//
// The two-kernel forms (dcl_bn.hip: reduce, then apply) run at the HBM rate but move every input twice: the per-channel
// sums over all N x H x W values have to be complete before the first output can be written.  Here 256 persistent
// workgroups (one per CU) keep their share of a channel IN REGISTERS between the two phases:
//   team      the H W / 1024 workgroups of one XCD that share a channel (a member owns 1024 consecutive pixels of every image:
//             one float4 per thread and image); 32 / T teams per XCD work on different channels;
//   phase 1   load dy, x (and the packed ReLU mask) of the channel, partial sums, block reduction, the member's two partial
//             sums to memory, one relaxed agent-scope atomic add on the channel's counter;
//   overlap   the NEXT channel's loads are issued into a second register set before the wait;
//   phase 2   wait until the counter shows T arrivals, add the T partials in member order (double: the same fixed-order sum as
//             part_sums), dx (and the residual gradient) from the registers.
// HBM traffic: dy + x + dx (+ dres) instead of 2 dy + 2 x + dx: 42 us against 68 us on 12 x 48 x 128 x 256
// (tools/probes/l2_reread.py is the model this was sized on).
// The team barrier needs all members resident: the grid is 256 workgroups of 256 threads, and the host side only takes this
// path on the device's default stream of a single-rank run, so that never more than ONE such kernel is in flight (two
// persistent kernels that each hold part of the CUs would wait for each other forever).  A spin that exceeds ~2^27 polls traps.
// Counters live in a ring of four regions; every launch zeroes the region two launches ahead.
#include "dcl_common.h"

namespace {

constexpr int OP_THREADS = 256;
constexpr int OP_NMAX = 12;             // images per channel a thread can hold (two register sets of 2 x 12 float4)
constexpr int OP_TMAX = 32;

struct OpArgs {
    const float *dy, *x;
    const unsigned long long *mask;     // packed ReLU mask (relu == 2) or null
    const float *mean, *invstd, *gamma, *beta;
    float *dx, *dres, *dbeta, *dgamma, *amax;
    int *ctr, *ctr_clear;
    float *part;                        // [C][T][2]
    int N, C, HW, T;
    float inv_count;
    int relu;                           // 0 none, 1 mask recomputed from x, 2 packed mask
};

__device__ __forceinline__ size_t mask_word(size_t plane, int hw4, int iv) { return (plane * (hw4 >> 6) + (iv >> 6)) * 4; }

struct OpSet {
    f32x4 g[OP_NMAX], x[OP_NMAX];
    unsigned mbits[OP_NMAX];            // relu == 2: the four sign bits of this lane's vector
};

__global__ __launch_bounds__(OP_THREADS) void k_bn_bwd_onepass(OpArgs a)
{
    __shared__ float sh[8];
    __shared__ float bc[2];
    const int tid = threadIdx.x, ln = tid & 63;
    for (int i = blockIdx.x * OP_THREADS + tid; i < a.C; i += gridDim.x * OP_THREADS)
        a.ctr_clear[i] = 0;
    const int xcd = blockIdx.x & 7, mi = blockIdx.x >> 3;          // 32 workgroups per XCD
    const int T = a.T, tpx = OP_TMAX / T;
    const int tq = mi / T, m = mi - tq * T;
    const int hw4 = a.HW >> 2;
    const int iv = m * OP_THREADS + tid;                            // this thread's vector inside every plane
    const int N = a.N;
    float am = 0.f;

    auto load = [&](OpSet &s, int c) {
#pragma unroll
        for (int n = 0; n < OP_NMAX; ++n) {
            const int nn = min(n, N - 1);                           // unconditional loads from clamped planes
            const size_t plane = (size_t)nn * a.C + c;
            const size_t off = plane * a.HW + 4 * (size_t)iv;
            s.g[n] = *(const f32x4 *)(a.dy + off);
            s.x[n] = *(const f32x4 *)(a.x + off);
            if (a.relu == 2) {
                const unsigned long long *mw = a.mask + mask_word(plane, hw4, iv);
                s.mbits[n] = (unsigned)((mw[0] >> ln) & 1) | ((unsigned)((mw[1] >> ln) & 1) << 1) |
                             ((unsigned)((mw[2] >> ln) & 1) << 2) | ((unsigned)((mw[3] >> ln) & 1) << 3);
            }
        }
    };
    // channels of this team: c = xcd + 8 (tq + tpx r)
    auto chan = [&](int r) { return xcd + 8 * (tq + tpx * r); };
    auto one = [&](OpSet &cur, OpSet &nxt, int r) {
        const int c = chan(r);
        const float mu = a.mean[c], is = a.invstd[c];
        const float gm = a.gamma ? a.gamma[c] : 1.f;
        const float asc = is * gm, ash = (a.beta ? a.beta[c] : 0.f) - mu * asc;
        // ---- phase 1: mask the gradient, partial sums
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int n = 0; n < OP_NMAX; ++n) {
            if (n < N) {
                f32x4 g = cur.g[n];
                const f32x4 xv = cur.x[n];
                if (a.relu == 2) {
                    const unsigned b = cur.mbits[n];
                    g.x = (b & 1) ? g.x : 0.f; g.y = (b & 2) ? g.y : 0.f; g.z = (b & 4) ? g.z : 0.f; g.w = (b & 8) ? g.w : 0.f;
                } else if (a.relu == 1) {
                    g.x = xv.x * asc + ash > 0.f ? g.x : 0.f; g.y = xv.y * asc + ash > 0.f ? g.y : 0.f;
                    g.z = xv.z * asc + ash > 0.f ? g.z : 0.f; g.w = xv.w * asc + ash > 0.f ? g.w : 0.f;
                }
                cur.g[n] = g;
                sa += (g.x + g.y) + (g.z + g.w);
                sb += (g.x * (xv.x - mu) + g.y * (xv.y - mu)) + (g.z * (xv.z - mu) + g.w * (xv.w - mu));
            }
        }
        sb *= is;
        sa = wave_sum(sa);
        sb = wave_sum(sb);
        if (ln == 0) {
            sh[tid >> 6] = sa;
            sh[4 + (tid >> 6)] = sb;
        }
        __syncthreads();
        if (tid == 0) {
            const float pa = (sh[0] + sh[1]) + (sh[2] + sh[3]), pb = (sh[4] + sh[5]) + (sh[6] + sh[7]);
            float *pp = a.part + ((size_t)c * T + m) * 2;
            __hip_atomic_store(pp, pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 1, pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the partials have reached memory before the arrival
            __hip_atomic_fetch_add(a.ctr + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- the next channel's loads run under the wait
        if (chan(r + 1) < a.C)
            load(nxt, chan(r + 1));
        if (tid == 0) {
            unsigned polls = 0;
            while (__hip_atomic_load(a.ctr + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T) {
                if (++polls > (1u << 27))
                    __builtin_trap();
            }
            double da = 0.0, db = 0.0;
            for (int t = 0; t < T; ++t) {
                da += (double)__hip_atomic_load(a.part + ((size_t)c * T + t) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                db += (double)__hip_atomic_load(a.part + ((size_t)c * T + t) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bc[0] = (float)da;
            bc[1] = (float)db;
            if (m == 0) {
                if (a.dbeta)
                    a.dbeta[c] = (float)da;
                if (a.dgamma)
                    a.dgamma[c] = (float)db;
            }
        }
        __syncthreads();
        const float mg = bc[0] * a.inv_count, mgx = bc[1] * a.inv_count;
        const float k = is * gm;
        // ---- phase 2: dx (and the residual's gradient) from the registers
#pragma unroll
        for (int n = 0; n < OP_NMAX; ++n) {
            if (n < N) {
                const f32x4 g = cur.g[n], xv = cur.x[n];
                f32x4 o;
                o.x = k * (g.x - mg - (xv.x - mu) * is * mgx);
                o.y = k * (g.y - mg - (xv.y - mu) * is * mgx);
                o.z = k * (g.z - mg - (xv.z - mu) * is * mgx);
                o.w = k * (g.w - mg - (xv.w - mu) * is * mgx);
                const size_t off = ((size_t)n * a.C + c) * a.HW + 4 * (size_t)iv;
                if (a.dres)
                    *(f32x4 *)(a.dres + off) = g;
                *(f32x4 *)(a.dx + off) = o;
                am = fmaxf(am, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
            }
        }
    };

    OpSet s0, s1;
    if (chan(0) < a.C)
        load(s0, chan(0));
    for (int r = 0; chan(r) < a.C; r += 2) {
        one(s0, s1, r);
        if (chan(r + 1) < a.C)
            one(s1, s0, r + 1);
    }
    if (a.amax) {
        am = fmaxf(am, 0.f);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            am = fmaxf(am, __shfl_xor(am, o, 64));
        if (ln == 0)
            atomicMax((int *)a.amax + (blockIdx.x & (DCL_AMAX_SLOTS - 1)), __float_as_int(am));
    }
}

}  // namespace

extern "C" int dcl_bn_bwd_onepass_supported(int N, int C, int HW, int relu)
{
    if (N < 1 || N > OP_NMAX || C < 8 || C > 4096 || HW % 1024 || relu < 0 || relu > 2)
        return 0;
    const int T = HW / 1024;
    return T >= 1 && T <= OP_TMAX && (OP_TMAX % T) == 0;
}

extern "C" int64_t dcl_bn_onepass_workspace_bytes(void)
{
    return 4 * (4096 * (int64_t)sizeof(int)) + (int64_t)4096 * OP_TMAX * 2 * sizeof(float);
}

// ws: dcl_bn_onepass_workspace_bytes() bytes, zero-initialised ONCE by the caller; seq: launch counter of that workspace
// (0, 1, 2, ...: selects the counter region).  relu: 0 none, 1 the mask is recomputed from x, 2 `y` is the packed mask.
extern "C" int dcl_bn_bwd_onepass(const float *dy, const float *x, const void *y_or_mask, const float *mean,
                                  const float *invstd, const float *gamma, const float *beta, double count, int N, int C,
                                  int HW, int relu, float *dx, float *dres, float *dbeta, float *dgamma, float *amax,
                                  void *ws, int64_t seq, void *stream)
{
    DCL_CHECK_ARG(dy && x && mean && invstd && dx && ws && count > 0, "bad arguments");
    DCL_CHECK_ARG(dcl_bn_bwd_onepass_supported(N, C, HW, relu), "unsupported shape (H W % 1024, H W / 1024 in 1..32 dividing 32, N <= 12)");
    DCL_CHECK_ARG(relu != 2 || y_or_mask, "relu = 2 needs the packed mask");
    DCL_CHECK_ARG((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0, "16-byte alignment");
    OpArgs a{};
    a.dy = dy; a.x = x; a.mask = (const unsigned long long *)(relu == 2 ? y_or_mask : nullptr);
    a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta;
    a.dx = dx; a.dres = dres; a.dbeta = dbeta; a.dgamma = dgamma; a.amax = amax;
    int *ctrs = (int *)ws;
    a.ctr = ctrs + (seq & 3) * 4096;
    a.ctr_clear = ctrs + ((seq + 2) & 3) * 4096;
    a.part = (float *)(ctrs + 4 * 4096);
    a.N = N; a.C = C; a.HW = HW; a.T = HW / 1024;
    a.inv_count = (float)(1.0 / count);
    a.relu = relu;
    hipLaunchKernelGGL(k_bn_bwd_onepass, dim3(256), dim3(OP_THREADS), 0, (hipStream_t)stream, a);
    DCL_LAUNCH_CHECK();
    return 0;
}
