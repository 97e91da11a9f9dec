// dcl_bn_onepass.hip -- batch-norm backward (and forward) that reads its inputs ONCE.
//
// The two-kernel forms (dcl_bn.hip: reduce, then apply) run at the HBM rate but move every input twice: the per-channel
// sums over all N x H x W values have to be complete before the first output can be written.  Here 256 persistent
// workgroups (one per CU) keep their share of a channel IN REGISTERS between the two phases:
//   team      the H W / 1024 workgroups of one XCD that share a channel (a member owns 1024 consecutive pixels of every image:
//             one float4 per thread and image); 32 / T teams per XCD work on different channels;
//   phase 1   load dy, x (and the packed ReLU mask) of the channel, partial sums, block reduction, the member's two partial
//             sums to memory, one relaxed agent-scope atomic add on the channel's counter;
//   overlap   three register sets: while channel r waits, channel r + 1 is already reduced and published, r + 2 is loading;
//   phase 2   wait until the counter shows T arrivals, add the T partials in member order (double: the same fixed-order sum as
//             part_sums), dx (and the residual gradient) from the registers.
// HBM traffic: dy + x + dx (+ dres) instead of 2 dy + 2 x + dx (tools/probes/l2_reread.py is the model this was sized on: 42 us
// against 68 us on 12 x 48 x 128 x 256); measured: -15 % without a residual output, no gain with one, and a LOSS inside the
// multi-stream training step -- the host side keeps it OFF by default (DCL_BN_ONEPASS=1 turns it on).
// The team barrier needs all members resident: the grid is 256 workgroups of 256 threads, and the host side only takes this
// path on the device's default stream of a single-rank run, so that never more than ONE such kernel is in flight (two
// persistent kernels that each hold part of the CUs would wait for each other forever).  A spin that exceeds ~2^27 polls traps.
// A member publishes its two partial sums as ONE 64-bit store into its slot; slots live in a ring of four regions and every
// launch resets the region two launches ahead to the all-ones "empty" pattern.
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
    unsigned long long *slots, *slots_clear;   // [C][T] {sum g, sum g xhat} of every member: this launch's region / the one to clear
    int N, C, HW, T;
    float inv_count;
    int relu;                           // 0 none, 1 mask recomputed from x, 2 packed mask
};

__device__ __forceinline__ size_t mask_word(size_t plane, int hw4, int iv) { return (plane * (hw4 >> 6) + (iv >> 6)) * 4; }

struct OpSet {
    f32x4 g[OP_NMAX], x[OP_NMAX];
    unsigned mbits[OP_NMAX];            // relu == 2: the four sign bits of this lane's vector
};

// LDS-only workgroup barrier: __syncthreads() also drains vmcnt on this target, i.e. the loads of the channels in flight
__device__ __forceinline__ void op_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(OP_THREADS) void k_bn_bwd_onepass(OpArgs a)
{
    __shared__ float sh[8];
    __shared__ float bc[2];
    const int tid = threadIdx.x, ln = tid & 63;
    // the WHOLE region: the launch that dirtied it may have had more channels x members than this one (a Bottleneck
    // alternates 64 and 256 channels; clearing only C * T of the current shape left stale "published" slots behind)
    for (int i = blockIdx.x * OP_THREADS + tid; i < 4096 * OP_TMAX; i += gridDim.x * OP_THREADS)
        a.slots_clear[i] = ~0ull;
    const int xcd = blockIdx.x & 7, mi = blockIdx.x >> 3;          // 32 workgroups per XCD
    const int T = a.T, tpx = OP_TMAX / T;
    const int tq = mi / T, m = mi - tq * T;
    const int hw4 = a.HW >> 2;
    const int N = a.N;
    float am = 0.f;

    // A member owns a CONTIGUOUS N x 1024-float range of the channel's [image][pixel] data (chunk q = m N + j: image q / T,
    // pixels (q % T) 1024 ...): 48 KB per tensor from one or two planes.  (With a slice of EVERY image per member each
    // workgroup touched 24 planes 94 MB apart per channel -- 2.5 TB/s; the translation caches did not keep up.)
    auto chunk = [&](int j, int &n, int &ivv) {
        const int q = m * N + j;
        n = q / T;
        ivv = (q - n * T) * OP_THREADS + tid;                       // vector index inside the plane
    };
    auto load = [&](OpSet &s, int c) {
#pragma unroll
        for (int j = 0; j < OP_NMAX; ++j) {
            int n, ivv;
            chunk(min(j, N - 1), n, ivv);                           // unconditional loads from clamped chunks
            const size_t plane = (size_t)n * a.C + c;
            const size_t off = plane * a.HW + 4 * (size_t)ivv;
            s.g[j] = *(const f32x4 *)(a.dy + off);
            s.x[j] = *(const f32x4 *)(a.x + off);
            if (a.relu == 2) {
                const unsigned long long *mw = a.mask + mask_word(plane, hw4, ivv);
                s.mbits[j] = (unsigned)((mw[0] >> ln) & 1) | ((unsigned)((mw[1] >> ln) & 1) << 1) |
                             ((unsigned)((mw[2] >> ln) & 1) << 2) | ((unsigned)((mw[3] >> ln) & 1) << 3);
            }
        }
    };
    // channels of this team: c = xcd + 8 (tq + tpx r)
    auto chan = [&](int r) { return xcd + 8 * (tq + tpx * r); };
    // stage B of channel r: mask the gradient, partial sums, publish this member's partials, arrive
    auto reduce_publish = [&](OpSet &cur, int r) {
        const int c = chan(r);
        const float mu = a.mean[c], is = a.invstd[c];
        const float gm = a.gamma ? a.gamma[c] : 1.f;
        const float asc = is * gm, ash = (a.beta ? a.beta[c] : 0.f) - mu * asc;
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
        op_barrier();                                            // the previous user of sh is done
        if (ln == 0) {
            sh[tid >> 6] = sa;
            sh[4 + (tid >> 6)] = sb;
        }
        op_barrier();
        if (tid == 0) {
            // ONE 64-bit relaxed store carries both partial sums: a slot is valid as soon as it differs from the all-ones
            // pattern its region was cleared to (a NaN sum is canonicalised so that it can never look like "empty") -- no
            // arrival counter, no ordering between stores, hence no s_waitcnt vmcnt(0) in the middle of the pipeline
            float pa = (sh[0] + sh[1]) + (sh[2] + sh[3]), pb = (sh[4] + sh[5]) + (sh[6] + sh[7]);
            unsigned ua = __float_as_uint(pa), ub = __float_as_uint(pb);
            ua = pa != pa ? 0x7fc00000u : ua;
            ub = pb != pb ? 0x7fc00000u : ub;
            __hip_atomic_store(a.slots + (size_t)c * T + m, ((unsigned long long)ub << 32) | ua, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // stage C of channel r: wait for the team, the T partials in member order (wave 0: one lane per member, a fixed
    // shuffle tree in double), dx (and the residual's gradient) from the registers
    auto wait_apply = [&](OpSet &cur, int r) {
        const int c = chan(r);
        const float mu = a.mean[c], is = a.invstd[c];
        const float gm = a.gamma ? a.gamma[c] : 1.f;
        if (tid < 64) {
            double da = 0.0, db = 0.0;
            if (tid < T) {
                unsigned long long v;
                unsigned polls = 0;
                while ((v = __hip_atomic_load(a.slots + (size_t)c * T + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == ~0ull) {
                    if (++polls > (1u << 27))
                        __builtin_trap();
                }
                da = (double)__uint_as_float((unsigned)v);
                db = (double)__uint_as_float((unsigned)(v >> 32));
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                da += __shfl_xor(da, o, 64);
                db += __shfl_xor(db, o, 64);
            }
            if (tid == 0) {
                bc[0] = (float)da;
                bc[1] = (float)db;
                if (m == 0) {
                    if (a.dbeta)
                        a.dbeta[c] = (float)da;
                    if (a.dgamma)
                        a.dgamma[c] = (float)db;
                }
            }
        }
        op_barrier();
        const float mg = bc[0] * a.inv_count, mgx = bc[1] * a.inv_count;
        const float k = is * gm;
#pragma unroll
        for (int n = 0; n < OP_NMAX; ++n) {
            if (n < N) {
                const f32x4 g = cur.g[n], xv = cur.x[n];
                f32x4 o;
                o.x = k * (g.x - mg - (xv.x - mu) * is * mgx);
                o.y = k * (g.y - mg - (xv.y - mu) * is * mgx);
                o.z = k * (g.z - mg - (xv.z - mu) * is * mgx);
                o.w = k * (g.w - mg - (xv.w - mu) * is * mgx);
                int nn, ivv;
                chunk(n, nn, ivv);
                const size_t off = ((size_t)nn * a.C + c) * a.HW + 4 * (size_t)ivv;
                if (a.dres)
                    *(f32x4 *)(a.dres + off) = g;
                *(f32x4 *)(a.dx + off) = o;
                am = fmaxf(am, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
            }
        }
        op_barrier();                                            // bc may be rewritten
    };

    // software pipeline over the team's channels with three register sets: while channel r waits for its team, channel
    // r + 1 has already been reduced and published and channel r + 2 is being loaded
    OpSet s0, s1, s2;
    auto has = [&](int r) { return chan(r) < a.C; };
    if (has(0)) {
        load(s0, chan(0));
        reduce_publish(s0, 0);
    }
    if (has(1))
        load(s1, chan(1));
    for (int r = 0; has(r); r += 3) {
        if (has(r + 1))
            reduce_publish(s1, r + 1);
        if (has(r + 2))
            load(s2, chan(r + 2));
        wait_apply(s0, r);
        if (!has(r + 1))
            break;
        if (has(r + 2))
            reduce_publish(s2, r + 2);
        if (has(r + 3))
            load(s0, chan(r + 3));
        wait_apply(s1, r + 1);
        if (!has(r + 2))
            break;
        if (has(r + 3))
            reduce_publish(s0, r + 3);
        if (has(r + 4))
            load(s1, chan(r + 4));
        wait_apply(s2, r + 2);
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
    return 4 * (int64_t)4096 * OP_TMAX * sizeof(unsigned long long);       // four regions of [4096][32] slots
}

// ws: dcl_bn_onepass_workspace_bytes() bytes, set to 0xFF bytes ONCE by the caller; seq: launch counter of that workspace
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
    unsigned long long *slots = (unsigned long long *)ws;
    const size_t region = (size_t)4096 * OP_TMAX;
    a.slots = slots + (seq & 3) * region;
    a.slots_clear = slots + ((seq + 2) & 3) * region;
    a.N = N; a.C = C; a.HW = HW; a.T = HW / 1024;
    a.inv_count = (float)(1.0 / count);
    a.relu = relu;
    hipLaunchKernelGGL(k_bn_bwd_onepass, dim3(256), dim3(OP_THREADS), 0, (hipStream_t)stream, a);
    DCL_LAUNCH_CHECK();
    return 0;
}
