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

// Variant with the weight fragments staged ONCE per workgroup in LDS (double-buffered, one barrier per chunk): the four
// waves of a workgroup use the same R column tiles, and k_tok_gemm's per-wave copies of them are what saturates the L1
// (8 KB of operands per 9 MFMAs and wave).  Each thread carries its share of chunk c + 2 in registers while chunk c + 1
// sits in the other buffer.
template <int R, int P>
__global__ __launch_bounds__(256) void k_tok_gemm_s(TgArgs a)
{
    constexpr int RW = 32 * P, NB = 2 * R;                   // NB 1-KiB fragment blocks per chunk: [tile][hi | lo]
    __shared__ __attribute__((aligned(16))) uint4 wl[2][NB * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x % a.ngroups;
    const long long row0 = (long long)(blockIdx.x / a.ngroups) * (4 * RW) + wave * RW;
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
        long long r = row0 + 32 * p + li;
        r = r < a.M ? r : a.M - 1;
        xp[p] = a.x + r * a.K + 8 * h;
    }
    // staging share of this thread: blocks b0 (and b0 + 4 when it exists) of every chunk
    const int b0 = wave;
    constexpr bool TWO = NB > 4;
    const bool has2 = TWO && b0 + 4 < NB;
    const size_t tstride = (size_t)a.nchunk * 128;
    auto wsrc = [&](int b, int c) {                           // block b = 2 * tile + part of chunk c
        return a.wp + (size_t)(g * R + (b >> 1)) * tstride + (size_t)c * 128 + (b & 1) * 64 + lane;
    };

    f32x16 acc[P][R];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int t = 0; t < R; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                acc[p][t][q] = 0.f;

    // prologue: chunk 0 into buffer 0, chunk 1 into registers, x of chunk 0
    const int c1 = a.nchunk > 1 ? 1 : 0;
    if (b0 < NB)
        wl[0][b0 * 64 + lane] = *wsrc(b0, 0);
    if (has2)
        wl[0][(b0 + 4) * 64 + lane] = *wsrc(b0 + 4, 0);
    uint4 s0 = b0 < NB ? *wsrc(b0, c1) : uint4{0, 0, 0, 0}, s1 = has2 ? *wsrc(b0 + 4, c1) : uint4{0, 0, 0, 0};
    // x fragments two chunks ahead as well (registers): a trip then waits for loads issued a full trip earlier, not for
    // the ones it has just issued -- at ~300 matrix-pipe cycles per trip and 2-3 waves per SIMD one trip of look-ahead
    // covers less than half of the memory latency
    f32x4 xa[P], xb[P], n1a[P], n1b[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        xa[p] = *(const f32x4 *)xp[p];
        xb[p] = *(const f32x4 *)(xp[p] + 4);
        n1a[p] = *(const f32x4 *)(xp[p] + 16 * c1);
        n1b[p] = *(const f32x4 *)(xp[p] + 16 * c1 + 4);
    }
    for (int c = 0; c < a.nchunk; ++c) {
        __syncthreads();                                     // buffer c % 2 is complete, buffer (c + 1) % 2 is free
        const int cur = c & 1;
        if (b0 < NB)
            wl[cur ^ 1][b0 * 64 + lane] = s0;
        if (has2)
            wl[cur ^ 1][(b0 + 4) * 64 + lane] = s1;
        const int c2 = c + 2 < a.nchunk ? c + 2 : a.nchunk - 1;
        if (b0 < NB)
            s0 = *wsrc(b0, c2);
        if (has2)
            s1 = *wsrc(b0 + 4, c2);
        f32x4 n2a[P], n2b[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            n2a[p] = *(const f32x4 *)(xp[p] + 16 * c2);
            n2b[p] = *(const f32x4 *)(xp[p] + 16 * c2 + 4);
        }
        uint4 bh[R], bl[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            bh[t] = wl[cur][(2 * t) * 64 + lane];
            bl[t] = wl[cur][(2 * t + 1) * 64 + lane];
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
            xa[p] = n1a[p];
            xb[p] = n1b[p];
            n1a[p] = n2a[p];
            n1b[p] = n2b[p];
        }
    }

    if (row0 >= a.M)
        return;
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

// Both operands through LDS: the counters say what bounds the two kernels above (tools/probes/tokgemm_one.py under
// rocprofv3 --pmc): 43 % of a wave's lifetime in s_waitcnt, the matrix pipe busy 25 %, and pinning the loads to one chunk
// (all L1 hits) changes nothing -- the x fragments are loaded in MFMA order, every lane its own 32 bytes of its own row,
// and the vector-memory front end looks up one cache line per lane (the bound of the first weight-gradient kernel).
// Here the workgroup's 128 rows x 32 k of x are fetched with eight lanes along each row's 128 bytes (one line per
// eight lanes), parked in LDS with a 144-byte row stride (16 lanes of a ds_read_b128 pass -> 16 bank groups), and read
// back in MFMA order; the weight fragments of the two chunks of a 32-wide k step sit next to them.  Double-buffered, one
// barrier per k step, the next step's 7 loads per thread in flight over the 18 MFMAs of the current one.
template <int R, int P>
__global__ __launch_bounds__(256) void k_tok_gemm_x(TgArgs a)
{
    // P row tiles per wave (32 P rows; a workgroup = 128 P rows): every weight fragment read from LDS feeds P MFMAs
    // (P = 2 moves 10 KB of operands per 18 MFMAs instead of 8 KB per 9, but leaves one wave per SIMD: slower, measured).
    constexpr int XROW = 144, XBUF = 128 * P * XROW, WBUF = 4 * R * 1024;     // bytes per buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 2 x (XBUF + WBUF)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int g = blockIdx.x % a.ngroups;
    const long long blk0 = (long long)(blockIdx.x / a.ngroups) * (128 * P);
    const long long row0 = blk0 + wave * 32 * P;
    float m = 0.f;
    for (int i = lane; i < a.xcount; i += 64)
        m = fmaxf(m, a.xamax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = fmaxf(m, __shfl_xor(m, o, 64));
    const float xs = pow2_scale(m);
    const int li = lane & 31, h = lane >> 5;
    const int nstep = a.nchunk >> 1;                          // k steps of 32
    const size_t tstride = (size_t)a.nchunk * 128;

    // staging shares: x -- 4 P rows (32 i + tid / 8), 16 bytes at tid % 8; weights -- R fragments
    const float *xsrc[4 * P];
#pragma unroll
    for (int i = 0; i < 4 * P; ++i) {
        long long r = blk0 + 32 * i + (tid >> 3);
        r = r < a.M ? r : a.M - 1;
        xsrc[i] = a.x + r * a.K + 4 * (tid & 7);
    }
    const uint4 *wsrc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int u = i * 256 + tid, blk = u >> 6, j = blk / (2 * R), b = blk % (2 * R);
        wsrc[i] = a.wp + (size_t)(g * R + (b >> 1)) * tstride + (size_t)j * 128 + (b & 1) * 64 + (u & 63);
    }
    const int xdst = (tid >> 3) * XROW + (tid & 7) * 16;     // + 32 i rows
    // (named registers, not arrays: the compiler keeps arrays that live across the loop edge as allocas and "promotes"
    // them to 16 KB of LDS plus scratch)
#define TG_STAGE(buf)                                                          \
    do {                                                                       \
        unsigned char *base_ = lds + (buf) * (XBUF + WBUF);                    \
        *(uint4 *)(base_ + xdst) = sx0;                                        \
        *(uint4 *)(base_ + xdst + 32 * XROW) = sx1;                            \
        *(uint4 *)(base_ + xdst + 64 * XROW) = sx2;                            \
        *(uint4 *)(base_ + xdst + 96 * XROW) = sx3;                            \
        if constexpr (P > 1) {                                                 \
            *(uint4 *)(base_ + xdst + 128 * XROW) = sx4;                       \
            *(uint4 *)(base_ + xdst + 160 * XROW) = sx5;                       \
            *(uint4 *)(base_ + xdst + 192 * XROW) = sx6;                       \
            *(uint4 *)(base_ + xdst + 224 * XROW) = sx7;                       \
        }                                                                      \
        *(uint4 *)(base_ + XBUF + tid * 16) = sw0;                             \
        if constexpr (R > 1)                                                   \
            *(uint4 *)(base_ + XBUF + (256 + tid) * 16) = sw1;                 \
        if constexpr (R > 2)                                                   \
            *(uint4 *)(base_ + XBUF + (512 + tid) * 16) = sw2;                 \
    } while (0)
#define TG_FETCH(st_)                                                          \
    do {                                                                       \
        sx0 = *(const uint4 *)(xsrc[0] + 32 * (st_));                          \
        sx1 = *(const uint4 *)(xsrc[1] + 32 * (st_));                          \
        sx2 = *(const uint4 *)(xsrc[2] + 32 * (st_));                          \
        sx3 = *(const uint4 *)(xsrc[3] + 32 * (st_));                          \
        if constexpr (P > 1) {                                                 \
            sx4 = *(const uint4 *)(xsrc[P > 1 ? 4 : 0] + 32 * (st_));          \
            sx5 = *(const uint4 *)(xsrc[P > 1 ? 5 : 0] + 32 * (st_));          \
            sx6 = *(const uint4 *)(xsrc[P > 1 ? 6 : 0] + 32 * (st_));          \
            sx7 = *(const uint4 *)(xsrc[P > 1 ? 7 : 0] + 32 * (st_));          \
        }                                                                      \
        sw0 = wsrc[0][(size_t)(st_) * 256];                                    \
        if constexpr (R > 1)                                                   \
            sw1 = wsrc[R > 1 ? 1 : 0][(size_t)(st_) * 256];                    \
        if constexpr (R > 2)                                                   \
            sw2 = wsrc[R > 2 ? 2 : 0][(size_t)(st_) * 256];                    \
    } while (0)

    f32x16 acc[P][R];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int t = 0; t < R; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                acc[p][t][q] = 0.f;

    uint4 sx0, sx1, sx2, sx3, sx4 = {0, 0, 0, 0}, sx5 = {0, 0, 0, 0}, sx6 = {0, 0, 0, 0}, sx7 = {0, 0, 0, 0};
    uint4 sw0, sw1 = {0, 0, 0, 0}, sw2 = {0, 0, 0, 0};
    // fragments of sub-chunk (st, j) out of buffer st % 2, into named register sets (no arrays across the loop edge)
    const int xoff = (wave * 32 * P + li) * XROW + h * 32, woff = XBUF + lane * 16;
#define TG_READ(buf, j, XA, XC, XA2, XC2, BH, BL)                                              \
    do {                                                                                       \
        const unsigned char *b_ = lds + (buf) * (XBUF + WBUF);                                 \
        XA = *(const f32x4 *)(b_ + xoff + (j) * 64);                                           \
        XC = *(const f32x4 *)(b_ + xoff + (j) * 64 + 16);                                      \
        if constexpr (P > 1) {                                                                 \
            XA2 = *(const f32x4 *)(b_ + xoff + 32 * XROW + (j) * 64);                          \
            XC2 = *(const f32x4 *)(b_ + xoff + 32 * XROW + (j) * 64 + 16);                     \
        }                                                                                      \
        _Pragma("unroll") for (int t = 0; t < R; ++t) {                                        \
            BH[t] = *(const uint4 *)(b_ + woff + (((j) * 2 * R + 2 * t) << 10));               \
            BL[t] = *(const uint4 *)(b_ + woff + (((j) * 2 * R + 2 * t + 1) << 10));           \
        }                                                                                      \
    } while (0)
#define TG_SPLIT(XA, XC, AH, AL)                                                               \
    do {                                                                                       \
        uint4 uh_, ul_;                                                                        \
        split2(XA.x, XA.y, xs, uh_.x, ul_.x);                                                  \
        split2(XA.z, XA.w, xs, uh_.y, ul_.y);                                                  \
        split2(XC.x, XC.y, xs, uh_.z, ul_.z);                                                  \
        split2(XC.z, XC.w, xs, uh_.w, ul_.w);                                                  \
        AH = __builtin_bit_cast(half8, uh_);                                                   \
        AL = __builtin_bit_cast(half8, ul_);                                                   \
    } while (0)
#define TG_MFMA(XA, XC, XA2, XC2, BH, BL)                                                      \
    do {                                                                                       \
        half8 ah[P], al[P];                                                                    \
        TG_SPLIT(XA, XC, ah[0], al[0]);                                                        \
        if constexpr (P > 1)                                                                   \
            TG_SPLIT(XA2, XC2, ah[P - 1], al[P - 1]);                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        _Pragma("unroll") for (int p = 0; p < P; ++p)                                          \
            _Pragma("unroll") for (int t = 0; t < R; ++t)                                      \
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[p], __builtin_bit_cast(half8, BH[t]), acc[p][t], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        _Pragma("unroll") for (int p = 0; p < P; ++p)                                          \
            _Pragma("unroll") for (int t = 0; t < R; ++t)                                      \
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[p], __builtin_bit_cast(half8, BL[t]), acc[p][t], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        _Pragma("unroll") for (int p = 0; p < P; ++p)                                          \
            _Pragma("unroll") for (int t = 0; t < R; ++t)                                      \
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[p], __builtin_bit_cast(half8, BH[t]), acc[p][t], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)

    // Schedule of a k step (two sub-chunks of 16): the fragment reads run ONE sub-chunk ahead of the MFMAs, so each MFMA
    // block covers the LDS latency of the next one's operands, and the workgroup barrier sits between the two blocks:
    //   read F(st, 1) | MFMA(st, 0) | park step st + 1 in the other buffer, fetch step st + 2 | barrier |
    //   read F(st + 1, 0) | MFMA(st, 1)
    // (a raw barrier behind an LDS-only wait: __syncthreads() also drains vmcnt, i.e. waits at every step for the global
    // loads that were issued just before it)
#define TG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    TG_FETCH(0);
    TG_STAGE(0);
    TG_FETCH(nstep > 1 ? 1 : 0);
    f32x4 xa0, xc0, xa1, xc1, ya0 = {0.f, 0.f, 0.f, 0.f}, yc0 = ya0, ya1 = ya0, yc1 = ya0;
    uint4 bh0[R], bl0[R], bh1[R], bl1[R];
    TG_BARRIER();
    TG_READ(0, 0, xa0, xc0, ya0, yc0, bh0, bl0);
    for (int st = 0; st < nstep; ++st) {
        const int cur = st & 1;
        TG_READ(cur, 1, xa1, xc1, ya1, yc1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
        TG_MFMA(xa0, xc0, ya0, yc0, bh0, bl0);
        TG_STAGE(cur ^ 1);
        TG_FETCH(st + 2 < nstep ? st + 2 : nstep - 1);
        TG_BARRIER();                          // the other buffer is complete; nobody reads this one any more
        TG_READ(cur ^ 1, 0, xa0, xc0, ya0, yc0, bh0, bl0);      // (after the last step: a harmless re-read)
        __builtin_amdgcn_sched_barrier(0);
        TG_MFMA(xa1, xc1, ya1, yc1, bh1, bl1);
    }
#undef TG_READ
#undef TG_SPLIT
#undef TG_MFMA
#undef TG_BARRIER

#undef TG_STAGE
#undef TG_FETCH
    if (row0 >= a.M)
        return;
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
    hipStream_t s = (hipStream_t)stream;
    if (K % 32 == 0 && !(g_tok_dbg & 12)) {
        // both operands through LDS (k_tok_gemm_x); the two kernels above remain for K % 32 = 16 and as references
        // one row tile per wave: two (98 KB of LDS, one workgroup = one wave per SIMD) are 25-50 % slower on every shape
        // (tools/probes/tokgemm_bound.py); kept for dcl_tok_gemm_set_rows(2)
        int PX = g_tok_p;
        if (PX <= 0)
            PX = 1;
        const long long rbx = (M + 128 * PX - 1) / (128 * PX);
        DCL_CHECK_ARG(rbx * a.ngroups < ((long long)1 << 31), "grid too large");
        const dim3 gridx((unsigned)(rbx * a.ngroups));
        const size_t ldsx = 2 * ((size_t)128 * PX * 144 + 4 * R * 1024);
#define DCL_TGX_CASE(r, p)                                                                              \
    if (R == r && PX == p) {                                                                            \
        if (ldsx > 64 * 1024)                                                                           \
            (void)hipFuncSetAttribute((const void *)k_tok_gemm_x<r, p>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)ldsx);                                                       \
        hipLaunchKernelGGL((k_tok_gemm_x<r, p>), gridx, dim3(256), ldsx, s, a);                         \
    }
        DCL_TGX_CASE(3, 2) DCL_TGX_CASE(3, 1) DCL_TGX_CASE(2, 2) DCL_TGX_CASE(2, 1) DCL_TGX_CASE(1, 2) DCL_TGX_CASE(1, 1)
#undef DCL_TGX_CASE
        DCL_LAUNCH_CHECK();
        return 0;
    }
    int P = g_tok_p;
    if (P <= 0)
        P = ((M + 255) / 256) * a.ngroups >= 512 ? 2 : 1;
    const long long rb = (M + 128 * P - 1) / (128 * P);
    DCL_CHECK_ARG(rb * a.ngroups < ((long long)1 << 31), "grid too large");
    const dim3 grid((unsigned)(rb * a.ngroups));
#define DCL_TG_CASE(r, p)                                                   \
    if (R == r && P == p) {                                                 \
        if (g_tok_dbg & 4)                                                  \
            hipLaunchKernelGGL((k_tok_gemm<r, p>), grid, dim3(256), 0, s, a); \
        else                                                                \
            hipLaunchKernelGGL((k_tok_gemm_s<r, p>), grid, dim3(256), 0, s, a); \
    }
    DCL_TG_CASE(3, 2) DCL_TG_CASE(3, 1) DCL_TG_CASE(2, 2) DCL_TG_CASE(2, 1) DCL_TG_CASE(1, 2) DCL_TG_CASE(1, 1)
#undef DCL_TG_CASE
    DCL_LAUNCH_CHECK();
    return 0;
}
