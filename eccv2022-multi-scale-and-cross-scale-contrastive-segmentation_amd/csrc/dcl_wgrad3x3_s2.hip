// dcl_wgrad3x3_s2.hip -- weight gradient of the 3x3 / STRIDE 2 / pad 1 convolution (f16x3, fp32-equivalent).
//
//   dw[co, ci, ky, kx] = sum_{n, yo, xo} dy[n, co, yo, xo] * x[n, ci, 2 yo + ky - 1, 2 xo + kx - 1]
//
// The stride-1 kernels handle this case by reading dy as if zeros were inserted at the odd coordinates, which keeps
// their data flow but spends three quarters of the matrix work on zeros (HRNet-W48 fuse layers: 0.10-0.12 of the
// roofline).  Here the GEMM K dimension is the OUTPUT pixels: a lane of v_mfma_f32_16x16x32_f16 supplies 8
// consecutive xo of one row,
//   A: dy[co][yo][xo0 .. xo0 + 7]                       -- one 32-byte run,
//   B: x[ci][2 yo + ky - 1][2 xo0 + kx - 1 + 2 e]       -- every second value of the 17-wide window V[-1 .. 15] at
//      column 2 xo0: kx = 1 takes the even values E = V[0], V[2], ..; kx = 2 the odd ones O = V[1], V[3], ..; kx = 0 is O
//      shifted by one element with the left halo V[-1] in front.  Split in pairs (V0, V2), (V1, V3), .. the f16 words
//      ARE the E and O fragments, and the kx = 0 fragment is the 16-bit funnel shift of neighbouring O words.
// A wave owns NCO x NCI tiles x 9 taps and walks down a 32-output-pixel strip.  Output row yo pairs with x rows
// 2 yo - 1 (ky = 0), 2 yo (ky = 1), 2 yo + 1 (ky = 2); row 2 yo + 1 is row 2 (yo + 1) - 1 of the next step, so its
// fragments are carried over: per step one dy row and two x rows are loaded, 27 NCO NCI MFMAs issued -- a quarter of
// the zero-inserted formulation's.  Work split, slabs and the fixed-order reductions are those of dcl_wgrad3x3.hip
// (four waves = four runs of the flat (image, strip, output row) sequence, combined through LDS; one slab per
// workgroup; k_wgrad_reduce sums the slabs).
#include <type_traits>

#include "dcl_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float F16_TARGET = 16384.0f;

__device__ __forceinline__ float pow2_scale(float amax)
{
    return amax == 0.f ? 1.f : exp2f(fminf(fmaxf(floorf(log2f(F16_TARGET / amax)), -100.f), 100.f));
}

struct WgradS2Args {
    const float *x, *dy;
    float *part;                 // [nx][9][Cout][Cin]
    const float *xamax, *gamax;
    int xcount, gcount;
    int N, Cin, Cout, H, W;      // x is [N, Cin, H, W]
    int Hd, Wd;                  // dy is [N, Cout, Hd, Wd]
    int strips, units, S, ncig, npairs, nx;
    const float *pre_sc, *pre_sh;   // PRE forms: the input operand is relu(x * pre_sc[ci] + pre_sh[ci]) (see k_wgrad3x3d, PRE); else NULL
};

__device__ __forceinline__ void split2(float v0, float v1, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(s), "v"(hi));
}

__device__ __forceinline__ void split1(float v0, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
}

__device__ __forceinline__ half8 as_half8(u32x4 v) { return __builtin_bit_cast(half8, v); }

// DEEP: operands are loaded two steps ahead of their use instead of one (second set of raw registers)
// PRE: x is the RAW output of the convolution in front of a training-mode norm, the operand relu(x sc[ci] + sh[ci]) -- the second
// convolution of a two-step down-sampling chain (reference models/HRNet.py:236-258: conv s2 -> bn -> relu -> conv s2) and the stem's
// conv2 (:333-338) without the normalised tensor in memory; a lane converts values of one input channel per ci tile, rows outside
// the image and halo values are masked through the operand scale, i.e. after the map (as in dcl_wgrad3x3d.hip).
template <int NCO, int NCI, bool DEEP, bool PRE = false>
__global__ __launch_bounds__(256, 1) void k_wgrad3x3_s2(WgradS2Args a)
{
    __shared__ float wm[8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q4 = lane >> 4, j = lane & 15;

    float sx, sg;
    {
        float mx = 0.f, mg = 0.f;
        for (int i = tid; i < a.xcount; i += 256)
            mx = fmaxf(mx, a.xamax[i]);
        for (int i = tid; i < a.gcount; i += 256)
            mg = fmaxf(mg, a.gamax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            mg = fmaxf(mg, __shfl_xor(mg, o, 64));
        }
        if (lane == 0) {
            wm[wave] = mx;
            wm[4 + wave] = mg;
        }
        __syncthreads();
        sx = pow2_scale(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
        sg = pow2_scale(fmaxf(fmaxf(wm[4], wm[5]), fmaxf(wm[6], wm[7])));
    }
    // XCD-aware decode as in k_wgrad3x3: the tile pairs of one pixel split share an XCD (and its L2)
    int pair, xsplit;
    {
        const int nx8 = a.nx & ~7, main_blocks = nx8 * a.npairs;
        if ((int)blockIdx.x < main_blocks) {
            const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
            pair = rest % a.npairs;
            xsplit = (rest / a.npairs) * 8 + xcd;
        } else {
            const int rest = blockIdx.x - main_blocks;
            pair = rest % a.npairs;
            xsplit = nx8 + rest / a.npairs;
        }
    }
    const int split = xsplit * 4 + wave;           // a wave past the last split gets an empty run and adds zeros
    const int cog = pair / a.ncig, cig = pair - cog * a.ncig;
    const int co0 = cog * NCO * 16, ci0 = cig * NCI * 16;
    const size_t plane = (size_t)a.H * a.W, dplane = (size_t)a.Hd * a.Wd;
    bool ci_ok[NCI];
#pragma unroll
    for (int u = 0; u < NCI; ++u)
        ci_ok[u] = ci0 + 16 * u < a.Cin;
    float psc[NCI], psh[NCI];               // PRE: the norm's map of this lane's channel of ci tile u
#pragma unroll
    for (int u = 0; u < NCI; ++u) {
        const int ch = ci_ok[u] ? ci0 + 16 * u + j : ci0 + j;
        psc[u] = PRE ? a.pre_sc[ch] : 1.f;
        psh[u] = PRE ? a.pre_sh[ch] : 0.f;
    }
    auto pre = [&](float v, int u) { return PRE ? fmaxf(__builtin_fmaf(v, psc[u], psh[u]), 0.f) : v; };

    f32x4 acc[NCO][NCI][9];
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k)
                acc[t][u][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long T = (long long)a.units * a.Hd;             // units = images x strips of 32 output pixels
    long long t = min(T, T * split / a.S);
    const long long t1 = min(T, T * (split + 1) / a.S);
    while (t < t1) {
        const int col = (int)(t / a.Hd);
        const int r0 = (int)(t - (long long)col * a.Hd);
        const int r1 = (int)min((long long)a.Hd, r0 + (t1 - t));
        t += r1 - r0;
        const int strip = col % a.strips;
        const int n = col / a.strips;
        const int xo = strip * 32 + 8 * q4;                     // first of the lane's 8 output pixels
        // unconditional loads from clamped addresses, masked through the operand scale (see dcl_wgrad3x3.hip)
        const bool oct_ok = xo < a.Wd;
        const int xoc = oct_ok ? xo : a.Wd - 8;
        const int dl = (oct_ok && xo > 0) ? -1 : 0;             // left halo offset from the clamped base
        const float sg_c = oct_ok ? sg : 0.f;
        const float sx_c = oct_ok ? sx : 0.f;
        const float sx_l = (oct_ok && xo > 0) ? sx : 0.f;       // x column -1 is outside the image
        const float *ap = a.dy + ((size_t)n * a.Cout + co0 + j) * dplane + xoc;
        const float *bp = a.x + ((size_t)n * a.Cin + ci0 + j) * plane + 2 * xoc;

        auto load_A = [&](int yo, f32x4 (&dst)[NCO][2]) {
            const int yc = min(yo, a.Hd - 1);
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2) {
                const float *p = ap + (size_t)t2 * 16 * dplane + (size_t)yc * a.Wd;
                dst[t2][0] = *(const f32x4 *)p;
                dst[t2][1] = *(const f32x4 *)(p + 4);
            }
        };
        auto cvt_A = [&](const f32x4 (&src)[NCO][2], half8 (&dst)[NCO][2]) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2) {
                unsigned h[4], l[4];
                split2(src[t2][0].x, src[t2][0].y, sg_c, h[0], l[0]);
                split2(src[t2][0].z, src[t2][0].w, sg_c, h[1], l[1]);
                split2(src[t2][1].x, src[t2][1].y, sg_c, h[2], l[2]);
                split2(src[t2][1].z, src[t2][1].w, sg_c, h[3], l[3]);
                dst[t2][0] = as_half8(u32x4{h[0], h[1], h[2], h[3]});
                dst[t2][1] = as_half8(u32x4{l[0], l[1], l[2], l[3]});
            }
        };
        // x row r: the 16 values at column 2 xo and the left halo
        auto load_B = [&](int r, f32x4 (&dst)[NCI][4], float (&l)[NCI]) {
            const int rc = min(max(r, 0), a.H - 1);
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                const float *p = bp + (size_t)(ci_ok[u] ? u : 0) * 16 * plane + (size_t)rc * a.W;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dst[u][e] = *(const f32x4 *)(p + 4 * e);
                l[u] = p[dl];
            }
        };
        auto cvt_B = [&](int r, const f32x4 (&src)[NCI][4], const float (&l)[NCI], half8 (&dst)[3][NCI][2]) {
            const bool row_ok = r >= 0 && r < a.H;
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                const float sc = (ci_ok[u] && row_ok) ? sx_c : 0.f, sl = (ci_ok[u] && row_ok) ? sx_l : 0.f;
                unsigned eh[4], el[4], oh[4], ol[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    split2(pre(src[u][e].x, u), pre(src[u][e].z, u), sc, eh[e], el[e]);          // (V[4e], V[4e + 2])
                    split2(pre(src[u][e].y, u), pre(src[u][e].w, u), sc, oh[e], ol[e]);          // (V[4e + 1], V[4e + 3])
                }
                unsigned hl, ql;
                split1(pre(l[u], u), sl, hl, ql);                                        // V[-1]
                unsigned zh[4], zl[4];                                           // (V[-1], V[1]), (V[3], V[5]), ..
                zh[0] = __builtin_amdgcn_perm(oh[0], hl, 0x05040100u);
                zl[0] = __builtin_amdgcn_perm(ol[0], ql, 0x05040100u);
#pragma unroll
                for (int e = 1; e < 4; ++e) {
                    zh[e] = __builtin_amdgcn_alignbit(oh[e], oh[e - 1], 16);
                    zl[e] = __builtin_amdgcn_alignbit(ol[e], ol[e - 1], 16);
                }
                dst[0][u][0] = as_half8(u32x4{zh[0], zh[1], zh[2], zh[3]});
                dst[0][u][1] = as_half8(u32x4{zl[0], zl[1], zl[2], zl[3]});
                dst[1][u][0] = as_half8(u32x4{eh[0], eh[1], eh[2], eh[3]});
                dst[1][u][1] = as_half8(u32x4{el[0], el[1], el[2], el[3]});
                dst[2][u][0] = as_half8(u32x4{oh[0], oh[1], oh[2], oh[3]});
                dst[2][u][1] = as_half8(u32x4{ol[0], ol[1], ol[2], ol[3]});
            }
        };

        // Software pipeline, unrolled by two so that every register array is indexed statically.  At the start of step
        // yo the fragments of dy row yo (A2[ph]), x row 2 yo - 1 (O[ph]) and x row 2 yo (Q) are ready; raw registers
        // hold dy row yo + 1 and x rows 2 yo + 1, 2 yo + 2.
        //   1. MFMAs of tap rows ky = 0 (O[ph]) and ky = 1 (Q); underneath them the VALU splits x row 2 yo + 1 into
        //      O[ph ^ 1] and dy row yo + 1 into A2[ph ^ 1],
        //   2. loads of x row 2 yo + 3 and dy row yo + 2 into the raw registers just consumed,
        //   3. MFMAs of tap row ky = 2 (O[ph ^ 1]); underneath, x row 2 yo + 2 is split into Q (its readers have issued),
        //   4. load of x row 2 yo + 4.
        half8 A2[2][NCO][2], O[2][3][NCI][2], Q[3][NCI][2];
        constexpr int ND = DEEP ? 2 : 1;
        f32x4 rawA[ND][NCO][2], rawB0[ND][NCI][4], rawB1[ND][NCI][4];
        float rawL0[ND][NCI], rawL1[ND][NCI];
        {
            f32x4 q0[NCI][4], q1[NCI][4], p0[NCO][2];
            float l0[NCI], l1[NCI];
            load_B(2 * r0 - 1, q0, l0);
            load_B(2 * r0, q1, l1);
            load_A(r0, p0);
            load_B(2 * r0 + 1, rawB1[0], rawL1[0]);
            load_A(r0 + 1, rawA[0]);
            load_B(2 * r0 + 2, rawB0[0], rawL0[0]);
            if (DEEP) {
                load_B(2 * r0 + 3, rawB1[ND - 1], rawL1[ND - 1]);
                load_A(r0 + 2, rawA[ND - 1]);
                load_B(2 * r0 + 4, rawB0[ND - 1], rawL0[ND - 1]);
            }
            cvt_B(2 * r0 - 1, q0, l0, O[0]);
            cvt_B(2 * r0, q1, l1, Q);
            cvt_A(p0, A2[0]);
        }
        auto mfma_row = [&](const half8 (&Af)[NCO][2], const half8 (&B)[3][NCI][2], auto KY) {
            constexpr int ky = decltype(KY)::value;
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                        for (int u = 0; u < NCI; ++u)
                            acc[t2][u][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                Af[t2][pass == 2 ? 1 : 0], B[kx][u][pass == 1 ? 1 : 0], acc[t2][u][ky * 3 + kx], 0, 0, 0);
        };
        auto step = [&](auto PH, int yo) {
            constexpr int ph = decltype(PH)::value, rs = DEEP ? ph : 0, ahead = DEEP ? 1 : 0;
            mfma_row(A2[ph], O[ph], std::integral_constant<int, 0>{});
            mfma_row(A2[ph], Q, std::integral_constant<int, 1>{});
            cvt_B(2 * yo + 1, rawB1[rs], rawL1[rs], O[ph ^ 1]);
            cvt_A(rawA[rs], A2[ph ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            load_B(2 * (yo + ahead) + 3, rawB1[rs], rawL1[rs]);
            load_A(yo + ahead + 2, rawA[rs]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(A2[ph], O[ph ^ 1], std::integral_constant<int, 2>{});
            cvt_B(2 * yo + 2, rawB0[rs], rawL0[rs], Q);
            __builtin_amdgcn_sched_barrier(0);
            load_B(2 * (yo + ahead) + 4, rawB0[rs], rawL0[rs]);
            __builtin_amdgcn_sched_barrier(0);
        };
        int yo = r0;
        for (; yo + 2 <= r1; yo += 2) {
            step(std::integral_constant<int, 0>{}, yo);
            step(std::integral_constant<int, 1>{}, yo + 1);
        }
        if (yo < r1)
            step(std::integral_constant<int, 0>{}, yo);
    }

    // (w0 + w1) + (w2 + w3) through LDS, one slab per workgroup
    {
        constexpr int NREG = NCO * NCI * 36;
        __shared__ float red[2][NREG][64];
        auto put = [&](int b) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            red[b][((t2 * NCI + u) * 9 + k) * 4 + q][lane] = acc[t2][u][k][q];
        };
        auto add = [&](int b) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[t2][u][k][q] += red[b][((t2 * NCI + u) * 9 + k) * 4 + q][lane];
        };
        if (wave & 1)
            put(wave >> 1);
        __syncthreads();
        if (!(wave & 1))
            add(wave >> 1);
        __syncthreads();
        if (wave == 2)
            put(0);
        __syncthreads();
        if (wave != 0)
            return;
        add(0);
    }
    const float inv = 1.0f / (sx * sg);
    float *out = a.part + (size_t)xsplit * 9 * a.Cout * a.Cin;
#pragma unroll
    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = co0 + 16 * t2 + 4 * q4 + q, ci = ci0 + 16 * u + j;
                    if (ci_ok[u])
                        out[((size_t)k * a.Cout + co) * a.Cin + ci] = acc[t2][u][k][q] * inv;
                }
}

// ---- LDS-DMA form (round 5): the x rows -- ten of the sixteen vector-memory instructions of a step above, each one touching 64
// different cache lines because its lanes run ACROSS 16 channel rows -- are fetched with global_load_lds_dwordx4, lanes running
// ALONG the rows: a row set (NCI x 16 channel rows of one x row: the 64 pixels of the strip plus a 4-pixel piece for the left halo,
// 17 pieces of 16 bytes per row) is 272 NCI pieces = 5 (9) wave instructions touching ~4 lines per 17 lanes instead of 64 per
// instruction; the data lands in a private ring of the wave ([slot][row set][row][piece]; 17 pieces per row: odd, so the 16 rows of a
// ds_read_b128 pass fall into 16 different bank groups) and is read back in MFMA order one to two steps later.  The dY rows (six
// instructions per step) stay on direct loads.  Same arithmetic, order of accumulation and slabs as k_wgrad3x3_s2: bitwise the same
// result.
__device__ __forceinline__ const float *uniform_ptr(const float *p)
{
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void dma16(const void *gbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait()
{
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int S2_BP = 17;                                 // pieces per staged x row: left-halo piece + 16 pieces of 4 pixels

template <int NCO, int NCI, bool PRE = false>
__global__ __launch_bounds__(256, 1) void k_wgrad3x3_s2d(WgradS2Args a)
{
    constexpr int NIR = (NCI * 16 * S2_BP + 63) / 64;      // DMA instructions per row set (5 | 9)
    constexpr int SETB = NIR * 1024, SLOTB = 2 * SETB;     // a group = x rows 2 g + 1 and 2 g + 2
    constexpr int NS = 3;
    constexpr int NREG = NCO * NCI * 36;
    constexpr int STAGEB = 4 * NS * SLOTB, REDB = 2 * NREG * 64 * 4;
    constexpr int SMEMB = STAGEB > REDB ? STAGEB : REDB;
    static_assert(SMEMB + 64 <= 160 * 1024, "LDS");
    constexpr int BUNDLE = 2 * NIR + 2 * NCO;              // vector-memory instructions of one look-ahead bundle
    static_assert(BUNDLE < 64, "vmcnt");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEMB];
    __shared__ float wm[8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q4 = lane >> 4, j = lane & 15;

    float sx, sg;
    {
        float mx = 0.f, mg = 0.f;
        for (int i = tid; i < a.xcount; i += 256)
            mx = fmaxf(mx, a.xamax[i]);
        for (int i = tid; i < a.gcount; i += 256)
            mg = fmaxf(mg, a.gamax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            mg = fmaxf(mg, __shfl_xor(mg, o, 64));
        }
        if (lane == 0) {
            wm[wave] = mx;
            wm[4 + wave] = mg;
        }
        __syncthreads();
        sx = pow2_scale(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
        sg = pow2_scale(fmaxf(fmaxf(wm[4], wm[5]), fmaxf(wm[6], wm[7])));
    }
    int pair, xsplit;
    {
        const int nx8 = a.nx & ~7, main_blocks = nx8 * a.npairs;
        if ((int)blockIdx.x < main_blocks) {
            const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
            pair = rest % a.npairs;
            xsplit = (rest / a.npairs) * 8 + xcd;
        } else {
            const int rest = blockIdx.x - main_blocks;
            pair = rest % a.npairs;
            xsplit = nx8 + rest / a.npairs;
        }
    }
    const int split = xsplit * 4 + wave;
    const int cog = pair / a.ncig, cig = pair - cog * a.ncig;
    const int co0 = cog * NCO * 16, ci0 = cig * NCI * 16;
    const size_t plane = (size_t)a.H * a.W, dplane = (size_t)a.Hd * a.Wd;
    bool ci_ok[NCI];
#pragma unroll
    for (int u = 0; u < NCI; ++u)
        ci_ok[u] = ci0 + 16 * u < a.Cin;
    float psc[NCI], psh[NCI];               // PRE: the norm's map of this lane's channel of ci tile u
#pragma unroll
    for (int u = 0; u < NCI; ++u) {
        const int ch = ci_ok[u] ? ci0 + 16 * u + j : ci0 + j;
        psc[u] = PRE ? a.pre_sc[ch] : 1.f;
        psh[u] = PRE ? a.pre_sh[ch] : 0.f;
    }
    auto pre = [&](float v, int u) { return PRE ? fmaxf(__builtin_fmaf(v, psc[u], psh[u]), 0.f) : v; };

    // DMA geometry of this lane, per instruction m of a row set: piece P = 64 m + lane in [row][piece] order (the tail repeats piece 0)
    unsigned chanB[NIR];
    int pieceB[NIR];                        // first pixel of the piece relative to 2 px0: -4, 0, 4, .. 60
#pragma unroll
    for (int m = 0; m < NIR; ++m) {
        int P = 64 * m + lane;
        if (P >= NCI * 16 * S2_BP)
            P = 0;
        const int row = P / S2_BP, pc = P - row * S2_BP;
        const int u = row >> 4;
        const int ch = (ci0 + 16 * u < a.Cin) ? ci0 + row : ci0 + (row & 15);        // ragged last ci group: tile 0 again (masked)
        chanB[m] = (unsigned)((size_t)ch * plane * 4);
        pieceB[m] = 4 * pc - 4;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)smem +
                          (unsigned)wave * (NS * SLOTB);
    const unsigned char *my = smem + wave * (NS * SLOTB);
    const unsigned brdo = (unsigned)((j * S2_BP + 1 + 4 * q4) * 16);      // lane (q4, j): row j, the 16 pixels at 2 (px0 + 8 q4)

    f32x4 acc[NCO][NCI][9];
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k)
                acc[t][u][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long T = (long long)a.units * a.Hd;
    long long t = min(T, T * split / a.S);
    const long long t1 = min(T, T * (split + 1) / a.S);
    while (t < t1) {
        const int col = (int)(t / a.Hd);
        const int r0 = (int)(t - (long long)col * a.Hd);
        const int r1 = (int)min((long long)a.Hd, r0 + (t1 - t));
        t += r1 - r0;
        const int strip = col % a.strips;
        const int n = col / a.strips;
        const int px0 = strip * 32, xo = px0 + 8 * q4;
        const bool oct_ok = xo < a.Wd;
        const int xoc = oct_ok ? xo : a.Wd - 8;
        const float sg_c = oct_ok ? sg : 0.f;
        const float sx_c = oct_ok ? sx : 0.f;
        const float sx_l = (oct_ok && xo > 0) ? sx : 0.f;
        const float *ap = a.dy + ((size_t)n * a.Cout + co0 + j) * dplane + xoc;
        const float *xn = a.x + (size_t)n * a.Cin * plane;
        unsigned offB[NIR];
#pragma unroll
        for (int m = 0; m < NIR; ++m)
            offB[m] = chanB[m] + 4u * (unsigned)min(max(2 * px0 + pieceB[m], 0), a.W - 4);

        auto load_A = [&](int yo, f32x4 (&dst)[NCO][2]) {
            const int yc = min(yo, a.Hd - 1);
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2) {
                const float *p = ap + (size_t)t2 * 16 * dplane + (size_t)yc * a.Wd;
                dst[t2][0] = *(const f32x4 *)p;
                dst[t2][1] = *(const f32x4 *)(p + 4);
            }
        };
        auto cvt_A = [&](const f32x4 (&src)[NCO][2], half8 (&dst)[NCO][2]) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2) {
                unsigned h[4], l[4];
                split2(src[t2][0].x, src[t2][0].y, sg_c, h[0], l[0]);
                split2(src[t2][0].z, src[t2][0].w, sg_c, h[1], l[1]);
                split2(src[t2][1].x, src[t2][1].y, sg_c, h[2], l[2]);
                split2(src[t2][1].z, src[t2][1].w, sg_c, h[3], l[3]);
                dst[t2][0] = as_half8(u32x4{h[0], h[1], h[2], h[3]});
                dst[t2][1] = as_half8(u32x4{l[0], l[1], l[2], l[3]});
            }
        };
        // group g = x rows 2 g + 1 (row set 0) and 2 g + 2 (row set 1) -> ring slot s
        auto dma_group = [&](int g, int s) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)s * SLOTB);
#pragma unroll
            for (int rs = 0; rs < 2; ++rs) {
                const float *bb = uniform_ptr(xn + (size_t)min(max(2 * g + 1 + rs, 0), a.H - 1) * a.W);
#pragma unroll
                for (int m = 0; m < NIR; ++m)
                    dma16(bb, offB[m], dst + rs * SETB + m * 1024);
            }
        };
        auto read_B = [&](int s, int rs, f32x4 (&dst)[NCI][4], float (&l)[NCI]) {
            const unsigned char *p = my + s * SLOTB + rs * SETB + brdo;
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                const unsigned char *pu = p + u * (16 * S2_BP * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dst[u][e] = *(const f32x4 *)(pu + 16 * e);
                l[u] = *(const float *)(pu - 4);
            }
        };
        auto cvt_B = [&](int r, const f32x4 (&src)[NCI][4], const float (&l)[NCI], half8 (&dst)[3][NCI][2]) {
            const bool row_ok = r >= 0 && r < a.H;
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                const float sc = (ci_ok[u] && row_ok) ? sx_c : 0.f, sl = (ci_ok[u] && row_ok) ? sx_l : 0.f;
                unsigned eh[4], el[4], oh[4], ol[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    split2(pre(src[u][e].x, u), pre(src[u][e].z, u), sc, eh[e], el[e]);
                    split2(pre(src[u][e].y, u), pre(src[u][e].w, u), sc, oh[e], ol[e]);
                }
                unsigned hl, ql;
                split1(pre(l[u], u), sl, hl, ql);
                unsigned zh[4], zl[4];
                zh[0] = __builtin_amdgcn_perm(oh[0], hl, 0x05040100u);
                zl[0] = __builtin_amdgcn_perm(ol[0], ql, 0x05040100u);
#pragma unroll
                for (int e = 1; e < 4; ++e) {
                    zh[e] = __builtin_amdgcn_alignbit(oh[e], oh[e - 1], 16);
                    zl[e] = __builtin_amdgcn_alignbit(ol[e], ol[e - 1], 16);
                }
                dst[0][u][0] = as_half8(u32x4{zh[0], zh[1], zh[2], zh[3]});
                dst[0][u][1] = as_half8(u32x4{zl[0], zl[1], zl[2], zl[3]});
                dst[1][u][0] = as_half8(u32x4{eh[0], eh[1], eh[2], eh[3]});
                dst[1][u][1] = as_half8(u32x4{el[0], el[1], el[2], el[3]});
                dst[2][u][0] = as_half8(u32x4{oh[0], oh[1], oh[2], oh[3]});
                dst[2][u][1] = as_half8(u32x4{ol[0], ol[1], ol[2], ol[3]});
            }
        };

        // Bundle b(g) = {x rows 2 g + 1, 2 g + 2 into slot g % 3, dY row g + 1 into rawA[g % 2]}: consumed during step g, issued
        // during step g - 2.  At the top of step g one younger bundle (b(g + 1)) may still be landing: s_waitcnt vmcnt(BUNDLE).
        half8 A2[2][NCO][2], O[2][3][NCI][2], Q[3][NCI][2];
        f32x4 rawA[2][NCO][2];
        auto bundle = [&](int g, auto SLOT, auto RA) {
            dma_group(g, decltype(SLOT)::value);
            load_A(g + 1, rawA[decltype(RA)::value]);
        };
        vm_wait<0>();                            // nothing of the previous column is still landing in the ring
        {
            // column start: group r0 - 1 (x rows 2 r0 - 1, 2 r0) and dY row r0 by themselves, then bundles b(r0), b(r0 + 1)
            f32x4 p0[NCO][2], q0[NCI][4], q1[NCI][4];
            float l0[NCI], l1[NCI];
            dma_group(r0 - 1, 2);
            load_A(r0, p0);
            vm_wait<0>();
            read_B(2, 0, q0, l0);
            read_B(2, 1, q1, l1);
            cvt_B(2 * r0 - 1, q0, l0, O[0]);
            cvt_B(2 * r0, q1, l1, Q);
            cvt_A(p0, A2[0]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // slot 2 has been read: it may be refilled
        }
        auto mfma_row = [&](const half8 (&Af)[NCO][2], const half8 (&B)[3][NCI][2], auto KY) {
            constexpr int ky = decltype(KY)::value;
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                        for (int u = 0; u < NCI; ++u)
                            acc[t2][u][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                Af[t2][pass == 2 ? 1 : 0], B[kx][u][pass == 1 ? 1 : 0], acc[t2][u][ky * 3 + kx], 0, 0, 0);
        };
        // step i of the column (output row yo = r0 + i): ph = i % 6 -> ring slot i % 3, fragment / raw-register set i % 2.
        //   tap row ky = 0 (x row 2 yo - 1) unfenced, the splits of x row 2 yo + 1 and dY row yo + 1 underneath it;
        //   tap row ky = 1 (x row 2 yo) in nine fenced (pass, kx) sub-blocks with the 2 NIR + 2 NCO vector-memory instructions of
        //   bundle b(yo + 2) dealt out among them (as ONE block between the tap rows they were issued with the matrix pipe idle);
        //   tap row ky = 2 (x row 2 yo + 1) unfenced, the split of x row 2 yo + 2 underneath it.
        auto step = [&](auto PH, int yo) {
            constexpr int ph = decltype(PH)::value, sl = ph % 3, ab = ph % 2, s2 = (ph + 2) % 3;
            f32x4 rb1[NCI][4], rb0[NCI][4];
            float l1[NCI], l0[NCI];
            vm_wait<BUNDLE>();                                          // bundle b(yo) has landed (b(yo + 1) may be in flight)
            read_B(sl, 0, rb1, l1);
            read_B(sl, 1, rb0, l0);
            mfma_row(A2[ab], O[ab], std::integral_constant<int, 0>{});
            cvt_B(2 * yo + 1, rb1, l1, O[ab ^ 1]);
            cvt_A(rawA[ab], A2[ab ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            {
                // bundle b(yo + 2): slot s2 held group yo - 1, read one step ago; rawA[ab] has just been converted
                const int g = yo + 2;
                const float *b0 = uniform_ptr(xn + (size_t)min(max(2 * g + 1, 0), a.H - 1) * a.W);
                const float *b1 = uniform_ptr(xn + (size_t)min(max(2 * g + 2, 0), a.H - 1) * a.W);
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)s2 * SLOTB);
                const int yc = min(g + 1, a.Hd - 1);
                constexpr int NMEM = 2 * NIR + 2 * NCO, PER = (NMEM + 8) / 9;
#pragma unroll
                for (int sb = 0; sb < 9; ++sb) {
                    const int pass = sb / 3, kx = sb % 3;
#pragma unroll
                    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                        for (int u = 0; u < NCI; ++u)
                            acc[t2][u][3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                A2[ab][t2][pass == 2 ? 1 : 0], Q[kx][u][pass == 1 ? 1 : 0], acc[t2][u][3 + kx], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < PER; ++e) {
                        const int m = sb * PER + e;
                        if (m < NIR)
                            dma16(b0, offB[m], dst + m * 1024);
                        else if (m < 2 * NIR)
                            dma16(b1, offB[m - NIR], dst + SETB + (m - NIR) * 1024);
                        else if (m < NMEM) {
                            const int t2 = (m - 2 * NIR) / 2, hf = (m - 2 * NIR) % 2;
                            rawA[ab][t2][hf] = *(const f32x4 *)(ap + (size_t)t2 * 16 * dplane + (size_t)yc * a.Wd + 4 * hf);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            mfma_row(A2[ab], O[ab ^ 1], std::integral_constant<int, 2>{});
            cvt_B(2 * yo + 2, rb0, l0, Q);
            __builtin_amdgcn_sched_barrier(0);
        };
        bundle(r0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        bundle(r0 + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        int yo = r0;
        for (; yo + 6 <= r1; yo += 6) {
            step(std::integral_constant<int, 0>{}, yo);
            step(std::integral_constant<int, 1>{}, yo + 1);
            step(std::integral_constant<int, 2>{}, yo + 2);
            step(std::integral_constant<int, 3>{}, yo + 3);
            step(std::integral_constant<int, 4>{}, yo + 4);
            step(std::integral_constant<int, 5>{}, yo + 5);
        }
        if (yo < r1)
            step(std::integral_constant<int, 0>{}, yo);
        if (yo + 1 < r1)
            step(std::integral_constant<int, 1>{}, yo + 1);
        if (yo + 2 < r1)
            step(std::integral_constant<int, 2>{}, yo + 2);
        if (yo + 3 < r1)
            step(std::integral_constant<int, 3>{}, yo + 3);
        if (yo + 4 < r1)
            step(std::integral_constant<int, 4>{}, yo + 4);
    }
    vm_wait<0>();

    // (w0 + w1) + (w2 + w3) through LDS (the staging rings are done: their memory is reused), one slab per workgroup
    {
        __syncthreads();
        float(*red)[NREG][64] = (float(*)[NREG][64])smem;
        auto put = [&](int b) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            red[b][((t2 * NCI + u) * 9 + k) * 4 + q][lane] = acc[t2][u][k][q];
        };
        auto add = [&](int b) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[t2][u][k][q] += red[b][((t2 * NCI + u) * 9 + k) * 4 + q][lane];
        };
        if (wave & 1)
            put(wave >> 1);
        __syncthreads();
        if (!(wave & 1))
            add(wave >> 1);
        __syncthreads();
        if (wave == 2)
            put(0);
        __syncthreads();
        if (wave != 0)
            return;
        add(0);
    }
    const float inv = 1.0f / (sx * sg);
    float *out = a.part + (size_t)xsplit * 9 * a.Cout * a.Cin;
#pragma unroll
    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = co0 + 16 * t2 + 4 * q4 + q, ci = ci0 + 16 * u + j;
                    if (ci_ok[u])
                        out[((size_t)k * a.Cout + co) * a.Cin + ci] = acc[t2][u][k][q] * inv;
                }
}

struct S2Plan {
    int nco, nci, ncig, npairs, units, S, nx;
};

S2Plan s2_plan(int N, int Cin, int Cout, int H, int W, int force_nco, int force_nci)
{
    const int cot = Cout / 16, cit = Cin / 16, Hd = (H - 1) / 2 + 1, Wd = W / 2;
    S2Plan p;
    // four tiles per wave at most (LDS reduction across the waves, registers); threes when the co tiles allow it
    p.nco = (cot % 3 == 0) ? 3 : (cot % 2 == 0) ? 2 : 1;
    p.nci = 1;
    if (p.nco < 3 && cit % 2 == 0)
        p.nci = 2;
    if (force_nco > 0 && force_nco <= 3 && cot % force_nco == 0) {
        p.nco = force_nco;
        p.nci = force_nci > 0 ? force_nci : 1;
    }
    if (p.nco * p.nci > 4)
        p.nci = 1;
    p.ncig = (cit + p.nci - 1) / p.nci;
    p.npairs = (cot / p.nco) * p.ncig;
    p.units = N * ((Wd + 31) / 32);
    p.nx = 256 / p.npairs;
    if (p.nx < 1)
        p.nx = 1;
    p.S = 4 * p.nx;
    if ((long long)p.S > (long long)p.units * Hd)
        p.S = p.units * Hd;
    p.nx = (p.S + 3) / 4;
    return p;
}

}  // namespace

bool dcl_wgrad_s2_supported(int H, int W) { return W % 16 == 0 && H >= 1; }

static bool g_s2_dma = true;
void dcl_wgrad_s2_set_dma(int on) { g_s2_dma = on != 0; }

int dcl_wgrad_s2_slabs(int N, int Cin, int Cout, int H, int W, int force_nco, int force_nci)
{
    return s2_plan(N, Cin, Cout, H, W, force_nco, force_nci).nx;
}

// kernel only; the caller sums the *nslab slabs (k_wgrad_reduce)
void dcl_wgrad_s2_launch(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W, const float *xamax,
                         int xcount, const float *gamax, int gcount, float *part, int force_nco, int force_nci,
                         hipStream_t s, int *nslab, const float *pre_sc, const float *pre_sh)
{
    const S2Plan p = s2_plan(N, Cin, Cout, H, W, force_nco, force_nci);
    WgradS2Args a;
    a.pre_sc = pre_sc;
    a.pre_sh = pre_sh;
    a.x = x; a.dy = dy; a.part = part; a.xamax = xamax; a.gamax = gamax; a.xcount = xcount; a.gcount = gcount;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.Hd = (H - 1) / 2 + 1;
    a.Wd = W / 2;
    a.strips = (a.Wd + 31) / 32;
    a.units = p.units; a.S = p.S; a.ncig = p.ncig; a.npairs = p.npairs; a.nx = p.nx;
    *nslab = p.nx;
    const dim3 grid((unsigned)(p.npairs * p.nx));
    // LDS-DMA staging of the x rows (k_wgrad3x3_s2d) unless switched off (dcl_wgrad3x3_set_variant(0)) or the rows are too far
    // apart for the 32-bit lane offsets
    // (one ci tile per wave only: with two, three ring slots of four waves do not fit the 160 KB of LDS)
    if (g_s2_dma && p.nci == 1 && (size_t)Cin * H * W * 4 < ((size_t)1 << 32)) {
#define DCL_S2D_CASE(o, i)                                                      \
    if (p.nco == o && p.nci == i) {                                             \
        if (pre_sc)                                                             \
            hipLaunchKernelGGL((k_wgrad3x3_s2d<o, i, true>), grid, dim3(256), 0, s, a);  \
        else                                                                    \
            hipLaunchKernelGGL((k_wgrad3x3_s2d<o, i>), grid, dim3(256), 0, s, a);        \
    }
        DCL_S2D_CASE(3, 1)
        DCL_S2D_CASE(2, 1)
        DCL_S2D_CASE(1, 1)
#undef DCL_S2D_CASE
        dcl_note_kernel(pre_sc ? "k_wgrad3x3_s2d_pre<%d,%d>" : "k_wgrad3x3_s2d<%d,%d>", p.nco, p.nci);
        return;
    }
#define DCL_S2_CASE(o, i, deep)                                                 \
    if (p.nco == o && p.nci == i) {                                             \
        if (pre_sc)                                                             \
            hipLaunchKernelGGL((k_wgrad3x3_s2<o, i, deep, true>), grid, dim3(256), 0, s, a);  \
        else                                                                    \
            hipLaunchKernelGGL((k_wgrad3x3_s2<o, i, deep>), grid, dim3(256), 0, s, a);        \
    }
    DCL_S2_CASE(3, 1, true)
    DCL_S2_CASE(2, 2, false)
    DCL_S2_CASE(2, 1, true)
    DCL_S2_CASE(1, 2, true)
    DCL_S2_CASE(1, 1, true)
#undef DCL_S2_CASE
    dcl_note_kernel(pre_sc ? "k_wgrad3x3_s2_pre<%d,%d>" : "k_wgrad3x3_s2<%d,%d>", p.nco, p.nci);
}
