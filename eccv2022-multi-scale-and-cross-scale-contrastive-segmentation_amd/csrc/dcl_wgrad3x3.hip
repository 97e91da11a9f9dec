// dcl_wgrad3x3.hip -- weight gradient of the 3x3 / stride 1 / pad 1 convolution on the f16 matrix pipe with
// fp32-equivalent arithmetic (f16x3: hi.hi + hi.lo + lo.hi of split-f16 operands, f32 accumulation).
//
//   dw[co, ci, ky, kx] = sum_{n, y, x} dy[n, co, y, x] * x[n, ci, y + ky - 1, x + kx - 1]
//
// GEMM view per tap: D[co][ci] += dY[co][pixel] * X_tap[pixel][ci], K = pixels.  In NCHW both operands have the K
// dimension (pixels of a row) contiguous, which is exactly what v_mfma_f32_16x16x32_f16 wants from a lane:
// lane (q, i) supplies 8 consecutive k for row / column i.  So there is NO LDS staging at all: every lane loads
// its 8 (+2 halo) f32 pixels straight from global memory (32-byte runs, 128-byte lines shared by the four
// k-octets of a wave), scales, splits into f16 (hi, lo) in registers and feeds the MFMA.  The three kx-shifted B
// fragments of an input row are three windows of the same 10 split values.
//
// One wave = NCO x NCI tiles of 16 x 16 (co x ci) x 9 taps of accumulators, walking down a 32-pixel-wide strip:
// per input row r it loads X[r] (B, 3 kx windows) and keeps dY rows r-1, r, r+1 (A) in registers -- row r pairs
// with dY row r + 1 - ky for tap row ky -- so each loaded fragment feeds 9 * NCO (B) or 9 * NCI (A) MFMAs x 3.
// The flat (image, strip, row) sequence is cut into S contiguous runs, one per wave; the four waves of a workgroup
// combine their partial dw through LDS in a fixed order and write one slab, k_wgrad_reduce sums the slabs in fixed
// order (deterministic, no float atomics).
#include <type_traits>

#include "dcl_common.h"
#include "dcl_wgrad.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr float F16_TARGET = 16384.0f;

__device__ __forceinline__ float pow2_scale(float amax)
{
    return amax == 0.f ? 1.f : exp2f(fminf(fmaxf(floorf(log2f(F16_TARGET / amax)), -100.f), 100.f));
}

// Packed f16 pair (lo half = element 0) of hi = f16(v * s) and of lo = f16(v * s - hi) for two values.  s is a power
// of two (or 0), so v * s is exact and the fused form computes the same value; written as v_fma_mix{lo,hi}_f16
// (f32 / f16 inputs, f32 arithmetic, f16 result into one half of the destination): 2 VALU instructions per value
// and no packing.  The compiler's own lowering of the C expression takes 3+ and the kernel is VALU-issue-bound.
__device__ __forceinline__ void split2(float v0, float v1, float s, unsigned &hi, unsigned &lo)
{
    // (mixlo leaves the upper half of its destination alone; mixhi fills it right after, so no initialisation)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(s), "v"(hi));
}

// one value: f16 hi / lo in the low halves of hi / lo (upper halves undefined)
__device__ __forceinline__ void split1(float v0, float s, unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ half8 as_half8(u32x4 v) { return __builtin_bit_cast(half8, v); }

// UPS: dy is the gradient of a STRIDE-2 convolution, i.e. the stride-1 formulation sees it zero-inserted at odd
// coordinates: row y of the virtual dy is row y / 2 of the stored one for even y (else zero), and an octet of 8
// virtual pixels is 4 stored values interleaved with zeros.
template <int NCO, int NCI, bool UPS>
__global__ __launch_bounds__(256, 1) void k_wgrad3x3(WgradArgs a)
{
    __shared__ float wm[8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q4 = lane >> 4, j = lane & 15;

    // operand scales from the producers' partial maxima
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
    // XCD-aware decode of the 1-D grid: consecutive workgroup ids go round-robin over the 8 XCDs, so
    // id = xcd + 8 * (pair + npairs * hi) puts every (co group, ci group) pair of one pixel split on the SAME XCD,
    // next to each other in dispatch order -- they stream the same dy / x rows, which then come out of that
    // XCD's L2 instead of being fetched once per pair.
    int pair, xsplit;
    if (a.rect_mode) {
        // Many tile pairs, one pixel split (large channel counts): every workgroup streams ALL pixels, so what
        // matters is which workgroups share an L2 while they do.  The (co group x ci group) grid is cut into
        // rectangles of rect_c x rect_i = 32 pairs -- they read rect_c + rect_i operand row sets instead of 64 --
        // and rectangle q of XCD k takes the ids k + 8 * (32 q .. 32 q + 31): one XCD (32 CUs), adjacent dispatch
        // slots.  (Head convolution, 15 x 45 pairs: FETCH_SIZE 28.0 GiB -> 16.5 GiB per launch, 12.9 -> 12.5 ms.)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int unit = (slot >> 5) * 8 + xcd, idx = slot & 31;       // unit = (pixel split, rectangle)
        const int ncog = a.npairs / a.ncig, rects_i = (a.ncig + a.rect_i - 1) / a.rect_i;
        const int nrect = ((ncog + a.rect_c - 1) / a.rect_c) * rects_i;
        const int rect = unit % nrect;
        xsplit = unit / nrect;
        const int cg = (rect / rects_i) * a.rect_c + idx / a.rect_i, ci = (rect % rects_i) * a.rect_i + idx % a.rect_i;
        if (xsplit >= a.nx || cg >= ncog || ci >= a.ncig)
            return;                             // padding of the last rectangles (the whole workgroup leaves)
        pair = cg * a.ncig + ci;
    } else {
        const int nx8 = a.nx & ~7, main_blocks = nx8 * a.npairs;
        if ((int)blockIdx.x < main_blocks) {
            const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
            pair = rest % a.npairs;
            xsplit = (rest / a.npairs) * 8 + xcd;
        } else {                                // the nx % 8 left-over pixel splits, in plain order
            const int rest = blockIdx.x - main_blocks;
            pair = rest % a.npairs;
            xsplit = nx8 + rest / a.npairs;
        }
    }
    const int split = xsplit * 4 + wave;           // a wave past the last split gets an empty run and adds zeros
    const int cog = pair / a.ncig, cig = pair - cog * a.ncig;
    const int co0 = cog * NCO * 16, ci0 = cig * NCI * 16;
    const size_t plane = (size_t)a.H * a.W;
    bool ci_ok[NCI];            // the last ci group is ragged when the tile count is not a multiple of NCI
#pragma unroll
    for (int u = 0; u < NCI; ++u)
        ci_ok[u] = ci0 + 16 * u < a.Cin;

    f32x4 acc[NCO][NCI][9];
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k)
                acc[t][u][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this split's share of the flat (image, strip, input row) sequence: rows [t0, t1), walked column by column
    const long long T = (long long)a.units * a.H;              // units = images x strips (columns)
    long long t = min(T, T * split / a.S);
    const long long t1 = min(T, T * (split + 1) / a.S);
    while (t < t1) {
        const int col = (int)(t / a.H);
        const int r0 = (int)(t - (long long)col * a.H);
        const int r1 = (int)min((long long)a.H, r0 + (t1 - t));
        t += r1 - r0;
        const int strip = col % a.strips;
        const int n = col / a.strips;
        const int px = strip * 32 + 8 * q4;
        // Loads are unconditional from clamped (always valid) addresses and masked through the operand scale
        // (0 instead of s) when they are converted one iteration later: a branch around a load would make the
        // compiler wait for it at the join, i.e. before the MFMAs it is supposed to overlap.
        const bool oct_ok = px < a.W;
        const int pxc = oct_ok ? px : a.W - 8;
        const int dl = (oct_ok && px > 0) ? -1 : 0, dr = px + 8 < a.W ? 8 : 7;     // halo offsets from the CLAMPED base
        const float sx_c = oct_ok ? sx : 0.f;                       // elements 1..8 of the B window
        const float sx_l = (oct_ok && px > 0) ? sx : 0.f;           // left halo (outside the image at x = -1)
        const float sx_r = (px + 8 < a.W) ? sx : 0.f;               // right halo
        const size_t dplane = (size_t)a.Hd * a.Wd;
        const float *ap = a.dy + ((size_t)n * a.Cout + co0 + j) * dplane + (UPS ? pxc / 2 : pxc);
        const float *bp = a.x + ((size_t)n * a.Cin + ci0 + j) * plane + pxc;     // + u * 16 * plane + r * W

        auto load_A = [&](int y, f32x4 (&dst)[NCO][2], float &scale) {
            if (UPS) {
                scale = (oct_ok && y >= 0 && !(y & 1) && (y >> 1) < a.Hd) ? sg : 0.f;
                const int yc = min(max(y >> 1, 0), a.Hd - 1);
#pragma unroll
                for (int t2 = 0; t2 < NCO; ++t2)
                    dst[t2][0] = *(const f32x4 *)(ap + (size_t)t2 * 16 * dplane + (size_t)yc * a.Wd);
            } else {
                scale = (oct_ok && y >= 0 && y < a.H) ? sg : 0.f;
                const int yc = min(max(y, 0), a.H - 1);
#pragma unroll
                for (int t2 = 0; t2 < NCO; ++t2) {
                    const float *p = ap + (size_t)t2 * 16 * dplane + (size_t)yc * a.W;
                    dst[t2][0] = *(const f32x4 *)p;
                    dst[t2][1] = *(const f32x4 *)(p + 4);
                }
            }
        };
        auto load_B = [&](int r, f32x4 (&dst)[NCI][2], float (&l)[NCI], float (&rr)[NCI]) {
            const int rc = min(r, a.H - 1);
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                const float *p = bp + (size_t)(ci_ok[u] ? u : 0) * 16 * plane + (size_t)rc * a.W;
                dst[u][0] = *(const f32x4 *)p;
                dst[u][1] = *(const f32x4 *)(p + 4);
                l[u] = p[dl];
                rr[u] = p[dr];
            }
        };
        auto cvt_A = [&](const f32x4 (&src)[NCO][2], float scale, half8 (&dst)[NCO][2]) {
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2) {
                unsigned h[4], l[4];
                if (UPS) {
                    // (a0, 0, a1, 0, a2, 0, a3, 0): one split value in the low half of every pair
                    split1(src[t2][0].x, scale, h[0], l[0]);
                    split1(src[t2][0].y, scale, h[1], l[1]);
                    split1(src[t2][0].z, scale, h[2], l[2]);
                    split1(src[t2][0].w, scale, h[3], l[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        h[e] &= 0xffffu;
                        l[e] &= 0xffffu;
                    }
                    dst[t2][0] = as_half8(u32x4{h[0], h[1], h[2], h[3]});
                    dst[t2][1] = as_half8(u32x4{l[0], l[1], l[2], l[3]});
                    continue;
                }
                split2(src[t2][0].x, src[t2][0].y, scale, h[0], l[0]);
                split2(src[t2][0].z, src[t2][0].w, scale, h[1], l[1]);
                split2(src[t2][1].x, src[t2][1].y, scale, h[2], l[2]);
                split2(src[t2][1].z, src[t2][1].w, scale, h[3], l[3]);
                dst[t2][0] = as_half8(u32x4{h[0], h[1], h[2], h[3]});
                dst[t2][1] = as_half8(u32x4{l[0], l[1], l[2], l[3]});
            }
        };
        // the three kx windows of an input row are windows of the same 10 split values w[-1 .. 8]: kx = 0 and
        // kx = 2 share the pairing (w-1,w0)(w1,w2)...(w7,w8); kx = 1 is the 16-bit funnel shift of neighbours
        auto cvt_B = [&](const f32x4 (&src)[NCI][2], const float (&l)[NCI], const float (&rr)[NCI],
                         half8 (&dst)[3][NCI][2]) {
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                const float sc = ci_ok[u] ? sx_c : 0.f, sl = ci_ok[u] ? sx_l : 0.f, sr = ci_ok[u] ? sx_r : 0.f;
                unsigned h[5], q[5];
                // the halo values have their own scale (0 outside the image): split them alone, then merge
                unsigned hl, ql, hr, qr, hm, qm;
                split1(l[u], sl, hl, ql);                            // w-1
                split1(src[u][0].x, sc, hm, qm);                     // w0
                h[0] = __builtin_amdgcn_perm(hm, hl, 0x05040100u);   // (lo16(hl), lo16(hm))
                q[0] = __builtin_amdgcn_perm(qm, ql, 0x05040100u);
                split2(src[u][0].y, src[u][0].z, sc, h[1], q[1]);
                split2(src[u][0].w, src[u][1].x, sc, h[2], q[2]);
                split2(src[u][1].y, src[u][1].z, sc, h[3], q[3]);
                split1(src[u][1].w, sc, hm, qm);                     // w7
                split1(rr[u], sr, hr, qr);                           // w8
                h[4] = __builtin_amdgcn_perm(hr, hm, 0x05040100u);
                q[4] = __builtin_amdgcn_perm(qr, qm, 0x05040100u);
                u32x4 w0h = {h[0], h[1], h[2], h[3]}, w0l = {q[0], q[1], q[2], q[3]};
                u32x4 w2h = {h[1], h[2], h[3], h[4]}, w2l = {q[1], q[2], q[3], q[4]};
                unsigned a1h[4], a1l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a1h[e] = __builtin_amdgcn_alignbit(h[e + 1], h[e], 16);
                    a1l[e] = __builtin_amdgcn_alignbit(q[e + 1], q[e], 16);
                }
                u32x4 w1h = {a1h[0], a1h[1], a1h[2], a1h[3]}, w1l = {a1l[0], a1l[1], a1l[2], a1l[3]};
                dst[0][u][0] = as_half8(w0h);
                dst[0][u][1] = as_half8(w0l);
                dst[1][u][0] = as_half8(w1h);
                dst[1][u][1] = as_half8(w1l);
                dst[2][u][0] = as_half8(w2h);
                dst[2][u][1] = as_half8(w2l);
            }
        };

        // Software pipeline, unrolled by six so that every register array is indexed statically and nothing is
        // copied between iterations.  dY rows live in a ring of three slots (row y in slot (y - r0 + 1) % 3), X rows
        // in two sets.  Step i (input row r = r0 + i):
        //   1. MFMAs of tap row ky = 2 (dY row r - 1, the oldest slot),
        //   2. the raw values loaded one step ago are split into f16 pairs: dY row r + 2 into that slot, X row
        //      r + 1 into the other B set -- VALU work that overlaps the remaining MFMAs,
        //   3. the loads of dY row r + 3 / X row r + 2 are issued into the raw registers just consumed,
        //   4. MFMAs of tap rows ky = 1, 0 (dY rows r, r + 1).
        half8 A[3][NCO][2], B[2][3][NCI][2];
        f32x4 rawA[NCO][2], rawB[NCI][2];
        float rawL[NCI], rawR[NCI], rawS;
        {
            f32x4 p0[NCO][2], p1[NCO][2], p2[NCO][2], q0[NCI][2];
            float s0, s1, s2, l0[NCI], rr0[NCI];
            load_A(r0 - 1, p0, s0);
            load_A(r0, p1, s1);
            load_A(r0 + 1, p2, s2);
            load_B(r0, q0, l0, rr0);
            load_A(r0 + 2, rawA, rawS);
            load_B(r0 + 1, rawB, rawL, rawR);
            cvt_A(p0, s0, A[0]);
            cvt_A(p1, s1, A[1]);
            cvt_A(p2, s2, A[2]);
            cvt_B(q0, l0, rr0, B[0]);
        }
        auto mfma_row = [&](auto SLOT, auto BSET, auto KY) {
            constexpr int slot = decltype(SLOT)::value, bs = decltype(BSET)::value, ky = decltype(KY)::value;
            // pass-major: all tiles hi.hi, then all hi.lo, then all lo.hi -- the three MFMAs of one accumulator are
            // 9 NCI instructions apart and each accumulates in place (dst == srcC)
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                        for (int u = 0; u < NCI; ++u)
                            acc[t2][u][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                A[slot][t2][pass == 2 ? 1 : 0], B[bs][kx][u][pass == 1 ? 1 : 0],
                                acc[t2][u][ky * 3 + kx], 0, 0, 0);
        };
        auto step = [&](auto PH, int r) {
            constexpr int ph = decltype(PH)::value;
            using I = std::integral_constant<int, ph % 3>;             // slot of dY row r - 1
            using I1 = std::integral_constant<int, (ph + 1) % 3>;      // row r
            using I2 = std::integral_constant<int, (ph + 2) % 3>;      // row r + 1
            using BS = std::integral_constant<int, ph % 2>;
            mfma_row(I{}, BS{}, std::integral_constant<int, 2>{});
            cvt_A(rawA, rawS, A[ph % 3]);
            cvt_B(rawB, rawL, rawR, B[(ph + 1) % 2]);
            // the loads stay between the two MFMA groups (left alone the scheduler sinks them to their use, one
            // step later, and the wave eats the memory latency every row)
            __builtin_amdgcn_sched_barrier(0);
            load_A(r + 3, rawA, rawS);
            load_B(r + 2, rawB, rawL, rawR);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(I1{}, BS{}, std::integral_constant<int, 1>{});
            mfma_row(I2{}, BS{}, std::integral_constant<int, 0>{});
        };
        int r = r0;
        for (; r + 6 <= r1; r += 6) {
            step(std::integral_constant<int, 0>{}, r);
            step(std::integral_constant<int, 1>{}, r + 1);
            step(std::integral_constant<int, 2>{}, r + 2);
            step(std::integral_constant<int, 3>{}, r + 3);
            step(std::integral_constant<int, 4>{}, r + 4);
            step(std::integral_constant<int, 5>{}, r + 5);
        }
        if (r < r1)
            step(std::integral_constant<int, 0>{}, r);
        if (r + 1 < r1)
            step(std::integral_constant<int, 1>{}, r + 1);
        if (r + 2 < r1)
            step(std::integral_constant<int, 2>{}, r + 2);
        if (r + 3 < r1)
            step(std::integral_constant<int, 3>{}, r + 3);
        if (r + 4 < r1)
            step(std::integral_constant<int, 4>{}, r + 4);
    }

    // The four waves of a workgroup hold partial sums of the SAME (co, ci) tiles: they are combined through LDS in
    // a fixed order, (w0 + w1) + (w2 + w3), so that one slab per workgroup (not per wave) goes to memory.
    // (Only for the variants with <= 3 tiles per tap: with 6 the extra live ranges make the register allocator spill
    // inside the main loop, so those keep one slab per wave.)
    constexpr bool LDSRED = NCO * NCI <= 4;
    if (LDSRED) {
        constexpr int NREG = LDSRED ? NCO * NCI * 36 : 1;
        __shared__ float red[2][NREG][64];
        auto put = [&](int b) {
#pragma unroll
            for (int t = 0; t < NCO; ++t)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            red[b][((t * NCI + u) * 9 + k) * 4 + q][lane] = acc[t][u][k][q];
        };
        auto add = [&](int b) {
#pragma unroll
            for (int t = 0; t < NCO; ++t)
#pragma unroll
                for (int u = 0; u < NCI; ++u)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[t][u][k][q] += red[b][((t * NCI + u) * 9 + k) * 4 + q][lane];
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
    else if (split >= a.S)
        return;
    // slab [xsplit | split][tap][co][ci]; accumulator register q of lane (q4, j) is (co = 4 q4 + q, ci = j) of its tile
    const float inv = 1.0f / (sx * sg);
    float *out = a.part + (size_t)(LDSRED ? xsplit : split) * 9 * a.Cout * a.Cin;
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = co0 + 16 * t + 4 * q4 + q, ci = ci0 + 16 * u + j;
                    if (ci_ok[u])
                        out[((size_t)k * a.Cout + co) * a.Cin + ci] = acc[t][u][k][q] * inv;
                }
}

// dw[co][ci][tap] = sum_s part[s][tap][co][ci].  Block = 32 outputs x 8 slab groups: group g adds slabs g, g+8, ...
// in order, the 8 partial sums are combined in order through LDS -> fixed summation tree, deterministic.
template <int K>
__global__ __launch_bounds__(256) void k_wgrad_reduce_t(const float *__restrict__ part, int S, int Cout, int Cin,
                                                       float *__restrict__ dw)
{
    // K groups of 32 outputs per workgroup: 1 for the narrow layers (many slabs, few outputs: parallelism over the
    // outputs matters), 4 for the wide ones (with one, the 1.3 M outputs of a 384-channel layer were 41 k workgroups of
    // a few loads each -- 16 us, bound by the workgroup dispatch rate)
    __shared__ float sh[K][8][32];
    const int total = 9 * Cout * Cin;
    const int lane = threadIdx.x & 31, g = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int idx = (blockIdx.x * K + k) * 32 + lane;       // (tap, co, ci) in slab order
        float s0 = 0.f, s1 = 0.f;
        if (idx < total) {
            int q = g;
            for (; q + 8 < S; q += 16) {
                s0 += part[(size_t)q * total + idx];
                s1 += part[(size_t)(q + 8) * total + idx];
            }
            if (q < S)
                s0 += part[(size_t)q * total + idx];
        }
        sh[k][g][lane] = s0 + s1;
    }
    __syncthreads();
    if (g < K) {
        const int idx = (blockIdx.x * K + g) * 32 + lane;
        if (idx < total) {
            float s = sh[g][0][lane];
#pragma unroll
            for (int q = 1; q < 8; ++q)
                s += sh[g][q][lane];
            const int ci = idx % Cin;
            const int co = (idx / Cin) % Cout;
            const int tap = idx / (Cin * Cout);
            dw[((size_t)co * Cin + ci) * 9 + tap] = s;
        }
    }
}

// Wide layers (few slabs, many outputs): one workgroup per (co, 32 ci), thread = (tap, ci).  The slab layout
// [tap][co][ci] is read in 128-byte runs as before, but dw[co][ci][tap] is written through an LDS tile as ONE contiguous
// 1152-byte run -- the per-output form writes 4 bytes every 36 (a 32-byte sector per value: 42 MB of write traffic for the
// 5 MB gradient of a 384-channel layer).  Same slab order: bitwise the same sums when S <= 8 (one group per output).
__global__ __launch_bounds__(320) void k_wgrad_reduce_wide(const float *__restrict__ part, int S, int Cout, int Cin,
                                                          float *__restrict__ dw)
{
    __shared__ float tile[32 * 9];
    const int co = blockIdx.y, ci0 = blockIdx.x * 32;
    const int t = threadIdx.x, tap = t >> 5, cl = t & 31;
    const size_t total = (size_t)9 * Cout * Cin;
    if (t < 288) {
        float s0 = 0.f;
        if (ci0 + cl < Cin) {
            const float *p = part + ((size_t)tap * Cout + co) * Cin + ci0 + cl;
            for (int q = 0; q < S; ++q)
                s0 += p[(size_t)q * total];
        }
        tile[cl * 9 + tap] = s0;
    }
    __syncthreads();
    const int nci = min(32, Cin - ci0);
    if (t < nci * 9)
        dw[((size_t)co * Cin + ci0) * 9 + t] = tile[t];
}

static void launch_wgrad_reduce(const float *part, int S, int Cout, int Cin, float *dw, hipStream_t st)
{
    const int total = 9 * Cout * Cin;
    if (total >= (1 << 18) && S <= 8)
        hipLaunchKernelGGL(k_wgrad_reduce_wide, dim3((Cin + 31) / 32, Cout), dim3(320), 0, st, part, S, Cout, Cin, dw);
    else if (total >= (1 << 18))
        hipLaunchKernelGGL(k_wgrad_reduce_t<4>, dim3((total + 127) / 128), dim3(256), 0, st, part, S, Cout, Cin, dw);
    else
        hipLaunchKernelGGL(k_wgrad_reduce_t<1>, dim3((total + 31) / 32), dim3(256), 0, st, part, S, Cout, Cin, dw);
}

}  // namespace

static int g_tile_nco = 0, g_tile_nci = 0;      // tuning override (dcl_wgrad3x3_set_tile), 0 = automatic
static int g_force_nx = 0;                      // tuning override: pixel splits per tile pair (0 = automatic)
static int g_wg_target = 256;                   // workgroups a launch aims at (pixel splits = target / tile pairs): one per CU;
                                                // tuning hook dcl_wgrad3x3_set_workgroup_target for in-step A/B runs, where the
                                                // launch shares the chip with the other branches' kernels
static int g_variant = -1;     // 0 = MFMA-order operand loads (this file), -1 / 2 = LDS-DMA staging (dcl_wgrad3x3d.hip)

// (A third variant -- workgroups handing the dY rows of a co group to each other through LDS, optional stream-K partition
// -- reached the same kernel times as the LDS-DMA kernel with slabs four times the size and lost in the training step; it
// was retired from the library in round 3: tools/probes/retired/dcl_wgrad3x3s.hip.)
// per-wave kernel: operands staged by LDS-DMA (dcl_wgrad3x3d.hip) unless variant 0 asks for the direct loads; the direct
// loads also serve tensors beyond the DMA kernel's 32-bit offsets and the zero-inserted stride-2 form
static bool use_dma(int Cin, int H, int W) { return g_variant != 0 && (size_t)Cin * H * W * 4 < ((size_t)1 << 32); }

// dcl_wgrad3x3_s2.hip
bool dcl_wgrad_s2_supported(int H, int W);
void dcl_wgrad_s2_set_dma(int on);
int dcl_wgrad_s2_slabs(int N, int Cin, int Cout, int H, int W, int force_nco, int force_nci);
void dcl_wgrad_s2_launch(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W, const float *xamax,
                         int xcount, const float *gamax, int gcount, float *part, int force_nco, int force_nci,
                         hipStream_t s, int *nslab, const float *pre_sc = nullptr, const float *pre_sh = nullptr);
static int g_s2_native = 1;     // stride 2: 1 = output-pixel formulation (dcl_wgrad3x3_s2.hip), 0 = zero-inserted dy

static int g_wave_band = 0;     // wave form: rows per column of the traversal (0 = whole strips: the default -- 32-row bands
                                // re-use the halo lines in L2 but pay a column prologue every 32 rows: 2.01 vs 1.91 ms on the head)
static int g_strip_group = 1;   // waves of a workgroup on adjacent strips (WgradArgs::grp)
static int g_wave_mode = 2;     // 129 .. 256 tile pairs of the (3, 1) tile: pixel splits dealt out to waves (dcl_wgrad3x3d.hip);
                                // 2 = a workgroup's waves take the same split of four neighbouring pairs, 1 = four splits of a pair

static void wgrad_plan(int N, int Cin, int Cout, int H, int W, int &nco, int &nci, int &S, int &units, bool *wave_mode = nullptr)
{
    const int cot = Cout / 16, cit = Cin / 16;
    // Tiles per wave, measured on the HRNet-W48 shapes at batch 12 (tools/wgrad_tiles.py, and bench.py with
    // DCL_WGRAD_TILE: the step decides, the slabs of the larger tiles cost HBM traffic that the standalone timing does not
    // show): three co tiles x one ci tile wherever the co tiles divide by three (48 ... 720 channels: step 122.3 ms
    // against 122.8 with (2, 2) at 192 and (3, 2) at 384 channels); else the four-tile wave (2, 2), (2, 1), (1, 2), (1, 1).
    if (cot % 3 == 0) {
        nco = 3;
        nci = 1;
    } else {
        nco = (cot % 2 == 0) ? 2 : 1;
        nci = (cit % 2 == 0) ? 2 : 1;
    }
    if (g_tile_nco > 0 && g_tile_nco <= 3 && cot % g_tile_nco == 0)
        nco = g_tile_nco;
    if (g_tile_nci > 0)
        nci = g_tile_nci;
    const int pairs = (cot / nco) * ((cit + nci - 1) / nci);
    units = N * ((W + 31) / 32);            // columns: (image, 32-pixel strip), H input rows each
    // One workgroup (4 waves = 4 splits of one pair) per CU is all that fits (a wave owns most of its SIMD's
    // registers): at most 256 workgroups, or the stragglers run as a second round and double the kernel time.
    int nx = g_wg_target / pairs;
    if (g_force_nx > 0)
        nx = g_force_nx;
    if (nx < 1)
        nx = 1;         // more tile pairs than CUs (head convolution): one pixel split; finer splits were tried and
                        // lose (3 splits fill the last round better but take 26 ms against 12.5 ms)
    S = 4 * nx;
    if ((long long)S > (long long)units * H)
        S = units * H;
    if (wave_mode) {
        // one workgroup per pair and CUs left over: the splits go to single waves, S = 128 / (pairs of one XCD) of them
        const int sw = 128 / ((pairs + 7) / 8);
        *wave_mode = g_wave_mode && g_force_nx == 0 && pairs > 128 && pairs <= 256 && sw > 4 &&
                     dcl_wgrad_dma_wave_mode_supported(nco, nci) && (long long)sw <= (long long)units * H;
        if (*wave_mode)
            S = sw;
    }
}

extern "C" int dcl_wgrad3x3_set_tile(int nco, int nci)
{
    if (nco < 0 || nco > 5 || nco == 4 || nci < 0 || nci > 2 || (nco == 3 && nci == 2))      // (the six-tile wave spilled: retired)
        return DCL_EINVAL;
    g_tile_nco = nco;
    g_tile_nci = nci;
    return 0;
}

extern "C" int dcl_wgrad3x3_set_variant(int variant)
{
    if (variant < -1 || variant > 2 || variant == 1)     // 1 was the retired kernel
        return DCL_EINVAL;
    g_variant = variant;
    dcl_wgrad_s2_set_dma(variant != 0);      // the stride-2 kernel's LDS-DMA form follows the same switch
    return 0;
}

extern "C" int dcl_wgrad3x3_set_workgroup_target(int n)
{
    g_wg_target = n >= 32 ? n : 256;
    return 0;
}

extern "C" int dcl_wgrad3x3_set_splits(int nx)
{
    g_force_nx = nx > 0 ? nx : 0;
    return 0;
}

extern "C" int dcl_wgrad3x3_set_wave_mode(int on)
{
    g_wave_mode = on < 0 ? 0 : on;
    return 0;
}

extern "C" int dcl_wgrad3x3_set_wave_band(int rows)
{
    g_wave_band = rows > 0 ? rows : 0;
    return 0;
}

extern "C" int dcl_wgrad3x3_set_strip_group(int on)
{
    g_strip_group = on ? 1 : 0;
    return 0;
}

extern "C" int dcl_wgrad3x3_set_stride2(int native)
{
    if (native < 0 || native > 2)
        return DCL_EINVAL;
    g_s2_native = native ? 1 : 0;
    dcl_wgrad_s2_set_dma(native != 2);       // 2: the round-4 kernel (every operand loaded in MFMA order)
    return 0;
}

extern "C" int dcl_wgrad3x3_splits(int N, int Cin, int Cout, int H, int W, int stride)
{
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (Cin & 15) || (Cout & 15) || stride < 1 || stride > 2)
        return 0;
    if (stride == 2 && g_s2_native && dcl_wgrad_s2_supported(H, W))
        return dcl_wgrad_s2_slabs(N, Cin, Cout, H, W, g_tile_nco, g_tile_nci);
    int nco, nci, S, units;
    bool wm = false;
    wgrad_plan(N, Cin, Cout, H, W, nco, nci, S, units, stride == 1 && use_dma(Cin > Cout ? Cin : Cout, H, W) ? &wm : nullptr);
    if (wm)
        return S;                                  // one slab per wave-level split
    return nco * nci <= 4 ? (S + 3) / 4 : S;      // one slab per workgroup (4 splits), or per wave (6-tile variant)
}

static int wgrad3x3_impl(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W, const float *xamax, int xcount,
                         const float *gamax, int gcount, int stride, float *part, float *dw, void *stream, const float *pre_sc,
                         const float *pre_sh);

extern "C" int dcl_wgrad3x3_f16x3(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W,
                                  const float *xamax, int xcount, const float *gamax, int gcount, int stride,
                                  float *part, float *dw, void *stream)
{
    return wgrad3x3_impl(x, dy, N, Cin, Cout, H, W, xamax, xcount, gamax, gcount, stride, part, dw, stream, nullptr, nullptr);
}

// 1 when dcl_wgrad3x3_pre_f16x3 serves the shape: stride 1 on the LDS-DMA kernel, stride 2 on the output-pixel formulation
// (the slab count is dcl_wgrad3x3_splits(..., stride))
extern "C" int dcl_wgrad3x3_pre_supported(int N, int Cin, int Cout, int H, int W, int stride)
{
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (Cin & 15) || (Cout & 15) || (W & 7))
        return 0;
    if (stride == 2)
        return g_s2_native && dcl_wgrad_s2_supported(H, W) ? 1 : 0;
    if (stride != 1)
        return 0;
    int nco, nci, S, units;
    wgrad_plan(N, Cin, Cout, H, W, nco, nci, S, units);
    return use_dma(Cin > Cout ? Cin : Cout, H, W) && dcl_wgrad_dma_supported(nco, nci) ? 1 : 0;
}

// The weight gradient of conv2d(relu(x * pre_sc[ci] + pre_sh[ci]), w, padding = 1) for the output gradient dy: x is the RAW
// tensor in front of the norm, xamax the absmax slots of the mapped tensor (dcl_bn_finalize_pre); see k_wgrad3x3d, PRE.
extern "C" int dcl_wgrad3x3_pre_f16x3(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W,
                                      const float *xamax, int xcount, const float *gamax, int gcount, const float *pre_sc,
                                      const float *pre_sh, int stride, float *part, float *dw, void *stream)
{
    DCL_CHECK_ARG(pre_sc && pre_sh, "null pointer");
    if (!dcl_wgrad3x3_pre_supported(N, Cin, Cout, H, W, stride)) {
        dcl_set_error("dcl_wgrad3x3_pre_f16x3: no kernel with the map for this shape / stride");
        return DCL_EUNSUPPORTED;
    }
    return wgrad3x3_impl(x, dy, N, Cin, Cout, H, W, xamax, xcount, gamax, gcount, stride, part, dw, stream, pre_sc, pre_sh);
}

static int wgrad3x3_impl(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W, const float *xamax, int xcount,
                         const float *gamax, int gcount, int stride, float *part, float *dw, void *stream, const float *pre_sc,
                         const float *pre_sh)
{
    DCL_CHECK_ARG(stride == 1 || stride == 2, "stride must be 1 or 2");
    DCL_CHECK_ARG(x && dy && xamax && gamax && part && dw, "null pointer");
    DCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && xcount > 0 && gcount > 0, "bad shape");
    DCL_CHECK_ARG(Cin > 0 && Cout > 0 && (Cin & 15) == 0 && (Cout & 15) == 0, "channel counts must be multiples of 16");
    DCL_CHECK_ARG((W & 7) == 0, "W must be a multiple of 8");
    DCL_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)dy)) & 15) == 0, "tensors must be 16-byte aligned");
    if (stride == 2 && g_s2_native && dcl_wgrad_s2_supported(H, W)) {
        int nslab = 0;
        dcl_wgrad_s2_launch(x, dy, N, Cin, Cout, H, W, xamax, xcount, gamax, gcount, part, g_tile_nco, g_tile_nci,
                            (hipStream_t)stream, &nslab, pre_sc, pre_sh);
        DCL_LAUNCH_CHECK();
        const int total = 9 * Cout * Cin;
        (void)total;
        launch_wgrad_reduce(part, nslab, Cout, Cin, dw, (hipStream_t)stream);
        DCL_LAUNCH_CHECK();
        return 0;
    }
    WgradArgs a;
    a.x = x;
    a.dy = dy;
    a.part = part;
    a.xamax = xamax;
    a.gamax = gamax;
    a.xcount = xcount;
    a.gcount = gcount;
    a.N = N;
    a.Cin = Cin;
    a.Cout = Cout;
    a.H = H;
    a.W = W;
    a.Hd = stride == 2 ? (H - 1) / 2 + 1 : H;
    a.Wd = stride == 2 ? W / 2 : W;
    a.strips = (W + 31) / 32;
    a.nseg = 1;
    a.pre_sc = pre_sc;
    a.pre_sh = pre_sh;
    int nco, nci;
    bool wm = false;
    wgrad_plan(N, Cin, Cout, H, W, nco, nci, a.S, a.units,
               stride == 1 && use_dma(Cin > Cout ? Cin : Cout, H, W) ? &wm : nullptr);
    a.wave_mode = wm ? g_wave_mode : 0;
    // adjacent strips for the waves of a workgroup (dcl_wgrad3x3d.hip): LDS-reduced tiles, unclamped split count, strips in fours / pairs
    a.grp = 1;
    a.band = (wm && g_wave_band > 0 && H % g_wave_band == 0 && H > g_wave_band) ? g_wave_band : H;
    if (g_strip_group && !wm && nco * nci <= 4 && a.S == 4 * ((a.S + 3) / 4) && a.S >= 4)
        a.grp = (a.strips % 4 == 0) ? 4 : ((a.strips % 2 == 0) ? 2 : 1);
    a.ncig = (Cin / 16 + nci - 1) / nci;
    a.npairs = (Cout / 16 / nco) * a.ncig;
    a.nx = (a.S + 3) / 4;                                    // workgroups per pair (4 splits each)
    dim3 grid((unsigned)(a.npairs * a.nx));      // exactly the populated workgroups, <= 256 whenever pairs <= 256
    a.rect_i = a.ncig >= 16 ? 16 : (a.ncig >= 8 ? 8 : (a.ncig >= 4 ? 4 : (a.ncig >= 2 ? 2 : 1)));
    a.rect_c = 32 / a.rect_i;
    a.rect_mode = a.npairs > 128 ? 1 : 0;
    if (a.rect_mode) {
        const int ncog = a.npairs / a.ncig;
        const int units_r = ((ncog + a.rect_c - 1) / a.rect_c) * ((a.ncig + a.rect_i - 1) / a.rect_i) * a.nx;
        grid = dim3((unsigned)(((units_r + 7) / 8) * 8 * 32));
    }
    hipStream_t s = (hipStream_t)stream;
    if (stride == 1 && use_dma(Cin > Cout ? Cin : Cout, H, W) && dcl_wgrad_dma_supported(nco, nci)) {
        dcl_wgrad_dma_launch(a, nco, nci, grid, s);
        DCL_LAUNCH_CHECK();
        const int total = 9 * Cout * Cin;
        (void)total;
        launch_wgrad_reduce(part, (nco * nci <= 4 && !wm) ? a.nx : a.S, Cout, Cin, dw, s);
        DCL_LAUNCH_CHECK();
        return 0;
    }
#define DCL_WG_CASE(o, i)                                                        \
    if (nco == o && nci == i) {                                                  \
        if (stride == 2)                                                         \
            hipLaunchKernelGGL((k_wgrad3x3<o, i, true>), grid, dim3(256), 0, s, a);  \
        else                                                                     \
            hipLaunchKernelGGL((k_wgrad3x3<o, i, false>), grid, dim3(256), 0, s, a); \
    }
    DCL_WG_CASE(2, 2)
    DCL_WG_CASE(1, 2)
    DCL_WG_CASE(3, 1)
    DCL_WG_CASE(2, 1)
    DCL_WG_CASE(1, 1)
#undef DCL_WG_CASE
    dcl_note_kernel("k_wgrad3x3<%d,%d,%s>", nco, nci, stride == 2 ? "true" : "false");
    DCL_LAUNCH_CHECK();
    const int total = 9 * Cout * Cin;
    (void)total;
    launch_wgrad_reduce(part, nco * nci <= 4 ? a.nx : a.S, Cout, Cin, dw, s);
    DCL_LAUNCH_CHECK();
    return 0;
}
