// dcl_wgrad3x3s.hip -- weight gradient of the 3x3 / pad 1 convolution, "shared dY" variant (round 2).
//
//   dw[co, ci, ky, kx] = sum_{n, y, x} dy[n, co, y, x] * x[n, ci, y + ky - 1, x + kx - 1]
//
// Same arithmetic and operand flow per wave as dcl_wgrad3x3.hip (f16x3 on v_mfma_f32_16x16x32_f16, K = 8 consecutive
// pixels of a row per lane, three kx windows out of one 10-value row fragment, rows of dY reused for the three ky).
// What changed is who loads what.  SQ counters on the per-wave kernel: 30-36 % of the wave cycles parked in s_waitcnt,
// matrix pipe 40-45 % busy: every wave pulls BOTH operands' rows straight through the vector-memory path -- 4 waves x
// 14 wave-loads x 16 lines per row step = ~88 B/cycle/CU against the ~64 B/cycle that path delivers.  Here a workgroup
// is (one pixel run) x (one group of NCO co tiles) x (4 waves, each owning NCI ci tiles):
//   * the dY rows of the co group are loaded, scaled and split ONCE per workgroup (tile t by wave t % 4) and parked
//     in LDS exactly in MFMA-fragment order (64 lanes x 16 B per (tile, hi | lo)): a 4-row ring, lane-linear
//     ds_write_b128 / ds_read_b128 -- conflict-free -- shared by the four waves;
//   * every wave loads only ITS X rows from memory (they are private to its ci tiles);
//   -> 2 NCO + 4 * 4 NCI wave-loads per row step and workgroup instead of 4 * (2 NCO + 4 NCI), a quarter of the dY
//      split VALU, and no A-fragment ring in registers (the six-tile wave (3, 2) fits without spills);
//   * the four waves accumulate DIFFERENT (co, ci) tiles over the same pixels, so nothing is reduced across waves:
//     one slab per pixel split, summed by k_wgrad_reduce in fixed order (deterministic).
// One s_barrier per row step separates the producers' write of row r + 2 from its first read (step r + 1) and the last
// read of row r - 2 (step r - 1) from its overwrite.
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

struct WgradSArgs {
    const float *x, *dy;
    float *part;                 // [nx][9][Cout][Cin]
    const float *xamax, *gamax;
    int xcount, gcount;
    int N, Cin, Cout, H, W;
    int Hd, Wd;                  // stored size of dy (stride 2: the even samples of the zero-inserted gradient)
    int strips, units, ncib, ptypes;
    int nwg, nx;                 // workgroups; nx > 0: nwg = ptypes * nx equal pixel splits (XCD-aware decode), 0: stream-K
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

// Work partition ("stream-K" over the pixel dimension): the ptypes x T (workgroup type, row step) items, type-major, are
// cut into nwg equal contiguous ranges; workgroup b owns items [sk_lo(b), sk_lo(b + 1)).  A range that crosses a type
// boundary is processed as two segments with separate accumulators; segment `b - sk_owner(first item of p)` of type p
// goes to that slab, so slab s of a type is always written by the same workgroup -> fixed summation order.
__host__ __device__ __forceinline__ long long sk_lo(long long b, long long wtot, int nwg) { return b * wtot / nwg; }

// workgroup that owns `item`, from a nearby guess; comparisons only (lo(b) <= item  <=>  b * wtot < (item + 1) * nwg)
__host__ __device__ __forceinline__ int sk_owner(long long item, int guess, long long wtot, int nwg)
{
    int b = guess < 0 ? 0 : (guess > nwg - 1 ? nwg - 1 : guess);
    const long long rhs = (item + 1) * nwg;
    while (b + 1 < nwg && (long long)(b + 1) * wtot < rhs)
        ++b;
    while (b > 0 && (long long)b * wtot >= rhs)
        --b;
    return b;
}

// first / last workgroup of type p (its items are [p T, (p + 1) T); wtot = ptypes * T)
__host__ __device__ __forceinline__ int sk_first(int p, int ptypes, long long T, long long wtot, int nwg)
{
    return sk_owner((long long)p * T, (int)((unsigned)(p * nwg) / (unsigned)ptypes), wtot, nwg);       // p * nwg < 2^31
}

__host__ __device__ __forceinline__ int sk_last(int p, int ptypes, long long T, long long wtot, int nwg)
{
    return sk_owner((long long)(p + 1) * T - 1, (int)((unsigned)((p + 1) * nwg) / (unsigned)ptypes), wtot, nwg);
}

template <int NCO, int NCI, bool UPS>
__global__ __launch_bounds__(256, 1) void k_wgrad3x3s(WgradSArgs a)
{
    __shared__ float wm[8];
    __shared__ __attribute__((aligned(16))) u32x4 Alds[4][NCO][2][64];       // [row slot][co tile][hi | lo][lane]
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
    const long long T = (long long)a.units * a.H, wtot = T * a.ptypes;
    // workgroup number b in type-major order.  Equal splits (nx > 0): XCD-aware decode (consecutive ids go round-robin
    // over the 8 XCDs) -- the `ptypes` workgroups of one pixel split get the same XCD and adjacent dispatch slots, they
    // stream the same dy / x rows out of one L2
    int b = blockIdx.x;
    if (a.nx > 0) {
        int ptype, xsplit;
        const int nx8 = a.nx & ~7, main_blocks = nx8 * a.ptypes;
        if ((int)blockIdx.x < main_blocks) {
            const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
            ptype = rest % a.ptypes;
            xsplit = (rest / a.ptypes) * 8 + xcd;
        } else {
            const int rest = blockIdx.x - main_blocks;
            ptype = rest % a.ptypes;
            xsplit = nx8 + rest / a.ptypes;
        }
        b = ptype * a.nx + xsplit;
    }
    const size_t plane = (size_t)a.H * a.W;
    // co tiles this wave produces (loads, splits, parks in LDS): t with t % 4 == wave
    constexpr int NMY = (NCO + 3) / 4;
    const float inv = 1.0f / (sx * sg);

    f32x4 acc[NCO][NCI][9];
    long long item = sk_lo(b, wtot, a.nwg);
    const long long item_end = sk_lo(b + 1, wtot, a.nwg);
    while (item < item_end) {           // one segment = one workgroup type (co group, ci block), a run of row steps
    const int ptype = (int)(item / T);
    long long t = item - (long long)ptype * T;
    const long long t1 = min(T, t + (item_end - item));
    item += t1 - t;
    const int cog = ptype / a.ncib, cib = ptype - cog * a.ncib;
    const int co0 = cog * NCO * 16;
    const int ci0 = (cib * 4 + wave) * NCI * 16;
    bool ci_ok[NCI];            // a wave past the last ci tile computes on zeroed operands and stores nothing
#pragma unroll
    for (int u = 0; u < NCI; ++u)
        ci_ok[u] = ci0 + 16 * u < a.Cin;
#pragma unroll
    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
        for (int u = 0; u < NCI; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k)
                acc[t2][u][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    while (t < t1) {
        const int col = (int)(t / a.H);
        const int r0 = (int)(t - (long long)col * a.H);
        const int r1 = (int)min((long long)a.H, r0 + (t1 - t));
        t += r1 - r0;
        const int strip = col % a.strips;
        const int n = col / a.strips;
        const int px = strip * 32 + 8 * q4;
        const bool oct_ok = px < a.W;
        const int pxc = oct_ok ? px : a.W - 8;
        const int dl = (oct_ok && px > 0) ? -1 : 0, dr = px + 8 < a.W ? 8 : 7;
        const float sx_c = oct_ok ? sx : 0.f;
        const float sx_l = (oct_ok && px > 0) ? sx : 0.f;
        const float sx_r = (px + 8 < a.W) ? sx : 0.f;
        const size_t dplane = (size_t)a.Hd * a.Wd;
        const float *ap = a.dy + ((size_t)n * a.Cout + co0 + j) * dplane + (UPS ? pxc / 2 : pxc);
        const float *bp = a.x + ((size_t)n * a.Cin + (ci_ok[0] ? ci0 : 0) + j) * plane + pxc;      // idle wave: tile 0

        // ---- producer side: dY row y of my tiles -> raw registers -> (scaled, split) fragments in LDS slot (y + 1) & 3
        auto load_A = [&](int y, f32x4 (&dst)[NMY][2], float &scale) {
            if (UPS) {
                scale = (oct_ok && y >= 0 && !(y & 1) && (y >> 1) < a.Hd) ? sg : 0.f;
                const int yc = min(max(y >> 1, 0), a.Hd - 1);
#pragma unroll
                for (int m = 0; m < NMY; ++m) {
                    const int t2 = min(wave + 4 * m, NCO - 1);
                    dst[m][0] = *(const f32x4 *)(ap + (size_t)t2 * 16 * dplane + (size_t)yc * a.Wd);
                }
            } else {
                scale = (oct_ok && y >= 0 && y < a.H) ? sg : 0.f;
                const int yc = min(max(y, 0), a.H - 1);
#pragma unroll
                for (int m = 0; m < NMY; ++m) {
                    const int t2 = min(wave + 4 * m, NCO - 1);
                    const float *p = ap + (size_t)t2 * 16 * dplane + (size_t)yc * a.W;
                    dst[m][0] = *(const f32x4 *)p;
                    dst[m][1] = *(const f32x4 *)(p + 4);
                }
            }
        };
        auto park_A = [&](int y, const f32x4 (&src)[NMY][2], float scale) {
            const int slot = (y + 1) & 3;
#pragma unroll
            for (int m = 0; m < NMY; ++m) {
                const int t2 = wave + 4 * m;
                if (t2 >= NCO)
                    continue;
                unsigned h[4], l[4];
                if (UPS) {
                    split1(src[m][0].x, scale, h[0], l[0]);
                    split1(src[m][0].y, scale, h[1], l[1]);
                    split1(src[m][0].z, scale, h[2], l[2]);
                    split1(src[m][0].w, scale, h[3], l[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        h[e] &= 0xffffu;
                        l[e] &= 0xffffu;
                    }
                } else {
                    split2(src[m][0].x, src[m][0].y, scale, h[0], l[0]);
                    split2(src[m][0].z, src[m][0].w, scale, h[1], l[1]);
                    split2(src[m][1].x, src[m][1].y, scale, h[2], l[2]);
                    split2(src[m][1].z, src[m][1].w, scale, h[3], l[3]);
                }
                Alds[slot][t2][0][lane] = u32x4{h[0], h[1], h[2], h[3]};
                Alds[slot][t2][1][lane] = u32x4{l[0], l[1], l[2], l[3]};
            }
        };
        // ---- consumer side: X rows of my ci tiles, private to this wave (as in the per-wave kernel)
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
        auto cvt_B = [&](const f32x4 (&src)[NCI][2], const float (&l)[NCI], const float (&rr)[NCI],
                         half8 (&dst)[3][NCI][2]) {
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                const float sc = ci_ok[u] ? sx_c : 0.f, sl = ci_ok[u] ? sx_l : 0.f, sr = ci_ok[u] ? sx_r : 0.f;
                unsigned h[5], q[5];
                unsigned hl, ql, hr, qr, hm, qm;
                split1(l[u], sl, hl, ql);
                split1(src[u][0].x, sc, hm, qm);
                h[0] = __builtin_amdgcn_perm(hm, hl, 0x05040100u);
                q[0] = __builtin_amdgcn_perm(qm, ql, 0x05040100u);
                split2(src[u][0].y, src[u][0].z, sc, h[1], q[1]);
                split2(src[u][0].w, src[u][1].x, sc, h[2], q[2]);
                split2(src[u][1].y, src[u][1].z, sc, h[3], q[3]);
                split1(src[u][1].w, sc, hm, qm);
                split1(rr[u], sr, hr, qr);
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

        half8 B[2][3][NCI][2];
        f32x4 rawA[NMY][2], rawB[NCI][2];
        float rawL[NCI], rawR[NCI], rawS;
        __syncthreads();                        // the previous column's last reads of the ring are done
        {
            f32x4 p0[NMY][2], p1[NMY][2], p2[NMY][2], q0[NCI][2];
            float s0, s1, s2, l0[NCI], rr0[NCI];
            load_A(r0 - 1, p0, s0);
            load_A(r0, p1, s1);
            load_A(r0 + 1, p2, s2);
            load_B(r0, q0, l0, rr0);
            load_A(r0 + 2, rawA, rawS);
            load_B(r0 + 1, rawB, rawL, rawR);
            park_A(r0 - 1, p0, s0);
            park_A(r0, p1, s1);
            park_A(r0 + 1, p2, s2);
            cvt_B(q0, l0, rr0, B[0]);
        }
        // the three MFMA passes of tap row ky for input row r: dY row r + 1 - ky out of its LDS slot
        auto mfma_row = [&](int yrow, auto BSET, auto KY) {
            constexpr int bs = decltype(BSET)::value, ky = decltype(KY)::value;
            const int slot = (yrow + 1) & 3;
            half8 Af[NCO][2];
#pragma unroll
            for (int t2 = 0; t2 < NCO; ++t2) {
                Af[t2][0] = as_half8(Alds[slot][t2][0][lane]);
                Af[t2][1] = as_half8(Alds[slot][t2][1][lane]);
            }
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int t2 = 0; t2 < NCO; ++t2)
#pragma unroll
                        for (int u = 0; u < NCI; ++u)
                            acc[t2][u][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                Af[t2][pass == 2 ? 1 : 0], B[bs][kx][u][pass == 1 ? 1 : 0], acc[t2][u][ky * 3 + kx], 0, 0,
                                0);
        };
        auto step = [&](auto PH, int r) {
            constexpr int ph = decltype(PH)::value;
            using BS = std::integral_constant<int, ph % 2>;
            // rows r - 1 .. r + 1 are in the ring (row r + 1 was parked during step r - 1): one barrier per step makes
            // it visible and frees the slot of row r - 2 for row r + 2
            __syncthreads();
            mfma_row(r - 1, BS{}, std::integral_constant<int, 2>{});
            park_A(r + 2, rawA, rawS);
            cvt_B(rawB, rawL, rawR, B[(ph + 1) % 2]);
            __builtin_amdgcn_sched_barrier(0);
            load_A(r + 3, rawA, rawS);
            load_B(r + 2, rawB, rawL, rawR);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(r, BS{}, std::integral_constant<int, 1>{});
            mfma_row(r + 1, BS{}, std::integral_constant<int, 0>{});
        };
        int r = r0;
        for (; r + 2 <= r1; r += 2) {
            step(std::integral_constant<int, 0>{}, r);
            step(std::integral_constant<int, 1>{}, r + 1);
        }
        if (r < r1)
            step(std::integral_constant<int, 0>{}, r);
        if ((r1 - r0) & 1) {
            // the B sets alternate per step: an odd run leaves the NEXT column's first row in set 1; nothing to fix --
            // every column re-converts its first row into B[0] above
        }
    }

    const int slab = b - sk_first(ptype, a.ptypes, T, wtot, a.nwg);
    float *out = a.part + (size_t)slab * 9 * a.Cout * a.Cin;
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
    }   // segment
}

// dw[co][ci][tap] = sum_s part[s][tap][co][ci] over the slabs the element's workgroup type wrote (sk_owner of its first
// and last item).  Block = 32 outputs x 8 slab groups, combined in fixed order through LDS as in k_wgrad_reduce.
__global__ __launch_bounds__(256) void k_wgrad_reduce_sk(const float *__restrict__ part, int Cout, int Cin, int co_span,
                                                        int ci_span, int ncib, int ptypes, long long T, long long wtot, int nwg, int nx,
                                                        float *__restrict__ dw)
{
    __shared__ float sh[8][32];
    const int total = 9 * Cout * Cin;
    const int lane = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + lane;       // (tap, co, ci) in slab order
    float s0 = 0.f, s1 = 0.f;
    int ci = 0, co = 0, tap = 0;
    if (idx < total) {
        ci = idx % Cin;
        co = (idx / Cin) % Cout;
        tap = idx / (Cin * Cout);
        const int ptype = (co / co_span) * ncib + ci / ci_span;
        const int S = nx > 0 ? nx : sk_last(ptype, ptypes, T, wtot, nwg) - sk_first(ptype, ptypes, T, wtot, nwg) + 1;
        int k = g;
        for (; k + 8 < S; k += 16) {
            s0 += part[(size_t)k * total + idx];
            s1 += part[(size_t)(k + 8) * total + idx];
        }
        if (k < S)
            s0 += part[(size_t)k * total + idx];
    }
    sh[g][lane] = s0 + s1;
    __syncthreads();
    if (g == 0 && idx < total) {
        float s = sh[0][lane];
#pragma unroll
        for (int q = 1; q < 8; ++q)
            s += sh[q][lane];
        dw[((size_t)co * Cin + ci) * 9 + tap] = s;
    }
}

}  // namespace

struct WgradSPlan {
    int nco, nci;       // co tiles per workgroup, ci tiles per wave
    int ncib, ptypes;   // ci blocks (4 waves x nci tiles), workgroup types = co groups x ci blocks
    int units;          // columns: (image, 32-pixel strip), H row steps each
    int nwg, nx;        // workgroups; nx > 0: equal pixel splits (nwg = ptypes * nx), 0: stream-K partition
    int nslab;          // slabs of 9 * Cout * Cin floats the workspace must hold
};

static int g_sk_mode = -1;      // tuning: -1 automatic, 0 equal pixel splits only, 1 stream-K always
static int g_sk_nwg = 256;      // workgroups of the stream-K partition (one per CU)

void dcl_wgrad_shared_tune(int sk_mode, int nwg)
{
    g_sk_mode = sk_mode;
    g_sk_nwg = nwg > 0 ? nwg : 256;
}

// Tile and partition for a shape.  force_nco / force_nci > 0 pin the tile (tuning hook).
WgradSPlan dcl_wgrad_shared_plan(int N, int Cin, int Cout, int H, int W, int force_nco, int force_nci)
{
    const int cot = Cout / 16, cit = Cin / 16;
    WgradSPlan pl;
    pl.nco = 1;
    pl.nci = 1;
    // cost model fitted to tools/wgrad_head.py on the HRNet-W48 shapes: time ~ (tile pairs per wave) x (row steps per
    // workgroup) x (1 + 0.15 wave-loads per tile pair and row step), a little extra for the slab traffic of big tiles
    double best = 1e30;
    for (int nco = 5; nco >= 1; --nco) {
        if (cot % nco || nco == 4)
            continue;
        for (int nci = 2; nci >= 1; --nci) {
            if (nco == 5 && nci == 2)
                continue;                                               // 90 accumulator tiles do not fit
            if (force_nco > 0 && (nco != force_nco || nci != (force_nci > 0 ? force_nci : nci)))
                continue;
            const int slots = (cit + nci - 1) / nci, ncib = (slots + 3) / 4;
            const int pt = (cot / nco) * ncib;
            const double nx = pt >= g_sk_nwg ? (double)g_sk_nwg / pt : (double)(g_sk_nwg / pt);
            const double loads = (2.0 * nco + 4.0 * 4 * nci) / (nco * 4.0 * nci);
            const double cost = nco * nci / nx * (1.0 + 0.15 * loads) * (1.0 + 0.02 * nco * nci);
            if (cost < best) {
                best = cost;
                pl.nco = nco;
                pl.nci = nci;
            }
        }
    }
    const int slots = (cit + pl.nci - 1) / pl.nci;
    pl.ncib = (slots + 3) / 4;
    pl.ptypes = (cot / pl.nco) * pl.ncib;
    pl.units = N * ((W + 31) / 32);
    const long long T = (long long)pl.units * H, wtot = T * pl.ptypes;
    // equal pixel splits when they fill >= 15/16 of the CUs, else cut the (type, row step) sequence into one range per CU
    int nx = pl.ptypes >= g_sk_nwg ? 0 : g_sk_nwg / pl.ptypes;
    if ((long long)nx > T)
        nx = (int)T;
    const bool equal_ok = nx > 0 && (nx * pl.ptypes * 16 >= g_sk_nwg * 15 || nx == T);
    if (g_sk_mode == 0 ? nx > 0 : (g_sk_mode == 1 ? false : equal_ok)) {
        pl.nx = nx;
        pl.nwg = nx * pl.ptypes;
        pl.nslab = nx;
    } else {
        pl.nx = 0;
        pl.nwg = (int)(wtot < g_sk_nwg ? wtot : g_sk_nwg);
        pl.nslab = 1;
        for (int p = 0; p < pl.ptypes; ++p) {
            const int n = sk_last(p, pl.ptypes, T, wtot, pl.nwg) - sk_first(p, pl.ptypes, T, wtot, pl.nwg) + 1;
            pl.nslab = n > pl.nslab ? n : pl.nslab;
        }
    }
    return pl;
}

int dcl_wgrad_shared_slabs(int N, int Cin, int Cout, int H, int W, int force_nco, int force_nci)
{
    return dcl_wgrad_shared_plan(N, Cin, Cout, H, W, force_nco, force_nci).nslab;
}

// kernel + slab reduction; part must hold dcl_wgrad_shared_slabs(...) slabs
void dcl_wgrad_shared_launch(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W, const float *xamax,
                             int xcount, const float *gamax, int gcount, int stride, float *part, float *dw, int force_nco,
                             int force_nci, hipStream_t s)
{
    const WgradSPlan pl = dcl_wgrad_shared_plan(N, Cin, Cout, H, W, force_nco, force_nci);
    WgradSArgs a;
    a.x = x; a.dy = dy; a.part = part; a.xamax = xamax; a.gamax = gamax; a.xcount = xcount; a.gcount = gcount;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.Hd = stride == 2 ? (H - 1) / 2 + 1 : H;
    a.Wd = stride == 2 ? W / 2 : W;
    a.strips = (W + 31) / 32;
    a.units = pl.units; a.ncib = pl.ncib; a.ptypes = pl.ptypes; a.nwg = pl.nwg; a.nx = pl.nx;
    const dim3 grid((unsigned)pl.nwg);
#define DCL_WGS_CASE(o, i)                                                          \
    if (pl.nco == o && pl.nci == i) {                                               \
        if (stride == 2)                                                            \
            hipLaunchKernelGGL((k_wgrad3x3s<o, i, true>), grid, dim3(256), 0, s, a);    \
        else                                                                        \
            hipLaunchKernelGGL((k_wgrad3x3s<o, i, false>), grid, dim3(256), 0, s, a);   \
    }
    DCL_WGS_CASE(5, 1)
    DCL_WGS_CASE(3, 2)
    DCL_WGS_CASE(3, 1)
    DCL_WGS_CASE(2, 2)
    DCL_WGS_CASE(2, 1)
    DCL_WGS_CASE(1, 2)
    DCL_WGS_CASE(1, 1)
#undef DCL_WGS_CASE
    const int total = 9 * Cout * Cin;
    const long long T = (long long)pl.units * H;
    hipLaunchKernelGGL(k_wgrad_reduce_sk, dim3((total + 31) / 32), dim3(256), 0, s, part, Cout, Cin, pl.nco * 16,
                       pl.nci * 64, pl.ncib, pl.ptypes, T, T * pl.ptypes, pl.nwg, pl.nx, dw);
}
