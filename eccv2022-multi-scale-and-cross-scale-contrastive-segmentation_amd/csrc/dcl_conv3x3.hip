// dcl_conv3x3.hip -- direct (implicit-GEMM) 3x3 / stride 1 / pad 1 convolution on the f16 matrix pipe with
// fp32-equivalent arithmetic ("f16x3"), for the BasicBlock convolutions that carry ~80 % of HRNet-W48's FLOPs
// (reference models/HRNet.py:32-60 builds them as nn.Conv2d(C, C, 3, 1, 1, bias=False), C in {48, 96, 192, 384}).
//
//   y[n, co, y, x] = sum_{ci, ky, kx} w[co, ci, ky, kx] * x[n, ci, y + ky - 1, x + kx - 1]
//
// GEMM view per tap: D[co][pixel] += W_tap[co][ci] * X_tap[ci][pixel]; both operands are split on the fly into
// hi = f16(v * s), lo = f16(v * s - hi) (s: a power of two from the tensor's absmax, read on the device) and every
// product is issued as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with f32 accumulation.
//
// Work decomposition (one workgroup = 4 waves):
//   * output tile: 4P rows x 32 columns of one image x R*32 output channels; wave w owns rows [P w, P w + P)
//     (P MFMA pixel tiles of 1 row x 32 columns) for all R channel tiles -> R*P accumulators of 16 registers.
//   * K loop over chunks of 16 input channels.  The (4P+2) x 34 input patch of a chunk is loaded from NCHW f32
//     with coalesced 4-byte loads (8 channels of one pixel per work item), split, and written to LDS as one
//     80-byte record per pixel: [16 hi | 16 lo | pad] -- channel-innermost, so the B operand of a tap (8
//     consecutive channels of the lane's shifted pixel) is ONE ds_read_b128, conflict-free (80 B = 5 x 16 B).
//     Double buffered, one barrier per chunk; the next chunk's global loads are in flight during the MFMAs.
//   * a B fragment depends on (tile row rr = p + ky, kx) only, so the (P+2) x 3 fragments of a chunk are read
//     once and used by every (p, ky) with p + ky = rr: 2 (P+2) 3 LDS reads feed 27 R P MFMAs.
//   * the A operand (weights) is pre-packed by k_pack_w3x3 into exact fragment order (16 B per lane, 1 KiB per
//     fragment) and streamed from L2 / L1 straight into registers one kx ahead; the four waves of a workgroup
//     read the same fragments (L1 hits).
//   * epilogue: accumulator register r of lane (h, li) is output channel jrow(r, h), pixel column li -> 128-B
//     row segments of the NCHW output.
// The same kernel computes the data gradient: dx = conv3x3(dy, w') with w'[ci, co, ky, kx] = w[co, ci, 2-ky, 2-kx]
// (k_pack_w3x3 with `transposed`).
#include "dcl_conv_body.h"

namespace {

template <int R, int P, int S>
__global__ __launch_bounds__(256, 1) void k_conv3x3(ConvArgs a)
{
    conv_body<R, P, S>(a, (int)blockIdx.x);
}

template <int R, int P>
__global__ __launch_bounds__(256, 1) void k_conv3x3_il(ConvArgs a)
{
    conv_body<R, P, 1, 0, true>(a, (int)blockIdx.x);
}

template <int R, int P>
__global__ __launch_bounds__(256, 2) void k_conv3x3_il_o2(ConvArgs a)
{
    conv_body<R, P, 1, 0, true>(a, (int)blockIdx.x);
}

// two workgroups per CU, waves split 2 (rows) x 2 (channel tiles): workgroup tile 2 P rows x 2 R channel tiles
template <int R, int P>
__global__ __launch_bounds__(256, 2) void k_conv3x3_il_ws2(ConvArgs a)
{
    conv_body<R, P, 1, 0, true, 2>(a, (int)blockIdx.x);
}

// stride 2 with the interleaved staging (one output row per wave: a B fragment group is one (kx, ky) tap)
template <int R>
__global__ __launch_bounds__(256, 1) void k_conv3x3_il_s2(ConvArgs a)
{
    conv_body<R, 1, 2, 0, true>(a, (int)blockIdx.x);
}

template <int R, int P>
__global__ __launch_bounds__(256, 1) void k_conv3x3_phases(ConvArgs a)
{
    conv_body<R, P, 1, 1>(a, (int)blockIdx.x);
}

template <int R, int P>
__global__ __launch_bounds__(256, 1) void k_conv1x1(ConvArgs a)
{
    conv_body<R, P, 1, 2>(a, (int)blockIdx.x);
}


// ---- "PM": data gradient of a stride-2 convolution, ALL FOUR output parity classes of a tile in one workgroup ----------------
// dx = conv3x3(dy zero-inserted at the odd coordinates, w').  Output pixel (2 i + py, 2 j + px) sees the taps whose input
// coordinate is even: patch row offset dr = 1 (stored row i) with weight row 1 for py = 0 and weight row 0 for py = 1, dr = 2
// (stored row i + 1) with weight row 2 for py = 1; the same along x.  k_conv3x3_phases gives each class its own workgroup --
// four workgroups stage (load, split, write to LDS) the same patch for 1, 2, 2 and 4 of the 9 taps: 0.09-0.13 of the roofline,
// bound by the staging.  Here one workgroup stages the patch ONCE and runs all 9 taps against it, each tap into the accumulator
// of its class (4 R P accumulator tiles per wave): the matrix work per staged patch of a stride-1 tile.  A B fragment
// (tile row p + dr, column offset dc) feeds the (2 if dr = 1 else 1) x (2 if dc = 1 else 1) taps that read it.  Epilogue: the
// two px classes of a lane are neighbouring pixels -> one 8-byte store (the per-class kernel scatters 4-byte stores).
template <int R, int P>
__device__ __forceinline__ void conv_pm_body(const ConvArgs &a)
{
    constexpr int WR = 4;
    constexpr int LW = TW + 2;
    constexpr int ROWS = WR * P + 2;
    constexpr int TP = ROWS * LW;
    constexpr int NITEM = (2 * TP + 255) / 256;
    constexpr int BUFB = TP * PIXB;
    static_assert(4 * R * P <= 12, "PM: at most 12 accumulator tiles per wave (16 spill: 256 accumulator registers + staging + fragments)");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFB];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, li = lane & 31;
    int bx, cg;
    {
        const int ngrp = a.groups;
        const int ntile = a.tiles_x * a.tiles_y * a.N, n8 = ntile & ~7, main_blocks = n8 * ngrp;
        if ((int)blockIdx.x < main_blocks) {
            const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
            cg = rest % ngrp;
            bx = xcd * (n8 >> 3) + rest / ngrp;
        } else {
            const int rest = blockIdx.x - main_blocks;
            cg = rest % ngrp;
            bx = n8 + rest / ngrp;
        }
    }
    const int tx = bx % a.tiles_x;
    bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int n = bx / a.tiles_y;
    const int x0 = tx * TW, y0 = ty * WR * P;               // stored-gradient coordinates of the tile
    const int T0 = cg * R;
    const size_t plane = (size_t)a.Hs * a.Ws, oplane = (size_t)a.Ho * a.Wo;
    const float *xb = a.x + (size_t)n * a.Cin * plane;
    int goff[NITEM], loff[NITEM];
    float gsc[NITEM];
    bool gok[NITEM];
#pragma unroll
    for (int m = 0; m < NITEM; ++m) {
        const int it = min(tid + 256 * m, 2 * TP - 1);
        const int oct = it >= TP ? 1 : 0;
        const int pix = it - oct * TP;
        const int r = pix / LW, c = pix - r * LW;
        const int gy = y0 + r - 1, gx = x0 + c - 1;
        const bool ok = gy >= 0 && gy < a.Hs && gx >= 0 && gx < a.Ws;
        loff[m] = pix * PIXB + oct * 16;
        goff[m] = ok ? gy * a.Ws + gx : 0;
        gok[m] = ok;
    }
    const bool ragged = (a.Cin & 15) != 0;
    float gA[NITEM][8];
    auto load_items = [&](int c, float (&g)[NITEM][8]) {
        if (!ragged || c + 1 < a.nchunk) {
#pragma unroll
            for (int m = 0; m < NITEM; ++m) {
                const float *xc = xb + (size_t)(16 * c + 8 * (tid + 256 * m >= TP ? 1 : 0)) * plane + goff[m];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    g[m][e] = xc[e * plane];
            }
        } else {
#pragma unroll
            for (int m = 0; m < NITEM; ++m) {
                const int ch0 = 16 * c + 8 * (tid + 256 * m >= TP ? 1 : 0);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    g[m][e] = xb[(size_t)min(ch0 + e, a.Cin - 1) * plane + goff[m]];
            }
        }
    };
    auto write_items = [&](unsigned char *buf, const float (&g)[NITEM][8]) {
#pragma unroll
        for (int m = 0; m < NITEM; ++m) {
            unsigned hh[4], ll[4];
            split2(g[m][0], g[m][1], gsc[m], hh[0], ll[0]);
            split2(g[m][2], g[m][3], gsc[m], hh[1], ll[1]);
            split2(g[m][4], g[m][5], gsc[m], hh[2], ll[2]);
            split2(g[m][6], g[m][7], gsc[m], hh[3], ll[3]);
            *(uint4 *)(buf + loff[m]) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
            *(uint4 *)(buf + loff[m] + 32) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
        }
    };
    const int mtiles = (a.Cout + 31) / 32;
    const uint4 *wa[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        wa[r] = a.wp + (size_t)min(T0 + r, mtiles - 1) * a.nchunk * 9 * 2 * 64 + lane;

    f32x16 acc[4][R][P];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    acc[k][r][p][q] = 0.f;

    load_items(0, gA);
    float xs;
    {
        float m = 0.f;
        for (int i = tid; i < a.xcount; i += 256)
            m = fmaxf(m, a.xamax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            m = fmaxf(m, __shfl_xor(m, o, 64));
        float *wm = (float *)lds;
        if (lane == 0)
            wm[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        __syncthreads();
        xs = pow2_scale(m);
    }
#pragma unroll
    for (int m = 0; m < NITEM; ++m)
        gsc[m] = gok[m] ? xs : 0.f;
    const int col0 = 2 * (x0 + li);
    if (a.addend || a.bias) {
        const float sc2 = xs * pow2_scale(a.wamax[0]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int row = 2 * (y0 + P * wave + p) + (k >> 1), col = col0 + (k & 1);
                    const int cob = (T0 + r) * 32 + 4 * h;
                    const bool ok = row < a.Ho && col < a.Wo && cob < a.Cout;
                    const size_t o0 = (((size_t)n * a.Cout + min(cob, a.Cout - 1)) * a.Ho + min(row, a.Ho - 1)) * a.Wo +
                                      min(col, a.Wo - 1);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int kk = ok ? min((q & 3) + 8 * (q >> 2), a.Cout - 1 - cob) : 0;
                        const float ad = (a.addend ? a.addend[o0 + (size_t)kk * oplane] : 0.f) +
                                         (a.bias ? a.bias[min(cob, a.Cout - 1) + kk] : 0.f);
                        acc[k][r][p][q] = ok ? ad * sc2 : 0.f;
                    }
                }
    }
    write_items(lds, gA);
    __syncthreads();

    // Weight fragments of tap group g = (dr, dc) -- 4, 2, 2, 1 taps -- travel one group ahead of their MFMAs in two register
    // sets (group g in set g & 1; group 0 of chunk c + 1 is fetched under the last group of chunk c): fetched right in front of
    // their use, every group started with an exposed L2 round trip (first version: 0.078-0.096 of the roofline)
    constexpr int WK[3] = {1, 0, 2};        // [0]: offset 1, class 0;  [1]: offset 1, class 1;  [2]: offset 2, class 1
    half8 Aq0[4][R][2], Aq1[2][R][2];          // set 0: groups 0 (4 taps) and 2 (2 taps); set 1: groups 1 (2 taps) and 3 (1 tap)
    auto load_group = [&](auto SET, auto G, int cc) {
        constexpr int set = decltype(SET)::value, g = decltype(G)::value, dr = 1 + g / 2, dc = 1 + g % 2;
        constexpr int nyv = dr == 1 ? 2 : 1, nxv = dc == 1 ? 2 : 1;
#pragma unroll
        for (int iy = 0; iy < nyv; ++iy)
#pragma unroll
            for (int ix = 0; ix < nxv; ++ix) {
                const int wky = dr == 1 ? WK[iy] : 2, wkx = dc == 1 ? WK[ix] : 2;
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        const half8 v = __builtin_bit_cast(half8, wa[r][(((size_t)cc * 9 + wky * 3 + wkx) * 2 + part) * 64]);
                        if constexpr (set == 0)
                            Aq0[iy * nxv + ix][r][part] = v;
                        else
                            Aq1[iy * nxv + ix][r][part] = v;
                    }
            }
    };
    load_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
    for (int c = 0; c < a.nchunk; ++c) {
        const unsigned char *cur = lds + (c & 1) * BUFB;
        const bool more = c + 1 < a.nchunk;
        __builtin_amdgcn_sched_barrier(0);
        load_items(min(c + 1, a.nchunk - 1), gA);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 4>([&](auto GC) {
            constexpr int g = decltype(GC)::value, dr = 1 + g / 2, dc = 1 + g % 2;
            constexpr int nyv = dr == 1 ? 2 : 1, nxv = dc == 1 ? 2 : 1;
            if constexpr (g < 3)
                load_group(std::integral_constant<int, (g + 1) & 1>{}, std::integral_constant<int, g + 1>{}, c);
            else
                load_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, min(c + 1, a.nchunk - 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const unsigned char *bp = cur + ((P * wave + p + dr) * LW + li + dc) * PIXB + h * 16;
                const half8 bh = *(const half8 *)bp, bl = *(const half8 *)(bp + 32);
#pragma unroll
                for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                    for (int iy = 0; iy < nyv; ++iy)
#pragma unroll
                        for (int ix = 0; ix < nxv; ++ix) {
                            const int cy = dr == 1 ? iy : 1, cx = dc == 1 ? ix : 1;       // class of this tap
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                acc[cy * 2 + cx][r][p] = DCL_MFMA(
                                    (g & 1) ? Aq1[iy * nxv + ix][r][pass == 2 ? 1 : 0] : Aq0[iy * nxv + ix][r][pass == 2 ? 1 : 0],
                                    pass == 1 ? bl : bh, acc[cy * 2 + cx][r][p]);
                        }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more)
            write_items(lds + ((c + 1) & 1) * BUFB, gA);
        __syncthreads();
    }

    const float inv = 1.0f / (xs * pow2_scale(a.wamax[0]));
    const bool pair_ok = (a.Wo & 1) == 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
                const int row = 2 * (y0 + P * wave + p) + cy;
                const int cob = (T0 + r) * 32 + 4 * h;
                if (row < a.Ho && col0 < a.Wo && cob < a.Cout) {
                    float *yp = a.y + (((size_t)n * a.Cout + cob) * a.Ho + row) * a.Wo + col0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int k = (q & 3) + 8 * (q >> 2);
                        if (cob + k < a.Cout) {
                            const float v0 = acc[cy * 2][r][p][q] * inv, v1 = acc[cy * 2 + 1][r][p][q] * inv;
                            float *o = yp + (size_t)k * oplane;
                            if (pair_ok) {                      // (col0 even, Wo even: 8-byte aligned, col0 + 1 < Wo)
                                float2 v;
                                v.x = v0;
                                v.y = v1;
                                *(float2 *)o = v;
                            } else {
                                o[0] = v0;
                                if (col0 + 1 < a.Wo)
                                    o[1] = v1;
                            }
                        }
                    }
                }
            }
}

template <int R, int P>
__global__ __launch_bounds__(256, 1) void k_conv3x3_pm(ConvArgs a)
{
    conv_pm_body<R, P>(a);
}

// weights -> fragment order.  transposed = 0: value(m, k, ky, kx) = w[m][k][ky][kx], w is [M][K][3][3];
// transposed = 1 (data gradient): value(m, k, ky, kx) = w[k][m][2 - ky][2 - kx], w is [K][M][3][3].
// One thread per (fragment pair, lane): fragment (T, c, t) lane (hh, i) holds value(32 T + i, 16 c + 8 hh + e, t).
// `transposed` bit 1 set: a 1x1 kernel, w is [M][K] ([K][M] transposed), ONE tap per (tile, chunk).
__device__ __forceinline__ void pack_item(const float *__restrict__ w, int M, int K, int transposed, int gid,
                                          float s, uint4 *__restrict__ wp)
{
    const int mtiles = (M + 31) / 32, nchunk = (K + 15) / 16;
    const int taps = (transposed & 2) ? 1 : 9;
    transposed &= 1;
    if (gid >= mtiles * nchunk * taps * 64)
        return;
    const int lane = gid & 63;
    int f = gid >> 6;
    const int t = f % taps;
    f /= taps;
    const int c = f % nchunk;
    const int T = f / nchunk;
    const int i = lane & 31, hh = lane >> 5;
    const int m = 32 * T + i;
    const int ky = t / 3, kx = t - 3 * ky;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 16 * c + 8 * hh + e;
        float v = 0.f;
        if (m < M && k < K) {
            if (taps == 1)
                v = transposed ? w[(size_t)k * M + m] : w[(size_t)m * K + k];
            else
                v = transposed ? w[(((size_t)k * M + m) * 3 + (2 - ky)) * 3 + (2 - kx)]
                               : w[(((size_t)m * K + k) * 3 + ky) * 3 + kx];
        }
        v *= s;
        hi[e] = (_Float16)v;
        lo[e] = (_Float16)(v - (float)hi[e]);
    }
    const size_t o = ((((size_t)T * nchunk + c) * taps + t) * 2) * 64 + lane;
    wp[o] = __builtin_bit_cast(uint4, hi);
    wp[o + 64] = __builtin_bit_cast(uint4, lo);
}

__global__ __launch_bounds__(256) void k_pack_w3x3(const float *__restrict__ w, int M, int K, int transposed,
                                                  const float *__restrict__ wamax, uint4 *__restrict__ wp)
{
    pack_item(w, M, K, transposed, blockIdx.x * 256 + threadIdx.x, pow2_scale(wamax[0]), wp);
}

// Multi-tensor forms (one launch for all weights of a model instead of three per convolution): job tables in
// device memory, blk2job[b] = job of workgroup b, job.first_block = its first workgroup.
struct PackJob {
    const float *w;
    uint4 *wp;
    const float *amax;
    int M, K, transposed, first_block;
};
struct AbsmaxJob {
    const float *x;
    float *out;
    long long n;
    int first_block, pad;
};

__global__ __launch_bounds__(256) void k_pack_multi(const PackJob *__restrict__ jobs,
                                                   const int *__restrict__ blk2job)
{
    const PackJob jb = jobs[blk2job[blockIdx.x]];
    pack_item(jb.w, jb.M, jb.K, jb.transposed, (blockIdx.x - jb.first_block) * 256 + threadIdx.x,
              pow2_scale(jb.amax[0]), jb.wp);
}

// each workgroup covers 4096 consecutive elements of its tensor
__global__ __launch_bounds__(256) void k_absmax_multi(const AbsmaxJob *__restrict__ jobs,
                                                     const int *__restrict__ blk2job)
{
    __shared__ float wmax[4];
    const AbsmaxJob jb = jobs[blk2job[blockIdx.x]];
    const long long base = (long long)(blockIdx.x - jb.first_block) * 4096;
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long i = base + q * 1024 + threadIdx.x * 4;
        if (i + 4 <= jb.n) {
            const f32x4 v = *(const f32x4 *)(jb.x + i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        } else {
            for (long long k = i; k < jb.n; ++k)
                m = fmaxf(m, fabsf(jb.x[k]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0)
        wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax((unsigned int *)jb.out, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// out[0] = max(out[0], max |x|): integer atomicMax on the float bits (order independent); out starts at 0.
// Fallback for tensors whose producer did not emit an absmax (the fused BN kernels do: dcl_bn.hip).
__global__ __launch_bounds__(256) void k_absmax(const float *__restrict__ x, size_t n, float *__restrict__ out)
{
    __shared__ float wmax[4];
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    float m = 0.f;
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    // four independent 16-B loads in flight per thread
    for (; i + 3 * stride + 4 <= n; i += 4 * stride) {
        const f32x4 v0 = *(const f32x4 *)(x + i), v1 = *(const f32x4 *)(x + i + stride),
                    v2 = *(const f32x4 *)(x + i + 2 * stride), v3 = *(const f32x4 *)(x + i + 3 * stride);
        const float m0 = fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w)));
        const float m1 = fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)));
        const float m2 = fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w)));
        const float m3 = fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)));
        m = fmaxf(m, fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
    }
    for (; i < n; i += stride) {
        if (i + 4 <= n) {
            const f32x4 v = *(const f32x4 *)(x + i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        } else {
            for (size_t j = i; j < n; ++j)
                m = fmaxf(m, fabsf(x[j]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0)
        wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax((unsigned int *)out, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

}  // namespace

// out[0] = max(a[0 .. na)) + max(b[0 .. nb)): an upper bound of max|a_tensor + b_tensor| from the two tensors' absmax
// partials (one wave; the partial buffers hold 1 .. DCL_AMAX_SLOTS values)
namespace {
__global__ __launch_bounds__(64) void k_amax_sum2(const float *__restrict__ a, int na, const float *__restrict__ b, int nb,
                                                 float *__restrict__ out)
{
    float ma = 0.f, mb = 0.f;
    for (int i = threadIdx.x; i < na; i += 64)
        ma = fmaxf(ma, a[i]);
    for (int i = threadIdx.x; i < nb; i += 64)
        mb = fmaxf(mb, b[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ma = fmaxf(ma, __shfl_xor(ma, o, 64));
        mb = fmaxf(mb, __shfl_xor(mb, o, 64));
    }
    if (threadIdx.x == 0)
        out[0] = ma + mb;
}
}  // namespace

extern "C" int dcl_amax_sum2(const float *a, int na, const float *b, int nb, float *out, void *stream)
{
    DCL_CHECK_ARG(a && b && out && na > 0 && nb > 0, "bad arguments");
    hipLaunchKernelGGL(k_amax_sum2, dim3(1), dim3(64), 0, (hipStream_t)stream, a, na, b, nb, out);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_absmax(const float *x, int64_t n, float *out, void *stream)
{
    DCL_CHECK_ARG(x && out && n > 0, "bad arguments");
    DCL_CHECK_ARG((((uintptr_t)x) & 15) == 0, "input must be 16-byte aligned");
    size_t blocks = ((size_t)n + 4095) / 4096;     // 16 elements per thread
    if (blocks > 4096)
        blocks = 4096;
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, out);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_conv3x3_pack(const float *w, int M, int K, int transposed, const float *wamax, void *wp,
                                void *stream)
{
    DCL_CHECK_ARG(w && wamax && wp && M > 0 && K > 0, "bad arguments");
    DCL_CHECK_ARG(transposed >= 0 && transposed <= 3, "transposed: bit 0 = data-gradient orientation, bit 1 = 1x1 kernel");
    const int mtiles = (M + 31) / 32, nchunk = (K + 15) / 16;
    const int total = mtiles * nchunk * ((transposed & 2) ? 1 : 9) * 64;
    hipLaunchKernelGGL(k_pack_w3x3, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, M, K,
                       transposed, wamax, (uint4 *)wp);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_absmax_multi(const void *jobs, const int32_t *blk2job, int nblocks, void *stream)
{
    DCL_CHECK_ARG(jobs && blk2job && nblocks > 0, "bad arguments");
    hipLaunchKernelGGL(k_absmax_multi, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                       (const AbsmaxJob *)jobs, blk2job);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_conv3x3_pack_multi(const void *jobs, const int32_t *blk2job, int nblocks, void *stream)
{
    DCL_CHECK_ARG(jobs && blk2job && nblocks > 0, "bad arguments");
    hipLaunchKernelGGL(k_pack_multi, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackJob *)jobs, blk2job);
    DCL_LAUNCH_CHECK();
    return 0;
}

static int g_conv_interleave = 2;      // 2 = staging interleaved with the MFMAs (k_conv3x3_il) + the (2, 2) tile's waves split
                                       // 2 x 2 (k_conv3x3_il_ws2); 1 = interleaved only; 0 = the fenced blocks

static int g_conv_s2_interleave = 1;   // stride-2 forward tiles on the interleaved staging as well (bit 2 of the hook's argument clears it)

extern "C" int dcl_conv3x3_set_interleave(int on)
{
    g_conv_interleave = on & 3;  // 0 fenced blocks, 1 interleaved, 2 interleaved + wave-split (2, 2) tile
    g_conv_s2_interleave = (on & 4) ? 0 : 1;        // + 4: stride-2 tiles keep the fenced blocks (A/B runs)
    return 0;
}

template <int R, int P, int S>
static int launch_conv(const ConvArgs &a0, hipStream_t stream)
{
    ConvArgs a = a0;
    // (phases: tiles over one parity class of the output, ceil(Ho / 2) x ceil(Wo / 2) pixels, four classes per tile)
    a.tiles_x = ((a.phases ? (a.Wo + 1) / 2 : a.Wo) + TW - 1) / TW;
    a.tiles_y = ((a.phases ? (a.Ho + 1) / 2 : a.Ho) + 4 * P - 1) / (4 * P);
    const int mtiles = (a.Cout + 31) / 32;
    a.groups = (mtiles + R - 1) / R;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N * a.groups * (a.phases == 1 ? 4 : 1)));
    if (a.pre_sc) {
        // the producer norm's map + ReLU inside the patch staging (dcl_conv3x3_pre.hip): interleaved tiles only
        const bool ws2 = S == 1 && R == 2 && P == 2;
        if (!g_conv_interleave || (ws2 && g_conv_interleave != 2) || (S == 2 && !g_conv_s2_interleave) || (a.Cin & 15) ||
            a.up != 1 || a.phases || a.onetap)
            return DCL_EUNSUPPORTED;
        return dcl_conv_pre_launch(a, R, P, S, ws2, grid, stream) ? 0 : DCL_EUNSUPPORTED;
    }
    if constexpr (S == 1 && 4 * R * P <= 12) {
        if (a.phases == 2) {                    // all four parity classes of a tile in one workgroup (conv_pm_body)
            hipLaunchKernelGGL((k_conv3x3_pm<R, P>), grid, dim3(256), 0, stream, a);
            dcl_note_kernel("k_conv3x3_pm<%d,%d>", R, P);
            return 0;
        }
    }
    if (a.phases == 2)
        return DCL_EUNSUPPORTED;
    if constexpr (S == 1) {
        if (a.phases == 1) {
            hipLaunchKernelGGL((k_conv3x3_phases<R, P>), grid, dim3(256), 0, stream, a);
            dcl_note_kernel("k_conv3x3_phases<%d,%d>", R, P);
            return 0;
        }
        if (a.onetap) {
            if constexpr (R == 3 && P == 4) {
                return DCL_EUNSUPPORTED;        // (conv_f16x3 never asks: P = 2 for one-tap (3, .) tiles)
            } else {
                hipLaunchKernelGGL((k_conv1x1<R, P>), grid, dim3(256), 0, stream, a);
                dcl_note_kernel("k_conv1x1<%d,%d>", R, P);
                return 0;
            }
        }
    }
    if constexpr (S == 1 && R == 2 && P == 2) {
        if (g_conv_interleave == 2 && (a.Cin & 15) == 0 && a.up == 1 && !a.phases && !a.onetap) {
            // (1, 4) waves, 2 x 2 per workgroup: the same 8 x 32 pixels x 64 channels per workgroup, same grid
            hipLaunchKernelGGL((k_conv3x3_il_ws2<1, 4>), grid, dim3(256), 0, stream, a);
            dcl_note_kernel("k_conv3x3_il_ws2<1,4>");
        } else if (g_conv_interleave && (a.Cin & 15) == 0 && a.up == 1)
        {
            hipLaunchKernelGGL((k_conv3x3_il_o2<R, P>), grid, dim3(256), 0, stream, a);
            dcl_note_kernel("k_conv3x3_il_o2<%d,%d>", R, P);
        } else {
            // (channel counts that are not multiples of 16 -- hrnet18: the fenced tile at one workgroup per CU; its two-per-CU
            // build kept 12 B of scratch)
            hipLaunchKernelGGL((k_conv3x3<R, P, S>), grid, dim3(256), 0, stream, a);
            dcl_note_kernel("k_conv3x3<%d,%d,%d>", R, P, S);
        }
    } else if constexpr (S == 1) {
        if (g_conv_interleave && (a.Cin & 15) == 0 && a.up == 1) {
            hipLaunchKernelGGL((k_conv3x3_il<R, P>), grid, dim3(256), 0, stream, a);
            dcl_note_kernel("k_conv3x3_il<%d,%d>", R, P);
        } else {
            if constexpr (R == 3 && P == 4) {
                return DCL_EUNSUPPORTED;        // (see conv_f16x3: (3, 4) only on the interleaved staging)
            } else {
                hipLaunchKernelGGL((k_conv3x3<R, P, S>), grid, dim3(256), 0, stream, a);
                dcl_note_kernel("k_conv3x3<%d,%d,%d>", R, P, S);
            }
        }
    } else {
        if constexpr (S == 2 && P == 1) {
            if (g_conv_interleave && g_conv_s2_interleave && (a.Cin & 15) == 0 && a.up == 1) {
                hipLaunchKernelGGL((k_conv3x3_il_s2<R>), grid, dim3(256), 0, stream, a);
                dcl_note_kernel("k_conv3x3_il_s2<%d>", R);
                return 0;
            }
        }
        hipLaunchKernelGGL((k_conv3x3<R, P, S>), grid, dim3(256), 0, stream, a);
        dcl_note_kernel("k_conv3x3<%d,%d,%d>", R, P, S);
    }
    return 0;
}

static int g_up2_phases = 2;    // in_up = 2: 2 = all four output parity classes of a tile in one workgroup (k_conv3x3_pm), 1 = one class
                                // per workgroup (k_conv3x3_phases), 0 = zero-inserted input
static int g_conv_min_wgs = 96;         // automatic tile: fewest workgroups a launch may have before the rows per wave are halved.
// 192 (one round of 256 CUs three quarters full) until round 4; 96 since: the 192 / 384-channel branch convolutions then run as 96
// workgroups with twice the rows per wave -- 22-33 % less CU-time per launch at 55 % more latency -- and leave the rest of the chip to
// the other branches' kernels: step 90.7 -> 89.7 ms over three alternating pairs (profiles/r04_ab_conv_min_wgs.json)

// ---- stem: 3x3 / stride 2 / pad 1 convolution with <= 4 input channels (reference models/HRNet.py:404-405: conv1 = 3 -> 64 on the
// image), plain fp32 FMAs.  The tile kernels pad the contraction to a 16-channel chunk -- 5x the multiply-adds for 3 channels, each of
// them three MFMA passes -- and spend 0.46 ms on a layer whose traffic (75 MB in, 403 MB out at batch 12 x 512 x 1024) is worth 0.1 ms.
// One thread per output pixel, all output channels (<= 64 per pass) in its registers; the 17 x 65 x Cin input patch of an 8 x 32
// output tile and the weights ([ci][ky][kx][co]: a tap's channels are one broadcast 16-byte LDS read per four) are staged in LDS.
namespace {
constexpr int SC_TH = 8, SC_TW = 32, SC_PH = 2 * SC_TH + 1, SC_PW = 2 * SC_TW + 1, SC_CO = 64;

__global__ __launch_bounds__(256) void k_conv3x3_s2_smallcin(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ y, int N, int Cin,
                                                            int Cout, int H, int W, int Ho, int Wo, int tiles_x, int tiles_y)
{
    __shared__ float patch[4][SC_PH][SC_PW + 1];
    __shared__ __attribute__((aligned(16))) float wl[4 * 9 * SC_CO];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    b /= tiles_y;
    const int ngrp = (Cout + SC_CO - 1) / SC_CO;
    const int cgp = b % ngrp, n = b / ngrp;
    const int oy0 = ty * SC_TH, ox0 = tx * SC_TW, co0 = cgp * SC_CO;
    for (int i = tid; i < Cin * SC_PH * SC_PW; i += 256) {
        const int ci = i / (SC_PH * SC_PW), r = (i / SC_PW) % SC_PH, c = i % SC_PW;
        const int gy = 2 * oy0 + r - 1, gx = 2 * ox0 + c - 1;
        patch[ci][r][c] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[(((size_t)n * Cin + ci) * H + gy) * W + gx] : 0.f;
    }
    for (int i = tid; i < Cin * 9 * SC_CO; i += 256) {
        const int co = i % SC_CO, t = i / SC_CO;            // t = ci * 9 + tap
        const int ci = t / 9, tap = t - 9 * ci;
        wl[i] = co0 + co < Cout ? w[((size_t)(co0 + co) * Cin + ci) * 9 + tap] : 0.f;
    }
    __syncthreads();
    const int ly = tid >> 5, lx = tid & 31;
    float acc[SC_CO];
#pragma unroll
    for (int k = 0; k < SC_CO; ++k)
        acc[k] = 0.f;
    for (int ci = 0; ci < Cin; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float v = patch[ci][2 * ly + ky][2 * lx + kx];
                const f32x4 *wp = (const f32x4 *)(wl + (ci * 9 + ky * 3 + kx) * SC_CO);
#pragma unroll
                for (int k4 = 0; k4 < SC_CO / 4; ++k4) {
                    const f32x4 wv = wp[k4];
                    acc[4 * k4 + 0] = fmaf(wv.x, v, acc[4 * k4 + 0]);
                    acc[4 * k4 + 1] = fmaf(wv.y, v, acc[4 * k4 + 1]);
                    acc[4 * k4 + 2] = fmaf(wv.z, v, acc[4 * k4 + 2]);
                    acc[4 * k4 + 3] = fmaf(wv.w, v, acc[4 * k4 + 3]);
                }
            }
    const int oy = oy0 + ly, ox = ox0 + lx;
    if (oy < Ho && ox < Wo) {
        float *yp = y + (((size_t)n * Cout + co0) * Ho + oy) * Wo + ox;
#pragma unroll
        for (int k = 0; k < SC_CO; ++k)
            if (co0 + k < Cout)
                yp[(size_t)k * Ho * Wo] = acc[k] + (bias ? bias[co0 + k] : 0.f);
    }
}
}  // namespace

extern "C" int dcl_conv3x3_s2_smallcin(const float *x, int N, int Cin, int H, int W, const float *w, int Cout, const float *bias,
                                       float *y, void *stream)
{
    DCL_CHECK_ARG(x && w && y && N > 0 && H > 0 && W > 0 && Cout > 0, "bad arguments");
    DCL_CHECK_ARG(Cin >= 1 && Cin <= 4, "this kernel takes 1 .. 4 input channels");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int tiles_x = (Wo + SC_TW - 1) / SC_TW, tiles_y = (Ho + SC_TH - 1) / SC_TH, ngrp = (Cout + SC_CO - 1) / SC_CO;
    const long long blocks = (long long)tiles_x * tiles_y * ngrp * N;
    DCL_CHECK_ARG(blocks < (1LL << 31), "too many tiles");
    hipLaunchKernelGGL(k_conv3x3_s2_smallcin, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, N, Cin,
                       Cout, H, W, Ho, Wo, tiles_x, tiles_y);
    dcl_note_kernel("k_conv3x3_s2_smallcin");
    DCL_LAUNCH_CHECK();
    return 0;
}

// ---- stem: weight gradient of the same layer (3x3 / stride 2 / pad 1, Cin * 9 <= 32 taps, Cout <= 64), fp32 on the matrix pipe.
//   dw[co][ci][ky][kx] = sum over (n, oy, ox) of gy[n][co][oy][ox] * x[n][ci][2 oy + ky - 1][2 ox + kx - 1]
// is a GEMM  D[co][tap] += A[co][pixel] B[pixel][tap]  with the PIXELS as the contraction: v_mfma_f32_32x32x2f32 takes two pixels per
// instruction (lane (i, k): A = gy of channel i at pixel k of the pair, B = tap i of that pixel), products and sums in fp32.  The
// library's kernel for this shape (igemm_wrw ... gkgs, 0.41 ms at batch 12 x 512 x 1024) splits the pixels and adds the pieces with
// atomics -- the one launch of a training step whose result changed from run to run; the traffic (75 MB of image, 403 MB of gradient)
// is worth 0.1 ms.  Persistent workgroups: workgroup b takes the 128-pixel row segments b, b + G, b + 2 G, ... (a FIXED assignment),
// stages gy [Cout][128] and the three image rows [Cin][3][258] of a segment in LDS with coalesced loads, its four waves take 32
// pixels each and keep their 32 x 32 accumulators (one per 32 output channels) across all segments; at the end the waves' tiles are
// added in LDS in wave order and the workgroup writes ONE slab; k_wgrad_stem_sum adds the slabs in index order in double.
namespace {
constexpr int SW_PIX = 128, SW_XW = 2 * SW_PIX + 2;      // staged image row: columns 2 ox0 - 1 .. 2 ox0 + 256
constexpr int SW_GS = SW_PIX + 1;                        // LDS row stride of the staged gradient (odd: channel rows fall into different banks)

template <int MT>                                        // MT = Cout / 32 (1 | 2)
__global__ __launch_bounds__(256, 2) void k_wgrad_stem(const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ part,
                                                      int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int segs_x, long nseg)
{
    __shared__ float gl[32 * MT * SW_GS];
    __shared__ float xl[3 * 3 * SW_XW + 32];
    __shared__ float red[3][32 * MT][33];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, k = lane >> 5, i = lane & 31;
    // tap of this lane: i = ci * 9 + ky * 3 + kx (lanes with i >= Cin * 9 feed zeros)
    const int ntap = Cin * 9;
    const int tci = i / 9, tky = (i % 9) / 3, tkx = i % 3;
    const bool tap_ok = i < ntap;
    const int xoff = tap_ok ? (tci * 3 + tky) * SW_XW + tkx : 0;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q)
            acc[m][q] = 0.f;
    const size_t gplane = (size_t)Ho * Wo, xplane = (size_t)H * W;
    // Staging registers: a segment's values are loaded UNCONDITIONALLY from clamped addresses (a branch around a load makes the
    // compiler wait for it on the spot: 40 dependent round trips per segment, 0.60 ms for the whole launch) while the previous
    // segment's matrix instructions run, and written to LDS -- zeros where the mask says "outside" -- behind a barrier.
    constexpr int GN = 32 * MT * SW_PIX / 256;                 // gradient values per thread and segment (16 | 32)
    constexpr int XN = (3 * 3 * SW_XW + 255) / 256;            // image values per thread and segment (10)
    float gr[GN], xr[XN];
    unsigned gmask = 0, xmask = 0;
    auto issue = [&](long sg) {
        const int sx = (int)(sg % segs_x);
        const long r = sg / segs_x;
        const int oy = (int)(r % Ho), n = (int)(r / Ho);
        const int ox0 = sx * SW_PIX;
        gmask = 0, xmask = 0;
        const float *gb = gy + (size_t)n * Cout * gplane + (size_t)oy * Wo + ox0;
#pragma unroll
        for (int u = 0; u < GN; ++u) {
            const int e = tid + 256 * u, c = e >> 7, pp = e & (SW_PIX - 1);
            const bool ok = c < Cout && ox0 + pp < Wo;
            gmask |= ok ? (1u << u) : 0u;
            gr[u] = gb[ok ? (size_t)c * gplane + pp : 0];
        }
        const float *xb = x + (size_t)n * Cin * xplane;
#pragma unroll
        for (int u = 0; u < XN; ++u) {
            const int e = tid + 256 * u, row = e / SW_XW, c = e - row * SW_XW;
            const int ci = row / 3, ky = row - 3 * ci;
            const int yy = 2 * oy + ky - 1, xx = 2 * ox0 + c - 1;
            const bool ok = row < Cin * 3 && yy >= 0 && yy < H && xx >= 0 && xx < W;
            xmask |= ok ? (1u << u) : 0u;
            xr[u] = xb[ok ? (size_t)ci * xplane + (size_t)yy * W + xx : 0];
        }
    };
    long s = blockIdx.x;
    if (s < nseg)
        issue(s);
    for (; s < nseg; s += gridDim.x) {
        __syncthreads();                                 // the previous segment's operand reads are done
#pragma unroll
        for (int u = 0; u < GN; ++u) {
            const int e = tid + 256 * u;
            gl[(e >> 7) * SW_GS + (e & (SW_PIX - 1))] = ((gmask >> u) & 1) ? gr[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < XN; ++u) {
            const int e = tid + 256 * u;
            if (e < 3 * 3 * SW_XW)
                xl[e] = ((xmask >> u) & 1) ? xr[u] : 0.f;
        }
        __syncthreads();
        if (s + gridDim.x < nseg)
            issue(s + gridDim.x);                        // the next segment's loads fly during this one's matrix instructions
        // this wave's 32 pixels, two per instruction: pixel p = 32 wave + 2 t + k
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {
            const int p = 32 * wave + 2 * t + k;
            const float b = tap_ok ? xl[xoff + 2 * p] : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(gl[(32 * m + i) * SW_GS + p], b, acc[m], 0, 0, 0);
        }
    }
    // waves 1 .. 3 hand their tiles to wave 0 through LDS, added in wave order; accumulator register q of lane (k, i) is row
    // (q & 3) + 8 (q >> 2) + 4 k (output channel), column i (tap)
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                red[wave - 1][32 * m + (q & 3) + 8 * (q >> 2) + 4 * k][i] = acc[m][q];
    }
    __syncthreads();
    if (wave == 0) {
        float *dst = part + (size_t)blockIdx.x * 32 * MT * 32;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * m + (q & 3) + 8 * (q >> 2) + 4 * k;
                dst[row * 32 + i] = ((acc[m][q] + red[0][row][i]) + red[1][row][i]) + red[2][row][i];
            }
    }
}

// one workgroup per output channel: thread (g, t) adds the slabs g, g + 8, g + 16, ... of tap t in double (eight loads in flight per
// trip: one thread per output walking all 512 slabs was 512 dependent round trips, 0.3 ms), the eight partial sums are added in
// order g = 0 .. 7 -- a fixed order either way
__global__ __launch_bounds__(256) void k_wgrad_stem_sum(const float *__restrict__ part, int nslab, int rows, int ntap, int Cout,
                                                       float *__restrict__ dw)
{
    __shared__ double ps[8][32];
    const int co = blockIdx.x, t = threadIdx.x & 31, g = threadIdx.x >> 5;
    double acc = 0.0;
    const float *p = part + (size_t)co * 32 + t;
    const size_t stride = (size_t)rows * 32;
    for (int sl = g; sl < nslab; sl += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = p[(size_t)min(sl + 8 * u, nslab - 1) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (sl + 8 * u < nslab)
                acc += (double)v[u];
    }
    ps[g][t] = acc;
    __syncthreads();
    if (g == 0 && t < ntap) {
        double a = ps[0][t];
#pragma unroll
        for (int u = 1; u < 8; ++u)
            a += ps[u][t];
        dw[co * ntap + t] = (float)a;                    // dw is [Cout][Cin][3][3] = [Cout][ntap]
    }
}
}  // namespace

extern "C" int dcl_wgrad3x3_s2_smallcin_workspace(int Cout)
{
    return 512 * ((Cout + 31) / 32) * 32 * 32;           // floats: one [Cout rounded up][32] slab per workgroup, 512 workgroups
}

extern "C" int dcl_wgrad3x3_s2_smallcin(const float *x, int N, int Cin, int H, int W, const float *gy, int Cout, float *part,
                                        float *dw, void *stream)
{
    DCL_CHECK_ARG(x && gy && part && dw && N > 0 && H > 0 && W > 0, "bad arguments");
    DCL_CHECK_ARG(Cin >= 1 && Cin * 9 <= 32, "this kernel takes up to 3 input channels (27 taps in one 32-column tile)");
    DCL_CHECK_ARG(Cout >= 1 && Cout <= 64, "this kernel takes up to 64 output channels");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int segs_x = (Wo + SW_PIX - 1) / SW_PIX;
    const long nseg = (long)segs_x * Ho * N;
    const int G = (int)(nseg < 512 ? nseg : 512);        // two workgroups per CU
    const int MT = (Cout + 31) / 32;
    if (MT == 1)
        hipLaunchKernelGGL(k_wgrad_stem<1>, dim3(G), dim3(256), 0, (hipStream_t)stream, x, gy, part, N, Cin, H, W, Cout, Ho, Wo, segs_x, nseg);
    else
        hipLaunchKernelGGL(k_wgrad_stem<2>, dim3(G), dim3(256), 0, (hipStream_t)stream, x, gy, part, N, Cin, H, W, Cout, Ho, Wo, segs_x, nseg);
    dcl_note_kernel("k_wgrad_stem");
    hipLaunchKernelGGL(k_wgrad_stem_sum, dim3(Cout), dim3(256), 0, (hipStream_t)stream, part, G, 32 * MT, Cin * 9, Cout, dw);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_conv3x3_set_min_workgroups(int n)
{
    g_conv_min_wgs = n > 0 ? n : 96;
    return 0;
}

extern "C" int dcl_conv3x3_set_up2_phases(int on)
{
    g_up2_phases = on < 0 ? 0 : (on > 2 ? 2 : on);       // 0 zero-inserted input, 1 one class per workgroup, 2 all four classes per workgroup
    return 0;
}

static int conv_f16x3(const float *x, int N, int Cin, int H, int W, const void *wp, int Cout, const float *xamax,
                      int xcount, const float *wamax, const float *addend, const float *bias, float *y, int stride,
                      int in_up, int Hout, int Wout, int tile_r, int tile_p, int onetap, void *stream,
                      const float *pre_sc = nullptr, const float *pre_sh = nullptr);
static void auto_tile(int N, int Cout, int Ho, int Wo, int nchunk, int stride, int phases, int onetap, int tile_r, int tile_p,
                      int &R, int &P);

extern "C" int dcl_conv3x3_f16x3(const float *x, int N, int Cin, int H, int W, const void *wp, int Cout,
                                 const float *xamax, int xcount, const float *wamax, const float *addend,
                                 const float *bias, float *y, int stride, int in_up, int Hout, int Wout,
                                 int tile_r, int tile_p, void *stream)
{
    return conv_f16x3(x, N, Cin, H, W, wp, Cout, xamax, xcount, wamax, addend, bias, y, stride, in_up, Hout, Wout, tile_r,
                      tile_p, 0, stream);
}

extern "C" int dcl_conv1x1_f16x3(const float *x, int N, int Cin, int H, int W, const void *wp, int Cout,
                                 const float *xamax, int xcount, const float *wamax, const float *addend,
                                 const float *bias, float *y, int tile_r, int tile_p, void *stream)
{
    return conv_f16x3(x, N, Cin, H, W, wp, Cout, xamax, xcount, wamax, addend, bias, y, 1, 1, 0, 0, tile_r, tile_p, 1,
                      stream);
}

// The convolution of relu(x * pre_sc[c] + pre_sh[c]) -- x: the RAW output of the convolution in front of a training-mode norm,
// (pre_sc, pre_sh): that norm's per-channel map from dcl_bn_finalize_pre, xamax: the absmax slots the same call wrote -- without
// the normalised tensor in memory (dcl_conv3x3_pre.hip).  3x3 / pad 1, stride 1 or 2, Cin % 16 == 0; DCL_EUNSUPPORTED (nothing
// launched) when the automatic tile has no such instantiation: ask dcl_conv3x3_pre_supported first.
extern "C" int dcl_conv3x3_pre_f16x3(const float *x, int N, int Cin, int H, int W, const void *wp, int Cout,
                                     const float *xamax, int xcount, const float *wamax, const float *pre_sc,
                                     const float *pre_sh, const float *bias, float *y, int stride, int tile_r, int tile_p,
                                     void *stream)
{
    DCL_CHECK_ARG(pre_sc && pre_sh, "null pointer");
    DCL_CHECK_ARG((Cin & 15) == 0, "Cin must be a multiple of 16");
    DCL_CHECK_ARG(((((uintptr_t)pre_sc) | ((uintptr_t)pre_sh)) & 3) == 0, "tables must be 4-byte aligned");
    return conv_f16x3(x, N, Cin, H, W, wp, Cout, xamax, xcount, wamax, nullptr, bias, y, stride, 1, 0, 0, tile_r, tile_p, 0,
                      stream, pre_sc, pre_sh);
}

extern "C" int dcl_conv3x3_pre_supported(int N, int Cin, int Cout, int H, int W, int stride)
{
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (Cin & 15) || (stride != 1 && stride != 2) || !g_conv_interleave)
        return 0;
    if ((size_t)8 * H * W + (size_t)H * W >= ((size_t)1 << 31))
        return 0;
    const int Ho = stride == 2 ? (H - 1) / 2 + 1 : H, Wo = stride == 2 ? (W - 1) / 2 + 1 : W;
    int R, P;
    auto_tile(N, Cout, Ho, Wo, Cin / 16, stride, 0, 0, 0, 0, R, P);
    const bool ws2 = stride == 1 && R == 2 && P == 2;
    if ((ws2 && g_conv_interleave != 2) || (stride == 2 && !g_conv_s2_interleave))
        return 0;
    return dcl_conv_pre_has_tile(R, P, stride, ws2) ? 1 : 0;
}

static void auto_tile(int N, int Cout, int Ho, int Wo, int nchunk, int stride, int phases, int onetap, int tile_r, int tile_p,
                      int &R, int &P)
{
    const int mtiles = (Cout + 31) / 32;
    if (phases == 2) {
        // merged parity classes: 4 R P accumulator tiles per wave, at most 12 (16 spill) -- (3, 1), (2, 1), (1, 2 | 1); tiles run
        // over the STORED gradient, ceil(Ho / 2) x ceil(Wo / 2)
        R = mtiles % 3 == 0 ? 3 : (mtiles == 1 ? 1 : 2);
        P = R == 1 ? 2 : 1;
        auto wgs = [&](int p) {
            return (long)(((Wo + 1) / 2 + TW - 1) / TW) * (((Ho + 1) / 2 + 4 * p - 1) / (4 * p)) * N * ((mtiles + R - 1) / R);
        };
        while (P > 1 && wgs(P) < g_conv_min_wgs)
            P >>= 1;
        if (tile_r > 0 && tile_p > 0 && 4 * tile_r * tile_p <= 12)
            R = tile_r, P = tile_p;
        return;
    }
    R = tile_r, P = tile_p;
    if (stride == 2)
        P = 1;
    if (R <= 0 || P <= 0 || (stride == 2 && tile_r <= 0)) {
        // measured on the four BasicBlock shapes of HRNet-W48 at batch 12 (tools/bench_conv3x3.py --tiles):
        // channel tiles per wave in threes when that leaves no padded tile, else pairs; the most rows per wave
        // that still give ~one workgroup per CU; tiny images fall back to single-tile waves to get enough
        // workgroups.
        R = (mtiles % 3 == 0 || mtiles >= 20) ? 3 : (mtiles == 1 ? 1 : 2);     // >= 20 tiles: one padded tile is < 5 %
        if (mtiles == 5 && stride == 1 && !onetap)
            R = 3;      // five tiles pad to six either way, and the (3, P) wave runs 324 P / 4 MFMAs per staged chunk against 216:
                        // the head's 720 -> 144 data gradient (45 chunks), (2, 4) tile 4.17 ms = 0.21 of the roofline
        auto wgs = [&](int r, int p) {
            if (phases)
                return (long)(((Wo + 1) / 2 + TW - 1) / TW) * (((Ho + 1) / 2 + 4 * p - 1) / (4 * p)) * N *
                       ((mtiles + r - 1) / r) * 4;
            return (long)((Wo + TW - 1) / TW) * ((Ho + 4 * p - 1) / (4 * p)) * N * ((mtiles + r - 1) / r);
        };
        P = stride == 2 ? 1 : 4;
        while (P > 1 && wgs(R, P) < g_conv_min_wgs)
            P >>= 1;
        if (R == 2 && P == 4 && stride == 1 && !phases && nchunk <= 4)
            P = 2;              // the (2, 2) tile runs two workgroups per CU (k_conv3x3_o2): 81 vs 100 us at 48 channels;
                                // a short K loop is mostly prologue / epilogue, which two co-resident workgroups
                                // hide.  With a long one (the 512-channel decoder convolutions of UPerNet) the (2, 4)
                                // tile's reuse of the weight fragments wins: 415 vs 386 TFLOP/s
        if (P == 1 && wgs(R, P) < 128)
            R = 1;
    }
}

static int conv_f16x3(const float *x, int N, int Cin, int H, int W, const void *wp, int Cout, const float *xamax,
                      int xcount, const float *wamax, const float *addend, const float *bias, float *y, int stride,
                      int in_up, int Hout, int Wout, int tile_r, int tile_p, int onetap, void *stream,
                      const float *pre_sc, const float *pre_sh)
{
    DCL_CHECK_ARG(x && wp && xamax && wamax && y, "null pointer");
    DCL_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && xcount > 0, "bad shape");
    DCL_CHECK_ARG((stride == 1 || stride == 2) && (in_up == 1 || in_up == 2) && !(stride == 2 && in_up == 2),
                  "stride / in_up must be 1 or 2 (not both 2)");
    DCL_CHECK_ARG((size_t)8 * H * W + (size_t)H * W < ((size_t)1 << 31), "image plane too large");
    ConvArgs a;
    a.x = x;
    a.wp = (const uint4 *)wp;
    a.y = y;
    a.addend = addend;
    a.bias = bias;
    a.xamax = xamax;
    a.wamax = wamax;
    a.xcount = xcount;
    a.N = N;
    a.Cin = Cin;
    a.Cout = Cout;
    a.Hs = H;
    a.Ws = W;
    a.up = in_up;
    a.phases = 0;
    a.onetap = onetap;
    a.pre_sc = pre_sc;
    a.pre_sh = pre_sh;
    if (in_up == 2 && g_up2_phases) {
        // one parity class of the output per workgroup, over the stored input (see conv_body, PH)
        DCL_CHECK_ARG(Hout > 0 && Wout > 0 && (Hout - 1) / 2 + 1 == H && (Wout - 1) / 2 + 1 == W,
                      "in_up = 2: H, W must be ceil(Hout / 2), ceil(Wout / 2)");
        a.up = 1;
        a.phases = 1;
        if (g_up2_phases >= 2) {
            // all four classes per workgroup where that still gives a round of workgroups (a quarter of the per-class grid):
            // on the HRNet-W48 shapes at batch 12 (gpurun_out/r4j, r4k) the merged form wins from 192 workgroups up -- 98 -> 71 us
            // (96 -> 48 at 64 x 128), 58 -> 44 (192 -> 48 at 32 x 64), 72 -> 58 (192 -> 96) -- and loses on the 16 x 32 maps (48-96
            // workgroups of 24-chunk K loops), which keep the per-class kernel
            int Rm, Pm;
            auto_tile(N, Cout, Hout, Wout, (Cin + 15) / 16, 1, 2, onetap, tile_r, tile_p, Rm, Pm);
            const long wg = (long)((W + TW - 1) / TW) * ((H + 4 * Pm - 1) / (4 * Pm)) * N * (((Cout + 31) / 32 + Rm - 1) / Rm);
            if (wg >= g_conv_min_wgs)
                a.phases = 2;
        }
        a.H = H;
        a.W = W;
    } else if (in_up == 2) {
        // the virtual (zero-inserted) input has the size of the output; the stored one must be its even samples
        DCL_CHECK_ARG(Hout > 0 && Wout > 0 && (Hout - 1) / 2 + 1 == H && (Wout - 1) / 2 + 1 == W,
                      "in_up = 2: H, W must be ceil(Hout / 2), ceil(Wout / 2)");
        a.H = Hout;
        a.W = Wout;
    } else {
        a.H = H;
        a.W = W;
    }
    a.Ho = a.phases ? Hout : (stride == 2 ? (a.H - 1) / 2 + 1 : a.H);
    a.Wo = a.phases ? Wout : (stride == 2 ? (a.W - 1) / 2 + 1 : a.W);
    DCL_CHECK_ARG((Hout <= 0 || Hout == a.Ho) && (Wout <= 0 || Wout == a.Wo), "Hout / Wout do not match the geometry");
    a.nchunk = (Cin + 15) / 16;
    int R, P;
    auto_tile(N, Cout, a.Ho, a.Wo, a.nchunk, stride, a.phases, onetap, tile_r, tile_p, R, P);
    // the (3, 4) tile exists on the interleaved staging only (its fenced and one-tap instantiations spilled: 288 / 104 B of scratch)
    if (R == 3 && P == 4 && !(g_conv_interleave && (Cin & 15) == 0 && a.up == 1 && !onetap && stride == 1))
        P = 2;
    hipStream_t s = (hipStream_t)stream;
#define DCL_CONV_CASE(r, p, st)                                                                              \
    if (R == r && P == p && stride == st) {                                                                  \
        if (launch_conv<r, p, st>(a, s) != 0) {                                                              \
            dcl_set_error("dcl_conv3x3: no kernel for tile (%d, %d), stride %d on this input", r, p, st);     \
            return DCL_EUNSUPPORTED;                                                                         \
        }                                                                                                    \
        DCL_LAUNCH_CHECK();                                                                                  \
        return 0;                                                                                            \
    }
    DCL_CONV_CASE(1, 1, 1)
    DCL_CONV_CASE(1, 2, 1)
    DCL_CONV_CASE(1, 4, 1)
    DCL_CONV_CASE(2, 1, 1)
    DCL_CONV_CASE(2, 2, 1)
    DCL_CONV_CASE(2, 4, 1)
    DCL_CONV_CASE(3, 1, 1)
    DCL_CONV_CASE(3, 2, 1)
    DCL_CONV_CASE(3, 4, 1)
    DCL_CONV_CASE(1, 1, 2)
    DCL_CONV_CASE(2, 1, 2)
    DCL_CONV_CASE(3, 1, 2)
#undef DCL_CONV_CASE
    dcl_set_error("dcl_conv3x3_f16x3: unsupported tile (R=%d, P=%d, stride=%d)", R, P, stride);
    return DCL_EINVAL;
}
