// dcl_resize.hip -- bilinear up-sampling of NCHW f32 maps, forward and (gather-form, deterministic)
// backward.  HRNet performs 35 of these per step (exchange-module fusions models/HRNet.py:279-282, the
// 720-channel concat :549-551 and the logits up-sampling :638 in the reference); 3.3 GB of output per
// step at BASELINE config 2.  PyTorch's kernel reaches ~270 GB/s there (one thread per output element,
// scalar stores) and its backward uses float atomics; this one writes 16 B per lane and gathers.
//
// Index arithmetic restates ATen's (UpSample.h: area_pixel_compute_scale / _source_index, f32):
//   align_corners: scale = (in-1)/(out-1) (0 if out == 1), src = scale * dst
//   otherwise    : scale = in/out,                         src = max(scale * (dst + 0.5) - 0.5, 0)
//   i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1
#include "dcl_common.h"

namespace {

struct Axis {
    float scale;
    int align;
};

__device__ __forceinline__ void src_index(const Axis a, int dst, int in_size, int &i0, int &i1, float &l0,
                                          float &l1)
{
    float s = a.align ? a.scale * (float)dst : fmaxf(a.scale * ((float)dst + 0.5f) - 0.5f, 0.f);
    i0 = (int)s;
    if (i0 > in_size - 1)
        i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

// The high-resolution tensor may be a channel slice [c0, c0 + C) of a wider [N, ctot, H, W] tensor (the concatenation
// the up-sampled maps go into): plane p = n * C + c of the low-resolution tensor <-> plane n * ctot + c0 + c there.
struct Slice {
    int C, ctot, c0;
};
__device__ __forceinline__ size_t wide_plane(const Slice &s, size_t p) { return (p / s.C) * s.ctot + s.c0 + p % s.C; }

// `tpr` threads (a power of two <= 256) per output row (n, c, oy), 256 / tpr rows per workgroup; each thread produces 4
// consecutive ox (narrow maps: a 256-pixel row keeps 64 threads busy, so four rows share a workgroup)
__global__ __launch_bounds__(256) void k_upsample_fwd(const float *__restrict__ x,
                                                     const float *__restrict__ addend, int h, int w,
                                                     int H, int W, Axis ay, Axis ax, int relu, float *__restrict__ y,
                                                     Slice sl, int tpr, int nrows)
{
    const int row = blockIdx.x * (256 / tpr) + threadIdx.x / tpr;              // (n*C + c) * H + oy
    if (row >= nrows)
        return;
    const int tx = threadIdx.x & (tpr - 1);
    const int oy = row % H;
    const size_t plane = row / H;
    int y0, y1;
    float ly0, ly1;
    src_index(ay, oy, h, y0, y1, ly0, ly1);
    const float *r0 = x + (plane * h + y0) * (size_t)w;
    const float *r1 = x + (plane * h + y1) * (size_t)w;
    float *out = y + (wide_plane(sl, plane) * H + oy) * (size_t)W;
    const float *add = addend ? addend + (size_t)row * W : nullptr;     // y = addend + upsample(x)
    const bool vec = (W & 3) == 0;
    for (int ox4 = tx * 4; ox4 < W; ox4 += tpr * 4) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ox = ox4 + k;
            if (ox < W) {
                int x0, x1;
                float lx0, lx1;
                src_index(ax, ox, w, x0, x1, lx0, lx1);
                v[k] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
            } else {
                v[k] = 0.f;
            }
        }
        if (vec) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            if (add) {
                const f32x4 a = *(const f32x4 *)(add + ox4);
                o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
            }
            if (relu) {
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            *(f32x4 *)(out + ox4) = o;
        } else {
            for (int k = 0; k < 4 && ox4 + k < W; ++k) {
                const float t = v[k] + (add ? add[ox4 + k] : 0.f);
                out[ox4 + k] = relu ? fmaxf(t, 0.f) : t;
            }
        }
    }
}

// weight of input index `i` in output index `o` along one axis
__device__ __forceinline__ float axis_weight(const Axis a, int o, int in_size, int i)
{
    int i0, i1;
    float l0, l1;
    src_index(a, o, in_size, i0, i1, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// candidate output range touching input index i (conservative; weights decide)
__host__ __device__ __forceinline__ void out_range(const Axis a, int i, int in_size, int out_size, int &lo, int &hi)
{
    // src(o) is non-decreasing in o; input i is touched when src(o) in (i-1, i+1)
    const float inv = a.scale > 0.f ? 1.f / a.scale : 0.f;
    float flo, fhi;
    if (a.align) {
        flo = ((float)i - 1.f) * inv;
        fhi = ((float)i + 1.f) * inv;
    } else {
        flo = ((float)i - 1.f + 0.5f) * inv - 0.5f;
        fhi = ((float)i + 1.f + 0.5f) * inv - 0.5f;
    }
    lo = (int)floorf(flo) - 1;
    hi = (int)ceilf(fhi) + 1;
    if (a.scale <= 0.f) {
        lo = 0;
        hi = out_size - 1;
    }
    lo = lo < 0 ? 0 : lo;
    hi = hi > out_size - 1 ? out_size - 1 : hi;
}

// one thread per input element: dx[n,c,iy,ix] = sum_{oy,ox} wy(iy,oy) * wx(ix,ox) * dy[n,c,oy,ox]
__global__ __launch_bounds__(256) void k_upsample_bwd(const float *__restrict__ dy, int h, int w, int H,
                                                     int W, Axis ay, Axis ax, size_t total,
                                                     float *__restrict__ dx, Slice sl)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total)
        return;
    const int ix = (int)(e % w);
    const size_t t = e / w;
    const int iy = (int)(t % h);
    const size_t plane = t / h;
    int oy_lo, oy_hi, ox_lo, ox_hi;
    out_range(ay, iy, h, H, oy_lo, oy_hi);
    out_range(ax, ix, w, W, ox_lo, ox_hi);
    const float *g = dy + wide_plane(sl, plane) * (size_t)H * W;
    // separable: the x-weights of this input column are evaluated once, not once per output row
    constexpr int MAXC = 16;
    float wxs[MAXC];
    const int nx = ox_hi - ox_lo + 1;
    if (nx <= MAXC) {
#pragma unroll
        for (int k = 0; k < MAXC; ++k)
            wxs[k] = k < nx ? axis_weight(ax, ox_lo + k, w, ix) : 0.f;
    }
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const float wy = axis_weight(ay, oy, h, iy);
        if (wy == 0.f)
            continue;
        float rowacc = 0.f;
        const float *gr = g + (size_t)oy * W + ox_lo;
        if (nx <= MAXC) {
#pragma unroll
            for (int k = 0; k < MAXC; ++k)
                if (k < nx)
                    rowacc += wxs[k] * gr[k];
        } else {                                   // down-scaling or huge factors: generic path
            for (int k = 0; k < nx; ++k)
                rowacc += axis_weight(ax, ox_lo + k, w, ix) * gr[k];
        }
        acc += wy * rowacc;
    }
    dx[e] = acc;
}

// Row form of the backward: one workgroup per INPUT row (n, c, iy).  Vertical pass: the output rows that touch iy
// (<= 2 * scale + 3 of them) are read once each, 16 B per lane, coalesced, and accumulated with their y-weights into
// an LDS row; horizontal pass: every input column gathers its x-footprint from that row.  Every output row is read by
// the two input rows it interpolates between -- 2x the tensor, coalesced -- where the per-element gather above reads
// each value four times through short uncoalesced runs (x8 fuse-layer up-sampling: 19 x 19 values per thread).
// Same weights, fixed summation order -> deterministic.
constexpr int ROWS_MAXW = 4096;

__global__ __launch_bounds__(256) void k_upsample_bwd_rows(const float *__restrict__ dy, int h, int w, int H, int W,
                                                          Axis ay, Axis ax, float *__restrict__ dx, Slice sl)
{
    // (dynamic: W floats -- a static 16-KiB row limited the CU to 10 of these one-wave workgroups)
    extern __shared__ __attribute__((aligned(16))) float tmp[];
    const int row = blockIdx.x;              // plane * h + iy
    const int iy = row % h;
    const size_t plane = row / h;
    int oy_lo, oy_hi;
    out_range(ay, iy, h, H, oy_lo, oy_hi);
    const float *g = dy + wide_plane(sl, plane) * (size_t)H * W;
    for (int ox4 = threadIdx.x * 4; ox4 < W; ox4 += blockDim.x * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float wy = axis_weight(ay, oy, h, iy);          // wave-uniform
            if (wy == 0.f)
                continue;
            const f32x4 v = *(const f32x4 *)(g + (size_t)oy * W + ox4);
            acc.x += wy * v.x;
            acc.y += wy * v.y;
            acc.z += wy * v.z;
            acc.w += wy * v.w;
        }
        *(f32x4 *)(tmp + ox4) = acc;
    }
    __syncthreads();
    for (int ix = threadIdx.x; ix < w; ix += blockDim.x) {
        int ox_lo, ox_hi;
        out_range(ax, ix, w, W, ox_lo, ox_hi);
        float acc = 0.f;
        for (int ox = ox_lo; ox <= ox_hi; ++ox)
            acc += axis_weight(ax, ox, w, ix) * tmp[ox];
        dx[(size_t)row * w + ix] = acc;
    }
}

Axis make_axis(int in_size, int out_size, int align)
{
    Axis a;
    a.align = align;
    if (align)
        a.scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    else
        a.scale = (float)in_size / (float)out_size;
    return a;
}

}  // namespace

static int upsample_fwd(const float *x, const float *addend, int planes, int h, int w, int H, int W, int align_corners,
                        int relu, float *y, Slice sl, void *stream)
{
    DCL_CHECK_ARG(x && y && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bad arguments");
    DCL_CHECK_ARG((long long)planes * H < 2147483647LL, "too many output rows");
    int tpr = 1;
    while (tpr < 256 && tpr * 4 < W)
        tpr *= 2;
    const int nrows = planes * H, rpb = 256 / tpr;
    hipLaunchKernelGGL(k_upsample_fwd, dim3((unsigned)((nrows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, x,
                       addend, h, w, H, W, make_axis(h, H, align_corners), make_axis(w, W, align_corners), relu, y, sl, tpr,
                       nrows);
    DCL_LAUNCH_CHECK();
    return 0;
}

static int upsample_bwd(const float *dy, int planes, int h, int w, int H, int W, int align_corners, float *dx, Slice sl,
                        void *stream)
{
    DCL_CHECK_ARG(dy && dx && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bad arguments");
    const size_t total = (size_t)planes * h * w;
    if ((W & 3) == 0 && W <= ROWS_MAXW && H >= h && W >= w && (long long)planes * h < 2147483647LL &&
        (((uintptr_t)dy) & 15) == 0) {
        int threads = ((W / 4 + 63) / 64) * 64;
        threads = threads < 64 ? 64 : (threads > 256 ? 256 : threads);
        hipLaunchKernelGGL(k_upsample_bwd_rows, dim3((unsigned)(planes * h)), dim3(threads), (size_t)W * sizeof(float),
                           (hipStream_t)stream, dy,
                           h, w, H, W, make_axis(h, H, align_corners), make_axis(w, W, align_corners), dx, sl);
        DCL_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_upsample_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, h, w, H, W, make_axis(h, H, align_corners),
                       make_axis(w, W, align_corners), total, dx, sl);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_upsample_bilinear_fwd(const float *x, const float *addend, int planes, int h, int w,
                                         int H, int W, int align_corners, int relu, float *y, void *stream)
{
    return upsample_fwd(x, addend, planes, h, w, H, W, align_corners, relu, y, Slice{planes, planes, 0}, stream);
}

extern "C" int dcl_upsample_bilinear_bwd(const float *dy, int planes, int h, int w, int H, int W,
                                         int align_corners, float *dx, void *stream)
{
    return upsample_bwd(dy, planes, h, w, H, W, align_corners, dx, Slice{planes, planes, 0}, stream);
}

// Up-sampling straight into / out of a channel slice of a wider tensor (the concatenation of the four HRNet branches,
// reference models/HRNet.py:549-553): y_wide[n, c0 + c] = up(x[n, c]); backward reads dy_wide[n, c0 + c] in place.
extern "C" int dcl_upsample_bilinear_fwd_slice(const float *x, int N, int C, int h, int w, int H, int W,
                                               int align_corners, float *y_wide, int ctot, int c0, void *stream)
{
    DCL_CHECK_ARG(N > 0 && C > 0 && ctot >= c0 + C && c0 >= 0, "bad slice");
    return upsample_fwd(x, nullptr, N * C, h, w, H, W, align_corners, 0, y_wide, Slice{C, ctot, c0}, stream);
}

extern "C" int dcl_upsample_bilinear_bwd_slice(const float *dy_wide, int ctot, int c0, int N, int C, int h, int w, int H,
                                               int W, int align_corners, float *dx, void *stream)
{
    DCL_CHECK_ARG(N > 0 && C > 0 && ctot >= c0 + C && c0 >= 0, "bad slice");
    return upsample_bwd(dy_wide, N * C, h, w, H, W, align_corners, dx, Slice{C, ctot, c0}, stream);
}

// ---- 3x3 convolution of a bilinearly up-sampled map WITHOUT the up-sampled map ("tap-up") --------------------------------
// The head of HRNet convolves cat(x0, up(x1), up(x2), up(x3)) with a 3x3 kernel (reference models/HRNet.py:549-553, :596-600).
// For an up-sampled input the two linear maps commute:
//     conv3x3(up(x), W)[co, Y, X] = sum_tap (S_tap U)(W_tap x)[co, Y, X],     z[tap] = W_tap x  (a 1x1 convolution at LOW resolution)
// with S_tap the shift by the tap offset (zero outside the image: the convolution's padding) and U the bilinear interpolation.
// The 9 x Cout channel products z run on a fraction of the pixels (1/16 and 1/64 for HRNet's two coarsest branches: 80 % of the
// head's multiply-adds), and what is left at full resolution is this gather: 9 taps x 4 neighbours per source.
//   forward : y[n, co, Y, X] (+)= sum_src sum_tap interp(z_src[n, tap * Co + co], (Y + ky - 1, X + kx - 1))     k_tapup_fwd
//   backward: dz[n, tap * Co + co, r, c] = sum_{Y, X} wy(Y + ky - 1 -> r) wx(X + kx - 1 -> c) dy[n, co, Y, X]     k_tapup_bwd
// Both are separable (vertical pass through LDS at the low-resolution columns, then the horizontal one), gather form,
// fixed summation order.  Index arithmetic: ATen's, as everywhere in this file.
namespace {

struct TapSrc {
    const float *z;        // plane (n, tap * Co + co) at z + n * sn + (tap * Co + co) * sc, [h][w] each
    long long sn, sc;      // batch / channel stride in floats ([N][9 Co][h][w]: 9 Co h w, h w; [9 Co][N][h][w]: h w, N h w)
    int h, w;
    Axis ay, ax;
    int wr, wc;            // LDS window reserved for this source: rows, columns (+ 1 pad column: wc counts it)
    int zoff;              // float offset of the window inside the workgroup's LDS
};

constexpr int TAP_RC = 32;         // output rows per workgroup (forward)
constexpr int TAP_TW = 256;        // output columns per workgroup = threads
constexpr size_t TAP_FWD_LDS_MAX = 128 * 1024;     // source windows of a forward tile (dynamic LDS; above 64 KB one workgroup per CU)

// Forward.  One workgroup = one (n, co) plane x 32 output rows x 256 output columns.  The low-resolution windows of BOTH
// sources (9 taps each) that the tile's shifted coordinates interpolate from are staged once (a duplicate of the last
// column pads every window row, so that the right-hand neighbour c0 + 1 always exists: at the border its weight only
// multiplies the same value); per output row the (row pair, weights) of each shifted row come from an LDS table, the
// (column pair, weights) of the thread's three shifted columns live in registers; 2 x 9 taps x 4 neighbours per output.
__global__ __launch_bounds__(TAP_TW) void k_tapup_fwd(TapSrc s0, TapSrc s1, int nsrc, int Co, int H, int W, int tiles_y,
                                                     int tiles_x, int tabfloats, float *__restrict__ y, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) float tap_lds[];
    // row table: for source si and shifted row Ys = Y0 - 1 + j (j in [0, RC + 2)): {r0 - ra, r1 - ra, l0, l1} (l = 0 outside)
    float *rtab = tap_lds;
    float *Zbase = tap_lds + tabfloats;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int plane = b / tiles_y;          // n * Co + co
    const int n = plane / Co, co = plane - n * Co;
    const int Y0 = ty * TAP_RC, X0 = tx * TAP_TW;
    const int t = threadIdx.x, X = X0 + t;
    int cix[2][3];
    float cw0[2][3], cw1[2][3];
#pragma unroll
    for (int si = 0; si < 2; ++si) {
        if (si >= nsrc)
            break;
        const TapSrc s = si == 0 ? s0 : s1;
        int ra, rb, ca, cb, d0, d1;
        float f0, f1;
        src_index(s.ay, max(Y0 - 1, 0), s.h, ra, d1, f0, f1);
        src_index(s.ay, min(Y0 + TAP_RC, H - 1), s.h, d0, rb, f0, f1);
        src_index(s.ax, max(X0 - 1, 0), s.w, ca, d1, f0, f1);
        src_index(s.ax, min(X0 + TAP_TW, W - 1), s.w, d0, cb, f0, f1);
        const int nr = rb - ra + 1, nc = cb - ca + 1;           // <= s.wr, s.wc - 1 (host-side bound)
        const float *zp = s.z + (size_t)n * s.sn + (size_t)co * s.sc;
        const size_t tapstride = (size_t)Co * s.sc;
        float *Zw = Zbase + s.zoff;                             // [9][wr][wc]
        {
            // flat element loop (many independent loads per thread); quotients by multiplication with the reciprocal, exact
            // for these ranges (e < 2^15, divisors < 2^9), instead of two integer divisions per element
            const float inv_c = 1.0f / (float)(nc + 1), inv_r = 1.0f / (float)nr;
            // eight loads in flight per thread: one load -> store per trip leaves the window staging bound by the
            // memory latency (~33 trips per thread)
            const int total = 9 * nr * (nc + 1);
            for (int e0 = t; e0 < total; e0 += 8 * TAP_TW) {
                float v[8];
                int dsti[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u * TAP_TW, total - 1);
                    const int row = (int)(((float)e + 0.5f) * inv_c);
                    const int c = e - row * (nc + 1);
                    const int tap = (int)(((float)row + 0.5f) * inv_r);
                    const int r = row - tap * nr;
                    dsti[u] = (tap * s.wr + r) * s.wc + c;
                    v[u] = zp[tap * tapstride + (size_t)(ra + r) * s.w + ca + min(c, nc - 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * TAP_TW < total)
                        Zw[dsti[u]] = v[u];
            }
        }
        for (int j = t; j < TAP_RC + 2; j += TAP_TW) {
            const int Ys = Y0 - 1 + j;
            int r0 = ra, r1 = ra;
            float l0 = 0.f, l1 = 0.f;
            if (Ys >= 0 && Ys < H)
                src_index(s.ay, Ys, s.h, r0, r1, l0, l1);
            float *e = rtab + (si * (TAP_RC + 2) + j) * 4;
            e[0] = __int_as_float((r0 - ra) * s.wc);
            e[1] = __int_as_float((r1 - ra) * s.wc);
            e[2] = l0;
            e[3] = l1;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int Xs = X + kx - 1;
            int c0 = ca, c1 = ca;
            float l0 = 0.f, l1 = 0.f;
            if (Xs >= 0 && Xs < W && X < W)
                src_index(s.ax, Xs, s.w, c0, c1, l0, l1);
            cix[si][kx] = c0 - ca;
            cw0[si][kx] = l0;
            cw1[si][kx] = (c1 == c0) ? 0.f : l1;                // right neighbour = c0 + 1 or (border) weightless
            if (c1 == c0)
                cw0[si][kx] = l0 + l1;
        }
    }
    __syncthreads();
    if (X >= W)
        return;
    float *yp = y + ((size_t)plane * H + Y0) * W + X;
    // Consecutive output rows interpolate from the SAME pair of source rows (4 of them at scale 4, 8 at scale 8): the
    // horizontally interpolated values of a (source, ky, kx) are kept in registers and re-read from LDS only when the
    // row pair of that shifted row changes (a wave-uniform test) -- per output ~14 LDS reads and ~45 FMA instead of 72 / 108.
    // The row-table entries of the three shifted rows live in registers and rotate: ONE new entry per source and output row
    // is read from LDS, a row ahead of its use (six dependent LDS reads + scalar tests per row had bound the loop).
    float ha[2][3][3], hb[2][3][3];
    f32x4 rt[2][3];
    int o0[2][3], o1[2][3];
    bool ch[2][3];
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            rt[si][ky] = *(const f32x4 *)(rtab + (si * (TAP_RC + 2) + ky) * 4);
            o0[si][ky] = __builtin_amdgcn_readfirstlane(__float_as_int(rt[si][ky].x));
            o1[si][ky] = __builtin_amdgcn_readfirstlane(__float_as_int(rt[si][ky].y));
            ch[si][ky] = true;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                ha[si][ky][kx] = hb[si][ky][kx] = 0.f;
        }
    for (int yy = 0; yy < TAP_RC && Y0 + yy < H; ++yy) {
        f32x4 nx[2];
#pragma unroll
        for (int si = 0; si < 2; ++si)          // entry of shifted row yy + 3 (the table is TAP_RC + 2 long: clamp)
            nx[si] = *(const f32x4 *)(rtab + (si * (TAP_RC + 2) + min(yy + 3, TAP_RC + 1)) * 4);
        float acc = 0.f;
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            if (si >= nsrc)
                break;
            const TapSrc s = si == 0 ? s0 : s1;
            const float *Zw = Zbase + s.zoff;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if (ch[si][ky]) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float *zt = Zw + (size_t)(ky * 3 + kx) * s.wr * s.wc + cix[si][kx];
                        ha[si][ky][kx] = cw0[si][kx] * zt[o0[si][ky]] + cw1[si][kx] * zt[o0[si][ky] + 1];
                        hb[si][ky][kx] = cw0[si][kx] * zt[o1[si][ky]] + cw1[si][kx] * zt[o1[si][ky] + 1];
                    }
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc += rt[si][ky].z * ha[si][ky][kx] + rt[si][ky].w * hb[si][ky][kx];
            }
        }
        yp[(size_t)yy * W] = accumulate ? yp[(size_t)yy * W] + acc : acc;
#pragma unroll
        for (int si = 0; si < 2; ++si) {        // rotate: (si, ky) of the next row is the entry one further down
            const int n0 = __builtin_amdgcn_readfirstlane(__float_as_int(nx[si].x));
            const int n1 = __builtin_amdgcn_readfirstlane(__float_as_int(nx[si].y));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int a0 = ky < 2 ? o0[si][ky + 1] : n0, a1 = ky < 2 ? o1[si][ky + 1] : n1;
                ch[si][ky] = a0 != o0[si][ky] || a1 != o1[si][ky];
            }
#pragma unroll
            for (int ky = 0; ky < 2; ++ky) {
                rt[si][ky] = rt[si][ky + 1];
                o0[si][ky] = o0[si][ky + 1];
                o1[si][ky] = o1[si][ky + 1];
            }
            rt[si][2] = nx[si];
            o0[si][2] = n0;
            o1[si][2] = n1;
        }
    }
}

constexpr int TAP_TRL = 4;         // low-resolution rows per workgroup (backward), at most

// Backward, one source per launch.  One workgroup = one (n, co) plane x trl low-resolution rows.  Vertical pass first, one
// thread per HIGH-resolution column (coalesced reads of dy straight from memory, no staging): the thread folds the rows of
// dy whose shifted coordinate touches the tile into 3 (ky) x trl sums with row weights from an LDS table; the sums go to
// LDS and the horizontal pass gathers each (tap, row, column) of dz from its footprint with column weights from a second
// table.  (The other order -- lanes along the low-resolution columns read dy rows with a stride of the scale factor --
// is 4- to 8-way bank conflicted on ten times the volume.)
// max|dz| of a workgroup -> one integer atomic max (values >= 0).  The backward GEMMs take their operand scale from it: the
// a-priori bound 4 s_y s_x max|dy| is 16-256 x the real maximum (the signs of dy under a norm cancel), i.e. 4-8 bits of the
// f16 split given away -- visible as 1.2e-2 in the conv_last weight gradient of the UPerNet fixture once the 2x map took this route.
__device__ __forceinline__ void tap_block_amax(float m, float *dst)
{
    __shared__ float wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0)
        wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax((unsigned int *)dst, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

__global__ __launch_bounds__(256) void k_tapup_bwd(const float *__restrict__ dy, int Co, int H, int W, int h, int w,
                                                  Axis ay, Axis ax, int trl, int tiles_r, int gr, int kxn,
                                                  float *__restrict__ dz, long long sn, long long sc, float *__restrict__ amax)
{
    float am = 0.f;
    extern __shared__ __attribute__((aligned(16))) float tap_lds[];
    float *wt = tap_lds;                                  // [gr][3][TAP_TRL] row weights
    float *wxt = wt + (size_t)gr * 3 * TAP_TRL;           // [w][kxn] column weights of shifted coordinate xlo[c] + k
    int *xlo = (int *)(wxt + (size_t)w * kxn);            // [w]
    float *Vt = (float *)(xlo + w);                       // [3 * TAP_TRL][W]
    const int tile = blockIdx.x % tiles_r, plane = blockIdx.x / tiles_r;
    const int n = plane / Co, co = plane - n * Co;
    const int r_a = tile * trl, r_b = min(r_a + trl, h) - 1;
    int lo, hi, d0, d1;
    out_range(ay, r_a, h, H, lo, d1);
    out_range(ay, r_b, h, H, d0, hi);
    const int Ya = max(lo - 1, 0), Yb = min(hi + 1, H - 1);       // rows of dy whose shifted coordinate may touch the tile
    const int nrows = min(Yb - Ya + 1, gr);
    const int t = threadIdx.x;
    for (int e = t; e < nrows * 3 * TAP_TRL; e += 256) {
        const int rr = e % TAP_TRL, ky = (e / TAP_TRL) % 3, j = e / (3 * TAP_TRL);
        const int Ys = Ya + j + ky - 1;
        wt[e] = (rr <= r_b - r_a && Ys >= 0 && Ys < H) ? axis_weight(ay, Ys, h, r_a + rr) : 0.f;
    }
    for (int c = t; c < w; c += 256) {
        int xs_lo, xs_hi;
        out_range(ax, c, w, W, xs_lo, xs_hi);
        xlo[c] = xs_lo;
        for (int k = 0; k < kxn; ++k)
            wxt[c * kxn + k] = (xs_lo + k <= xs_hi) ? axis_weight(ax, xs_lo + k, w, c) : 0.f;
    }
    __syncthreads();
    const float *g = dy + ((size_t)plane * H + Ya) * W;
    for (int X = t; X < W; X += 256) {
        float acc[3][TAP_TRL];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int rr = 0; rr < TAP_TRL; ++rr)
                acc[ky][rr] = 0.f;
        for (int j = 0; j < nrows; ++j) {
            const float v = g[(size_t)j * W + X];
            const float *wj = wt + j * 3 * TAP_TRL;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int rr = 0; rr < TAP_TRL; ++rr)
                    acc[ky][rr] += wj[ky * TAP_TRL + rr] * v;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int rr = 0; rr < TAP_TRL; ++rr)
                Vt[(ky * TAP_TRL + rr) * W + X] = acc[ky][rr];
    }
    __syncthreads();
    const int nr = r_b - r_a + 1;
    float *out = dz + (size_t)n * sn + (size_t)co * sc;
    const size_t tapstride = (size_t)Co * sc;
    for (int e = t; e < 9 * nr * w; e += 256) {
        const int c = e % w, rr = (e / w) % nr, tap = e / (w * nr);
        const int ky = tap / 3, kx = tap - 3 * ky;
        const float *vrow = Vt + (size_t)(ky * TAP_TRL + rr) * W;
        const float *wc = wxt + c * kxn;
        const int x0 = xlo[c] - (kx - 1);                         // dy column of shifted coordinate xlo[c]
        float a = 0.f;
        for (int k = 0; k < kxn; ++k) {
            const int Xd = x0 + k;
            if (Xd >= 0 && Xd < W)
                a += wc[k] * vrow[Xd];
        }
        out[tap * tapstride + (size_t)(r_a + rr) * w + c] = a;
        am = fmaxf(am, fabsf(a));
    }
    if (amax)
        tap_block_amax(am, amax);
}

// Backward, second form (the default wherever its column window fits NW4 <= 8 float4s).  Same tiles and summation structure;
// what changed is the arithmetic per element (the first form spent ~1 000 instructions per thread in its horizontal pass:
// 9 outputs x kxn terms, each with two compares, two scalar LDS reads -- 4-way bank conflicted at a column stride of the scale
// factor -- and its address arithmetic; 2.5 ms per launch on 12 x 720 x 128 x 256 against ~0.4 ms of HBM time):
//   vertical   per low-resolution row rr only the FR = footprint + 2 rows of dy that can reach it are read (not every row of
//              the tile for every rr): 3 sums (ky) per row, the three row weights of (rr, j) in ONE uniform 16-byte LDS read;
//   horizontal one item = (column c, ky, rr) produces the three kx taps from ONE window of 4 NW4 values of the vertical sums,
//              read as aligned float4s (lanes along c: neighbouring lanes read neighbouring float4s), against a per-column
//              weight row wf[c][m] = weight of shifted coordinate (window start + m - 1) that is built once per workgroup:
//              out[kx] = sum_m wf[m + kx] V[m] -- no compares (zero-padded V rows, zero weights outside the image).
template <int NW4>
__global__ __launch_bounds__(256) void k_tapup_bwd_w(const float *__restrict__ dy, int Co, int H, int W, int h, int w, Axis ay,
                                                    Axis ax, int trl, int tiles_r, int FR, int WP, float *__restrict__ dz,
                                                    long long sn, long long sc, int planes, int ppw, float *__restrict__ amax)
{
    float am = 0.f;
    // Round 4: a workgroup takes `ppw` consecutive planes of its row tile (the weight tables -- axis_weight has a division per
    // entry -- are built once for all of them), and the vertical pass reads dy as float4s with every row of a footprint in
    // flight: one item = (low-resolution row, four columns), up to 8 x 16 bytes per thread outstanding.  Before, a thread had
    // four 4-byte loads in flight and half the threads idled on a 128-column map: 3.5 ms per launch on 16 x 512 x 128 x 128 where
    // the bytes take 0.1 (the kernel was bound by the latency of ~20 dependent load groups per workgroup).
    constexpr int PADL = 4, TS = (NW4 + 1) | 1;            // weight row of a column: TS float4s (odd: conflict-free b128 reads)
    extern __shared__ __attribute__((aligned(16))) float tap_lds[];
    f32x4 *wt = (f32x4 *)tap_lds;                           // [TAP_TRL][FR] {w(ky = 0), w(1), w(2), 0}
    f32x4 *wf = wt + TAP_TRL * FR;                          // [w][TS]
    float *Vt = (float *)(wf + (size_t)w * TS);             // [3 * TAP_TRL][WP], column X at PADL + X
    int *a4s = (int *)(Vt + (size_t)3 * TAP_TRL * WP);      // [w] window start (padded index, multiple of 4)
    int *ylo = a4s + w;                                     // [TAP_TRL] first dy row of a low-resolution row's footprint
    const int tile = blockIdx.x % tiles_r, plane0 = (blockIdx.x / tiles_r) * ppw;
    const int r_a = tile * trl, nr = min(r_a + trl, h) - r_a;
    const int t = threadIdx.x;
    for (int e = t; e < TAP_TRL * FR; e += 256) {
        const int rr = e / FR, j = e - rr * FR;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int lo = 0, hi = 0;
        if (rr < nr) {
            out_range(ay, r_a + rr, h, H, lo, hi);
            const int Y = lo - 1 + j;
            if (Y >= 0 && Y < H && Y <= hi + 1) {
                v.x = (Y - 1 >= 0) ? axis_weight(ay, Y - 1, h, r_a + rr) : 0.f;
                v.y = axis_weight(ay, Y, h, r_a + rr);
                v.z = (Y + 1 < H) ? axis_weight(ay, Y + 1, h, r_a + rr) : 0.f;
            }
        }
        wt[e] = v;
        if (j == 0)
            ylo[rr] = lo - 1;
    }
    for (int c = t; c < w; c += 256) {
        int xs_lo, xs_hi;
        out_range(ax, c, w, W, xs_lo, xs_hi);
        a4s[c] = (xs_lo - 1 + PADL) & ~3;
    }
    for (int e = t; e < 3 * TAP_TRL * (WP - W); e += 256) {                 // zero pads of the V rows
        const int row = e / (WP - W), k = e - row * (WP - W);
        Vt[(size_t)row * WP + (k < PADL ? k : W + k)] = 0.f;
    }
    __syncthreads();
    for (int e = t; e < w * TS * 4; e += 256) {
        const int c = e / (TS * 4), m = e - c * (TS * 4);
        const int Xs = a4s[c] - PADL + m - 1;
        int xs_lo, xs_hi;
        out_range(ax, c, w, W, xs_lo, xs_hi);
        ((float *)wf)[e] = (m < 4 * NW4 + 2 && Xs >= xs_lo && Xs <= xs_hi) ? axis_weight(ax, Xs, w, c) : 0.f;
    }
    for (int pp = 0; pp < ppw && plane0 + pp < planes; ++pp) {
    const int plane = plane0 + pp;
    const int n = plane / Co, co = plane - n * Co;
    __syncthreads();                        // the previous plane's horizontal pass has read Vt (first trip: the tables are built)
    const float *g = dy + (size_t)plane * H * W;
    if ((W & 3) == 0) {
        const int W4 = W >> 2;
        for (int item = t; item < TAP_TRL * W4; item += 256) {
            const int rr = item / W4, x4 = item - rr * W4;
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
            if (rr < nr) {
                const int y0 = ylo[rr];
                const f32x4 *wr = wt + rr * FR;
                const float *gc = g + 4 * x4;
                for (int j = 0; j < FR; j += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)             // (rows past the footprint: clamped address, never used)
                        v[u] = *(const f32x4 *)(gc + (size_t)min(max(y0 + min(j + u, FR - 1), 0), H - 1) * W);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (j + u < FR) {
                            const f32x4 q = wr[j + u];
                            a0 += q.x * v[u];
                            a1 += q.y * v[u];
                            a2 += q.z * v[u];
                        }
                    }
                }
            }
            *(f32x4 *)(Vt + (size_t)(0 * TAP_TRL + rr) * WP + PADL + 4 * x4) = a0;
            *(f32x4 *)(Vt + (size_t)(1 * TAP_TRL + rr) * WP + PADL + 4 * x4) = a1;
            *(f32x4 *)(Vt + (size_t)(2 * TAP_TRL + rr) * WP + PADL + 4 * x4) = a2;
        }
    } else
    for (int X = t; X < W; X += 256) {
#pragma unroll
        for (int rr = 0; rr < TAP_TRL; ++rr) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            if (rr < nr) {
                const int y0 = ylo[rr];
                const f32x4 *wr = wt + rr * FR;
                int j = 0;
                for (; j + 4 <= FR; j += 4) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        v[u] = g[(size_t)min(max(y0 + j + u, 0), H - 1) * W + X];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f32x4 q = wr[j + u];
                        a0 += q.x * v[u];
                        a1 += q.y * v[u];
                        a2 += q.z * v[u];
                    }
                }
                for (; j < FR; ++j) {
                    const float v = g[(size_t)min(max(y0 + j, 0), H - 1) * W + X];
                    const f32x4 q = wr[j];
                    a0 += q.x * v;
                    a1 += q.y * v;
                    a2 += q.z * v;
                }
            }
            Vt[(size_t)(0 * TAP_TRL + rr) * WP + PADL + X] = a0;
            Vt[(size_t)(1 * TAP_TRL + rr) * WP + PADL + X] = a1;
            Vt[(size_t)(2 * TAP_TRL + rr) * WP + PADL + X] = a2;
        }
    }
    __syncthreads();
    float *out = dz + (size_t)n * sn + (size_t)co * sc;
    const size_t tapstride = (size_t)Co * sc;
    for (int e = t; e < 3 * nr * w; e += 256) {
        const int c = e % w, q = e / w, rr = q % nr, ky = q / nr;
        const f32x4 *vrow = (const f32x4 *)(Vt + (size_t)(ky * TAP_TRL + rr) * WP + a4s[c]);
        const f32x4 *wc = wf + (size_t)c * TS;
        float wv[4 * NW4 + 4];
#pragma unroll
        for (int m = 0; m <= NW4; ++m) {
            const f32x4 q4 = wc[m];
            wv[4 * m] = q4.x;
            wv[4 * m + 1] = q4.y;
            wv[4 * m + 2] = q4.z;
            wv[4 * m + 3] = q4.w;
        }
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int m = 0; m < NW4; ++m) {
            const f32x4 v = vrow[m];
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                o0 += wv[4 * m + u] * vv[u];
                o1 += wv[4 * m + u + 1] * vv[u];
                o2 += wv[4 * m + u + 2] * vv[u];
            }
        }
        float *op = out + (size_t)(3 * ky) * tapstride + (size_t)(r_a + rr) * w + c;
        op[0] = o0;
        op[tapstride] = o1;
        op[2 * tapstride] = o2;
        am = fmaxf(am, fmaxf(fabsf(o0), fmaxf(fabsf(o1), fabsf(o2))));
    }
    }   // planes of this workgroup
    if (amax)
        tap_block_amax(am, amax);
}

// conservative bound of the low-resolution window a run of `n_out` consecutive output coordinates (+ one on either side)
// interpolates from
int tap_window(int in_size, int out_size, int n_out)
{
    const double ratio = out_size > 1 ? (double)in_size / (double)out_size : 1.0;
    int wdw = (int)((n_out + 2) * ratio + 4.0);
    return wdw > in_size ? in_size : (wdw < 2 ? 2 : wdw);
}

// output coordinates whose interpolation can touch one input index (bound of out_range's span), + margin
int tap_footprint(int in_size, int out_size)
{
    const double inv = in_size > 0 ? (double)out_size / (double)in_size : 1.0;
    int fp = (int)(2.0 * (inv > 1.0 ? inv : 1.0) + 6.0);
    return fp > out_size ? out_size : fp;
}

}  // namespace

// LDS a workgroup of the current device may ask for with the dynamic-size opt-in (gfx950: 160 KiB per CU)
static size_t tapup_device_lds_limit()
{
    int devid = 0, v = 0;
    if (hipGetDevice(&devid) != hipSuccess)
        return 64 * 1024;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, devid) == hipSuccess && v > 0)
        return (size_t)v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, devid) == hipSuccess && v > 0)
        return (size_t)v;
    return 64 * 1024;
}

extern "C" int dcl_tapup_fwd(const float *z0, int h0, int w0, const float *z1, int h1, int w1, int N, int Co, int H, int W,
                             int align_corners, int channel_major, float *y, int accumulate, void *stream)
{
    DCL_CHECK_ARG(z0 && y && N > 0 && Co > 0 && H > 0 && W > 0 && h0 > 0 && w0 > 0, "bad arguments");
    DCL_CHECK_ARG(!z1 || (h1 > 0 && w1 > 0), "bad second source");
    TapSrc s[2] = {};
    const float *zs[2] = {z0, z1};
    const int hs[2] = {h0, h1}, ws[2] = {w0, w1};
    const int nsrc = z1 ? 2 : 1;
    const int tabfloats = 2 * (TAP_RC + 2) * 4;
    int zfloats = 0;
    for (int i = 0; i < nsrc; ++i) {
        s[i].z = zs[i];
        s[i].h = hs[i];
        s[i].w = ws[i];
        s[i].sn = channel_major ? (long long)hs[i] * ws[i] : (long long)9 * Co * hs[i] * ws[i];
        s[i].sc = channel_major ? (long long)N * hs[i] * ws[i] : (long long)hs[i] * ws[i];
        s[i].ay = make_axis(hs[i], H, align_corners);
        s[i].ax = make_axis(ws[i], W, align_corners);
        s[i].wr = tap_window(hs[i], H, TAP_RC);
        s[i].wc = tap_window(ws[i], W, TAP_TW) + 1;
        s[i].zoff = zfloats;
        zfloats += 9 * s[i].wr * s[i].wc;
    }
    const size_t lds = (size_t)(tabfloats + zfloats) * sizeof(float);
    DCL_CHECK_ARG(lds <= TAP_FWD_LDS_MAX, "source maps too large for the tap-up tile (LDS): convolve the up-sampled map instead");
    if (lds > 64 * 1024) {              // (a source only 2x coarser than the output: 95 KB for the 32 x 256 tile, one workgroup per CU)
        // the opt-in is per DEVICE (a second GPU in the process needs its own), and a device without that much LDS must say no
        // here, not in the launch
        static bool attr_done[64] = {};
        int devid = 0;
        if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64)
            devid = 0;
        if (!attr_done[devid]) {
            DCL_CHECK_ARG(lds <= tapup_device_lds_limit(), "tap-up tile needs more LDS than this device offers");
            const hipError_t e = hipFuncSetAttribute((const void *)k_tapup_fwd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)TAP_FWD_LDS_MAX);
            if (e != hipSuccess) {
                dcl_set_error("dcl_tapup_fwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d): %s", (int)TAP_FWD_LDS_MAX,
                              hipGetErrorString(e));
                return (int)e;
            }
            attr_done[devid] = true;
        }
    }
    const int tiles_y = (H + TAP_RC - 1) / TAP_RC, tiles_x = (W + TAP_TW - 1) / TAP_TW;
    const long long blocks = (long long)N * Co * tiles_y * tiles_x;
    DCL_CHECK_ARG(blocks < (1LL << 31), "too many tiles");
    hipLaunchKernelGGL(k_tapup_fwd, dim3((unsigned)blocks), dim3(TAP_TW), lds, (hipStream_t)stream, s[0], s[1], nsrc, Co, H, W,
                       tiles_y, tiles_x, tabfloats, y, accumulate);
    DCL_LAUNCH_CHECK();
    return 0;
}

static int g_tapup_bwd_form = 2;        // 2 = k_tapup_bwd_w (windowed horizontal pass), 1 = k_tapup_bwd
static int g_tapup_bwd_ppw = 4;         // planes per workgroup of k_tapup_bwd_w (dcl_tapup_set_bwd_form(16 + p): p = 1, 2, 4, 8)

static int tapup_bwd_impl(const float *dy, int N, int Co, int H, int W, int h, int w, int align_corners, int channel_major,
                          float *dz, void *stream, bool query, float *amax = nullptr);

extern "C" int dcl_tapup_bwd(const float *dy, int N, int Co, int H, int W, int h, int w, int align_corners,
                             int channel_major, float *dz, void *stream)
{
    DCL_CHECK_ARG(dy && dz && N > 0 && Co > 0 && H > 0 && W > 0 && h > 0 && w > 0, "bad arguments");
    return tapup_bwd_impl(dy, N, Co, H, W, h, w, align_corners, channel_major, dz, stream, false);
}

extern "C" int dcl_tapup_bwd_amax(const float *dy, int N, int Co, int H, int W, int h, int w, int align_corners,
                                  int channel_major, float *dz, float *dz_amax, void *stream)
{
    DCL_CHECK_ARG(dy && dz && dz_amax && N > 0 && Co > 0 && H > 0 && W > 0 && h > 0 && w > 0, "bad arguments");
    return tapup_bwd_impl(dy, N, Co, H, W, h, w, align_corners, channel_major, dz, stream, false, dz_amax);
}

static size_t tapup_fwd_lds(const int *hs, const int *ws, int nsrc, int H, int W)
{
    size_t zfloats = 0;
    for (int i = 0; i < nsrc; ++i)
        zfloats += (size_t)9 * tap_window(hs[i], H, TAP_RC) * (tap_window(ws[i], W, TAP_TW) + 1);
    return ((size_t)2 * (TAP_RC + 2) * 4 + zfloats) * sizeof(float);
}

// 1 when dcl_tapup_fwd(z0 [h0, w0], z1 [h1, w1] (h1 = 0: one source)) and dcl_tapup_bwd of each source fit their tiles for an
// [H, W] output, else 0 -- asked BEFORE the step commits to the split form (models/ops.py conv3x3_over_upsampled falls back to
// convolving the materialised concatenation), so that an oversized map cannot fail mid-step, possibly only in the backward.
extern "C" int dcl_tapup_supported(int h0, int w0, int h1, int w1, int H, int W, int align_corners)
{
    if (h0 <= 0 || w0 <= 0 || H <= 0 || W <= 0 || h1 < 0 || w1 < 0)
        return 0;
    const int hs[2] = {h0, h1}, ws[2] = {w0, w1};
    const int nsrc = (h1 > 0 && w1 > 0) ? 2 : 1;
    const size_t need = tapup_fwd_lds(hs, ws, nsrc, H, W);
    if (need > TAP_FWD_LDS_MAX || (need > 64 * 1024 && need > tapup_device_lds_limit()))
        return 0;
    for (int i = 0; i < nsrc; ++i)
        if (tapup_bwd_impl(nullptr, 1, 1, H, W, hs[i], ws[i], align_corners, 1, nullptr, nullptr, true) != 0)
            return 0;
    return 1;
}

static int tapup_bwd_impl(const float *dy, int N, int Co, int H, int W, int h, int w, int align_corners, int channel_major,
                          float *dz, void *stream, bool query, float *amax)
{
    const Axis ay = make_axis(h, H, align_corners), ax = make_axis(w, W, align_corners);
    const int kxn = tap_footprint(w, W);
    int trl = TAP_TRL;
    auto rows_for = [&](int tr) {                   // rows of dy a tile of tr low-resolution rows can touch (+ tap shift)
        int r = tap_footprint(h, H) + (int)((tr - 1) * ((double)H / (double)h > 1.0 ? (double)H / (double)h : 1.0)) + 4;
        return r > H ? H : r;
    };
    auto lds_for = [&](int tr) {
        return ((size_t)rows_for(tr) * 3 * TAP_TRL + (size_t)w * kxn + (size_t)w + (size_t)3 * TAP_TRL * W) * sizeof(float);
    };
    while (trl > 1 && lds_for(trl) > 48 * 1024)
        trl >>= 1;
    const bool form1_fits = lds_for(trl) <= 64 * 1024;      // (required below only if the windowed form is not taken)
    const int gr = rows_for(trl);
    const int tiles_r = (h + trl - 1) / trl;
    const long long blocks = (long long)N * Co * tiles_r;
    DCL_CHECK_ARG(blocks < (1LL << 31), "too many tiles");
    const long long sn = channel_major ? (long long)h * w : (long long)9 * Co * h * w;
    const long long sc = channel_major ? (long long)N * h * w : (long long)h * w;
    // second form: 4 low-resolution rows per tile, a column window of NW4 float4s
    // exact spans of out_range() over the rows / columns (+ 1: host and device may round a quotient differently);
    // tap_footprint()'s 2 s + 6 is too small for align_corners on few source pixels (3 -> 24: s = 11.5, not 8)
    int span_r = 1, span_c = 1;
    for (int r = 0; r < h; ++r) {
        int lo, hi;
        out_range(ay, r, h, H, lo, hi);
        span_r = hi - lo + 1 > span_r ? hi - lo + 1 : span_r;
    }
    for (int c = 0; c < w; ++c) {
        int lo, hi;
        out_range(ax, c, w, W, lo, hi);
        span_c = hi - lo + 1 > span_c ? hi - lo + 1 : span_c;
    }
    const int nw4 = (span_c + 1 + 5 + 3) / 4, FR = span_r + 1 + 2, PADL = 4;
    const int WP = (W + PADL + 4 * nw4 + 4 + 3) & ~3;
    const int ts = (nw4 + 1) | 1;
    const size_t lds2 = ((size_t)TAP_TRL * FR * 4 + (size_t)w * ts * 4 + (size_t)3 * TAP_TRL * WP + (size_t)w + TAP_TRL) * sizeof(float);
    if (g_tapup_bwd_form == 2 && nw4 >= 2 && nw4 <= 8 && lds2 <= 64 * 1024) {
        if (query)
            return 0;
        const int tiles2 = (h + TAP_TRL - 1) / TAP_TRL;
        // planes per workgroup: 4 while that still leaves ~8 workgroups per CU
        const long long planes = (long long)N * Co;
        int ppw = g_tapup_bwd_ppw;
        while (ppw > 1 && (planes + ppw - 1) / ppw * tiles2 < 2048)
            ppw >>= 1;
        const long long blocks2 = (planes + ppw - 1) / ppw * tiles2;
        DCL_CHECK_ARG(blocks2 < (1LL << 31) && planes < (1LL << 31), "too many tiles");
#define DCL_TAPUP_CASE(K)                                                                                                   \
    case K:                                                                                                                 \
        hipLaunchKernelGGL(k_tapup_bwd_w<K>, dim3((unsigned)blocks2), dim3(256), lds2, (hipStream_t)stream, dy, Co, H, W, h, w, \
                           ay, ax, TAP_TRL, tiles2, FR, WP, dz, sn, sc, (int)planes, ppw, amax);                             \
        break;
        switch (nw4) {
            DCL_TAPUP_CASE(2)
            DCL_TAPUP_CASE(3)
            DCL_TAPUP_CASE(4)
            DCL_TAPUP_CASE(5)
            DCL_TAPUP_CASE(6)
            DCL_TAPUP_CASE(7)
            DCL_TAPUP_CASE(8)
        }
#undef DCL_TAPUP_CASE
        DCL_LAUNCH_CHECK();
        return 0;
    }
    if (query)
        return form1_fits ? 0 : DCL_EUNSUPPORTED;
    DCL_CHECK_ARG(form1_fits, "maps too wide for the tap-up backward tile (LDS)");
    hipLaunchKernelGGL(k_tapup_bwd, dim3((unsigned)blocks), dim3(256), lds_for(trl), (hipStream_t)stream, dy, Co, H, W, h, w, ay,
                       ax, trl, tiles_r, gr, kxn, dz, sn, sc, amax);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_tapup_set_bwd_form(int form)
{
    if (form > 16 && form <= 24) {          // tuning hook: planes per workgroup of the second form
        g_tapup_bwd_ppw = form - 16;
        return 0;
    }
    if (form != 1 && form != 2)
        return DCL_EINVAL;
    g_tapup_bwd_form = form;
    return 0;
}

// ---- out = a + b (+ c) (+ d): the gradient of a tensor with several consumers in ONE pass (models/ops.py _FanOut) instead of
// autograd's chain of two-input adds (k + 1 tensor passes instead of 3 (k - 1))
namespace {
__global__ __launch_bounds__(256) void k_add_n(const float *__restrict__ a, const float *__restrict__ b,
                                              const float *__restrict__ c, const float *__restrict__ d, size_t n4,
                                              size_t n, float *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f32x4 v = ((const f32x4 *)a)[i];
        const f32x4 w = ((const f32x4 *)b)[i];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        if (c) {
            const f32x4 u = ((const f32x4 *)c)[i];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (d) {
            const f32x4 u = ((const f32x4 *)d)[i];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        ((f32x4 *)out)[i] = v;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256)
            out[i] = a[i] + b[i] + (c ? c[i] : 0.f) + (d ? d[i] : 0.f);
}
}  // namespace

extern "C" int dcl_add_n(const float *a, const float *b, const float *c, const float *d, int64_t n, float *out,
                         void *stream)
{
    DCL_CHECK_ARG(a && b && out && n > 0 && (c || !d), "bad arguments (a, b required; d only with c)");
    DCL_CHECK_ARG(((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d) | ((uintptr_t)out)) & 15) == 0,
                  "tensors must be 16-byte aligned");
    const size_t n4 = (size_t)n / 4;
    size_t blocks = (n4 + 256 * 4 - 1) / (256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
    hipLaunchKernelGGL(k_add_n, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, d, n4, (size_t)n, out);
    DCL_LAUNCH_CHECK();
    return 0;
}
