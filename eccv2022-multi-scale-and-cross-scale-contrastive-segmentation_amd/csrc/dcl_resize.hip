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
__device__ __forceinline__ void out_range(const Axis a, int i, int in_size, int out_size, int &lo, int &hi)
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
