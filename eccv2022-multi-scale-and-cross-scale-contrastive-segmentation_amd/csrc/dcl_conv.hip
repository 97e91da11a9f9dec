// dcl_conv.hip -- im2col (3x3, stride 1, pad 1) fused with the (hi, lo) f16 split used by the f16x3 GEMM
// formulation of HRNet's segmentation-head convolution (models/ops.py: Conv3x3F16x3).
//
// out_hi/out_lo [N, C*9, H*W] halves (torch.nn.functional.unfold layout: row (c, ky, kx), column = pixel):
//   v = x[n, c, y + ky - 1, x + kx - 1] * scale   (0 outside the image)
//   hi = f16(v), lo = f16(v - hi)                  (22 mantissa bits together)
// `scale` is a device scalar (a power of two chosen from the tensor's absmax so that hi stays in f16 range
// and lo stays normal).  HBM-bound: every input element is read 9x (L2 hits), 2 x 2 B written per output
// element, 16-B stores.
#include "dcl_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// grid: (ceil(W/8 * H / 256), C*3, N); each thread loads one 10-pixel window x[c, y+ky-1, x0-1 .. x0+8] (two
// 16-B loads + two scalars) and produces 8 consecutive pixels of the three rows (c, ky, kx = 0..2)
__global__ __launch_bounds__(256) void k_im2col3x3_split(const float *__restrict__ x, int C, int H, int W,
                                                        const float *__restrict__ scale,
                                                        _Float16 *__restrict__ out_hi,
                                                        _Float16 *__restrict__ out_lo)
{
    const int cky = blockIdx.y;                 // c*3 + ky
    const int n = blockIdx.z;
    const int c = cky / 3, ky = cky - 3 * c;
    const int w8 = W >> 3;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= w8 * H)
        return;
    const int y = t / w8, x0 = (t - y * w8) << 3;
    const int sy = y + ky - 1;
    const float s = scale[0];
    float win[10];
    if (sy >= 0 && sy < H) {
        const float *src = x + (((size_t)n * C + c) * H + sy) * W + x0;
        const f32x4 a = *(const f32x4 *)src, b = *(const f32x4 *)(src + 4);
        win[0] = x0 > 0 ? src[-1] * s : 0.f;
        win[1] = a.x * s; win[2] = a.y * s; win[3] = a.z * s; win[4] = a.w * s;
        win[5] = b.x * s; win[6] = b.y * s; win[7] = b.z * s; win[8] = b.w * s;
        win[9] = x0 + 8 < W ? src[8] * s : 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 10; ++j)
            win[j] = 0.f;
    }
    _Float16 hv[10], lv[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        hv[j] = (_Float16)win[j];
        lv[j] = (_Float16)(win[j] - (float)hv[j]);
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = hv[j + kx];
            lo[j] = lv[j + kx];
        }
        const size_t o = (((size_t)n * C * 9 + (cky * 3 + kx)) * H + y) * W + x0;
        *(half8 *)(out_hi + o) = hi;
        *(half8 *)(out_lo + o) = lo;
    }
}

// elementwise split of a contiguous f32 tensor: hi = f16(x * s), lo = f16(x * s - hi)
__global__ __launch_bounds__(256) void k_split_f16(const float *__restrict__ x, size_t n,
                                                  const float *__restrict__ scale,
                                                  _Float16 *__restrict__ hi, _Float16 *__restrict__ lo)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n)
        return;
    const float s = scale[0];
    if (i + 8 <= n) {
        const f32x4 a = *(const f32x4 *)(x + i), b = *(const f32x4 *)(x + i + 4);
        const float v[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
        half8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j] = (_Float16)v[j];
            l[j] = (_Float16)(v[j] - (float)h[j]);
        }
        *(half8 *)(hi + i) = h;
        *(half8 *)(lo + i) = l;
    } else {
        for (size_t j = i; j < n; ++j) {
            const float v = x[j] * s;
            const _Float16 h = (_Float16)v;
            hi[j] = h;
            lo[j] = (_Float16)(v - (float)h);
        }
    }
}

}  // namespace

extern "C" int dcl_im2col3x3_split(const float *x, int N, int C, int H, int W, const float *scale,
                                   void *out_hi, void *out_lo, void *stream)
{
    DCL_CHECK_ARG(x && scale && out_hi && out_lo && N > 0 && C > 0 && H > 0 && W > 0, "bad arguments");
    DCL_CHECK_ARG((W & 7) == 0, "W must be a multiple of 8");
    DCL_CHECK_ARG(C * 3 <= 65535 && N <= 65535, "too many channels / images for the launch grid");
    DCL_CHECK_ARG((((uintptr_t)x) & 15) == 0, "input must be 16-byte aligned");
    dim3 grid(((W >> 3) * H + 255) / 256, C * 3, N);
    hipLaunchKernelGGL(k_im2col3x3_split, grid, dim3(256), 0, (hipStream_t)stream, x, C, H, W, scale,
                       (_Float16 *)out_hi, (_Float16 *)out_lo);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_split_f16(const float *x, int64_t n, const float *scale, void *hi, void *lo, void *stream)
{
    DCL_CHECK_ARG(x && scale && hi && lo && n > 0, "bad arguments");
    DCL_CHECK_ARG((((uintptr_t)x) & 15) == 0, "input must be 16-byte aligned");
    const size_t groups = ((size_t)n + 7) / 8;
    hipLaunchKernelGGL(k_split_f16, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, (size_t)n, scale, (_Float16 *)hi, (_Float16 *)lo);
    DCL_LAUNCH_CHECK();
    return 0;
}
