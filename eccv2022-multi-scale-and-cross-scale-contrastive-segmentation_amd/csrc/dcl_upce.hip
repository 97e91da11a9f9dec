// dcl_upce.hip -- fused bilinear up-sampling + class-weighted cross-entropy.
//
// Replaces, for the segmentation logits, F.interpolate(logits_1/4, size, 'bilinear', align_corners) (reference
// models/HRNet.py:638) followed by nn.CrossEntropyLoss(weight, ignore_index) (losses/LossWrapper.py:26-30, :82) and
// their autograd backward.  At BASELINE config 2 the up-sampled logits are 12 x 19 x 512 x 1024 f32 = 478 MB; the
// unfused chain writes / reads that tensor (and its gradient) ~9 times per step.  Here the full-resolution logits exist
// only in registers:
//   forward : one workgroup per OUTPUT row; the two source rows of every class are staged in LDS; each thread owns 4
//             consecutive pixels, interpolates the class logits on the fly (ATen's index arithmetic, dcl_resize.hip),
//             online log-sum-exp, picks the target logit, emits  lse [N,H,W] (saved for the backward), the arg-max
//             class (uint8 map: the confusion matrix needs nothing else) and per-row partial sums {sum w_t (lse - v_t),
//             sum w_t}: loss = sum / sum, PyTorch's weighted-mean reduction (ignored pixels in neither);
//   backward: one workgroup per LOW-RES row (gather form, deterministic, no atomics): vertical pass over the output
//             rows that touch it -- g_c = w_t (softmax_c - [c = t]) recomputed from the staged low-res rows and the
//             saved lse, accumulated with the y-weights into an LDS tile [classes][W] -- then a horizontal pass in which
//             every low-res column gathers its x-footprint.  The upstream gradient and 1 / sum w are a device scalar.
// Classes are processed in chunks that fit LDS (19 Cityscapes classes: one chunk; 150 ADE20K classes: several).
// HBM-bound: reads z (30 MB), labels and lse; writes lse / prediction / dz.
#include <type_traits>

#include "dcl_common.h"

namespace {

struct Axis {
    float scale;
    int align;
};

__device__ __forceinline__ void src_index(const Axis a, int dst, int in_size, int &i0, int &i1, float &l0, float &l1)
{
    float s = a.align ? a.scale * (float)dst : fmaxf(a.scale * ((float)dst + 0.5f) - 0.5f, 0.f);
    i0 = (int)s;
    if (i0 > in_size - 1)
        i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

__device__ __forceinline__ float axis_weight(const Axis a, int o, int in_size, int i)
{
    int i0, i1;
    float l0, l1;
    src_index(a, o, in_size, i0, i1, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

__device__ __forceinline__ void out_range(const Axis a, int i, int in_size, int out_size, int &lo, int &hi)
{
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

// exp(t) = 2^(t log2 e) on v_exp_f32 (1 ulp) with the rounding error of the product folded back in: 7 instructions
// against ~13 for expf, ~1e-7 relative.  t is clamped at -104 (the result is below the smallest denormal either way; a
// comparison, so that NaN stays NaN), which also turns exp(-inf) into a plain 0 without the inf - inf of the remainder.
__device__ __forceinline__ float exp_fast(float t)
{
    t = t < -104.f ? -104.f : t;
    const float p = t * 1.44269502f;
    float e = fmaf(t, 1.44269502f, -p);
    e = fmaf(t, 1.92596299e-8f, e);
    const float r = __builtin_amdgcn_exp2f(p);
    return fmaf(r, e * 0.693147182f, r);
}

struct UpceArgs {
    const float *z;             // [N, C, h, w]
    const long long *target;    // [N, H, W]
    const float *weight;        // [C] or null
    float *lse;                 // [N, H, W]
    unsigned char *pred;        // [N, H, W] or null
    float *partial;             // fwd: [N*H, 2]
    const float *gscale;        // bwd: device scalar: grad_out / sum w
    float *dz;                  // bwd: [N, C, h, w]
    int N, C, h, w, H, W, ignore, cc;
    Axis ay, ax;
};

constexpr int FWD_LDS_FLOATS = 6144;        // 24 KiB: class chunk x 2 source rows x w (measured: 8-24 KiB 1.13 ms, 40-48 KiB
                                            // 1.32 ms for 150 classes at 128^2 -> 512^2: more workgroups per CU)
// PIX pixels per thread: 4 for rows of >= 1024 pixels, 2 below (a 512-pixel row would leave half the workgroup idle);
// the class-chunk stage is sized by the launch (cc * 2 w floats)
template <int PIX>
__global__ __launch_bounds__(256) void k_upce_fwd(UpceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float zs[];
    __shared__ float red[2][4];
    const int row = blockIdx.x;                     // n * H + oy
    const int oy = row % a.H, n = row / a.H;
    int y0, y1;
    float ly0, ly1;
    src_index(a.ay, oy, a.h, y0, y1, ly0, ly1);
    const size_t plane = (size_t)a.h * a.w;
    float num = 0.f, den = 0.f;
    for (int trip = 0; trip * 256 * PIX < a.W; ++trip) {            // uniform trip count: barriers inside
        const int px0 = trip * 256 * PIX + threadIdx.x * PIX;
        int x0[PIX], x1[PIX], tgt[PIX];
        float lx0[PIX], lx1[PIX], m[PIX], l[PIX], vt[PIX], best[PIX];
        int arg[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const int ox = min(px0 + k, a.W - 1);
            src_index(a.ax, ox, a.w, x0[k], x1[k], lx0[k], lx1[k]);
            const long long t = px0 + k < a.W ? a.target[(size_t)row * a.W + ox] : (long long)a.ignore;
            tgt[k] = (t < 0 || t > 0x7fffffff) ? -1 : (int)t;
            m[k] = -INFINITY;
            l[k] = 0.f;
            vt[k] = 0.f;
            best[k] = -INFINITY;
            arg[k] = 0;
        }
        for (int c0 = 0; c0 < a.C; c0 += a.cc) {
            const int nc = min(a.cc, a.C - c0);
            __syncthreads();
            // stage rows y0, y1 of classes c0 .. c0 + nc: one (class, row) per wave and trip, lanes along the row (no
            // index divisions: they cost as much as the interpolation of a pixel)
            for (int cr = threadIdx.x >> 6; cr < nc * 2; cr += 4) {
                const float *src = a.z + ((size_t)n * a.C + c0 + (cr >> 1)) * plane + (size_t)((cr & 1) ? y1 : y0) * a.w;
                for (int x = threadIdx.x & 63; x < a.w; x += 64)
                    zs[cr * a.w + x] = src[x];
            }
            __syncthreads();
            // four classes per trip: their 16 LDS gathers are in flight together, one rescale of the running sum per
            // block (5 exponentials for 4 classes instead of 8), arg-max and target pick in class order, branch-free
            for (int c = 0; c < nc; c += 4) {
                float v[4][PIX];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cu = min(c + u, nc - 1);
                    const float *r0 = zs + (cu * 2) * a.w, *r1 = r0 + a.w;
#pragma unroll
                    for (int k = 0; k < PIX; ++k) {
                        const float t = ly0 * (lx0[k] * r0[x0[k]] + lx1[k] * r0[x1[k]]) +
                                        ly1 * (lx0[k] * r1[x0[k]] + lx1[k] * r1[x1[k]]);
                        v[u][k] = c + u < nc ? t : -INFINITY;
                    }
                }
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    const float mn = fmaxf(m[k], fmaxf(fmaxf(v[0][k], v[1][k]), fmaxf(v[2][k], v[3][k])));
                    l[k] = l[k] * exp_fast(m[k] - mn) + ((exp_fast(v[0][k] - mn) + exp_fast(v[1][k] - mn)) +
                                                         (exp_fast(v[2][k] - mn) + exp_fast(v[3][k] - mn)));
                    m[k] = mn;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float vv = v[u][k];
                        vt[k] = ((c + u < nc) & (c0 + c + u == tgt[k])) ? vv : vt[k];
                        // torch.argmax rule: the first maximum; a NaN beats everything that is not NaN
                        const bool take = (vv > best[k]) | ((vv != vv) & (best[k] == best[k]));
                        best[k] = take ? vv : best[k];
                        arg[k] = take ? c0 + c + u : arg[k];
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const int ox = px0 + k;
            if (ox >= a.W)
                continue;
            const float lse = m[k] + logf(l[k]);
            a.lse[(size_t)row * a.W + ox] = lse;
            if (a.pred)
                a.pred[(size_t)row * a.W + ox] = (unsigned char)arg[k];
            if (tgt[k] != a.ignore && tgt[k] >= 0 && tgt[k] < a.C) {
                const float wt = a.weight ? a.weight[tgt[k]] : 1.f;
                num += wt * (lse - vt[k]);
                den += wt;
            }
        }
    }
    num = wave_sum(num);
    den = wave_sum(den);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = num;
        red[1][wv] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.partial[(size_t)row * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        a.partial[(size_t)row * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// fixed-order reduction of the per-row partials: out = {sum num / sum den, sum den}
__global__ __launch_bounds__(256) void k_upce_finish(const float *__restrict__ partial, int rows, float *__restrict__ out)
{
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < rows; i += 256) {
        a += partial[(size_t)i * 2];
        b += partial[(size_t)i * 2 + 1];
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(sh[0][0] / sh[1][0]);          // 0 / 0 = NaN when every pixel is ignored, like PyTorch
        out[1] = (float)sh[1][0];
    }
}

constexpr int BWD_MAX_ROWS = 80;           // output rows in the footprint of a low-res row: scale factors up to ~36
constexpr int BWD_LDS_FLOATS = 36864;       // 144 KiB: 3 staged low-res rows + the [classes][W] tile of a class chunk

__global__ __launch_bounds__(256) void k_upce_bwd(UpceArgs a)
{
    // dynamic: cc * (3 w + W) floats, sized by the launch so that two workgroups share a CU when the class count allows
    // (the kernel is latency-bound: exp, gathers; a static 144-KiB array meant one workgroup per CU)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int row = blockIdx.x;                     // n * h + iy
    const int iy = row % a.h, n = row / a.h;
    const size_t plane = (size_t)a.h * a.w;
    int oy_lo, oy_hi;
    out_range(a.ay, iy, a.h, a.H, oy_lo, oy_hi);
    const float gs = a.gscale[0];
    const int ya = max(iy - 1, 0);                  // staged rows ya .. ya + 2 (clamped to the image)
    // per output row of the footprint: {wy, weights of the three staged rows in its interpolation, times log2 e} --
    // wave-uniform values the compiler would otherwise recompute on the vector ALU for every class block
    __shared__ __attribute__((aligned(16))) float rowtab[BWD_MAX_ROWS][4];
    for (int j = threadIdx.x; j <= oy_hi - oy_lo; j += 256) {
        const int oy = oy_lo + j;
        int y0, y1;
        float ly0, ly1;
        src_index(a.ay, oy, a.h, y0, y1, ly0, ly1);
        const int r0 = y0 - ya, r1 = y1 - ya;        // both in 0 .. 2 for a row with a non-zero weight
        rowtab[j][0] = axis_weight(a.ay, oy, a.h, iy);
#pragma unroll
        for (int r = 0; r < 3; ++r)
            rowtab[j][1 + r] = ((r0 == r ? ly0 : 0.f) + (r1 == r ? ly1 : 0.f)) * 1.44269502f;
    }
    {                                               // one class chunk per workgroup (blockIdx.y)
        const int c0 = blockIdx.y * a.cc;
        const int nc = min(a.cc, a.C - c0);
        float *zr = lds;                            // [nc][3][w]
        float *tile = lds + nc * 3 * a.w;           // [nc][W]
        __syncthreads();
        for (int c = threadIdx.x >> 6; c < nc; c += 4)             // one class per wave and trip, lanes along the rows
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float *src = a.z + ((size_t)n * a.C + c0 + c) * plane + (size_t)min(ya + r, a.h - 1) * a.w;
                for (int x = threadIdx.x & 63; x < a.w; x += 64)
                    zr[(c * 3 + r) * a.w + x] = src[x];
            }
        __syncthreads();
        // vertical pass: tile[c][ox] = sum_oy wy(oy -> iy) * w_t (softmax_c - [c = t]).  Eight (tail: four) classes at a time: their
        // three staged rows are interpolated horizontally ONCE per column (6 LDS gathers per class and column instead of
        // 4 per class and output pixel); an output row then costs three FMAs with wave-uniform row weights per class,
        // one exponential, and the sums stay in registers until the block is done.
        for (int ox = threadIdx.x; ox < a.W; ox += 256) {
            int x0, x1;
            float lx0, lx1;
            src_index(a.ax, ox, a.w, x0, x1, lx0, lx1);
            auto block = [&](auto ubt, const int cb) {
                constexpr int UB = decltype(ubt)::value;
                float hl[UB][3], acc[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int cu = min(cb + u, nc - 1);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float *b = zr + (cu * 3 + r) * a.w;
                        hl[u][r] = lx0 * b[x0] + lx1 * b[x1];
                    }
                    acc[u] = 0.f;
                }
                for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                    const f32x4 rt = *(const f32x4 *)rowtab[oy - oy_lo];
                    if (rt.x == 0.f)                             // uniform
                        continue;
                    const size_t p = ((size_t)n * a.H + oy) * a.W + ox;
                    const long long t = a.target[p];
                    const bool valid = t != a.ignore && t >= 0 && t < a.C;
                    const int ti = valid ? (int)t : -1;
                    const float coef = valid ? rt.x * (a.weight ? a.weight[ti] : 1.f) * gs : 0.f;
                    const float lse2 = a.lse[p] * 1.44269502f;
                    // softmax_c = 2^((v - lse) log2 e), v <= lse: plain v_exp_f32 -- the rounding of the scaled argument
                    // moves a probability by at most |t| e^t 6e-8 <= 2.2e-8 absolute
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const float v2 = rt.y * hl[u][0] + rt.z * hl[u][1] + rt.w * hl[u][2];
                        const float g = __builtin_amdgcn_exp2f(v2 - lse2) - (c0 + cb + u == ti ? 1.f : 0.f);
                        acc[u] += coef * g;
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (cb + u < nc)
                        tile[(cb + u) * a.W + ox] = acc[u];
            };
            int cb = 0;
            for (; nc - cb > 4; cb += 8)
                block(std::integral_constant<int, 8>{}, cb);
            if (cb < nc)
                block(std::integral_constant<int, 4>{}, cb);
        }
        __syncthreads();
        // horizontal pass: dz[n, c, iy, ix] = sum_ox wx(ox -> ix) * tile[c][ox]
        // (lanes along the low-res row, classes over the waves: the footprint and its <= 12 weights are computed once per
        // column and serve every class of the chunk; wider footprints -- scale factors above 4 -- recompute them)
        for (int ix = threadIdx.x & 63; ix < a.w; ix += 64) {
            int ox_lo, ox_hi;
            out_range(a.ax, ix, a.w, a.W, ox_lo, ox_hi);
            const int cnt = ox_hi - ox_lo + 1;
            float *dst = a.dz + ((size_t)n * a.C + c0) * plane + (size_t)iy * a.w + ix;
            if (cnt <= 12) {
                float wx[12];
#pragma unroll
                for (int j = 0; j < 12; ++j)
                    wx[j] = j < cnt ? axis_weight(a.ax, ox_lo + j, a.w, ix) : 0.f;
                for (int c = threadIdx.x >> 6; c < nc; c += 4) {
                    const float *tr = tile + c * a.W + ox_lo;
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < 12; ++j)
                        acc += wx[j] * tr[j < cnt ? j : 0];
                    dst[(size_t)c * plane] = acc;
                }
            } else {
                for (int c = threadIdx.x >> 6; c < nc; c += 4) {
                    float acc = 0.f;
                    for (int ox = ox_lo; ox <= ox_hi; ++ox)
                        acc += axis_weight(a.ax, ox, a.w, ix) * tile[c * a.W + ox];
                    dst[(size_t)c * plane] = acc;
                }
            }
        }
    }
}

}  // namespace

static int g_upce_fwd_kb = 0;          // tuning override: KiB of LDS for the forward's class chunk (0 = 48)

extern "C" int dcl_upsample_ce_set_fwd_lds(int kib)
{
    g_upce_fwd_kb = kib > 0 && kib <= 48 ? kib : 0;
    return 0;
}

extern "C" int dcl_upsample_ce_fwd(const float *z, int N, int C, int h, int w, int H, int W, int align_corners,
                                   const int64_t *target, const float *weight, int ignore_index, float *lse,
                                   uint8_t *pred, float *partial, float *out2, void *stream)
{
    DCL_CHECK_ARG(z && target && lse && partial && out2, "null pointer");
    DCL_CHECK_ARG(N > 0 && C > 0 && C <= 255 && h > 0 && w > 0 && H >= h && W >= w, "bad shape (C <= 255, up-sampling only)");
    DCL_CHECK_ARG(2 * w <= FWD_LDS_FLOATS, "low-resolution row too wide for the LDS stage");
    UpceArgs a = {};
    a.z = z; a.target = (const long long *)target; a.weight = weight; a.lse = lse; a.pred = pred; a.partial = partial;
    a.N = N; a.C = C; a.h = h; a.w = w; a.H = H; a.W = W; a.ignore = ignore_index;
    a.ay = make_axis(h, H, align_corners);
    a.ax = make_axis(w, W, align_corners);
    a.cc = (g_upce_fwd_kb > 0 ? g_upce_fwd_kb * 256 : FWD_LDS_FLOATS) / (2 * w);
    if (a.cc > C)
        a.cc = C;
    if (a.cc < 1)
        a.cc = 1;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_bytes = (size_t)a.cc * 2 * w * sizeof(float);
    if (W >= 1024)
        hipLaunchKernelGGL(k_upce_fwd<4>, dim3((unsigned)(N * H)), dim3(256), lds_bytes, st, a);
    else
        hipLaunchKernelGGL(k_upce_fwd<2>, dim3((unsigned)(N * H)), dim3(256), lds_bytes, st, a);
    DCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_upce_finish, dim3(1), dim3(256), 0, st, partial, N * H, out2);
    DCL_LAUNCH_CHECK();
    return 0;
}

static int g_upce_bwd_cc = 0;          // tuning override of the backward's class chunk (0 = automatic)

extern "C" int dcl_upsample_ce_set_bwd_chunk(int cc)
{
    g_upce_bwd_cc = cc > 0 ? cc : 0;
    return 0;
}

extern "C" int dcl_upsample_ce_bwd(const float *z, int N, int C, int h, int w, int H, int W, int align_corners,
                                   const int64_t *target, const float *weight, int ignore_index, const float *lse,
                                   const float *gscale, float *dz, void *stream)
{
    DCL_CHECK_ARG(z && target && lse && gscale && dz, "null pointer");
    DCL_CHECK_ARG(N > 0 && C > 0 && C <= 255 && h > 0 && w > 0 && H >= h && W >= w, "bad shape");
    DCL_CHECK_ARG(3 * w + W <= BWD_LDS_FLOATS, "rows too wide for the LDS tile");
    DCL_CHECK_ARG(2 * ((H + h - 1) / h) + 8 <= BWD_MAX_ROWS, "vertical scale factor above 36");
    UpceArgs a = {};
    a.z = z; a.target = (const long long *)target; a.weight = weight; a.lse = const_cast<float *>(lse);
    a.gscale = gscale; a.dz = dz;
    a.N = N; a.C = C; a.h = h; a.w = w; a.H = H; a.W = W; a.ignore = ignore_index;
    a.ay = make_axis(h, H, align_corners);
    a.ax = make_axis(w, W, align_corners);
    // class chunk = classes per workgroup (grid.y walks the chunks): one register block of eight classes when that fits
    // 36 KiB of LDS (3-4 workgroups per CU), else one of four; measured (tools/probes/upce_chunk_sweep.sh): 150 classes
    // at 128^2 -> 512^2: 8 per chunk 817 us, 4: 1293, 12: 1251, 16: 1453; at 32^2 -> 512^2: 8: 534, 4: 898, 16: 604;
    // 19 classes at 128 x 256 -> 512 x 1024 (7 KiB per class): 4 per chunk 280 us, 7: 293, 8: 380
    const int per_class = 3 * w + W;
    int cc = 8 * per_class * 4 <= 36 * 1024 ? 8 : 4;
    if ((size_t)cc * per_class > BWD_LDS_FLOATS)
        cc = BWD_LDS_FLOATS / per_class;
    a.cc = cc > C ? C : cc;
    if (g_upce_bwd_cc > 0)
        a.cc = g_upce_bwd_cc > C ? C : g_upce_bwd_cc;
    DCL_CHECK_ARG(a.cc >= 1 && (size_t)a.cc * per_class <= BWD_LDS_FLOATS, "class chunk does not fit the LDS");
    const size_t lds_bytes = (size_t)a.cc * per_class * sizeof(float);
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)k_upce_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(k_upce_bwd, dim3((unsigned)(N * h), (unsigned)((C + a.cc - 1) / a.cc)), dim3(256), lds_bytes,
                       (hipStream_t)stream, a);
    DCL_LAUNCH_CHECK();
    return 0;
}
