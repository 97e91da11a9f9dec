// dcl_sampling.hip -- K1 (label stride-sample + histograms) and K2 (rank-select).
// Integer / byte work, HBM- and latency-bound; results are bit-exact by construction.
#include "dcl_common.h"

// ATen's legacy 'nearest' source index (UpSampleKernel: nearest_idx), restated:
// identity when sizes match, >>1 when out == 2*in, else min(floorf(dst * (float)in/out), in-1).
__device__ inline int nearest_src(int dst, int in_size, int out_size, float scale)
{
    if (out_size == in_size)
        return dst;
    if (out_size == 2 * in_size)
        return dst >> 1;
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// One workgroup (256 threads) per DCL_SEG-pixel segment of one image.
// Replaces DenseContrastiveLossV2.py:205 and :100-103.
__global__ __launch_bounds__(256) void k_label_hist(const int64_t *__restrict__ label, int H, int W,
                                                   int h, int w, int K, int nseg,
                                                   uint8_t *__restrict__ lbl_s,
                                                   int32_t *__restrict__ seg_hist,
                                                   int32_t *__restrict__ counts)
{
    __shared__ int hist[256];
    const int seg = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    const int hw = h * w;
    const int p = seg * DCL_SEG + tid;
    if (p < hw) {
        const int i = p / w, j = p - i * w;
        const float sh = (float)H / (float)h, sw = (float)W / (float)w;
        const int si = nearest_src(i, H, h, sh), sj = nearest_src(j, W, w, sw);
        const int64_t c = label[((int64_t)b * H + si) * W + sj];
        const bool ok = (c >= 0 && c < K);
        lbl_s[(int64_t)b * hw + p] = ok ? (uint8_t)c : (uint8_t)255;
        if (ok)
            atomicAdd(&hist[(int)c], 1);
    }
    __syncthreads();
    if (tid < K) {
        const int v = hist[tid];
        seg_hist[((int64_t)b * nseg + seg) * K + tid] = v;
        if (v)
            atomicAdd(&counts[b * K + tid], v);
    }
}

extern "C" int dcl_label_hist(const int64_t *label, int n, int H, int W, int scale, int K,
                              uint8_t *lbl_s, int32_t *seg_hist, int32_t *counts, void *stream)
{
    DCL_CHECK_ARG(label && lbl_s && seg_hist && counts, "null pointer");
    DCL_CHECK_ARG(n > 0 && H > 0 && W > 0 && scale > 0, "bad sizes");
    DCL_CHECK_ARG(K > 0 && K <= DCL_MAX_CLASSES, "K must be in [1, 255]");
    const int h = H / scale, w = W / scale;
    DCL_CHECK_ARG(h > 0 && w > 0, "scale larger than the label map");
    const int nseg = (h * w + DCL_SEG - 1) / DCL_SEG;
    dim3 grid(nseg, n);
    hipLaunchKernelGGL(k_label_hist, grid, dim3(256), 0, (hipStream_t)stream, label, H, W, h, w, K,
                       nseg, lbl_s, seg_hist, counts);
    DCL_LAUNCH_CHECK();
    return 0;
}

// One workgroup per (image, class) pair.  Phase 1: exclusive prefix of the pair's class column of
// seg_hist over segments (LDS).  Phase 2: each thread resolves one requested rank: binary search
// for its segment, then a byte scan of that segment's DCL_SEG labels.
// Replaces DenseContrastiveLossV2.py:117-122 (nonzero + perm[:V] indexing).
#define RS_MAX_SEG 8192
__global__ __launch_bounds__(256) void k_rank_select(const uint8_t *__restrict__ lbl_s,
                                                    const int32_t *__restrict__ seg_hist, int hw,
                                                    int K, int nseg,
                                                    const int32_t *__restrict__ pair_b,
                                                    const int32_t *__restrict__ pair_k, int V,
                                                    const int32_t *__restrict__ sel,
                                                    int32_t *__restrict__ pix)
{
    extern __shared__ int seg_start[];   // [nseg + 1]
    __shared__ int part[256];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int b = pair_b[t], k = pair_k[t];
    const int32_t *col = seg_hist + (int64_t)b * nseg * K + k;
    // chunked scan: thread tid owns segments [tid*per, (tid+1)*per)
    const int per = (nseg + 255) / 256;
    int s0 = tid * per, s1 = min(s0 + per, nseg), acc = 0;
    for (int s = s0; s < s1; ++s)
        acc += col[(int64_t)s * K];
    part[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) {
            int v = part[i];
            part[i] = run;
            run += v;
        }
    }
    __syncthreads();
    acc = part[tid];
    for (int s = s0; s < s1; ++s) {
        seg_start[s] = acc;
        acc += col[(int64_t)s * K];
    }
    if (s1 == nseg && s0 < nseg)
        seg_start[nseg] = acc;
    if (nseg == 0 && tid == 0)
        seg_start[0] = 0;
    __syncthreads();
    const uint8_t *row = lbl_s + (int64_t)b * hw;
    for (int v = tid; v < V; v += 256) {
        const int r = sel[(int64_t)t * V + v];
        // largest s with seg_start[s] <= r
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (seg_start[mid] <= r)
                lo = mid;
            else
                hi = mid - 1;
        }
        int q = r - seg_start[lo];
        const int p0 = lo * DCL_SEG, p1 = min(p0 + DCL_SEG, hw);
        int found = -1;
        for (int p = p0; p < p1; ++p) {
            if (row[p] == (uint8_t)k) {
                if (q == 0) {
                    found = p;
                    break;
                }
                --q;
            }
        }
        pix[(int64_t)t * V + v] = found;
    }
}

extern "C" int dcl_rank_select(const uint8_t *lbl_s, const int32_t *seg_hist, int n, int hw, int K,
                               const int32_t *pair_b, const int32_t *pair_k, int T, int V,
                               const int32_t *sel, int32_t *pix, void *stream)
{
    DCL_CHECK_ARG(lbl_s && seg_hist && pair_b && pair_k && sel && pix, "null pointer");
    DCL_CHECK_ARG(n > 0 && hw > 0 && K > 0 && K <= DCL_MAX_CLASSES, "bad sizes");
    if (T == 0 || V == 0)
        return 0;
    const int nseg = (hw + DCL_SEG - 1) / DCL_SEG;
    DCL_CHECK_ARG(nseg <= RS_MAX_SEG, "feature map too large (more than 8192*256 pixels)");
    hipLaunchKernelGGL(k_rank_select, dim3(T), dim3(256), (nseg + 1) * sizeof(int),
                       (hipStream_t)stream, lbl_s, seg_hist, hw, K, nseg, pair_b, pair_k, V, sel, pix);
    DCL_LAUNCH_CHECK();
    return 0;
}
