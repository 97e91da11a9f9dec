// dcl_metrics.hip -- per-step training metrics: argmax over the class planes of the
// logits + (predicted, target) 2-D histogram in ONE pass over the logits.
//
// Replaces t_get_confusion_matrix (reference utils/torch_utils.py:157-183): NCHW -> CNHW transpose copy, argmax,
// two one-hot matrices ([N*H*W, C] int64: ~1 GB at 12 x 512 x 1024 x 20) and a float matmul, called after every
// training step (managers/HRNet_Manager.py:117-121).  Here: HBM-bound, N*C*H*W*4 bytes read once (478 MB at the
// benchmark shape), integer arithmetic only -> bit-exact and order-independent.
//
//   * a work item owns 4 consecutive pixels: one 16-byte load per class plane (coalesced across the wave), running
//     (max, argmax) in registers; torch.argmax semantics: the FIRST maximal index wins, NaN counts as the maximum;
//   * the histogram lives in LDS ([C, cols] int32, cols = C + 1 when the dataset has an ignore id: the reference
//     one-hots the target with C + 1 classes and drops the last column) and is flushed with one integer atomicAdd
//     per non-zero cell per workgroup; matrices larger than the LDS budget (C * cols > 24 576) go straight to global
//     atomics;
//   * targets outside [0, cols) -- the reference's F.one_hot raises on them -- are not counted and reported in an
//     out-of-range counter the host checks with its per-step D2H.
#include "dcl_common.h"

namespace {

constexpr int LDS_CELLS = 24576;        // 96 KiB of int32: ADE20K's 150 x 151 matrix fits

struct CmArgs {
    const float *logits;        // [N, C, HW]
    const void *target;         // [N, HW] int64 | int32 | uint8
    int tbytes;                 // 8 | 4 | 1
    int N, C, HW, cols;
    int *cm;                    // [C, cols] int32, accumulated (zero it for a fresh matrix)
    int *oob;                   // [1] count of targets outside [0, cols)
};

__device__ __forceinline__ int load_target(const void *t, int tbytes, size_t i)
{
    if (tbytes == 8) {
        const long long v = ((const long long *)t)[i];
        return (v < 0 || v > 0x7fffffff) ? -1 : (int)v;
    }
    if (tbytes == 4)
        return ((const int *)t)[i];
    return (int)((const unsigned char *)t)[i];
}

// torch.argmax update rule for one candidate: replace when v > best, or when v is NaN and best is not
__device__ __forceinline__ void consider(float v, int c, float &best, int &arg)
{
    const bool take = (v > best) || (v != v && best == best);
    best = take ? v : best;
    arg = take ? c : arg;
}

// CELLS: LDS histogram capacity (0 = global atomics); the small instantiation keeps 8 workgroups per CU resident
template <int CELLS>
__global__ __launch_bounds__(256) void k_confusion(CmArgs a)
{
    constexpr bool USE_LDS = CELLS > 0;
    __shared__ int hist[USE_LDS ? CELLS : 1];
    const int cells = a.C * a.cols;
    if (USE_LDS) {
        for (int i = threadIdx.x; i < cells; i += 256)
            hist[i] = 0;
        __syncthreads();
    }
    const int quads = (a.HW + 3) / 4;                  // work items per image
    const long long total = (long long)a.N * quads;
    int bad = 0;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
        const int n = (int)(it / quads);
        const int p0 = (int)(it - (long long)n * quads) * 4;
        const float *lp = a.logits + (size_t)n * a.C * a.HW + p0;
        float best[4];
        int arg[4] = {0, 0, 0, 0};
        const bool vec = (p0 + 4 <= a.HW) && ((a.HW & 3) == 0);
        if (vec) {
            f32x4 v = *(const f32x4 *)lp;
            best[0] = v.x; best[1] = v.y; best[2] = v.z; best[3] = v.w;
            for (int c = 1; c < a.C; ++c) {
                v = *(const f32x4 *)(lp + (size_t)c * a.HW);
                consider(v.x, c, best[0], arg[0]);
                consider(v.y, c, best[1], arg[1]);
                consider(v.z, c, best[2], arg[2]);
                consider(v.w, c, best[3], arg[3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = min(p0 + e, a.HW - 1);
                best[e] = a.logits[((size_t)n * a.C) * a.HW + p];
                for (int c = 1; c < a.C; ++c)
                    consider(a.logits[((size_t)n * a.C + c) * a.HW + p], c, best[e], arg[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (p0 + e >= a.HW)
                continue;
            const int t = load_target(a.target, a.tbytes, (size_t)n * a.HW + p0 + e);
            if ((unsigned)t >= (unsigned)a.cols) {
                ++bad;
                continue;
            }
            if (USE_LDS)
                atomicAdd(&hist[arg[e] * a.cols + t], 1);
            else
                atomicAdd(&a.cm[arg[e] * a.cols + t], 1);
        }
    }
    if (bad)
        atomicAdd(a.oob, bad);
    if (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < cells; i += 256) {
            const int v = hist[i];
            if (v)
                atomicAdd(&a.cm[i], v);
        }
    }
}

// pa, pac, mIoU from the confusion matrix in ONE launch (the torch formulation is ~25 tiny kernels, each followed by
// a dispatch gap: ~1 ms of mostly idle GPU at the end of every step).  One workgroup, thread c owns class c.
//   pa   = sum_c cm[c][c] / sum cm                                   (utils/torch_utils.py:208-209)
//   pac  = mean_c cm[c][c] / max(rowsum_c, 1 if rowsum_c == 0)        (:210-212)
//   miou = mean_c cm[c][c] / (rowsum_c + colsum_c - cm[c][c]), NaN -> 0 (:268-275), c over the C real classes
// Integer sums are exact; the float arithmetic follows torch's (sums of float-converted integers, f32 division).
__global__ __launch_bounds__(256) void k_metrics_from_cm(const int *__restrict__ cm, int C, int ld, float *__restrict__ out)
{
    __shared__ float sh[3][4];
    const int c = threadIdx.x;
    float diag = 0.f, pacv = 0.f, iou = 0.f;
    long long tot = 0;
    if (c < C) {
        long long row = 0, col = 0;
        for (int j = 0; j < C; ++j) {
            row += cm[c * ld + j];
            col += cm[j * ld + c];
        }
        tot = row;
        const int d = cm[c * ld + c];
        diag = (float)d;
        pacv = diag / (row == 0 ? 1.0f : (float)row);
        const float den = (float)col + (float)row - diag;      // torch: row_sum (dim 0) + col_sum (dim 1) - diagonal, f32
        const float v = diag / den;
        iou = (v != v) ? 0.f : v;
    }
    float ftot = (float)tot;                                    // per-class row totals < 2^24 at any realistic batch
    diag = wave_sum(diag);
    pacv = wave_sum(pacv);
    iou = wave_sum(iou);
    ftot = wave_sum(ftot);
    __shared__ float st[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[0][w] = diag;
        sh[1][w] = pacv;
        sh[2][w] = iou;
        st[w] = ftot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float d = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        const float t = (st[0] + st[1]) + (st[2] + st[3]);
        out[0] = d / t;
        out[1] = ((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3])) / (float)C;
        out[2] = ((sh[2][0] + sh[2][1]) + (sh[2][2] + sh[2][3])) / (float)C;
    }
}

// confusion matrix from an arg-max map (uint8, produced by the fused up-sampling + cross-entropy forward): the
// histogram half of k_confusion only, 1 + target bytes per pixel
template <int CELLS>                   // LDS histogram cells (2048: two workgroups per CU; LDS_CELLS: one); 0: global atomics
__global__ __launch_bounds__(256) void k_confusion_pred(const unsigned char *__restrict__ pred, const void *target,
                                                       int tbytes, long long total, int C, int cols, int *cm, int *oob)
{
    __shared__ int hist[CELLS > 0 ? CELLS : 1];
    const int cells = C * cols;
    const bool use_lds = CELLS > 0;
    if (use_lds) {
        for (int i = threadIdx.x; i < cells; i += 256)
            hist[i] = 0;
        __syncthreads();
    }
    int bad = 0;
    auto count = [&](int t, int p) {
        if ((unsigned)t >= (unsigned)cols || p >= C) {
            bad += (unsigned)t >= (unsigned)cols ? 1 : 0;
            return;
        }
        if (use_lds)
            atomicAdd(&hist[p * cols + t], 1);
        else
            atomicAdd(&cm[p * cols + t], 1);
    };
    // four consecutive pixels per thread and trip: one 4-byte load of the predictions, the four targets' loads issued together
    const long long total4 = total & ~3LL;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < total4; i += (long long)gridDim.x * 1024) {
        const unsigned p4 = *(const unsigned *)(pred + i);
        int t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            t[k] = load_target(target, tbytes, (size_t)i + k);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            count(t[k], (int)((p4 >> (8 * k)) & 255u));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(total - total4))
        count(load_target(target, tbytes, (size_t)(total4 + threadIdx.x)), pred[total4 + threadIdx.x]);
    if (bad)
        atomicAdd(oob, bad);
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < cells; i += 256)
            if (hist[i])
                atomicAdd(&cm[i], hist[i]);
    }
}

}  // namespace

extern "C" int dcl_confusion_matrix_pred(const uint8_t *pred, int64_t total, const void *target, int target_bytes, int C,
                                         int cols, int32_t *cm, int32_t *oob, void *stream)
{
    DCL_CHECK_ARG(pred && target && cm && oob && total > 0 && C > 0 && (cols == C || cols == C + 1), "bad arguments");
    DCL_CHECK_ARG(target_bytes == 8 || target_bytes == 4 || target_bytes == 1, "target must be int64, int32 or uint8");
    DCL_CHECK_ARG((((uintptr_t)pred) & 3) == 0, "pred must be 4-byte aligned");
    // every workgroup ends with one global atomic per non-empty cell of its LDS histogram, all workgroups onto the same C x cols
    // addresses: with 2 048 workgroups that flush WAS the kernel (1.0 ms on 12 x 512 x 1024 pixels: ~2 000 same-address atomics
    // queue up per cell) -- two workgroups per CU, four pixels per thread and trip
    long long blocks = (total / 4 + 255) / 256;
    const long long cells = (long long)C * cols;
    const long long cap = cells <= 2048 ? 512 : 256;        // (the 96-KiB histogram: one workgroup per CU)
    if (blocks > cap)
        blocks = cap;
    if (blocks < 1)
        blocks = 1;
    // 150 x 151 cells (ADE20K): one global atomic per PIXEL took 1.0 ms on 16 x 640 x 640; the LDS histogram's flush issues its
    // atomics on consecutive addresses (whole cache lines per wave instruction)
    if (cells <= 2048)
        hipLaunchKernelGGL(k_confusion_pred<2048>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, target,
                           target_bytes, (long long)total, C, cols, cm, oob);
    else if (cells <= LDS_CELLS)
        hipLaunchKernelGGL(k_confusion_pred<LDS_CELLS>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, target,
                           target_bytes, (long long)total, C, cols, cm, oob);
    else
        hipLaunchKernelGGL(k_confusion_pred<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, target,
                           target_bytes, (long long)total, C, cols, cm, oob);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_metrics_from_cm(const int32_t *cm, int C, int ld, float *out3, void *stream)
{
    DCL_CHECK_ARG(cm && out3 && C > 0 && C <= 256 && ld >= C, "bad arguments (C <= 256)");
    hipLaunchKernelGGL(k_metrics_from_cm, dim3(1), dim3(256), 0, (hipStream_t)stream, cm, C, ld, out3);
    DCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int dcl_confusion_matrix(const float *logits, int N, int C, int HW, const void *target, int target_bytes,
                                    int cols, int32_t *cm, int32_t *oob, void *stream)
{
    DCL_CHECK_ARG(logits && target && cm && oob, "null pointer");
    DCL_CHECK_ARG(N > 0 && C > 0 && HW > 0 && (cols == C || cols == C + 1), "bad shape (cols must be C or C + 1)");
    DCL_CHECK_ARG(target_bytes == 8 || target_bytes == 4 || target_bytes == 1, "target must be int64, int32 or uint8");
    DCL_CHECK_ARG((((uintptr_t)logits) & 15) == 0, "logits must be 16-byte aligned");
    CmArgs a;
    a.logits = logits;
    a.target = target;
    a.tbytes = target_bytes;
    a.N = N;
    a.C = C;
    a.HW = HW;
    a.cols = cols;
    a.cm = cm;
    a.oob = oob;
    const long long items = (long long)N * ((HW + 3) / 4);
    long long blocks = (items + 255) / 256;
    if (blocks > 2048)
        blocks = 2048;              // 8 workgroups per CU, each flushes its LDS histogram once
    if ((long long)C * cols <= 2048)
        hipLaunchKernelGGL((k_confusion<2048>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else if ((long long)C * cols <= LDS_CELLS)
        hipLaunchKernelGGL((k_confusion<LDS_CELLS>), dim3((unsigned)(blocks > 256 ? 256 : blocks)), dim3(256), 0,
                           (hipStream_t)stream, a);      // 96 KiB of LDS: one workgroup per CU
    else
        hipLaunchKernelGGL((k_confusion<0>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    DCL_LAUNCH_CHECK();
    return 0;
}
