// dcl_conv3x3_pre.hip -- the direct f16x3 3x3 convolution on a NORMALISED input that is never written.
//
// The reference's residual blocks run conv1 -> bn1 -> relu -> conv2 (models/HRNet.py:77-93 BasicBlock, :117-137 Bottleneck).  As
// separate kernels the norm's apply pass reads the raw convolution output z1 and writes y1 = relu(sc z1 + sh), which conv2 and
// conv2's weight gradient then read back: two tensor passes per block (k_bn_apply<true,false>: 134 launches, 11.9 GB of a
// 512 x 1024 x 12 HRNet-W48 step) whose only product is a tensor that can be recomputed from z1 with one fma + one max per value.
// These kernels are the tiles of dcl_conv3x3.hip (same conv_body, same schedule, same arithmetic) with that map applied while the
// input patch is staged (conv_body<..., PRE = true>): the convolution is handed z1 and the per-channel (sc, sh) table the
// statistics' finalisation wrote (dcl_bn_finalize_pre, dcl_bn.hip), the operand scale comes from the exact absmax of y1 that the
// finalisation derives from the per-channel extrema of z1 (the map is monotone per channel).  Results are bitwise those of
// dcl_bn_apply_parts followed by dcl_conv3x3_f16x3 with the same tile.
#include "dcl_conv_body.h"

namespace {

template <int R, int P>
__global__ __launch_bounds__(256, 1) void k_conv3x3_il_pre(ConvArgs a)
{
    conv_body<R, P, 1, 0, true, 1, true>(a, (int)blockIdx.x);
}

// two workgroups per CU, waves split 2 (rows) x 2 (channel tiles): the 48 / 64-channel layers
template <int R, int P>
__global__ __launch_bounds__(256, 2) void k_conv3x3_il_ws2_pre(ConvArgs a)
{
    conv_body<R, P, 1, 0, true, 2, true>(a, (int)blockIdx.x);
}

// stride 2 (one output row per wave): the second convolution of a two-step down-sampling chain of a fuse layer (reference
// models/HRNet.py:236-258: conv s2 -> bn -> relu -> conv s2) and the stem's conv2 (:333-338)
template <int R>
__global__ __launch_bounds__(256, 1) void k_conv3x3_il_s2_pre(ConvArgs a)
{
    conv_body<R, 1, 2, 0, true, 1, true>(a, (int)blockIdx.x);
}

}  // namespace

bool dcl_conv_pre_has_tile(int R, int P, int stride, bool ws2)
{
    if (stride == 2)
        return P == 1 && R >= 1 && R <= 3;
    if (ws2)
        return true;
    return (R == 3 && (P == 4 || P == 2 || P == 1)) || (R == 2 && (P == 4 || P == 1)) || (R == 1 && (P == 4 || P == 2 || P == 1));
}

bool dcl_conv_pre_launch(const ConvArgs &a, int R, int P, int stride, bool ws2, dim3 grid, hipStream_t stream)
{
    if (!dcl_conv_pre_has_tile(R, P, stride, ws2))
        return false;
    if (stride == 2) {
        if (R == 3)
            hipLaunchKernelGGL((k_conv3x3_il_s2_pre<3>), grid, dim3(256), 0, stream, a);
        else if (R == 2)
            hipLaunchKernelGGL((k_conv3x3_il_s2_pre<2>), grid, dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((k_conv3x3_il_s2_pre<1>), grid, dim3(256), 0, stream, a);
        dcl_note_kernel("k_conv3x3_il_s2_pre<%d>", R);
        return true;
    }
    if (ws2) {
        hipLaunchKernelGGL((k_conv3x3_il_ws2_pre<1, 4>), grid, dim3(256), 0, stream, a);
        dcl_note_kernel("k_conv3x3_il_ws2_pre<1,4>");
        return true;
    }
#define DCL_PRE_CASE(r, p)                                                              \
    if (R == r && P == p) {                                                             \
        hipLaunchKernelGGL((k_conv3x3_il_pre<r, p>), grid, dim3(256), 0, stream, a);   \
        dcl_note_kernel("k_conv3x3_il_pre<%d,%d>", r, p);                               \
        return true;                                                                    \
    }
    DCL_PRE_CASE(3, 4)
    DCL_PRE_CASE(3, 2)
    DCL_PRE_CASE(3, 1)
    DCL_PRE_CASE(2, 4)
    DCL_PRE_CASE(2, 1)
    DCL_PRE_CASE(1, 4)
    DCL_PRE_CASE(1, 2)
    DCL_PRE_CASE(1, 1)
#undef DCL_PRE_CASE
    return false;
}
