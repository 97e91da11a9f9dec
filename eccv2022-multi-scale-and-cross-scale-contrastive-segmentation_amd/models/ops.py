"""Operator-level replacements inside the models where the library's default kernel is far from the
hardware roofline on MI355X (measured, see profiles/)."""
import torch
import torch.nn.functional as F


class _Conv3x3GemmWrw(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 convolution whose WEIGHT gradient is computed as im2col + batched GEMM.

    HRNet's segmentation head (models/HRNet.py:596-600 in the reference: conv3x3 720 -> 720 at 1/4
    resolution) holds 47 % of the model's FLOPs.  For its weight gradient (a 720 x 6480 x 393,216 GEMM
    at batch 12, 512x1024) MIOpen's default fp32 solver runs at 44 TFLOP/s (83.7 ms); unfolding the
    input and calling rocBLAS' batched SGEMM runs the same contraction at 113 TFLOP/s (32-35 ms
    including the im2col), bit-compatible up to fp32 summation order (2.5e-5 of max).  Forward and
    input-gradient stay on MIOpen (they already run at ~110 TFLOP/s)."""

    @staticmethod
    def forward(ctx, x, weight, bias, chunk):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.chunk = chunk
        return F.conv2d(x, weight, bias, stride=1, padding=1)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.nn.grad.conv2d_input(x.shape, weight, gy, stride=1, padding=1)
        if ctx.needs_input_grad[1]:
            n = x.shape[0]
            gyf = gy.flatten(2)                                        # [N, Cout, HW]
            for i in range(0, n, ctx.chunk):
                cols = F.unfold(x[i:i + ctx.chunk], 3, padding=1)      # [n, Cin*9, HW]
                part = torch.bmm(gyf[i:i + ctx.chunk], cols.transpose(1, 2)).sum(0)
                gw = part if gw is None else gw + part
            gw = gw.view_as(weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb, None


def conv3x3_gemm_wrw(x, conv: torch.nn.Conv2d, chunk: int = 4):
    """Apply ``conv`` (3x3, stride 1, padding 1, groups 1) with the GEMM weight-gradient path."""
    assert conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) \
        and conv.groups == 1 and conv.dilation == (1, 1)
    return _Conv3x3GemmWrw.apply(x, conv.weight, conv.bias, chunk)


class _UpsampleBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, addend, H, W, align_corners):
        import ctypes
        from .. import _lib
        L = _lib.lib()
        n, c, h, w = x.shape
        y = torch.empty((n, c, H, W), dtype=torch.float32, device=x.device)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.dcl_upsample_bilinear_fwd(_lib.ptr(x), _lib.ptr(addend), n * c, h, w, H, W,
                                               1 if align_corners else 0, _lib.ptr(y), st),
                   "dcl_upsample_bilinear_fwd")
        ctx.shape, ctx.align = (n, c, h, w), bool(align_corners)
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes
        from .. import _lib
        L = _lib.lib()
        n, c, h, w = ctx.shape
        dy = dy.contiguous()
        H, W = dy.shape[-2:]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(L.dcl_upsample_bilinear_bwd(_lib.ptr(dy), n * c, h, w, H, W, 1 if ctx.align else 0,
                                                   _lib.ptr(dx), st), "dcl_upsample_bilinear_bwd")
        return dx, (dy if ctx.needs_input_grad[1] else None), None, None, None


def upsample_bilinear(x, size, align_corners, add=None):
    """``add + F.interpolate(x, size, mode='bilinear', align_corners=...)`` (``add`` optional) on the HIP
    kernels of csrc/dcl_resize.hip for CUDA / float32 / contiguous NCHW inputs (16-B stores forward with the
    addend folded in, deterministic gather backward); PyTorch's own kernels otherwise."""
    H, W = int(size[0]), int(size[1])
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() \
            and not torch.is_autocast_enabled() and (H, W) != tuple(x.shape[-2:]) \
            and (add is None or (add.is_contiguous() and add.dtype == torch.float32
                                 and tuple(add.shape) == tuple(x.shape[:2]) + (H, W))):
        return _UpsampleBilinear.apply(x, add, H, W, bool(align_corners))
    y = x if (H, W) == tuple(x.shape[-2:]) else F.interpolate(x, size=(H, W), mode='bilinear',
                                                              align_corners=align_corners)
    return y if add is None else add + y
