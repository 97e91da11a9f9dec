"""Operator-level replacements inside the models where the library's default kernel is far from the
hardware roofline on MI355X (measured, see profiles/)."""
import os

import torch
import torch.nn.functional as F


class _UpsampleBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, addend, H, W, align_corners, relu):
        from .. import _lib
        L = _lib.lib()
        n, c, h, w = x.shape
        y = torch.empty((n, c, H, W), dtype=torch.float32, device=x.device)
        st = _lib.stream_ptr(x.device)
        _lib.check(L.dcl_upsample_bilinear_fwd(_lib.ptr(x), _lib.ptr(addend), n * c, h, w, H, W,
                                               1 if align_corners else 0, 1 if relu else 0, _lib.ptr(y), st),
                   "dcl_upsample_bilinear_fwd")
        ctx.shape, ctx.align, ctx.relu = (n, c, h, w), bool(align_corners), bool(relu)
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _lib
        L = _lib.lib()
        n, c, h, w = ctx.shape
        if ctx.relu:                                   # gradient of the fused ReLU: dy where y > 0
            (y,) = ctx.saved_tensors
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)
        dy = dy.contiguous()
        H, W = dy.shape[-2:]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
            st = _lib.stream_ptr(dy.device)
            _lib.check(L.dcl_upsample_bilinear_bwd(_lib.ptr(dy), n * c, h, w, H, W, 1 if ctx.align else 0,
                                                   _lib.ptr(dx), st), "dcl_upsample_bilinear_bwd")
        return dx, (dy if ctx.needs_input_grad[1] else None), None, None, None, None


class _UpsampleConcat(torch.autograd.Function):
    """cat([t0, up(t1), up(t2), ...], dim=1) with every up-sampled map written straight into its channel slice of the
    result (no separate maps, no cat copy of them) and, backward, read straight out of the slice of the incoming
    gradient (no .contiguous() copies of the narrow views torch.cat's backward hands out)."""

    @staticmethod
    def forward(ctx, align_corners, *ts):
        from .. import _lib
        L = _lib.lib()
        n, _, H, W = ts[0].shape
        ctot = sum(t.shape[1] for t in ts)
        out = torch.empty((n, ctot, H, W), dtype=torch.float32, device=ts[0].device)
        st = _lib.stream_ptr(out.device)
        c0 = 0
        for t in ts:
            c, h, w = t.shape[1:]
            if (h, w) == (H, W):
                out[:, c0:c0 + c].copy_(t)
            else:
                _lib.check(L.dcl_upsample_bilinear_fwd_slice(_lib.ptr(t), n, c, h, w, H, W, 1 if align_corners else 0,
                                                             _lib.ptr(out), ctot, c0, st),
                           "dcl_upsample_bilinear_fwd_slice")
            c0 += c
        ctx.shapes, ctx.align = [tuple(t.shape) for t in ts], bool(align_corners)
        return out

    @staticmethod
    def backward(ctx, dy):
        from .. import _lib
        L = _lib.lib()
        dy = dy.contiguous()
        n, ctot, H, W = dy.shape
        st = _lib.stream_ptr(dy.device)
        grads, c0 = [], 0
        for i, (_, c, h, w) in enumerate(ctx.shapes):
            g = None
            if ctx.needs_input_grad[1 + i]:
                if (h, w) == (H, W):
                    g = dy[:, c0:c0 + c]
                else:
                    g = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
                    _lib.check(L.dcl_upsample_bilinear_bwd_slice(_lib.ptr(dy), ctot, c0, n, c, h, w, H, W,
                                                                 1 if ctx.align else 0, _lib.ptr(g), st),
                               "dcl_upsample_bilinear_bwd_slice")
            grads.append(g)
            c0 += c
        return (None, *grads)


class _FanOut(torch.autograd.Function):
    """k aliases of one tensor for k consumers; the backward sums the k gradients in ONE kernel (k + 1 tensor passes)
    where autograd's own accumulation chains k - 1 two-input adds (3 (k - 1) passes)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        from .. import _lib
        L = _lib.lib()
        gs = [g.contiguous() for g in gs]
        while len(gs) > 1:
            part, gs = gs[:4], gs[4:]
            out = torch.empty_like(part[0])
            p = [_lib.ptr(t) for t in part] + [None] * (4 - len(part))
            _lib.check(L.dcl_add_n(p[0], p[1], p[2], p[3], out.numel(), _lib.ptr(out), _lib.stream_ptr(out.device)),
                       "dcl_add_n")
            gs = [out] + gs
        return gs[0], None


from ..debug import cfg as _dbg      # noqa: E402  (A/B switches of the tuning tools: one object, mscs_amd/debug.py)
_FANOUT = _dbg.fanout
_UPSAMPLE_TAG = _dbg.upsample_tag


def fan_out(x, k):
    """k aliases of x whose gradients are summed by one kernel (see _FanOut); the absmax tag travels along."""
    if k < 3 or not (_FANOUT and x.is_cuda and x.dtype == torch.float32 and x.requires_grad and torch.is_grad_enabled()):
        return [x] * k
    outs = _FanOut.apply(x, k)
    tag = getattr(x, "_dcl_amax", None)
    if tag is not None:
        for o in outs:
            o._dcl_amax = (o._version, tag[1])
    return list(outs)


def upsample_concat(ts, align_corners):
    """``torch.cat([ts[0]] + [F.interpolate(t, ts[0].shape[-2:], mode='bilinear', align_corners=...) for t in ts[1:]], 1)``
    (reference models/HRNet.py:549-553) in one pass over the result for CUDA / float32 / contiguous maps."""
    size = ts[0].shape[-2:]
    if HIP_UPSAMPLE and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.is_contiguous() for t in ts) \
            and not torch.is_autocast_enabled():
        from . import amax as _amax
        out = _UpsampleConcat.apply(bool(align_corners), *ts)
        # absmax side channel for the head convolution: bilinear interpolation is a convex combination, so max|up(t)|
        # <= max|t| and the maxima of the (small, already tagged) inputs bound the result -- no pass over its 1.1 GB
        return _amax.tag(out, torch.cat([_amax.amax_of(t) for t in ts]))
    return torch.cat([ts[0]] + [upsample_bilinear(t, size, align_corners) for t in ts[1:]], 1)


HIP_UPSAMPLE = True        # False: F.interpolate everywhere (library_kernels_only(), the eager comparator of bench.py)


class library_kernels_only:
    """Context manager for the eager-structure comparator (bench.py ``eager_gpu_step_ms``): inside it the model code
    of this package runs on stock PyTorch-ROCm kernels only -- F.interpolate instead of the HIP resize kernels and
    one stream instead of one per HRNet branch.  (The convolution / norm classes are selected at construction:
    graph keys branch_conv='library', head_conv='library', fused_bn=False, gemm_conv1x1=False.)"""

    def __enter__(self):
        import importlib
        _h = importlib.import_module(__package__ + '.HRNet')        # the module (the package exports the class too)
        global HIP_UPSAMPLE
        self.prev = (HIP_UPSAMPLE, _h._BRANCH_STREAMS)
        HIP_UPSAMPLE, _h._BRANCH_STREAMS = False, False
        return self

    def __exit__(self, *exc):
        import importlib
        _h = importlib.import_module(__package__ + '.HRNet')
        global HIP_UPSAMPLE
        HIP_UPSAMPLE, _h._BRANCH_STREAMS = self.prev
        return False


def upsample_bilinear(x, size, align_corners, add=None, relu=False):
    """``add + F.interpolate(x, size, mode='bilinear', align_corners=...)`` (``add`` optional; ``relu``: followed by a
    ReLU, fused into the same pass) on the HIP
    kernels of csrc/dcl_resize.hip for CUDA / float32 / contiguous NCHW inputs (16-B stores forward with the
    addend folded in, deterministic gather backward); PyTorch's own kernels otherwise."""
    H, W = int(size[0]), int(size[1])
    if HIP_UPSAMPLE and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() \
            and not torch.is_autocast_enabled() and (H, W) != tuple(x.shape[-2:]) \
            and (add is None or (add.is_contiguous() and add.dtype == torch.float32
                                 and tuple(add.shape) == tuple(x.shape[:2]) + (H, W))):
        y = _UpsampleBilinear.apply(x, add, H, W, bool(align_corners), bool(relu))
        # absmax side channel without a pass over y: interpolation is a convex combination, so max|up(x)| <= max|x|;
        # with an addend the sum of the two maxima bounds the result (a ReLU on top only shrinks it)
        from . import amax as _amax
        from .. import _lib
        if not _UPSAMPLE_TAG:
            return y
        tx, ta = _amax.tag_of(x), (_amax.tag_of(add) if add is not None else None)
        if tx is not None and add is None:
            _amax.tag(y, tx)
        elif tx is not None and ta is not None:
            buf = torch.empty(1, dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().dcl_amax_sum2(_lib.ptr(tx), tx.numel(), _lib.ptr(ta), ta.numel(), _lib.ptr(buf),
                                                _lib.stream_ptr(x.device)), "dcl_amax_sum2")
            _amax.tag(y, buf)
        return y
    y = x if (H, W) == tuple(x.shape[-2:]) else F.interpolate(x, size=(H, W), mode='bilinear',
                                                              align_corners=align_corners)
    y = y if add is None else add + y
    return F.relu(y) if relu else y


# ---- direct f16x3 3x3 convolution (csrc/dcl_conv3x3.hip) ----------------------------------------------------------

def _stream(t):
    import ctypes
    from .. import _lib
    return _lib.stream_ptr(t.device)


def conv3x3_pack(weight, wamax, transposed=False):
    """Weights [Co, Ci, 3, 3] (or [Co, Ci, 1, 1]) -> MFMA fragment order (f16 hi / lo), for the forward (M = Co,
    K = Ci) or, with ``transposed``, for the data gradient (M = Ci, K = Co, taps flipped)."""
    from .. import _lib
    co, ci = weight.shape[0], weight.shape[1]
    taps = weight.shape[2] * weight.shape[3]
    assert taps in (1, 9)
    m, k = (ci, co) if transposed else (co, ci)
    nbytes = ((m + 31) // 32) * ((k + 15) // 16) * taps * 2 * 64 * 16
    wp = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    _lib.check(_lib.lib().dcl_conv3x3_pack(_lib.ptr(weight), m, k, (1 if transposed else 0) | (2 if taps == 1 else 0),
                                           _lib.ptr(wamax), _lib.ptr(wp), _stream(weight)), "dcl_conv3x3_pack")
    return wp


def conv1x1_launch(x, wp, cout, xamax, wamax, out, tile_r=0, tile_p=0, addend=None, bias=None):
    """1x1 convolution (or its data gradient, with transposed fragments) on the one-tap mode of the direct kernel."""
    from .. import _lib
    n, c, h, w = x.shape
    _lib.check(_lib.lib().dcl_conv1x1_f16x3(_lib.ptr(x), n, c, h, w, _lib.ptr(wp), cout, _lib.ptr(xamax),
                                            xamax.numel(), _lib.ptr(wamax), _lib.ptr(addend), _lib.ptr(bias),
                                            _lib.ptr(out), tile_r, tile_p, _stream(x)), "dcl_conv1x1_f16x3")
    return out


def conv1x1_direct(x, weight, transposed=False):
    """y = conv2d(x, weight [Co, Ci, 1, 1]) (or, ``transposed``, its data gradient applied to x) on the f16x3 kernel."""
    from .amax import amax_of
    x, weight = x.contiguous(), weight.contiguous()
    wamax = amax_of(weight)
    wp = conv3x3_pack(weight, wamax, transposed)
    cout = weight.shape[1] if transposed else weight.shape[0]
    out = torch.empty((x.shape[0], cout, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
    return conv1x1_launch(x, wp, cout, amax_of(x), wamax, out)


def _conv1x1_by_gemm(rows, k, x, both_row_contiguous):
    """A 1x1 convolution direction as a batched dcl_gemm_f16x3 over the images ([rows, k] x [k, H W] per image): pays where
    the produced channel count fills a 256-row tile (tools/conv1x1_shapes.py: 64 -> 256 forward at 128 x 256, batch 12:
    109 us against 186 library / 290 tile kernel; 256 -> 64 data gradient 120 against 160 / 290; 192 -> 256 forward 17
    against 32 / 33; 64 -> 64 loses: 87 against 50)."""
    hw = x.shape[2] * x.shape[3]
    return GEMM_CONV1X1 and rows >= 192 and k >= 32 and (both_row_contiguous or k % 32 == 0) and hw % 4 == 0 \
        and rows % 4 == 0 and max(rows, k) * hw * 4 < (1 << 32)


GEMM_CONV1X1 = _dbg.gemm_conv1x1     # (DCL_GEMM_CONV1X1=0: the library / tile-kernel paths, for A/B runs)
LIB_CONV1X1_ADDEND = _dbg.lib_conv1x1_addend     # ... by the library's GEMM with beta = 1 (DCL_LIB_CONV1X1_ADDEND=0: tile kernel)
GEMM_CONV1X1_ADDEND = _dbg.gemm_conv1x1_addend   # residual gradient accumulated by the GEMM (DCL_GEMM_CONV1X1_ADDEND=0: tile kernel)


def conv1x1_gemm(x, weight2, out, xamax, wamax, transposed=False, accumulate=False):
    """out[n] [rows, H W] (+)= W x[n] (forward: rows = Co, W = weight2 [Co, Ci]) or W^T x[n] (``transposed``: the data
    gradient, rows = Ci) as one batched split-f16 GEMM; x, out NCHW contiguous.  ``accumulate``: added to what ``out`` holds
    (the residual branch's gradient), in place -- an absmax tag of ``out`` does not describe the sum and is dropped."""
    n, k, h, w = x.shape
    hw = h * w
    co, ci = weight2.shape
    rows = ci if transposed else co
    gemm_f16x3(weight2, not transposed, ci, x, False, hw, rows, hw, k, out, hw, wamax, xamax, batch=n,
               strides=(0, k * hw, rows * hw), splitk=1, accumulate=accumulate)
    if accumulate and getattr(out, "_dcl_amax", None) is not None:
        out._dcl_amax = None
    return out


def _conv1x1_by_library(x, y):
    """Forward / data gradient of a 1x1 convolution: above 64 MB of input + output the layer is bound by HBM and the
    library's streaming GEMM moves the bytes faster than the tile kernel, whose patch staging is built for the 3x3
    case (tools/conv1x1_shapes.py: 64 -> 256 at 128 x 256, batch 12: 182 vs 226 us); below, the direct kernel wins or
    ties and brings the absmax side channel and the fused bias / residual-gradient epilogue."""
    return (x.numel() + y.numel()) * 4 > (64 << 20)


def conv1x1_wgrad_supported(x, cout):
    return x.shape[1] % 16 == 0 and cout % 16 == 0 and x.shape[3] % 8 == 0 \
        and max(x.shape[1], cout) * x.shape[2] * x.shape[3] * 4 < (1 << 32)


def conv1x1_wgrad(x, gy):
    """dw [Co, Ci, 1, 1] of a 1x1 convolution, f16x3 (csrc/dcl_wgrad3x3d.hip, k_wgrad1x1d)."""
    from .. import _lib
    from .amax import amax_of
    n, ci, h, w = x.shape
    co = gy.shape[1]
    L = _lib.lib()
    splits = L.dcl_wgrad1x1_splits(n, ci, co, h, w)
    if splits <= 0:
        raise RuntimeError("conv1x1_wgrad: unsupported shape")
    part = torch.empty(splits * co * ci, dtype=torch.float32, device=x.device)
    dw = torch.empty((co, ci, 1, 1), dtype=torch.float32, device=x.device)
    xa, ga = amax_of(x), amax_of(gy)
    _lib.check(L.dcl_wgrad1x1_f16x3(_lib.ptr(x), _lib.ptr(gy), n, ci, co, h, w, _lib.ptr(xa), xa.numel(),
                                    _lib.ptr(ga), ga.numel(), _lib.ptr(part), _lib.ptr(dw), _stream(x)),
               "dcl_wgrad1x1_f16x3")
    return dw


def conv3x3_launch(x, wp, cout, xamax, wamax, out, tile_r=0, tile_p=0, addend=None, stride=1, in_up=1, bias=None):
    """stride = 2: stride-2 convolution; in_up = 2: x is the gradient of a stride-2 convolution (its data gradient
    is the stride-1 transposed convolution of x with zeros inserted at the odd coordinates; out has the size of the
    convolution's input)."""
    from .. import _lib
    n, c, h, w = x.shape
    _lib.check(_lib.lib().dcl_conv3x3_f16x3(_lib.ptr(x), n, c, h, w, _lib.ptr(wp), cout, _lib.ptr(xamax),
                                            xamax.numel(), _lib.ptr(wamax), _lib.ptr(addend), _lib.ptr(bias),
                                            _lib.ptr(out), stride, in_up, out.shape[2], out.shape[3], tile_r, tile_p, _stream(x)),
               "dcl_conv3x3_f16x3")
    return out


def conv3x3_bnstats_tiles(x, cout):
    """Pixel tiles of the epilogue-statistics form of the 3x3 / stride-1 convolution on x (0 = not available for the shape)."""
    from .. import _lib
    n, c, h, w = x.shape
    return int(_lib.lib().dcl_conv3x3_bnstats_tiles(n, c, cout, h, w))


def conv3x3_launch_bnstats(x, wp, cout, xamax, wamax, out, pivot, ntile, addend=None, bias=None):
    """conv3x3_launch (stride 1) whose epilogue also leaves the batch-norm partial sums of ``out``: returns (part f32
    [cout * ntile * 2], pivot_out f32 [cout]) for ``dcl_bn_apply_parts(ns=ntile)`` -- the norm behind the convolution then
    runs without its statistics pass (csrc/dcl_conv3x3.hip conv_body ST; reference models/HRNet.py:77-93 conv -> bn)."""
    from .. import _lib
    n, c, h, w = x.shape
    ws = torch.empty((cout * ntile * 2 + cout,), dtype=torch.float32, device=x.device)
    part, pivot_out = ws[:cout * ntile * 2], ws[cout * ntile * 2:]
    _lib.check(_lib.lib().dcl_conv3x3_bnstats_f16x3(_lib.ptr(x), n, c, h, w, _lib.ptr(wp), cout, _lib.ptr(xamax),
                                                    xamax.numel(), _lib.ptr(wamax), _lib.ptr(addend), _lib.ptr(bias),
                                                    _lib.ptr(out), _lib.ptr(pivot), _lib.ptr(part), _lib.ptr(pivot_out),
                                                    _stream(x)), "dcl_conv3x3_bnstats_f16x3")
    return part, pivot_out


SMALL_CIN_STEM = _dbg.small_cin_stem     # the stem's 3 -> 64 stride-2 convolution on its own fp32 kernel (DCL_SMALL_CIN_STEM=0: tile kernel)
CONV_BN_STATS = _dbg.conv_bn_stats       # batch-norm statistics in the producing convolution's epilogue (DCL_CONV_BN_STATS=0: off)


class GradToken:
    """Carries the gradient of a residual connection from the norm layer that produces it (``bn(y, residual=x,
    grad_token=tok)`` stores it here instead of handing it to autograd) to the convolution that also consumes x
    (``conv(x, grad_token=tok)`` adds it in the epilogue of its data-gradient kernel): one tensor add per
    residual block disappears.  Only valid when both consumers see the SAME tensor x and the convolution's
    backward runs after the norm's (it is earlier in the forward)."""
    __slots__ = ("dres",)

    def __init__(self):
        self.dres = None


def conv3x3_direct(x, weight, transposed=False, stride=1, out_hw=None):
    """y = conv2d(x, weight, stride=stride, padding=1) (or, ``transposed``, its data gradient applied to x; for
    stride 2 ``out_hw`` is the size of the convolution's input) on the f16x3 direct kernel; x [N, C, H, W] f32
    contiguous, weight [Co, Ci, 3, 3] f32 contiguous."""
    from .amax import amax_of
    x = x.contiguous()
    weight = weight.contiguous()
    wamax = amax_of(weight)
    wp = conv3x3_pack(weight, wamax, transposed)
    cout = weight.shape[1] if transposed else weight.shape[0]
    if transposed:
        oh, ow = (x.shape[2], x.shape[3]) if stride == 1 else out_hw
        out = torch.empty((x.shape[0], cout, oh, ow), dtype=torch.float32, device=x.device)
        return conv3x3_launch(x, wp, cout, amax_of(x), wamax, out, in_up=stride)
    oh, ow = (x.shape[2] - 1) // stride + 1, (x.shape[3] - 1) // stride + 1
    out = torch.empty((x.shape[0], cout, oh, ow), dtype=torch.float32, device=x.device)
    return conv3x3_launch(x, wp, cout, amax_of(x), wamax, out, stride=stride)


def conv3x3_wgrad_supported(x, cout, stride=1):
    """channel counts in multiples of 16; widths in multiples of 8 (stride 1: any width -- conv3x3_wgrad pads the rows)"""
    return x.shape[1] % 16 == 0 and cout % 16 == 0 and (x.shape[3] % 8 == 0 or stride == 1)


def conv3x3_wgrad(x, gy, stride=1):
    """dw [Co, Ci, 3, 3] = weight gradient of conv2d(x, w, stride=stride, padding=1) for the output gradient gy,
    on the f16x3 kernel of csrc/dcl_wgrad3x3.hip (x [N, Ci, H, W], gy [N, Co, Ho, Wo], contiguous f32)."""
    from .. import _lib
    from .amax import amax_of, tag
    if stride == 1 and x.shape[3] % 8:
        # the kernel walks the rows in octets: zero columns on the right change nothing (x: the convolution's own padding;
        # gy: no output there) -- e.g. the 20 x 20 maps of a 640 x 640 input at stride 32
        pad = 8 - x.shape[3] % 8
        xa, ga = amax_of(x), amax_of(gy)
        x, gy = torch.nn.functional.pad(x, (0, pad)), torch.nn.functional.pad(gy, (0, pad))
        tag(x, xa), tag(gy, ga)
    n, ci, h, w = x.shape
    co = gy.shape[1]
    L = _lib.lib()
    splits = L.dcl_wgrad3x3_splits(n, ci, co, h, w, stride)
    if splits <= 0:
        raise RuntimeError("conv3x3_wgrad: unsupported shape")
    part = torch.empty(splits * 9 * co * ci, dtype=torch.float32, device=x.device)
    dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=x.device)
    xa, ga = amax_of(x), amax_of(gy)
    assert gy.shape[2] == (h - 1) // stride + 1 and gy.shape[3] == (w - 1) // stride + 1
    _lib.check(L.dcl_wgrad3x3_f16x3(_lib.ptr(x), _lib.ptr(gy), n, ci, co, h, w, _lib.ptr(xa), xa.numel(),
                                    _lib.ptr(ga), ga.numel(), stride, _lib.ptr(part), _lib.ptr(dw), _stream(x)),
               "dcl_wgrad3x3_f16x3")
    return dw


class _Conv3x3Direct(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 convolution on the f16x3 (fp32-equivalent) kernels: forward and data gradient through
    csrc/dcl_conv3x3.hip, weight gradient through csrc/dcl_wgrad3x3.hip (channel counts that are not multiples
    of 16: ATen / MIOpen)."""

    @staticmethod
    def forward(ctx, x, weight, mod, token=None, bias=None, stats_for=None):
        from .amax import amax_of
        ctx.token = token
        ctx.has_bias = bias is not None
        ctx.stride = st = mod.stride[0]
        ctx.k1 = k1 = mod.kernel_size == (1, 1)
        wamax, wp, _ = mod.packed_weights()
        out = torch.empty((x.shape[0], weight.shape[0], (x.shape[2] - 1) // st + 1, (x.shape[3] - 1) // st + 1),
                          dtype=torch.float32, device=x.device)
        if k1 and _conv1x1_by_gemm(weight.shape[0], weight.shape[1], x, False):
            conv1x1_gemm(x, weight.view(weight.shape[0], -1), out, amax_of(x), wamax)
            if bias is not None:
                out += bias.view(1, -1, 1, 1)
        elif k1 and _conv1x1_by_library(x, out):
            n, ci, h, w = x.shape
            torch.matmul(weight.view(-1, ci), x.view(n, ci, h * w), out=out.view(n, -1, h * w))
            if bias is not None:
                out += bias.view(1, -1, 1, 1)
        elif k1:
            conv1x1_launch(x, wp, weight.shape[0], amax_of(x), wamax, out, bias=bias)
        elif st == 2 and weight.shape[1] <= 4 and SMALL_CIN_STEM:
            # the stem's convolution on the image: 3 input channels, fp32 FMAs (csrc/dcl_conv3x3.hip k_conv3x3_s2_smallcin)
            from .. import _lib
            n, ci, h, w = x.shape
            _lib.check(_lib.lib().dcl_conv3x3_s2_smallcin(_lib.ptr(x), n, ci, h, w, _lib.ptr(weight.contiguous()),
                                                          weight.shape[0], _lib.ptr(bias), _lib.ptr(out), _stream(x)),
                       "dcl_conv3x3_s2_smallcin")
        else:
            ntile = conv3x3_bnstats_tiles(x, weight.shape[0]) if (stats_for is not None and st == 1) else 0
            if ntile > 0:
                # the norm layer behind this convolution gets its partial sums from the epilogue (handed over by the module:
                # DirectConv2d.forward tags the output)
                part, pivot = conv3x3_launch_bnstats(x, wp, weight.shape[0], amax_of(x), wamax, out,
                                                     stats_for.running_mean, ntile, bias=bias)
                mod._bnstats = (part, ntile, pivot, stats_for.running_mean.data_ptr())
            else:
                conv3x3_launch(x, wp, weight.shape[0], amax_of(x), wamax, out, stride=st, bias=bias)
        ctx.save_for_backward(x, weight)
        ctx.mod = mod
        return out

    @staticmethod
    def backward(ctx, gy):
        from .amax import amax_of
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            wamax, _, wpt = ctx.mod.packed_weights()
            gx = torch.empty_like(x)
            addend = None
            if ctx.token is not None and ctx.token.dres is not None:
                addend, ctx.token.dres = ctx.token.dres, None        # gradient of the residual branch, fused in
            if ctx.k1 and _conv1x1_by_gemm(weight.shape[1], weight.shape[0], gy, True) and (
                    addend is None or (GEMM_CONV1X1_ADDEND and addend.shape == x.shape and addend.is_contiguous()
                                       and addend.dtype == torch.float32)):
                # with a residual gradient (experiment, off by default): accumulated INTO it by the GEMM's epilogue (C += ...).
                # Measured on the 256-channel gradients of layer 1's Bottlenecks: 467 us against 372 for the tile kernel's
                # fused addend -- with K = 64 the launch is all epilogue, and the epilogue now also reads 403 MB
                if addend is not None:
                    gx = addend
                conv1x1_gemm(gy, weight.view(weight.shape[0], -1), gx, amax_of(gy), wamax, transposed=True,
                             accumulate=addend is not None)
            elif ctx.k1 and _conv1x1_by_library(x, gy) and addend is None:
                n, ci, h, w = x.shape
                torch.matmul(weight.view(-1, ci).t(), gy.view(n, -1, h * w), out=gx.view(n, ci, h * w))
            elif ctx.k1 and LIB_CONV1X1_ADDEND and _conv1x1_by_library(x, gy) and weight.shape[1] >= 128 \
                    and addend.shape == x.shape and addend.is_contiguous() and addend.dtype == torch.float32:
                # HBM-bound size with a residual gradient: the library's GEMM with beta = 1 accumulates INTO it (layer 1's
                # 256-channel gradients: read 100 + 403 MB, write 403 MB; the tile kernel's fused addend runs at half the HBM rate)
                n, ci, h, w = x.shape
                gx = addend
                gx.view(n, ci, h * w).baddbmm_(weight.view(-1, ci).t().unsqueeze(0).expand(n, ci, weight.shape[0]),
                                               gy.view(n, -1, h * w))
                if getattr(gx, "_dcl_amax", None) is not None:
                    gx._dcl_amax = None
            elif ctx.k1:
                # (with a residual gradient to add, the tile kernel's fused epilogue beats library GEMM + add kernel
                # also above the size where the GEMM alone is faster)
                conv1x1_launch(gy, wpt, weight.shape[1], amax_of(gy), wamax, gx, addend=addend)
            else:
                conv3x3_launch(gy, wpt, weight.shape[1], amax_of(gy), wamax, gx, addend=addend, in_up=ctx.stride)
        if ctx.needs_input_grad[1]:
            if ctx.k1:
                if conv1x1_wgrad_supported(x, weight.shape[0]):
                    gw = conv1x1_wgrad(x, gy)
                else:
                    n, ci, h, w = x.shape
                    gw = torch.bmm(gy.view(n, -1, h * w), x.view(n, ci, h * w).transpose(1, 2)).sum(0).view_as(weight)
            elif conv3x3_wgrad_supported(x, weight.shape[0], ctx.stride):
                gw = conv3x3_wgrad(x, gy, ctx.stride)
            else:
                st = ctx.stride
                gw = torch.ops.aten.convolution_backward(gy, x, weight, None, [st, st], [1, 1], [1, 1], False,
                                                         [0, 0], 1, [False, True, False])[1]
        gb = gy.sum((0, 2, 3)) if (ctx.has_bias and ctx.needs_input_grad[4]) else None
        return gx, gw, None, None, gb, None


class DirectConv2d(torch.nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict) whose 3x3, pad-1, stride-1 or stride-2 case runs on the direct f16x3
    kernels for contiguous fp32 CUDA inputs; every other configuration falls through to nn.Conv2d.forward."""

    def eligible(self, x):
        return (((self.kernel_size == (3, 3) and self.stride in ((1, 1), (2, 2)) and self.padding == (1, 1))
                 or (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)))
                and self.dilation == (1, 1) and self.groups == 1
                and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and self.weight.dtype == torch.float32 and not torch.is_autocast_enabled()
                and x.is_contiguous())

    def packed_weights(self):
        """(max|w|, forward fragments, data-gradient fragments), rebuilt when the weight tensor was modified."""
        from .amax import amax_of
        w = self.weight
        key = (w._version, w.data_ptr())
        cache = getattr(self, "_packed", None)
        if cache is None or cache[0] != key:
            wd = w.detach()
            wamax = amax_of(wd)
            cache = (key, wamax, conv3x3_pack(wd, wamax, False), conv3x3_pack(wd, wamax, True))
            self._packed = cache
        return cache[1], cache[2], cache[3]

    def forward(self, x, grad_token=None, stats_for=None):
        """``stats_for``: the FusedBatchNorm2d that consumes the result next (training, one rank): the convolution's epilogue
        then leaves the norm's partial sums and the output carries them (``_dcl_bnstats``) -- no statistics pass over y."""
        if self.eligible(x):
            if stats_for is not None and not (CONV_BN_STATS and stats_for.takes_conv_stats(x)):
                stats_for = None
            self._bnstats = None
            out = _Conv3x3Direct.apply(x, self.weight, self, grad_token, self.bias, stats_for)
            got, self._bnstats = self._bnstats, None
            if got is not None:
                out._dcl_bnstats = (out._version,) + got
            return out
        return super().forward(x)

    def fuses_residual_grad(self, x):
        """True when a GradToken may be used for x: the direct path runs and x needs a gradient."""
        return self.eligible(x) and x.requires_grad and torch.is_grad_enabled()


class ConvPackGroup:
    """All DirectConv2d weights of a model packed by TWO launches per optimizer step (multi-tensor absmax, then
    multi-tensor pack of both orientations) instead of three small launches per convolution.  ``refresh()`` is
    called at the start of the model's forward; it does nothing while no weight was modified."""

    def __init__(self, module: torch.nn.Module):
        self.convs = [m for m in module.modules() if isinstance(m, DirectConv2d)]
        self.key = None
        self.tables = None

    def _build(self, dev):
        import numpy as np
        absjobs = np.zeros(len(self.convs), dtype=[("x", "<u8"), ("out", "<u8"), ("n", "<i8"), ("fb", "<i4"), ("pad", "<i4")])
        packjobs = np.zeros(2 * len(self.convs), dtype=[("w", "<u8"), ("wp", "<u8"), ("amax", "<u8"), ("M", "<i4"),
                                                        ("K", "<i4"), ("tr", "<i4"), ("fb", "<i4")])
        self.amax = torch.zeros(len(self.convs), dtype=torch.float32, device=dev)
        self.wp = []
        ab2j, pb2j = [], []
        for i, m in enumerate(self.convs):
            w = m.weight
            co, ci = w.shape[0], w.shape[1]
            nblk = (w.numel() + 4095) // 4096
            absjobs[i] = (w.data_ptr(), self.amax[i:i + 1].data_ptr(), w.numel(), len(ab2j), 0)
            ab2j += [i] * nblk
            pair = []
            taps = w.shape[2] * w.shape[3] if w.dim() == 4 else 1
            for tr in (0, 1):
                mm, kk = (ci, co) if tr else (co, ci)
                frags = ((mm + 31) // 32) * ((kk + 15) // 16) * taps
                buf = torch.empty(frags * 2 * 64 * 16, dtype=torch.uint8, device=dev)
                packjobs[2 * i + tr] = (w.data_ptr(), buf.data_ptr(), self.amax[i:i + 1].data_ptr(), mm, kk,
                                        tr | (2 if taps == 1 else 0), len(pb2j))
                pb2j += [2 * i + tr] * ((frags * 64 + 255) // 256)
                pair.append(buf)
            self.wp.append(pair)
        to_dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(dev)
        self.tables = (to_dev(absjobs), torch.tensor(ab2j, dtype=torch.int32, device=dev), len(ab2j),
                       to_dev(packjobs), torch.tensor(pb2j, dtype=torch.int32, device=dev), len(pb2j))
        self.ptrs = tuple(m.weight.data_ptr() for m in self.convs)

    def refresh(self):
        from .. import _lib
        if not self.convs or not self.convs[0].weight.is_cuda or self.convs[0].weight.dtype != torch.float32:
            return
        key = tuple(m.weight._version for m in self.convs)
        ptrs = tuple(m.weight.data_ptr() for m in self.convs)
        if self.tables is None or ptrs != self.ptrs:
            self._build(self.convs[0].weight.device)
            self.key = None
        if key == self.key:
            return
        L = _lib.lib()
        aj, ab, an, pj, pb, pn = self.tables
        st = _stream(self.amax)
        self.amax.zero_()
        _lib.check(L.dcl_absmax_multi(_lib.ptr(aj), _lib.ptr(ab), an, st), "dcl_absmax_multi")
        _lib.check(L.dcl_conv3x3_pack_multi(_lib.ptr(pj), _lib.ptr(pb), pn, st), "dcl_conv3x3_pack_multi")
        for i, m in enumerate(self.convs):
            w = m.weight
            m._packed = ((w._version, w.data_ptr()), self.amax[i:i + 1], self.wp[i][0], self.wp[i][1])
        self.key = key


class LinearTagGroup:
    """absmax tags of all TokenLinear weights of a model by ONE launch per optimizer step (dcl_absmax_multi) instead of one
    small dcl_absmax launch per Linear; ``refresh()`` at the start of the model's forward does nothing while no weight
    was modified."""

    def __init__(self, module: torch.nn.Module):
        self.lins = [m for m in module.modules() if isinstance(m, TokenLinear)]
        self.key = None
        self.tables = None

    def _build(self, dev):
        import numpy as np
        jobs = np.zeros(len(self.lins), dtype=[("x", "<u8"), ("out", "<u8"), ("n", "<i8"), ("fb", "<i4"), ("pad", "<i4")])
        self.amax = torch.zeros(len(self.lins), dtype=torch.float32, device=dev)
        b2j = []
        for i, m in enumerate(self.lins):
            w = m.weight
            jobs[i] = (w.data_ptr(), self.amax[i:i + 1].data_ptr(), w.numel(), len(b2j), 0)
            b2j += [i] * ((w.numel() + 4095) // 4096)
        self.tables = (torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev),
                       torch.tensor(b2j, dtype=torch.int32, device=dev), len(b2j))
        self.ptrs = tuple(m.weight.data_ptr() for m in self.lins)

    def refresh(self):
        from .. import _lib
        from . import amax as _am
        if not self.lins or not self.lins[0].weight.is_cuda or self.lins[0].weight.dtype != torch.float32:
            return
        key = tuple(m.weight._version for m in self.lins)
        ptrs = tuple(m.weight.data_ptr() for m in self.lins)
        if self.tables is None or ptrs != self.ptrs:
            self._build(self.lins[0].weight.device)
            self.key = None
        if key == self.key:
            return
        jobs, b2j, nb = self.tables
        self.amax.zero_()
        _lib.check(_lib.lib().dcl_absmax_multi(_lib.ptr(jobs), _lib.ptr(b2j), nb, _stream(self.amax)), "dcl_absmax_multi")
        for i, m in enumerate(self.lins):
            _am.tag(m.weight, self.amax[i:i + 1])
        self.key = key


def use_direct_conv3x3(module: torch.nn.Module) -> torch.nn.Module:
    """Switch every plain nn.Conv2d with a 3x3 / stride 1 or 2 / pad 1 geometry to DirectConv2d in place."""
    for m in module.modules():
        if type(m) is torch.nn.Conv2d and m.kernel_size == (3, 3) and m.stride in ((1, 1), (2, 2)) \
                and m.padding == (1, 1) \
                and m.dilation == (1, 1) and m.groups == 1:
            m.__class__ = DirectConv2d
    return module


def use_direct_conv1x1(module: torch.nn.Module) -> torch.nn.Module:
    """Switch every plain (or GemmConv1x1) 1x1 / stride 1 / pad 0 / groups 1 nn.Conv2d to DirectConv2d in place: all
    three directions on the f16x3 kernels (one-tap mode of csrc/dcl_conv3x3.hip, k_wgrad1x1d)."""
    for m in module.modules():
        if type(m) in (torch.nn.Conv2d, GemmConv1x1) and m.kernel_size == (1, 1) and m.stride == (1, 1) \
                and m.padding == (0, 0) and m.dilation == (1, 1) and m.groups == 1:
            m.__class__ = DirectConv2d
    return module


# ---- head convolution over a concatenation of up-sampled maps, without the up-sampled maps ---------------------------

def _coarse_offsets(c0, channels):
    """First input channel of every coarse map (``channels``: their channel counts): ``c0`` is either the first one's (the
    maps follow each other in the weight) or a tuple with one offset per map."""
    if isinstance(c0, (tuple, list)):
        return [int(v) for v in c0]
    offs, off = [], int(c0)
    for c in channels:
        offs.append(off)
        off += c
    return offs


class _CoarseTaps(torch.autograd.Function):
    """addend [N, Co, H, W] = sum over the coarse maps x_b of conv3x3(up(x_b), weight[:, slice_b], padding=1), computed as
    z_b = W_b x_b (split-f16 GEMMs [9 Co, C_b] x [C_b, h w] at LOW resolution, dcl_gemm_f16x3) followed by the tap-wise
    bilinear gather of csrc/dcl_resize.hip (k_tapup_fwd); backward: the gather's adjoint (k_tapup_bwd), then two GEMMs per map
    (dx_b = W_b^T dz_b, dW_b = dz_b x_b^T).  ``weight`` is the FULL [Co, Cin, 3, 3] parameter, ``c0`` the first input channel
    of the first coarse map; its gradient comes back full-size (zero outside the coarse slices).  ``gemm = False`` (class
    switch): the library's fp32 GEMMs.

    Layout of the tap products (round 4, ``image_major``, default): z / dz are [N][9 Co][h w] and every GEMM is BATCHED over
    the images -- x_b and dx_b are used / produced as the NCHW tensors they are (no [C_b, N h w] transpose copies), and a
    256-row operand tile of dz spans 256 x 8 KiB instead of 256 x 96 KiB.  With the channel-major layout of round 3 ([9 Co][N h
    w]: ONE GEMM over all images) the two backward GEMMs of the 1/16-resolution map streamed their 637-MB operand at 0.66 TB/s
    -- 0.94 + 0.85 ms for 61 GFLOP each, 0.08 of the f16x3 roofline, against 0.3 for the same kernel on compact operands
    (profiles/r04_kernel_table_*.json: k_gemm rows)."""

    gemm = _dbg.gemm_head_taps
    image_major = _dbg.head_taps_image_major

    @staticmethod
    def forward(ctx, align, H, W, c0, weight, *ts):
        from .. import _lib
        from . import amax as _am
        L = _lib.lib()
        Co = weight.shape[0]
        n = ts[0].shape[0]
        y = torch.empty((n, Co, H, W), dtype=torch.float32, device=weight.device)
        st = _lib.stream_ptr(y.device)
        use_gemm = _CoarseTaps.gemm and all(t.shape[1] % 32 == 0 and (n * t.shape[2] * t.shape[3]) % 32 == 0 for t in ts)
        img = bool(_CoarseTaps.image_major and use_gemm and all((t.shape[2] * t.shape[3]) % 32 == 0 and t.is_contiguous()
                                                                for t in ts))
        wam = _am.amax_of(weight) if use_gemm else None
        offs = _coarse_offsets(c0, [t.shape[1] for t in ts])
        saved, zs, xams = [], [], []
        for t, off in zip(ts, offs):
            cb, h, w = t.shape[1:]
            hw = h * w
            P = n * hw
            xam = _am.amax_of(t) if use_gemm else None
            wb = weight[:, off:off + cb].permute(2, 3, 0, 1).reshape(9 * Co, cb)           # [(tap, co), ci]
            if img:
                # z[n] [9 Co, h w] = W_b x[n]: A = wb (k-major), B = x[n] [C_b, h w] (row-contiguous), one launch for all images
                xc = t
                z = torch.empty((n, 9 * Co, hw), dtype=torch.float32, device=y.device)
                gemm_f16x3(wb, True, cb, t, False, hw, 9 * Co, hw, cb, z, hw, wam, xam, batch=n,
                           strides=(0, cb * hw, 9 * Co * hw), splitk=1)
            else:
                xc = t.transpose(0, 1).reshape(cb, P)                                       # [C_b, N h w] (one copy)
                if use_gemm:
                    z = torch.empty((9 * Co, P), dtype=torch.float32, device=y.device)
                    gemm_f16x3(wb, True, cb, xc, False, P, 9 * Co, P, cb, z, P, wam, xam, splitk=1)
                else:
                    z = torch.mm(wb, xc)
            zs.append((z, h, w))
            saved += [xc, wb]
            xams.append(xam)
        for i in range(0, len(zs), 2):
            z0, h0, w0 = zs[i]
            z1, h1, w1 = zs[i + 1] if i + 1 < len(zs) else (None, 0, 0)
            _lib.check(L.dcl_tapup_fwd(_lib.ptr(z0), h0, w0, _lib.ptr(z1), h1, w1, n, Co, H, W, 1 if align else 0,
                                       0 if img else 1, _lib.ptr(y), 1 if i else 0, st), "dcl_tapup_fwd")
        ctx.save_for_backward(*saved)
        ctx.geom = (bool(align), H, W, c0, tuple(weight.shape), [tuple(t.shape) for t in ts])
        ctx.ams = (wam, xams) if use_gemm else None
        ctx.img = img
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _lib
        from . import amax as _am
        L = _lib.lib()
        align, H, W, c0, wshape, shapes = ctx.geom
        Co = wshape[0]
        img = ctx.img
        dy = dy.contiguous()
        st = _lib.stream_ptr(dy.device)
        gw = torch.zeros(wshape, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[4] else None
        dyam = _am.amax_of(dy) if ctx.ams is not None else None
        grads = []
        offs = _coarse_offsets(c0, [sh[1] for sh in shapes])
        for i, (n, cb, h, w) in enumerate(shapes):
            off = offs[i]
            xc, wb = ctx.saved_tensors[2 * i], ctx.saved_tensors[2 * i + 1]
            hw = h * w
            P = n * hw
            dz = torch.empty((n, 9 * Co, hw) if img else (9 * Co, P), dtype=torch.float32, device=dy.device)
            gx = None
            if ctx.ams is not None:
                wam, xams = ctx.ams
                # max|dz| measured by the gather's adjoint itself (round 4).  Until then the a-priori bound 4 s_y s_x max|dy| (the
                # bilinear weights of one source pixel sum to s_y s_x in the interior, < 2 s per axis at a clamped border) set the
                # operand scale of the two GEMMs below: 16-256 x the real maximum, 4-8 bits of the f16 split
                dzam = _am.zeros(1, dy.device)
                _lib.check(L.dcl_tapup_bwd_amax(_lib.ptr(dy), n, Co, H, W, h, w, 1 if align else 0, 0 if img else 1, _lib.ptr(dz),
                                                _lib.ptr(dzam), st), "dcl_tapup_bwd_amax")
            else:
                _lib.check(L.dcl_tapup_bwd(_lib.ptr(dy), n, Co, H, W, h, w, 1 if align else 0, 0 if img else 1, _lib.ptr(dz), st),
                           "dcl_tapup_bwd")
            if ctx.ams is not None:
                if img:
                    if ctx.needs_input_grad[5 + i]:
                        # dx[n] [C_b, h w] = W_b^T dz[n]: both operands row-contiguous (the contraction 9 Co may be ragged); the
                        # result IS the NCHW gradient
                        gx = torch.empty((n, cb, h, w), dtype=torch.float32, device=dy.device)
                        gemm_f16x3(wb, False, cb, dz, False, hw, cb, hw, 9 * Co, gx, hw, wam, dzam, batch=n,
                                   strides=(0, 9 * Co * hw, cb * hw), splitk=1)
                    if gw is not None:
                        # dW_b = sum_n dz[n] [9 Co, h w] x[n]^T: both k-major (the pixels), one slab per image, fixed-order sum
                        part = torch.empty((n, 9 * Co, cb), dtype=torch.float32, device=dy.device)
                        gemm_f16x3(dz, True, hw, xc, True, hw, 9 * Co, cb, hw, part, cb, dzam, xams[i], batch=n,
                                   strides=(9 * Co * hw, cb * hw, 9 * Co * cb), splitk=1)
                        gwb = part.sum(0) if n > 1 else part[0]
                        gw[:, off:off + cb] = gwb.view(3, 3, Co, cb).permute(2, 3, 0, 1)
                else:
                    if ctx.needs_input_grad[5 + i]:
                        gxc = torch.empty((cb, P), dtype=torch.float32, device=dy.device)
                        # dx_b [C_b, P] = W_b^T dz: both operands row-contiguous (the contraction 9 Co = 6480 may be ragged)
                        gemm_f16x3(wb, False, cb, dz, False, P, cb, P, 9 * Co, gxc, P, wam, dzam, splitk=_dbg.head_dx_splitk)
                        gx = gxc.view(cb, n, h, w).transpose(0, 1).contiguous()
                    if gw is not None:
                        gwb = torch.empty((9 * Co, cb), dtype=torch.float32, device=dy.device)
                        gemm_f16x3(dz, True, P, xc, True, P, 9 * Co, cb, P, gwb, cb, dzam, xams[i])
                        gw[:, off:off + cb] = gwb.view(3, 3, Co, cb).permute(2, 3, 0, 1)
            else:
                if ctx.needs_input_grad[5 + i]:
                    gx = torch.mm(wb.t(), dz).view(cb, n, h, w).transpose(0, 1).contiguous()
                if gw is not None:
                    gw[:, off:off + cb] = torch.mm(dz, xc.t()).view(3, 3, Co, cb).permute(2, 3, 0, 1)
            grads.append(gx)
        return (None, None, None, None, gw, *grads)


class _Conv3x3Addend(torch.autograd.Function):
    """y = conv2d(x, weight, bias, padding=1) + addend on the direct f16x3 kernels with an explicit weight tensor (a slice of a
    module's parameter): forward / data gradient through csrc/dcl_conv3x3.hip (the addend and the bias enter as the
    accumulators' start values), weight gradient through csrc/dcl_wgrad3x3d.hip; the addend's gradient is the output's."""

    @staticmethod
    def forward(ctx, x, weight, bias, addend):
        from .amax import amax_of
        w = weight.contiguous()
        wamax = amax_of(w)
        out = torch.empty((x.shape[0], w.shape[0], x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
        conv3x3_launch(x, conv3x3_pack(w, wamax), w.shape[0], amax_of(x), wamax, out, addend=addend, bias=bias)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, gy):
        from .amax import amax_of
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None

        def dgrad():
            wamax = amax_of(w)
            g = torch.empty_like(x)
            conv3x3_launch(gy, conv3x3_pack(w, wamax, True), w.shape[1], amax_of(gy), wamax, g)
            return g

        def wgrad():
            if conv3x3_wgrad_supported(x, w.shape[0]):
                return conv3x3_wgrad(x, gy)
            return torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])[1]

        if getattr(ctx, "wgrad_first", False):      # (_HeadSplit: the weight gradient leaves CUs free for the side stream)
            gw = wgrad() if ctx.needs_input_grad[1] else None
            gx = dgrad() if ctx.needs_input_grad[0] else None
        else:
            gx = dgrad() if ctx.needs_input_grad[0] else None
            gw = wgrad() if ctx.needs_input_grad[1] else None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb, (gy if ctx.needs_input_grad[3] else None)


class _ShimCtx:
    """Stand-in for an autograd context when one Function's forward / backward bodies are composed inside another."""

    def __init__(self, needs=()):
        self.needs_input_grad = tuple(needs)
        self.saved_tensors = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts


_HEAD_SIDE = {}


def _head_side_stream(device):
    key = (device.type, device.index)
    st = _HEAD_SIDE.get(key)
    if st is None:
        st = _HEAD_SIDE[key] = torch.cuda.Stream(device=device)
    return st


class _HeadSplit(torch.autograd.Function):
    """_CoarseTaps + _Conv3x3Addend as ONE autograd node, so that the backward can run the coarse maps' half (two tap-gather
    adjoints, four GEMMs: ~2.6 ms) on a side stream NEXT TO the fine part's (data gradient + the 135-workgroup weight
    gradient that leaves 121 CUs idle for 2.9 ms); as two nodes the engine orders the second behind everything the first
    enqueued.  Both halves only read the incoming gradient.  ``weight`` is the full parameter; its gradient is assembled here
    (fine slice + coarse slices) instead of by autograd's slice / add nodes."""

    overlap = _dbg.head_overlap        # 0 off, 1 on, 2 on with the fine part's weight gradient first

    @staticmethod
    def forward(ctx, align, H, W, layout, hi, weight, bias, *coarse):
        fine_ranges, coarse_offs = layout
        cctx, fctx = _ShimCtx(), _ShimCtx()
        addend = _CoarseTaps.forward(cctx, align, H, W, tuple(coarse_offs), weight, *coarse)
        w_f = weight[:, fine_ranges[0][0]:fine_ranges[0][1]] if len(fine_ranges) == 1 else \
            torch.cat([weight[:, a:b] for a, b in fine_ranges], 1)
        out = _Conv3x3Addend.forward(fctx, hi, w_f, bias, addend)
        ctx.save_for_backward(*cctx.saved_tensors, *fctx.saved_tensors)
        ctx.nc = len(cctx.saved_tensors)
        ctx.c = (cctx.geom, cctx.ams, cctx.img)
        ctx.f = fctx.has_bias
        ctx.fine_ranges = tuple(fine_ranges)
        return out

    @staticmethod
    def backward(ctx, gy):
        need = ctx.needs_input_grad
        cctx = _ShimCtx((False,) * 4 + (need[5],) + tuple(need[7:]))
        cctx.saved_tensors = ctx.saved_tensors[:ctx.nc]
        cctx.geom, cctx.ams, cctx.img = ctx.c
        fctx = _ShimCtx((need[4], need[5], need[6], False))
        fctx.saved_tensors = ctx.saved_tensors[ctx.nc:]
        fctx.has_bias = ctx.f
        gy = gy.contiguous()
        if _HeadSplit.overlap and gy.is_cuda:
            from . import amax as _am
            _am.amax_of(gy)                                  # (the tag both halves read: computed once, on this stream)
            main = torch.cuda.current_stream(gy.device)
            side = _head_side_stream(gy.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                rc = _CoarseTaps.backward(cctx, gy)
            _am.record_stream(gy, side)
            fctx.wgrad_first = _HeadSplit.overlap == 2
            rf = _Conv3x3Addend.backward(fctx, gy)
            main.wait_stream(side)
            for t in rc:
                if t is not None:
                    t.record_stream(main)
        else:
            rc = _CoarseTaps.backward(cctx, gy)
            rf = _Conv3x3Addend.backward(fctx, gy)
        gw = rc[4]
        if gw is not None and rf[1] is not None:
            off = 0
            for a, b in ctx.fine_ranges:
                gw[:, a:b] = rf[1][:, off:off + b - a]
                off += b - a
        return (None, None, None, None, rf[0], gw, rf[2], *rc[5:])


class LazyConcat:
    """``torch.cat([ts[0]] + [interpolate(t, ts[0] size) for t in ts[1:]], 1)`` that has not been formed (reference
    models/HRNet.py:549-553): the head convolution of this repo consumes the parts (``conv3x3_over_upsampled``); anything
    else calls ``materialize()``."""

    def __init__(self, ts, align_corners):
        self.ts, self.align_corners = list(ts), bool(align_corners)
        self._full = None

    @property
    def shape(self):
        t0 = self.ts[0]
        return torch.Size((t0.shape[0], sum(t.shape[1] for t in self.ts)) + tuple(t0.shape[2:]))

    def materialize(self):
        if self._full is None:
            self._full = upsample_concat(self.ts, self.align_corners)
        return self._full


HEAD_SPLIT_MIN_SCALE = _dbg.head_split_min_scale


def conv3x3_over_upsampled(ts, align_corners, weight, bias, min_scale=None):
    """``conv2d(cat([ts[0]] + [up(t) for t in ts[1:]], 1), weight, bias, padding=1)`` (bilinear ``up`` to ts[0]'s size)
    without up-sampling the maps that are at least ``min_scale`` times coarser than ts[0]:

        conv3x3(up(x), W) = sum_tap (shift_tap . up)(W_tap x)        (the convolution acts on channels, up on pixels)

    so their channel products are 1x1 convolutions at LOW resolution (library fp32 GEMMs producing 9 * Co maps per source)
    and the rest is the tap-wise bilinear gather of ``_CoarseTaps``; the finer maps are concatenated and convolved directly, with
    the gathered sum as the convolution's addend.  For HRNet-W48's head (48 + 96 + 192 + 384 channels at scales 1, 2, 4,
    8) 80 % of the multiply-adds move to 1/16 and 1/64 of the pixels; UPerNet's fusion convolution (P2, P5, P4, P3 -- the maps
    may come in any order, ts[0] is the full-resolution one) 47 %.  Equal to the reference formulation up to fp32
    round-off (tests/test_hip_parity.py::test_head_conv_over_upsampled_matches_fp64)."""
    t0 = ts[0]
    n, _, H, W = t0.shape
    offs, off = [], 0
    for t in ts:
        offs.append(off)
        off += t.shape[1]
    if min_scale is None:
        min_scale = HEAD_SPLIT_MIN_SCALE

    def goes_coarse(t):
        # >= 4x coarser: always.  2x coarser: the tap products of such a map are 9/4 of an output map per output channel and
        # cross HBM five times per step (written and read forward, written and read twice backward) -- worth it for a wide map
        # (UPerNet's 512-channel P3: config 4 79.1 -> 74.3 ms, config 5 218.3 -> 210.5), not for HRNet's 96-channel branch
        # (89.4 -> 95.2 ms: 2.5 GB of tap products for a third of the fine convolution's work).  min_scale forces either rule.
        if min_scale:
            return t.shape[-1] * min_scale <= W
        return t.shape[-1] * 4 <= W or (t.shape[-1] * 2 <= W and t.shape[1] >= 256)
    is_fine = [t is t0 or not goes_coarse(t) for t in ts]
    fine = [t for t, f in zip(ts, is_fine) if f]
    fine_ranges = tuple((o, o + t.shape[1]) for t, o, f in zip(ts, offs, is_fine) if f)
    # coarse maps coarsest first: the tap gather takes them in pairs, and a map only 2x coarser (min_scale = 2) fills a forward
    # tile's LDS window by itself -- it goes last, alone or behind a small one
    order = sorted((i for i, f in enumerate(is_fine) if not f), key=lambda i: ts[i].shape[-1])
    coarse = [ts[i] for i in order]
    coarse_offs = tuple(offs[i] for i in order)
    if coarse:
        # the tap gather's tiles are sized by LDS: ask the library BEFORE committing to the split form (it would otherwise
        # raise mid-step, for some shapes only in the backward) and convolve the materialised concatenation instead
        from .. import _lib
        L = _lib.lib()
        for i in range(0, len(coarse), 2):
            a = coarse[i]
            b = coarse[i + 1] if i + 1 < len(coarse) else None
            if not L.dcl_tapup_supported(a.shape[2], a.shape[3], b.shape[2] if b is not None else 0,
                                         b.shape[3] if b is not None else 0, H, W, 1 if align_corners else 0):
                x = LazyConcat(list(ts), align_corners).materialize()
                return _Conv3x3Addend.apply(x, weight, bias, None)
    hi = upsample_concat(fine, align_corners) if len(fine) > 1 else t0
    if not coarse:
        return _Conv3x3Addend.apply(hi, weight, bias, None)
    return _HeadSplit.apply(bool(align_corners), H, W, (fine_ranges, coarse_offs), hi, weight, bias, *coarse)


# ---- 1x1 convolutions as plain batched GEMMs ----------------------------------------------------------------------

class _Conv1x1Gemm(torch.autograd.Function):
    """1x1 / stride 1 convolution on NCHW as batched f32 GEMMs in all three directions.  The library's own weight
    gradient for this case goes through an NHWC implicit-GEMM kernel with layout transposes around it (7 ms per
    HRNet-W48 step); in NCHW it is simply dW = sum_n gy_n [Co, HW] @ x_n^T [HW, Ci]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        n, ci, h, w = x.shape
        co = weight.shape[0]
        y = torch.empty((n, co, h, w), dtype=x.dtype, device=x.device)       # returned as a base tensor, not a view:
        if _conv1x1_by_gemm(co, ci, x, False):                                # callers relu_() it
            from .amax import amax_of
            conv1x1_gemm(x, weight.view(co, ci), y, amax_of(x), amax_of(weight))
        else:
            torch.matmul(weight.view(co, ci), x.view(n, ci, h * w), out=y.view(n, co, h * w))
        if bias is not None:
            y += bias.view(1, co, 1, 1)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        n, ci, h, w = x.shape
        co = weight.shape[0]
        gy = gy.contiguous()
        g2 = gy.view(n, co, h * w)
        gx = gw = gb = None
        from .amax import amax_of
        hw = h * w
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            if _conv1x1_by_gemm(ci, co, gy, True):
                conv1x1_gemm(gy, weight.view(co, ci), gx, amax_of(gy), amax_of(weight), transposed=True)
            else:
                torch.matmul(weight.view(co, ci).t(), g2, out=gx.view(n, ci, h * w))
        if ctx.needs_input_grad[1]:
            if GEMM_CONV1X1 and hw % 32 == 0 and co >= 64 and ci >= 64 and ci % 4 == 0 and co * hw * 4 < (1 << 32) \
                    and ci * hw * 4 < (1 << 32):
                # dW = sum_n gy_n [Co, HW] x_n^T: one batched GEMM (both operands k-major: the pixel axis), k-split slabs
                # per image, then the fixed-order sum over the images
                part = torch.empty((n, co, ci), dtype=torch.float32, device=x.device)
                gemm_f16x3(gy, True, hw, x, True, hw, co, ci, hw, part, ci, amax_of(gy), amax_of(x), batch=n,
                           strides=(co * hw, ci * hw, co * ci))
                gw = (part.sum(0) if n > 1 else part[0]).view_as(weight)
            else:
                gw = torch.bmm(g2, x.view(n, ci, h * w).transpose(1, 2)).sum(0).view_as(weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum((0, 2))
        return gx, gw, gb


class _Conv1x1FromTokens(torch.autograd.Function):
    """1x1 convolution (no bias) of a map given TOKEN-MAJOR, result NCHW: y[b] [Co, HW] = W [Co, C] . tok[b]^T with tok [B, HW, C]
    -- a Swin stage output as its LayerNorm wrote it (models/Swin.TokenMap) feeding a lateral convolution of the UPerNet decoder
    (reference models/Swin.py:452-455 permute + contiguous, then models/UPerNet.py:88-92 fpn_in).  All three products are batched
    split-f16 GEMMs on the operands as they lie: forward (W k-major, tokens k-major), data gradient dtok[b] [HW, C] = dy[b]^T W
    (both row-contiguous; written token-major, i.e. contiguous for the LayerNorm's backward), weight gradient dW = sum_b dy[b] tok[b]
    (dy k-major over the pixels, tokens row-contiguous).  No layout copy in either direction."""

    @staticmethod
    def forward(ctx, tok, weight, H, W):
        from . import amax as _am
        b, hw, c = tok.shape
        co = weight.shape[0]
        y = torch.empty((b, co, H, W), dtype=torch.float32, device=tok.device)
        ca = _am.zeros(1, tok.device)
        gemm_f16x3(weight, True, c, tok, True, c, co, hw, c, y, hw, _am.amax_of(weight), _am.amax_of(tok), batch=b,
                   strides=(0, hw * c, co * hw), c_amax=ca)
        _am.tag(y, ca)
        ctx.save_for_backward(tok, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import amax as _am
        tok, weight = ctx.saved_tensors
        b, hw, c = tok.shape
        co = weight.shape[0]
        g = _am.carry(gy, gy.contiguous())
        gt = gw = None
        if ctx.needs_input_grad[0]:
            gt = torch.empty_like(tok)
            cg = _am.zeros(1, tok.device)
            gemm_f16x3(g, False, hw, weight, False, c, hw, c, co, gt, c, _am.amax_of(g), _am.amax_of(weight), batch=b,
                       strides=(co * hw, 0, hw * c), c_amax=cg)
            _am.tag(gt, cg)
        if ctx.needs_input_grad[1]:
            part = torch.empty((b, co, c), dtype=torch.float32, device=tok.device)
            gemm_f16x3(g, True, hw, tok, False, c, co, c, hw, part, c, _am.amax_of(g), _am.amax_of(tok), batch=b,
                       strides=(co * hw, hw * c, co * c))
            gw = (part.sum(0) if b > 1 else part[0]).view_as(weight)
        return gt, gw, None, None


def conv1x1_from_tokens_ok(tok, conv, H, W):
    if not (GEMM_CONV1X1 and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and tok.is_cuda and tok.dtype == torch.float32 and tok.dim() == 3 and tok.is_contiguous()
            and conv.weight.dtype == torch.float32 and torch.is_grad_enabled() and not torch.is_autocast_enabled()):
        return False
    b, hw, c = tok.shape
    co = conv.out_channels
    return (hw == H * W and c == conv.in_channels and c % 32 == 0 and hw % 32 == 0 and co % 4 == 0
            and max(co, c) * hw * 4 < (1 << 32) and gemm_supported(co, hw, c, c, True, c, True)
            and gemm_supported(hw, c, co, hw, False, c, False) and gemm_supported(co, c, hw, hw, True, c, False))


def conv1x1_from_tokens(tok, conv, H, W):
    """conv(tokens as an NCHW map) -> [B, Co, H, W]; see _Conv1x1FromTokens."""
    return _Conv1x1FromTokens.apply(tok, conv.weight, int(H), int(W))


class _FeatureDropoutConv1x1(torch.autograd.Function):
    """conv1x1(dropout2d(x)) without the two passes over x: Dropout2d multiplies whole channels by a per-(sample, channel) factor
    m (0 or 1 / (1 - p)), and a 1x1 convolution is linear in its input channels, so y_n = (W . diag(m_n)) x_n -- the factor moves
    into a per-sample copy of the (tiny) weight matrix.  Backward: dx_n = (W diag(m_n))^T dy_n, dW = sum_n (dy_n x_n^T) diag(m_n).
    UPerNet's classifier (reference models/UPerNet.py:66-68: conv3x3 block -> Dropout2d -> conv1x1) on 16 x 512 x 160 x 160:
    2 x 3 passes over 839 MB less per step.  ``noise`` is the [N, C] factor tensor, drawn by the caller exactly as
    ``F.dropout2d`` draws it (same generator consumption)."""

    @staticmethod
    def forward(ctx, x, weight, bias, noise):
        n, ci, h, w = x.shape
        co = weight.shape[0]
        wb = weight.view(1, co, ci) * noise.view(n, 1, ci)
        y = torch.empty((n, co, h, w), dtype=x.dtype, device=x.device)
        torch.bmm(wb, x.view(n, ci, h * w), out=y.view(n, co, h * w))
        if bias is not None:
            y += bias.view(1, co, 1, 1)
        ctx.save_for_backward(x, wb, noise)
        ctx.has_bias = bias is not None
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wb, noise = ctx.saved_tensors
        n, ci, h, w = x.shape
        co = wb.shape[1]
        g2 = gy.contiguous().view(n, co, h * w)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            torch.bmm(wb.transpose(1, 2), g2, out=gx.view(n, ci, h * w))
        if ctx.needs_input_grad[1]:
            hw = h * w
            if GEMM_CONV1X1 and hw % 32 == 0 and co >= 64 and ci >= 64 and ci % 4 == 0 and co * hw * 4 < (1 << 32) \
                    and ci * hw * 4 < (1 << 32):
                from .amax import amax_of                       # (as _Conv1x1Gemm.backward: per-image products on the split-f16 GEMM;
                part = torch.empty((n, co, ci), dtype=torch.float32, device=x.device)   # the library runs this shape at 10-40 TFLOP/s)
                gy_c = g2.view(n, co, h, w)
                gemm_f16x3(gy_c, True, hw, x, True, hw, co, ci, hw, part, ci, amax_of(gy_c), amax_of(x), batch=n,
                           strides=(co * hw, ci * hw, co * ci))
            else:
                part = torch.bmm(g2, x.view(n, ci, hw).transpose(1, 2))              # [n, co, ci]
            gw = (part * noise.view(n, 1, ci)).sum(0).view(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum((0, 2))
        return gx, gw, gb, None


def dropout2d_conv1x1(x, drop, conv):
    """``conv(drop(x))`` for nn.Dropout2d followed by a plain 1x1 convolution; in training with p > 0 on contiguous fp32 CUDA
    maps the channel factors ride in per-sample weights (_FeatureDropoutConv1x1), otherwise the two modules run as they are."""
    if (drop.training and 0.0 < drop.p < 1.0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32
            and torch.is_grad_enabled() and not torch.is_autocast_enabled() and _dbg.fold_dropout2d):
        n, c = x.shape[:2]
        # F.dropout2d's draw: noise = empty([N, C, 1, 1]).bernoulli_(1 - p).div_(1 - p)
        noise = x.new_empty((n, c, 1, 1)).bernoulli_(1.0 - drop.p).div_(1.0 - drop.p)
        return _FeatureDropoutConv1x1.apply(x, conv.weight, conv.bias, noise.view(n, c))
    return conv(drop(x))


class _Conv1x1ToNHWC(torch.autograd.Function):
    """The projection heads' last 1x1 convolution (reference models/Projector.py:56-63) with its output written PIXEL-MAJOR:
    y[n, pix, :] = W x[n, :, pix] + b as one batched split-f16 GEMM per direction.  The result is handed out as the reference's
    [n, d, h, w] tensor with channels-last strides (same shape, same values), so that the contrastive loss -- the only reader of
    the embedding (reference losses/DenseContrastiveLossV2.py:113-124) -- gathers ONE contiguous 1-KiB row per sampled pixel
    (K3) and scatters one per pixel in the backward (K6) instead of 256 strided 4-byte accesses (K3 FETCH_SIZE on the NCHW map:
    294 MB per launch for 10 MB of rows, profiles/r03_loss_pmc_fetch.csv).  Backward: the gradient arrives in the same layout
    (K6 writes the strides it was given); dx and dW are GEMMs over it, no layout copy in either direction."""

    @staticmethod
    def forward(ctx, x, weight, bias, wamax):
        from .amax import amax_of
        n, ci, h, w = x.shape
        co = weight.shape[0]
        hw = h * w
        y = torch.empty((n, h, w, co), dtype=torch.float32, device=x.device)
        # both operands with the contraction index (the input channels) OUTERMOST: a ragged K (48 channels = 1.5 k-steps) is
        # only legal for row-contiguous operands, so the (tiny) weight goes in transposed, [ci, co]
        wt = weight.t().contiguous()
        gemm_f16x3(x, False, hw, wt, False, co, hw, co, ci, y, co, amax_of(x), wamax, bias=bias, batch=n,
                   strides=(ci * hw, 0, hw * co), splitk=1)
        ctx.save_for_backward(x, weight, wamax)
        ctx.has_bias = bias is not None
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        from .amax import amax_of, carry
        x, weight, wamax = ctx.saved_tensors
        n, ci, h, w = x.shape
        co = weight.shape[0]
        hw = h * w
        g = carry(gy, gy.permute(0, 2, 3, 1))
        if not g.is_contiguous():
            g = g.contiguous()
        ga = amax_of(g)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            gemm_f16x3(weight, False, ci, g, True, co, ci, hw, co, gx, hw, wamax, ga, batch=n,
                       strides=(0, hw * co, ci * hw), splitk=1)
        if ctx.needs_input_grad[1]:
            part = torch.empty((n, co, ci), dtype=torch.float32, device=x.device)
            gemm_f16x3(g, False, co, x, True, hw, co, ci, hw, part, ci, ga, amax_of(x), batch=n,
                       strides=(hw * co, ci * hw, co * ci))
            gw = (part.sum(0) if n > 1 else part[0]).view_as(weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.reshape(-1, co).sum(0)
        return gx, gw, gb, None


def conv1x1_nhwc_supported(x, conv):
    """True when ``conv1x1_to_nhwc`` applies: a plain 1x1 convolution on a contiguous fp32 CUDA map whose three GEMM shapes the
    split-f16 kernel takes."""
    if not (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and conv.weight.dtype == torch.float32 and not torch.is_autocast_enabled()):
        return False
    n, ci, h, w = x.shape
    co, hw = conv.out_channels, h * w
    if max(ci, co) * hw * 4 >= (1 << 32):
        return False
    return (gemm_supported(hw, co, ci, hw, False, co, False) and gemm_supported(ci, hw, co, ci, False, co, True)
            and gemm_supported(co, ci, hw, co, False, hw, True))


def conv1x1_to_nhwc(x, conv):
    """conv(x) as an [n, d, h, w] tensor with channels-last strides (see _Conv1x1ToNHWC)."""
    if isinstance(conv, DirectConv2d):
        wamax = conv.packed_weights()[0]
    else:
        from .amax import amax_of
        wamax = amax_of(conv.weight.detach())
    return _Conv1x1ToNHWC.apply(x, conv.weight.view(conv.out_channels, -1), conv.bias, wamax)


class GemmConv1x1(torch.nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict) whose 1x1 / stride 1 / unpadded case runs as batched GEMMs for
    contiguous fp32 CUDA inputs in training; anything else falls through to nn.Conv2d.forward."""

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
                and self.weight.dtype == torch.float32 and not torch.is_autocast_enabled()
                and torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
            return _Conv1x1Gemm.apply(x, self.weight, self.bias)
        return super().forward(x)


def use_gemm_conv1x1(module: torch.nn.Module) -> torch.nn.Module:
    """Switch every plain 1x1 / stride 1 / pad 0 / groups 1 nn.Conv2d of a module tree to GemmConv1x1 in place."""
    for m in module.modules():
        if type(m) is torch.nn.Conv2d and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0) \
                and m.dilation == (1, 1) and m.groups == 1:
            m.__class__ = GemmConv1x1
    return module


# ---- token-major Linear (Swin: models/Swin.py qkv / proj / fc1 / fc2 / reduction) --------------------------------------

# ---- split-f16 GEMM (csrc/dcl_gemm.hip) ---------------------------------------------------------------------------

def gemm_supported(M, N, K, lda, a_kmajor, ldb, b_kmajor):
    from .. import _lib
    return bool(_lib.lib().dcl_gemm_supported(M, N, K, lda, int(a_kmajor), ldb, int(b_kmajor)))


def gemm_f16x3(a, a_kmajor, lda, b, b_kmajor, ldb, M, N, K, out, ldc, a_amax, b_amax, bias=None, batch=1,
               strides=(0, 0, 0), accumulate=False, c_amax=None, splitk=0, a_rowsum=None):
    """out[b][m][n] (+)= bias[n] + sum_k A[b](m, k) B[b](n, k) on dcl_gemm_f16x3 (fp32-equivalent split-f16 MFMA).

    ``a`` / ``b`` are the tensors whose storage holds the operands (used for their data pointers); X_kmajor says whether
    element (row, k) sits at X[row * ldx + k] (True) or X[k * ldx + row] (False).  ``a_amax`` / ``b_amax``: 1-D float
    tensors whose maxima bound max|A| / max|B| (models.amax.amax_of).  splitk = 0: the library's suggestion."""
    from .. import _lib
    L = _lib.lib()
    if splitk == 0:
        splitk = L.dcl_gemm_suggest_splitk(M, N, K, batch)
    ws = None
    if splitk > 1:
        ws = torch.empty(L.dcl_gemm_workspace_floats(M, N, batch, splitk), dtype=torch.float32, device=out.device)
    p = _lib.ptr
    _lib.check(L.dcl_gemm_f16x3(p(a), lda, int(a_kmajor), strides[0], p(b), ldb, int(b_kmajor), strides[1], M, N, K, batch,
                                p(a_amax), a_amax.numel(), p(b_amax), b_amax.numel(), p(bias), p(out), ldc, strides[2],
                                int(accumulate), p(c_amax), splitk, p(ws), p(a_rowsum), _stream(out)), "dcl_gemm_f16x3")
    return out


def gemm_f16x3_ascaled(a, a_kmajor, lda, b, ldb, M, N, K, out, a_amax, b_amax, a_scale, group, c_amax=None, a_rowsum=None,
                       ep=0, aux=None):
    """dcl_gemm_f16x3_ascaled: out [M, N] = (A with a per-token factor) . B, B row-contiguous ([K, N] read as its transpose);
    a_scale [tokens / group] multiplies row t of a k-major A or k row t of a row-contiguous A (see include/dcl_hip.h)."""
    from .. import _lib
    L = _lib.lib()
    splitk = 1 if ep else L.dcl_gemm_suggest_splitk(M, N, K, 1)
    ws = torch.empty(L.dcl_gemm_workspace_floats(M, N, 1, splitk), dtype=torch.float32, device=out.device) if splitk > 1 else None
    p = _lib.ptr
    _lib.check(L.dcl_gemm_f16x3_ascaled(p(a), lda, int(a_kmajor), p(b), ldb, 0, M, N, K, p(a_amax), a_amax.numel(), p(b_amax),
                                        b_amax.numel(), p(out), N, p(c_amax), splitk, p(ws), p(a_rowsum), p(a_scale), int(group),
                                        int(ep), p(aux), _stream(out)), "dcl_gemm_f16x3_ascaled")
    return out


def linear_f16x3(x2, weight, bias=None, tag_out=True):
    """y [M, N] = x2 [M, K] weight[N, K]^T + bias (the forward of nn.Linear on contiguous rows)."""
    from . import amax as _am
    m, k = x2.shape
    n = weight.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device=x2.device)
    ca = _am.zeros(1, x2.device) if tag_out else None
    gemm_f16x3(x2, True, k, weight, True, k, m, n, k, y, n, _am.amax_of(x2), _am.amax_of(weight), bias=bias, c_amax=ca)
    if tag_out:
        _am.tag(y, ca)
    return y


def linear_dgrad_f16x3(gy2, weight, scale=None, group=1):
    """dx [M, K] = (gy2 [M, N], rows scaled by scale[row // group]) weight[N, K]."""
    from . import amax as _am
    m, n = gy2.shape
    k = weight.shape[1]
    gx = torch.empty((m, k), dtype=torch.float32, device=gy2.device)
    ca = _am.zeros(1, gy2.device)
    if scale is not None:
        gemm_f16x3_ascaled(gy2, True, n, weight, k, m, k, n, gx, _am.amax_of(gy2), _am.amax_of(weight), scale, group, c_amax=ca)
    else:
        gemm_f16x3(gy2, True, n, weight, False, k, m, k, n, gx, k, _am.amax_of(gy2), _am.amax_of(weight), c_amax=ca)
    return _am.tag(gx, ca)


def linear_wgrad_f16x3(gy2, x2, want_bias=False, scale=None, group=1):
    """dW [N, K] = gy2 [M, N]^T x2 [M, K] (contraction over the M rows, k-split slabs summed in fixed order); with
    ``want_bias`` also db [N] = the column sums of gy2, accumulated by the threads that stage the dy^T operand (no extra
    pass over gy2): returns (dW, db)."""
    from . import amax as _am
    m, n = gy2.shape
    k = x2.shape[1]
    gw = torch.empty((n, k), dtype=torch.float32, device=gy2.device)
    gb = torch.empty((n,), dtype=torch.float32, device=gy2.device) if want_bias else None
    if scale is not None:       # rows of gy2 (the contraction index here) scaled by scale[row // group], the bias gradient too
        gemm_f16x3_ascaled(gy2, False, n, x2, k, n, k, m, gw, _am.amax_of(gy2), _am.amax_of(x2), scale, group, a_rowsum=gb)
    else:
        gemm_f16x3(gy2, False, n, x2, False, k, n, k, m, gw, k, _am.amax_of(gy2), _am.amax_of(x2), a_rowsum=gb)
    return (gw, gb) if want_bias else gw


class _TokenLinear(torch.autograd.Function):
    """y = x W^T + b on [tokens, K] rows: forward, data gradient and weight gradient on dcl_gemm_f16x3.

    The library's fp32 GEMMs run these shapes at 60-110 TFLOP/s (forward / data gradient) and 10-40 TFLOP/s (dW = dY^T X
    reduces over 10^4..10^5 tokens into a tiny [N, K] output: one macro tile per output tile, the whole token axis
    serial); the split-f16 kernel reaches 190-360 with errors below the library's (tools/gemm_shapes.py), the weight
    gradient as k-split slabs summed in fixed order (deterministic).  Operand scales come from absmax tags: the GEMM's own
    epilogue tags its result, LayerNorm tags its output, GELU hands its input's bound through; anything else costs
    one dcl_absmax pass."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import amax as _am
        n, k = weight.shape
        x2 = _am.carry(x, x.reshape(-1, k))                 # (a view is a new tensor object: the tag travels along)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        y = linear_f16x3(x2, weight, bias)
        return _am.carry(y, y.view(*x.shape[:-1], n))

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        n, k = weight.shape
        from . import amax as _am
        gy2 = _am.carry(gy, gy.reshape(-1, n))
        if not gy2.is_contiguous():
            gy2 = _am.carry(gy2, gy2.contiguous())
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            g = linear_dgrad_f16x3(gy2, weight)
            gx = _am.carry(g, g.view(ctx.xshape))
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_gb:
                gw, gb = linear_wgrad_f16x3(gy2, x2, want_bias=True)
            else:
                gw = linear_wgrad_f16x3(gy2, x2)
        elif want_gb:
            gb = gy2.sum(0)
        return gx, gw, gb


def gemm_f16x3_ep(a, b, b_kmajor, M, N, K, out, a_amax, b_amax, ep, bias=None, c_amax=None, out2=None, aux=None,
                  rowscale=None, rows_per_scale=1):
    """dcl_gemm_f16x3_ep: out [M, N] = epilogue(a [M, K] (k-major rows) . b (+ bias)); b is W [N, K] (b_kmajor) or W [K, N] read as
    its transpose.  ep 1: out = v, out2 = gelu(v); ep 2: out = v * gelu'(aux); ep 3: out = aux + rowscale[row // rows_per_scale] * v."""
    from .. import _lib
    p = _lib.ptr
    _lib.check(_lib.lib().dcl_gemm_f16x3_ep(p(a), K, 1, p(b), K if b_kmajor else N, int(b_kmajor), M, N, K, p(a_amax),
                                            a_amax.numel(), p(b_amax), b_amax.numel(), p(bias), p(out), N, p(c_amax), int(ep),
                                            p(out2), p(aux), p(rowscale), int(rows_per_scale), _stream(out)), "dcl_gemm_f16x3_ep")
    return out


def _ep_gemm_ok(m, n, k):
    """The fused-epilogue entry takes one pass over the contraction (no k-split) and 32-bit element offsets into C."""
    from .. import _lib
    return m * n < (1 << 30) and _lib.lib().dcl_gemm_suggest_splitk(m, n, k, 1) == 1


class _TokenLinearResidual(torch.autograd.Function):
    """shortcut + scale * (x W^T + b) in ONE launch: the residual sum of a Swin block's attention half (reference
    models/Swin.py:318, shortcut + drop_path(proj(...))) rides in the projection GEMM's epilogue.  scale: [B] per-sample DropPath
    factors (mask / keep) or None; bound = 1 / keep bounds |scale|."""

    @staticmethod
    def forward(ctx, x, weight, bias, shortcut, scale, bound):
        from . import amax as _am
        n, k = weight.shape
        x2 = _am.carry(x, x.reshape(-1, k))
        m = x2.shape[0]
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        ca = _am.zeros(1, x.device)
        gemm_f16x3_ep(x2, weight, True, m, n, k, y, _am.amax_of(x2), _am.amax_of(weight), 3, bias=bias, c_amax=ca,
                      aux=shortcut, rowscale=scale, rows_per_scale=m // scale.numel() if scale is not None else 1)
        _am.tag(y, ca)
        ctx.save_for_backward(x2, weight, scale)
        ctx.has_bias = bias is not None
        ctx.xshape, ctx.bound = x.shape, float(bound)
        return _am.carry(y, y.view(*x.shape[:-1], n))

    @staticmethod
    def backward(ctx, gy):
        from . import amax as _am
        x2, weight, scale = ctx.saved_tensors
        n, k = weight.shape
        gb2, sc, grp = _branch_grad(gy, n, scale, ctx.bound)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            g = linear_dgrad_f16x3(gb2, weight, sc, grp)
            gx = _am.carry(g, g.view(ctx.xshape))
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_gb:
                gw, gb = linear_wgrad_f16x3(gb2, x2, want_bias=True, scale=sc, group=grp)
            else:
                gw = linear_wgrad_f16x3(gb2, x2, scale=sc, group=grp)
        elif want_gb:
            gb = _scaled_rows(gy, n, scale, ctx.bound).sum(0)
        return gx, gw, gb, (gy if ctx.needs_input_grad[3] else None), None, None


ASCALE_IN_GEMM = _dbg.gemm_ascale     # the per-sample factor of a branch's gradient as an operand scale of the backward GEMMs


def _branch_grad(gy, n, scale, bound):
    """(rows, scale, group) for the backward GEMMs of a branch whose output was scaled per sample: the incoming gradient as
    contiguous [M, n] rows UNSCALED plus the factors for dcl_gemm_f16x3_ascaled when the GEMMs can apply them (whole k-steps of 32
    tokens per sample, factors that keep the scaled operand inside the f16 split's range), else the scaled rows and no factors."""
    from . import amax as _am
    if scale is None:
        return _scaled_rows(gy, n, None, bound), None, 1
    g2 = _am.carry(gy, gy.reshape(-1, n))
    if not g2.is_contiguous():
        g2 = _am.carry(g2, g2.contiguous())
    grp = g2.shape[0] // scale.numel()
    if ASCALE_IN_GEMM and grp % 32 == 0 and grp * scale.numel() == g2.shape[0] and bound <= 3.9 and g2.shape[0] % 32 == 0 \
            and scale.numel() <= 64:
        return g2, scale, grp
    return _scaled_rows(gy, n, scale, bound), None, 1


def _scaled_rows(gy, n, scale, bound):
    """gy as contiguous [M, n] rows, times the per-sample factors (the branch's share of a residual sum's gradient); the absmax
    tag travels along (|g scale| <= |g| * bound)."""
    from . import amax as _am
    g2 = _am.carry(gy, gy.reshape(-1, n))
    if not g2.is_contiguous():
        g2 = _am.carry(g2, g2.contiguous())
    if scale is None:
        return g2
    out = (g2.view(scale.numel(), -1, n) * scale.view(-1, 1, 1)).view(-1, n)
    t = _am.tag_of(g2)
    if t is not None:
        _am.tag(out, t * bound)
    return out


class _FusedMlp(torch.autograd.Function):
    """fc2(gelu(fc1(x))) (+ shortcut, scaled per sample) of a Swin Mlp (reference models/Swin.py:62-76, :321) with the
    element-wise passes inside the GEMMs: fc1's epilogue writes the pre-activation h AND gelu(h) (no GELU kernel), fc2's epilogue
    adds the residual (no addcmul), and in the backward fc2's data gradient leaves its epilogue already multiplied by gelu'(h)
    (no gelu_backward kernel) -- per block three passes over [tokens, 4 C] and one over [tokens, C] less."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, shortcut, scale, bound):
        from . import amax as _am
        hd, k = w1.shape
        n = w2.shape[0]
        x2 = _am.carry(x, x.reshape(-1, k))
        m = x2.shape[0]
        dev = x.device
        h = torch.empty((m, hd), dtype=torch.float32, device=dev)
        a = torch.empty((m, hd), dtype=torch.float32, device=dev)
        ch = _am.zeros(1, dev)
        gemm_f16x3_ep(x2, w1, True, m, hd, k, h, _am.amax_of(x2), _am.amax_of(w1), 1, bias=b1, c_amax=ch, out2=a)
        _am.tag(a, ch)                          # |gelu(v)| <= |v|
        if shortcut is not None:
            y = torch.empty((m, n), dtype=torch.float32, device=dev)
            cy = _am.zeros(1, dev)
            gemm_f16x3_ep(a, w2, True, m, n, hd, y, ch, _am.amax_of(w2), 3, bias=b2, c_amax=cy, aux=shortcut, rowscale=scale,
                          rows_per_scale=m // scale.numel() if scale is not None else 1)
            _am.tag(y, cy)
        else:
            y = linear_f16x3(a, w2, b2)
        ctx.save_for_backward(x2, h, a, w1, w2, scale)
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        ctx.xshape, ctx.bound, ctx.residual = x.shape, float(bound), shortcut is not None
        return _am.carry(y, y.view(*x.shape[:-1], n))

    @staticmethod
    def backward(ctx, gy):
        from . import amax as _am
        x2, h, a, w1, w2, scale = ctx.saved_tensors
        hd, k = w1.shape
        n = w2.shape[0]
        m = x2.shape[0]
        g2, sc, grp = _branch_grad(gy, n, scale if ctx.residual else None, ctx.bound)
        need = ctx.needs_input_grad
        gx = gw1 = gb1 = gw2 = gb2 = None
        # fc2: weight / bias gradient from (dy, a); data gradient with gelu'(h) applied in its epilogue = fc1's dy
        if need[3]:
            if ctx.has_b2 and need[4]:
                gw2, gb2 = linear_wgrad_f16x3(g2, a, want_bias=True, scale=sc, group=grp)
            else:
                gw2 = linear_wgrad_f16x3(g2, a, scale=sc, group=grp)
        elif ctx.has_b2 and need[4]:
            gb2 = _scaled_rows(gy, n, scale if ctx.residual else None, ctx.bound).sum(0)
        if need[0] or need[1] or (ctx.has_b1 and need[2]):
            gh = torch.empty((m, hd), dtype=torch.float32, device=gy.device)
            cg = _am.zeros(1, gy.device)
            if sc is not None:
                gemm_f16x3_ascaled(g2, True, n, w2, hd, m, hd, n, gh, _am.amax_of(g2), _am.amax_of(w2), sc, grp, c_amax=cg, ep=2, aux=h)
            else:
                gemm_f16x3_ep(g2, w2, False, m, hd, n, gh, _am.amax_of(g2), _am.amax_of(w2), 2, c_amax=cg, aux=h)
            _am.tag(gh, cg)
            if need[0]:
                g = linear_dgrad_f16x3(gh, w1)
                gx = _am.carry(g, g.view(ctx.xshape))
            if need[1]:
                if ctx.has_b1 and need[2]:
                    gw1, gb1 = linear_wgrad_f16x3(gh, x2, want_bias=True)
                else:
                    gw1 = linear_wgrad_f16x3(gh, x2)
            elif ctx.has_b1 and need[2]:
                gb1 = gh.sum(0)
        return gx, gw1, gb1, gw2, gb2, (gy if (ctx.residual and need[5]) else None), None, None


FUSED_MLP = _dbg.fused_mlp  # module switch (A/B runs, tests of the unfused path): False = TokenLinear -> tagged_gelu -> TokenLinear


def fused_mlp_ok(x, fc1, fc2):
    """Both Linears of a Swin Mlp on the split-f16 GEMM with fused epilogues: the TokenLinear conditions for each, and
    products that need no k-split."""
    if not (FUSED_MLP and isinstance(fc1, TokenLinear) and isinstance(fc2, TokenLinear) and fc1.f16x3 and fc2.f16x3
            and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and fc1.weight.dtype == torch.float32 and fc1.weight.requires_grad and fc2.weight.requires_grad
            and _token_gemm_ok(x, fc1.weight)):
        return False
    hd, k = fc1.weight.shape
    n = fc2.weight.shape[0]
    m = x.numel() // k
    # (few tokens per hidden column -- the last stage: the erf evaluations sit exposed at the end of long tiles and cost more than
    # the cache-resident element-wise kernels they replace: +45 / +55 us per launch at 6 400 x 6 144, tools/probes/gemm_ep_time.py)
    return (fc2.weight.shape[1] == hd and n % 32 == 0 and fc2.weight.is_contiguous() and m * max(n, hd) * 4 < (1 << 32)
            and m >= 4 * hd and _ep_gemm_ok(m, hd, k) and _ep_gemm_ok(m, n, hd) and _ep_gemm_ok(m, hd, n))


def fused_mlp(x, fc1, fc2, shortcut=None, scale=None, bound=1.0):
    """fc2(gelu(fc1(x))), or shortcut + scale * that (scale [B] per-sample factors or None); see _FusedMlp."""
    if shortcut is not None:
        shortcut = shortcut.contiguous()
    return _FusedMlp.apply(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, shortcut,
                           scale.reshape(-1).contiguous() if scale is not None else None, bound)


def linear_residual_ok(x, lin):
    n, k = lin.weight.shape
    return (FUSED_MLP and isinstance(lin, TokenLinear) and lin.f16x3 and x.is_cuda and x.dtype == torch.float32
            and torch.is_grad_enabled() and not torch.is_autocast_enabled() and lin.weight.dtype == torch.float32
            and lin.weight.requires_grad and _token_gemm_ok(x, lin.weight) and _ep_gemm_ok(x.numel() // k, n, k))


def linear_residual(x, lin, shortcut, scale=None, bound=1.0):
    """shortcut + scale * lin(x) with the sum in the GEMM's epilogue (see _TokenLinearResidual)."""
    return _TokenLinearResidual.apply(x, lin.weight, lin.bias, shortcut.contiguous(),
                                      scale.reshape(-1).contiguous() if scale is not None else None, bound)


def _token_gemm_ok(x, weight):
    """Shapes the split-f16 GEMM takes for all three products of a Linear: every extent a multiple of 32 (each is the
    contraction of one of them), the token rows contiguous."""
    n, k = weight.shape
    m = x.numel() // k
    return m >= 1024 and m % 32 == 0 and k % 32 == 0 and n % 32 == 0 and x.is_contiguous() and weight.is_contiguous() \
        and m * max(n, k) * 4 < (1 << 32)


class TokenLinear(torch.nn.Linear):
    """nn.Linear (same parameters / state_dict keys) for token-major fp32 CUDA rows in training: the three GEMMs on the
    split-f16 kernel (csrc/dcl_gemm.hip); anything else is nn.Linear.forward.  ``f16x3 = False`` (class switch, the
    eager comparator of tools / tests) keeps the library's GEMMs."""

    f16x3 = True

    def forward(self, x):
        if (self.f16x3 and x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32
                and self.weight.requires_grad and torch.is_grad_enabled() and not torch.is_autocast_enabled()
                and _token_gemm_ok(x, self.weight)):
            return _TokenLinear.apply(x, self.weight, self.bias)
        return super().forward(x)


# ---- LayerNorm over token-major rows (csrc/dcl_layernorm.hip) -----------------------------------------------------

class _TaggedGelu(torch.autograd.Function):
    """nn.GELU() (exact erf form) that hands absmax BOUNDS through in both directions: |gelu(v)| <= |v| and
    |gelu'(v)| <= 1.13, so the tag of the input bounds the output and 1.13 x the tag of the incoming gradient bounds the
    outgoing one -- the fc2 / fc1 GEMMs of a Swin Mlp find their operand scales without a pass over the 4C-wide tensors."""

    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return torch.nn.functional.gelu(h)

    @staticmethod
    def backward(ctx, gy):
        from . import amax as _am
        (h,) = ctx.saved_tensors
        gx = torch.ops.aten.gelu_backward(gy, h, approximate="none")
        t = _am.tag_of(gy)
        if t is not None:
            _am.tag(gx, t * 1.13)
        return gx


def tagged_gelu(h):
    """nn.GELU() (exact erf form); on CUDA fp32 the absmax tags travel through it (see _TaggedGelu)."""
    if not (h.is_cuda and h.dtype == torch.float32):
        return torch.nn.functional.gelu(h)
    from . import amax as _am
    out = _TaggedGelu.apply(h) if (h.requires_grad and torch.is_grad_enabled()) else torch.nn.functional.gelu(h)
    t = _am.tag_of(h)
    if t is not None:
        _am.tag(out, t)
    return out


def _amax_mod():
    from . import amax
    return amax


def _ln_forward(x, weight, bias, eps):
    from .. import _lib
    from . import amax as _am
    c = x.shape[-1]
    m = x.numel() // c
    y = torch.empty_like(x)
    stats = torch.empty((2, m), dtype=torch.float32, device=x.device)
    ybuf = _am.zeros(_am.SLOTS, x.device)       # absmax tag of y for the Linear behind the norm (the GEMM's operand scale)
    _am.tag(y, ybuf)
    _lib.check(_lib.lib().dcl_layernorm_fwd(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), m, c, float(eps),
                                            _lib.ptr(y), _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(ybuf),
                                            _stream(x)), "dcl_layernorm_fwd")
    return y, stats


def _ln_backward(gy, x, weight, stats, addend=None):
    """(gx (+ addend), dgamma, dbeta); gx is tagged with its absmax (emitted by the kernel)."""
    from .. import _lib
    from . import amax as _am
    c = x.shape[-1]
    m = x.numel() // c
    gy = gy.contiguous()
    if addend is not None:
        addend = addend.contiguous()
    L = _lib.lib()
    gx = torch.empty_like(x)
    parts = torch.empty((L.dcl_layernorm_bwd_parts(m, c), 2, c), dtype=torch.float32, device=x.device)
    gwb = torch.empty((2, c), dtype=torch.float32, device=x.device)
    gam = _am.zeros(_am.SLOTS, x.device)
    _lib.check(L.dcl_layernorm_bwd(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(weight), _lib.ptr(stats[0]),
                                   _lib.ptr(stats[1]), m, c, _lib.ptr(gx), _lib.ptr(parts), _lib.ptr(gwb),
                                   _lib.ptr(addend), _lib.ptr(gam), _stream(x)), "dcl_layernorm_bwd")
    _am.tag(gx, gam)
    return gx, gwb[0], gwb[1]


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, stats = _ln_forward(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, stats = ctx.saved_tensors
        gx, gw, gb = _ln_backward(gy, x, weight, stats)
        return gx, gw, gb, None


class _LayerNormResidualFn(torch.autograd.Function):
    """(LayerNorm(x), alias of x): for ``x -> norm -> branch`` with ``x`` also feeding the residual sum behind the branch
    (both halves of a Swin block, reference models/Swin.py:286-321).  The backward receives the branch's gradient AND the
    shortcut's and adds them inside the norm's backward kernel -- autograd's own sum of the two would be one more
    element-wise pass (3 tensor passes) per norm."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, stats = _ln_forward(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, stats)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gres):
        x, weight, stats = ctx.saved_tensors
        if gy is None:
            return gres, None, None, None
        gx, gw, gb = _ln_backward(gy, x, weight, stats, addend=gres)
        return gx, gw, gb, None


class FusedLayerNorm(torch.nn.LayerNorm):
    """nn.LayerNorm over the last axis (same parameters / state_dict keys) on the HIP kernels of
    csrc/dcl_layernorm.hip for contiguous fp32 CUDA rows of a supported length; anything else (CPU, autocast,
    no affine, several normalised axes) is nn.LayerNorm.forward."""

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and len(self.normalized_shape) == 1 and self.weight is not None
                and self.bias is not None and self.weight.dtype == torch.float32 and x.is_contiguous()
                and not torch.is_autocast_enabled() and x.numel() > 0):
            from .. import _lib
            if _lib.lib().dcl_layernorm_supported(x.shape[-1]):
                return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)

    def with_shortcut(self, x):
        """(self(x), x') where x' carries x into the residual sum: on the HIP path the two gradients of x meet inside the
        norm's backward kernel (see _LayerNormResidualFn); elsewhere x' is x."""
        if (x.is_cuda and x.dtype == torch.float32 and len(self.normalized_shape) == 1 and self.weight is not None
                and self.bias is not None and self.weight.dtype == torch.float32 and x.is_contiguous()
                and not torch.is_autocast_enabled() and x.numel() > 0 and torch.is_grad_enabled() and x.requires_grad):
            from .. import _lib
            if _lib.lib().dcl_layernorm_supported(x.shape[-1]):
                return _LayerNormResidualFn.apply(x, self.weight, self.bias, self.eps)
        return self(x), x


# ---- Swin window attention (csrc/dcl_winattn.hip) -----------------------------------------------------------------

class _WindowAttention(torch.autograd.Function):
    """softmax(q k^T * scale + bias (+ shift mask)) v over 7 x 7 windows of tokens kept in their natural order."""

    @staticmethod
    def forward(ctx, qkv, qkv_bias, bias, H, W, heads, shift, scale):
        from .. import _lib
        L = _lib.lib()
        B, Ltok, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty((B, Ltok, C), dtype=torch.float32, device=qkv.device)
        nW = ((H + 6) // 7) * ((W + 6) // 7)
        lse = torch.empty((B, nW, heads, 49), dtype=torch.float32, device=qkv.device)
        _lib.check(L.dcl_winattn_fwd(_lib.ptr(qkv), _lib.ptr(qkv_bias), _lib.ptr(bias), B, H, W, C, heads, shift,
                                     scale, _lib.ptr(out), _lib.ptr(lse), _lib.stream_ptr(qkv.device)),
                   "dcl_winattn_fwd")
        ctx.save_for_backward(qkv, qkv_bias, bias, lse)
        ctx.geom = (H, W, heads, shift, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .. import _lib
        L = _lib.lib()
        qkv, qkv_bias, bias, lse = ctx.saved_tensors
        H, W, heads, shift, scale = ctx.geom
        B, Ltok, C3 = qkv.shape
        C = C3 // 3
        dout = dout.contiguous()
        npad = L.dcl_winattn_npad(H, W)
        nwaves = L.dcl_winattn_bwd_waves(B, H, W, heads)
        dqkv = torch.empty_like(qkv)
        dpad = torch.empty((B, npad, C3), dtype=torch.float32, device=qkv.device) if npad else None
        part = torch.empty((nwaves, 49, 49), dtype=torch.float32, device=qkv.device)
        from . import amax as _am
        gam = _am.zeros(_am.SLOTS, qkv.device)      # max|dqkv|: the operand scale of the qkv Linear's backward GEMMs
        _lib.check(L.dcl_winattn_bwd(_lib.ptr(qkv), _lib.ptr(qkv_bias), _lib.ptr(bias), _lib.ptr(lse), _lib.ptr(dout),
                                     B, H, W, C, heads, shift, scale, _lib.ptr(dqkv), _lib.ptr(dpad), _lib.ptr(part),
                                     _lib.ptr(gam), _lib.stream_ptr(qkv.device)), "dcl_winattn_bwd")
        _am.tag(dqkv, gam)
        dbias = part.view(nwaves // heads, heads, 49, 49).sum(0) if ctx.needs_input_grad[2] else None
        dqb = None
        if ctx.needs_input_grad[1]:
            dqb = dpad.sum((0, 1)) if npad else torch.zeros_like(qkv_bias)
        return dqkv, dqb, dbias, None, None, None, None, None


def window_attention(qkv, qkv_bias, bias, H, W, heads, shift, scale):
    """qkv [B, H*W, 3C] (projection of the tokens in natural order), qkv_bias [3C] (qkv of zero-padded tokens),
    bias [heads, 49, 49] -> [B, H*W, C]; window 7, head_dim 32, fp32 (csrc/dcl_winattn.hip)."""
    return _WindowAttention.apply(qkv.contiguous(), qkv_bias.contiguous(), bias.contiguous(), int(H), int(W),
                                  int(heads), int(shift), float(scale))


# ---- fused up-sampling + cross-entropy (csrc/dcl_upce.hip) --------------------------------------------------------

class _UpsampleCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, target, weight, ignore_index, H, W, align_corners, holder):
        from .. import _lib
        L = _lib.lib()
        n, c, h, w = z.shape
        dev = z.device
        lse = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        pred = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
        partial = torch.empty((n * H, 2), dtype=torch.float32, device=dev)
        out2 = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.check(L.dcl_upsample_ce_fwd(_lib.ptr(z), n, c, h, w, H, W, 1 if align_corners else 0, _lib.ptr(target),
                                         _lib.ptr(weight), int(ignore_index), _lib.ptr(lse), _lib.ptr(pred),
                                         _lib.ptr(partial), _lib.ptr(out2), _lib.stream_ptr(dev)),
                   "dcl_upsample_ce_fwd")
        ctx.save_for_backward(z, target, weight, lse, out2)
        ctx.geom = (H, W, bool(align_corners), int(ignore_index))
        if holder is not None:
            holder["pred"] = pred
        return out2[0].clone()

    @staticmethod
    def backward(ctx, gout):
        from .. import _lib
        L = _lib.lib()
        z, target, weight, lse, out2 = ctx.saved_tensors
        H, W, align, ignore = ctx.geom
        n, c, h, w = z.shape
        gscale = (gout.reshape(1).to(torch.float32) / out2[1:2]).contiguous()
        dz = torch.empty_like(z)
        _lib.check(L.dcl_upsample_ce_bwd(_lib.ptr(z), n, c, h, w, H, W, 1 if align else 0, _lib.ptr(target),
                                         _lib.ptr(weight), ignore, _lib.ptr(lse), _lib.ptr(gscale), _lib.ptr(dz),
                                         _lib.stream_ptr(z.device)), "dcl_upsample_ce_bwd")
        return dz, None, None, None, None, None, None, None


class UpsampledLogits:
    """Logits that exist at 1/4 resolution only: ``lowres`` [N, C, h, w] (part of the autograd graph) plus the size
    and align_corners flag of the bilinear up-sampling the reference applies to them (models/HRNet.py:638).  Returned
    by HRNet when graph['lazy_logits'] is set (an extension: the default returns the up-sampled tensor like the
    reference); this repo's LossWrapper / TwoScaleLoss / metrics consume it through the fused kernels of
    csrc/dcl_upce.hip, anything else calls ``materialize()``."""

    def __init__(self, lowres, size, align_corners):
        self.lowres, self.size, self.align_corners = lowres, (int(size[0]), int(size[1])), bool(align_corners)
        self.pred = None                                   # uint8 [N, H, W] arg-max map, filled by cross_entropy()
        self._full = None

    @property
    def shape(self):
        return torch.Size((self.lowres.shape[0], self.lowres.shape[1]) + self.size)

    @property
    def device(self):
        return self.lowres.device

    def materialize(self):
        if self._full is None:
            self._full = upsample_bilinear(self.lowres, self.size, self.align_corners)
        return self._full

    def cross_entropy(self, target, weight=None, ignore_index=-100):
        """nn.CrossEntropyLoss(weight, ignore_index)(materialize(), target) without materialising."""
        z = self.lowres
        if not (z.is_cuda and z.dtype == torch.float32 and z.shape[1] <= 255 and not torch.is_autocast_enabled()):
            return F.cross_entropy(self.materialize(), target, weight=weight, ignore_index=ignore_index)
        holder = {}
        loss = _UpsampleCE.apply(z.contiguous(), target.contiguous().long(), None if weight is None else
                                 weight.to(device=z.device, dtype=torch.float32).contiguous(), int(ignore_index),
                                 self.size[0], self.size[1], self.align_corners, holder)
        self.pred = holder["pred"]
        return loss
