"""Direct split-f16 (f16x3) convolutions: 3x3 / 1x1 forward, data gradient, weight gradient (csrc/dcl_conv3x3.hip, dcl_wgrad3x3*.hip) --
reference models/HRNet.py:56-137 (BasicBlock / Bottleneck convolutions), :216-261 (fuse layers), :333-338 (stem)."""
import torch
import torch.nn.functional as F

from ..debug import cfg as _dbg      # A/B switches of the tuning tools: one object (mscs_amd/debug.py)
from .ops_common import _stream
from .ops_linear import gemm_f16x3


# ---- direct f16x3 3x3 convolution (csrc/dcl_conv3x3.hip) ----------------------------------------------------------

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


SMALL_CIN_STEM = _dbg.small_cin_stem     # the stem's 3 -> 64 stride-2 convolution on its own fp32 kernel (DCL_SMALL_CIN_STEM=0: tile kernel)


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


def conv3x3_wgrad_pre(x, gy, pre_sc, pre_sh, pre_amax, stride=1):
    """dw of conv2d(relu(x * sc[c] + sh[c]), w, stride=stride, padding=1) for the output gradient gy, x the RAW tensor in front of
    the norm: csrc/dcl_wgrad3x3d.hip (stride 1), csrc/dcl_wgrad3x3_s2.hip (stride 2), PRE forms."""
    from .. import _lib
    from .amax import amax_of
    n, ci, h, w = x.shape
    co = gy.shape[1]
    L = _lib.lib()
    splits = L.dcl_wgrad3x3_splits(n, ci, co, h, w, stride)
    if splits <= 0:
        raise RuntimeError("conv3x3_wgrad_pre: unsupported shape")
    part = torch.empty(splits * 9 * co * ci, dtype=torch.float32, device=x.device)
    dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=x.device)
    ga = amax_of(gy)
    _lib.check(L.dcl_wgrad3x3_pre_f16x3(_lib.ptr(x), _lib.ptr(gy), n, ci, co, h, w, _lib.ptr(pre_amax), pre_amax.numel(),
                                        _lib.ptr(ga), ga.numel(), _lib.ptr(pre_sc), _lib.ptr(pre_sh), stride, _lib.ptr(part),
                                        _lib.ptr(dw), _stream(x)), "dcl_wgrad3x3_pre_f16x3")
    return dw


def stem_wgrad_supported(x, cout, stride):
    return stride == 2 and x.shape[1] * 9 <= 32 and cout <= 64 and x.is_contiguous() and x.dtype == torch.float32


def stem_wgrad(x, gy):
    """dw [Co, Ci, 3, 3] of conv2d(x, w, stride=2, padding=1) for Ci <= 3 input channels (the stem's convolution on the image) on
    csrc/dcl_conv3x3.hip k_wgrad_stem: fp32 products on the matrix pipe, fixed summation order."""
    from .. import _lib
    L = _lib.lib()
    n, ci, h, w = x.shape
    co = gy.shape[1]
    part = torch.empty(L.dcl_wgrad3x3_s2_smallcin_workspace(co), dtype=torch.float32, device=x.device)
    dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=x.device)
    _lib.check(L.dcl_wgrad3x3_s2_smallcin(_lib.ptr(x), n, ci, h, w, _lib.ptr(gy), co, _lib.ptr(part), _lib.ptr(dw), _stream(x)),
               "dcl_wgrad3x3_s2_smallcin")
    return dw


class _Conv3x3Direct(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 convolution on the f16x3 (fp32-equivalent) kernels: forward and data gradient through
    csrc/dcl_conv3x3.hip, weight gradient through csrc/dcl_wgrad3x3.hip (channel counts that are not multiples
    of 16: ATen / MIOpen)."""

    @staticmethod
    def forward(ctx, x, weight, mod, token=None, bias=None):
        from .amax import amax_of, pre_of
        ctx.token = token
        ctx.has_bias = bias is not None
        ctx.stride = st = mod.stride[0]
        ctx.k1 = k1 = mod.kernel_size == (1, 1)
        wamax, wp, _ = mod.packed_weights()
        out = torch.empty((x.shape[0], weight.shape[0], (x.shape[2] - 1) // st + 1, (x.shape[3] - 1) // st + 1),
                          dtype=torch.float32, device=x.device)
        pre = pre_of(x)
        ctx.pre = pre is not None
        if pre is not None:
            # x's storage holds the raw input z of the norm in front of this convolution; the operand is relu(z sc + sh), formed
            # in the kernel's patch staging (csrc/dcl_conv3x3_pre.hip).  The caller asked fuses_input_norm() first.
            from .. import _lib
            if not mod.fuses_input_norm(x):
                raise RuntimeError("DirectConv2d: handed a deferred norm output it does not take (fuses_input_norm() is False)")
            n, ci, h, w = x.shape
            _lib.check(_lib.lib().dcl_conv3x3_pre_f16x3(_lib.ptr(x), n, ci, h, w, _lib.ptr(wp), weight.shape[0],
                                                        _lib.ptr(pre.amax), pre.amax.numel(), _lib.ptr(wamax), _lib.ptr(pre.sc),
                                                        _lib.ptr(pre.sh), _lib.ptr(bias), _lib.ptr(out), st, 0, 0, _stream(x)),
                       "dcl_conv3x3_pre_f16x3")
            ctx.save_for_backward(x, weight, pre.sc, pre.sh, pre.amax)
            ctx.mod = mod
            return out
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
            conv3x3_launch(x, wp, weight.shape[0], amax_of(x), wamax, out, stride=st, bias=bias)
        ctx.save_for_backward(x, weight)
        ctx.mod = mod
        return out

    @staticmethod
    def backward(ctx, gy):
        from .amax import amax_of
        if ctx.pre:
            x, weight, pre_sc, pre_sh, pre_amax = ctx.saved_tensors
        else:
            x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            wamax, _, wpt = ctx.mod.packed_weights()
            gx = torch.empty_like(x)
            addend = None
            if ctx.token is not None and ctx.token.dres is not None:
                addend, ctx.token.dres = ctx.token.dres, None        # gradient of the residual branch, fused in
            if ctx.k1 and _conv1x1_by_gemm(weight.shape[1], weight.shape[0], gy, True) and addend is None:
                # (with a residual gradient the tile kernel's fused addend wins: accumulating into it in the GEMM's epilogue was
                # measured at 467 us against 372 on the 256-channel gradients of layer 1's Bottlenecks and left the product in round 6)
                conv1x1_gemm(gy, weight.view(weight.shape[0], -1), gx, amax_of(gy), wamax, transposed=True)
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
        if ctx.needs_input_grad[1] and ctx.pre:
            gw = conv3x3_wgrad_pre(x, gy, pre_sc, pre_sh, pre_amax, ctx.stride)
        elif ctx.needs_input_grad[1]:
            if ctx.k1:
                if conv1x1_wgrad_supported(x, weight.shape[0]):
                    gw = conv1x1_wgrad(x, gy)
                else:
                    n, ci, h, w = x.shape
                    gw = torch.bmm(gy.view(n, -1, h * w), x.view(n, ci, h * w).transpose(1, 2)).sum(0).view_as(weight)
            elif conv3x3_wgrad_supported(x, weight.shape[0], ctx.stride):
                gw = conv3x3_wgrad(x, gy, ctx.stride)
            elif stem_wgrad_supported(x, weight.shape[0], ctx.stride):
                # the stem's 3-channel input (reference models/HRNet.py:404-405): the library's kernel for this shape splits the
                # pixels and adds the pieces with ATOMICS (igemm_wrw ... gkgs) -- the one launch of a training step whose result
                # changed from run to run (tools/probes/step_repro.py: after one step this weight differed by 8e-8, everything else
                # was bitwise equal; after three steps every tensor differed)
                gw = stem_wgrad(x, gy)
            elif weight.shape[1] < 16 and weight.shape[0] % 16 == 0 \
                    and (x.shape[3] % 8 == 0 or ctx.stride == 1):
                # other narrow inputs: channels zero-padded to 16, then the split-f16 weight-gradient kernel (fixed summation order too)
                xp = x.new_zeros((x.shape[0], 16, x.shape[2], x.shape[3]))
                xp[:, :weight.shape[1]] = x
                gw = conv3x3_wgrad(xp, gy, ctx.stride)[:, :weight.shape[1]].contiguous()
            else:
                st = ctx.stride
                gw = torch.ops.aten.convolution_backward(gy, x, weight, None, [st, st], [1, 1], [1, 1], False,
                                                         [0, 0], 1, [False, True, False])[1]
        gb = gy.sum((0, 2, 3)) if (ctx.has_bias and ctx.needs_input_grad[4]) else None
        return gx, gw, None, None, gb


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

    def forward(self, x, grad_token=None):
        if self.eligible(x):
            return _Conv3x3Direct.apply(x, self.weight, self, grad_token, self.bias)
        from .amax import refuse_pre
        refuse_pre(x, "DirectConv2d (library path)")
        return super().forward(x)

    def fuses_input_norm(self, z):
        """True when this convolution takes ``relu(bn(z))`` as a deferred norm output (amax.PreAct): the norm's map + ReLU run in
        the operand staging of its forward and weight-gradient kernels and the normalised tensor is never written.  3x3 / stride
        1 or 2 / pad 1 on the direct kernels, channel counts in sixteens, rows in octets (stride 2: sixteens; the weight gradient's
        row padding would pad the RAW tensor, whose zeros do not map to zeros)."""
        from .. import _lib
        if not (_dbg.fuse_bn_apply and self.kernel_size == (3, 3) and self.stride in ((1, 1), (2, 2)) and self.eligible(z)):
            return False
        n, ci, h, w = z.shape
        co, st = self.weight.shape[0], self.stride[0]
        if ci % 16 or co % 16 or w % (8 if st == 1 else 16):
            return False
        L = _lib.lib()
        return bool(L.dcl_conv3x3_pre_supported(n, ci, co, h, w, st)) and bool(L.dcl_wgrad3x3_pre_supported(n, ci, co, h, w, st))

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
    from .ops_conv1x1 import GemmConv1x1
    for m in module.modules():
        if type(m) in (torch.nn.Conv2d, GemmConv1x1) and m.kernel_size == (1, 1) and m.stride == (1, 1) \
                and m.padding == (0, 0) and m.dilation == (1, 1) and m.groups == 1:
            m.__class__ = DirectConv2d
    return module


# ---- head convolution over a concatenation of up-sampled maps, without the up-sampled maps ---------------------------
