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
