"""The head convolution over up-sampled maps without the up-sampled maps (reference models/HRNet.py:549-553, :596-600; UPerNet.py:96-101):
channel products of the coarse maps at their own resolution + tap-wise bilinear gather (csrc/dcl_resize.hip k_tapup_*)."""
import torch
import torch.nn.functional as F

from ..debug import cfg as _dbg      # A/B switches of the tuning tools: one object (mscs_amd/debug.py)
from .ops_conv import conv3x3_launch, conv3x3_pack, conv3x3_wgrad, conv3x3_wgrad_supported
from .ops_linear import gemm_f16x3
from .ops_resize import upsample_concat


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
    round-off (tests/test_model_ops_parity.py::test_head_conv_over_upsampled_matches_fp64)."""
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


# ---- the head's norm folded into its classifier -------------------------------------------------------------------------------

class _HeadNormClassifier(torch.autograd.Function):
    """``conv1x1(bn(z), W)`` of the HRNet head (reference models/HRNet.py:596-600: conv3x3 -> BatchNorm -> conv1x1 with NO activation
    in between) without the normalised tensor: a training-mode norm is a per-channel affine map, a 1x1 convolution is linear, so
    ``W (sc z + sh) = (W diag(sc)) z + W sh``.  Forward: the norm's statistics pass and its finalisation (csrc/dcl_bn.hip: mean,
    invstd, running statistics and the map exactly as FusedBatchNorm2d computes them; the SyncBatchNorm exchange in between on several
    ranks), then the classifier GEMM on z with rescaled weights.  Backward: G = dl z^T (the classifier's weight-gradient product) and
    s = sum dl give dW, dgamma, dbeta and the two channel sums of the norm's backward; dz = W'^T dl + c1 z + c0 is ONE pass
    (dcl_head_norm_dz) instead of the classifier's data-gradient GEMM, the norm's reduce and its apply.  At batch 12 x 128 x 256 x 720
    channels: six passes over 1.13 GB less per step and two 1.13-GB tensors less in memory."""

    @staticmethod
    def forward(ctx, z, gamma, beta, weight, bn):
        from .. import _lib
        from . import amax as _amax
        from .fused_bn import _all_reduce_async, _check_equal_batch, _world
        L = _lib.lib()
        n, c, h, w = z.shape
        hw = h * w
        k = weight.shape[0]
        dev = z.device
        st = _lib.stream_ptr(dev)
        world = _world() if bn.sync else 1
        count = float(n * hw * world)
        ns = L.dcl_bn_num_slices(n, c)
        ws = torch.empty((c * ns * 2 + 5 * c,), dtype=torch.float32, device=dev)
        part = ws[:c * ns * 2]
        mean, invstd, pivot, sc, sh = (ws[c * ns * 2 + i * c:c * ns * 2 + (i + 1) * c] for i in range(5))
        _lib.check(L.dcl_bn_stats_part(_lib.ptr(z), n, c, hw, _lib.ptr(part), _lib.ptr(bn.running_mean), _lib.ptr(pivot), st),
                   "dcl_bn_stats_part")
        if world > 1:
            _check_equal_batch(n, dev)
            _all_reduce_async(part).wait()
        _lib.check(L.dcl_bn_finalize_pre(_lib.ptr(part), None, ns, count, float(bn.eps), float(bn.momentum), _lib.ptr(gamma),
                                         _lib.ptr(beta), c, _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(bn.running_mean),
                                         _lib.ptr(bn.running_var), _lib.ptr(bn.num_batches_tracked), _lib.ptr(pivot), _lib.ptr(sc),
                                         _lib.ptr(sh), None, st), "dcl_bn_finalize_pre")
        w2 = weight.view(k, c)
        out = torch.empty((n, k, h, w), dtype=torch.float32, device=dev)
        torch.matmul(w2 * sc, z.view(n, c, hw), out=out.view(n, k, hw))
        out += (w2 @ sh).view(1, k, 1, 1)
        ctx.save_for_backward(z, w2, gamma, mean, invstd, sc, sh)
        ctx.world, ctx.count = world, count
        return out

    @staticmethod
    def backward(ctx, dl):
        from .. import _lib
        from . import amax as _amax
        from .fused_bn import _all_reduce_async
        L = _lib.lib()
        z, w2, gamma, mean, invstd, sc, sh = ctx.saved_tensors
        n, c, h, w = z.shape
        hw = h * w
        k = w2.shape[0]
        dev = z.device
        dl = dl.contiguous()
        s = dl.sum((0, 2, 3))                                                          # [K]
        g = torch.bmm(dl.view(n, k, hw), z.view(n, c, hw).transpose(1, 2)).sum(0)      # [K, C] = sum dl z^T
        # gradient of the norm's output dy = W^T dl, never formed: its channel sums follow from G and s
        dbeta = (w2 * s.view(k, 1)).sum(0)                                             # sum dy
        dgamma = (w2 * (g - mean.view(1, c) * s.view(k, 1))).sum(0) * invstd           # sum dy xhat
        dw = (g * sc.view(1, c) + s.view(k, 1) * sh.view(1, c)).view(k, c, 1, 1)
        sums = torch.stack([dbeta, dgamma])
        if ctx.world > 1:
            # dz needs the sums over ALL ranks; dgamma / dbeta stay this rank's (DDP averages parameter gradients)
            sums = sums.clone()
            _all_reduce_async(sums).wait()
        c1 = -sc * invstd * sums[1] / ctx.count
        c0 = -sc * sums[0] / ctx.count - c1 * mean
        kp = (k + 3) // 4 * 4
        wt = torch.zeros((c, kp), dtype=torch.float32, device=dev)
        wt[:, :k] = (w2 * sc.view(1, c)).t()
        dz = torch.empty_like(z)
        am = _amax.zeros(_amax.SLOTS, dev)
        _lib.check(L.dcl_head_norm_dz(_lib.ptr(dl), _lib.ptr(z), _lib.ptr(wt), _lib.ptr(c0.contiguous()), _lib.ptr(c1.contiguous()),
                                      n, k, c, hw, _lib.ptr(dz), _lib.ptr(am), _lib.stream_ptr(dev)), "dcl_head_norm_dz")
        _amax.tag(dz, am)
        return dz, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None, \
            dw if ctx.needs_input_grad[3] else None, None


FOLD_HEAD_NORM = _dbg.fold_head_norm     # (DCL_FOLD_HEAD_NORM=0: the norm writes its output, the classifier reads it)


def head_norm_classifier_ok(z, bn, conv):
    """conv1x1(bn(z)) can run folded: a fused norm on its training path followed DIRECTLY by a bias-free 1x1 convolution with at most
    32 outputs (the classifier), contiguous fp32 NCHW with whole pixel quads."""
    from .fused_bn import FusedBatchNorm2d
    return (FOLD_HEAD_NORM and isinstance(bn, FusedBatchNorm2d) and bn._fusable(z, None) and isinstance(conv, torch.nn.Conv2d)
            and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and conv.bias is None and conv.weight.dtype == torch.float32 and 1 <= conv.weight.shape[0] <= 32
            and (z.shape[2] * z.shape[3]) % 4 == 0 and z.data_ptr() % 16 == 0)


def head_norm_classifier(z, bn, conv):
    """``conv(bn(z))`` for a training-mode norm and a bias-free 1x1 convolution, see _HeadNormClassifier."""
    return _HeadNormClassifier.apply(z, bn.weight, bn.bias, conv.weight, bn)
