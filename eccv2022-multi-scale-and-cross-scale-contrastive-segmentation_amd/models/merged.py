"""One launch per kernel stage for the coarse branches of an HRNet exchange module at one block depth.

Reference models/HRNet.py:263-287 runs the branches of a ``HighResolutionModule`` one after the other; each is a chain of
BasicBlocks (conv3x3 -> bn -> relu -> conv3x3 -> bn -> += -> relu, :77-93).  At W48 / batch 12 the branch-1..3 convolutions
put 192 / 96 / 96 workgroups on 256 CUs and their norms stream 9-38 MB each: launched singly -- even on one HIP stream per
branch -- those kernels run mostly one after the other with most of the chip idle (profiles/r04_step_kernels.csv: exactly one
kernel running for 75 of 101 busy ms).  Here the branches' kernels of one depth are ONE launch per stage, job-table forms of
the same kernel bodies (csrc/dcl_conv3x3.hip ``k_conv3x3_il_multi``, csrc/dcl_bn.hip ``k_bn_*_multi``): heaviest workgroups
first, so the dispatcher packs them over the CUs.  Measured on the three coarse W48 shapes (tools/probes/conv_multi_time.py):
213 us for three launches in a row -> 130 us merged.

Arithmetic and kernels are those of the single-layer path: results are bitwise the same for the same convolution tile
(tests/test_merged_branches.py).  Autograd sees one node per stage and depth (``_ConvGroupFn``, ``_BNMergedFn``), so the
backward runs the same merged launches (data gradients merged; weight gradients, which already fill the chip, singly)."""
import ctypes

import torch

from .. import _lib
from . import amax as _amax
from .fused_bn import _PACKED_RELU_MASK, _all_reduce_async, _check_equal_batch, _world

_vp, _i = ctypes.c_void_p, ctypes.c_int


class ConvJob(ctypes.Structure):
    """include/dcl_hip.h dcl_conv_job"""
    _fields_ = [("x", _vp), ("wp", _vp), ("xamax", _vp), ("wamax", _vp), ("addend", _vp), ("bias", _vp), ("y", _vp),
                ("N", _i), ("Cin", _i), ("Cout", _i), ("H", _i), ("W", _i), ("xcount", _i), ("tile_p", _i), ("reserved", _i)]


class BnJob(ctypes.Structure):
    """include/dcl_hip.h dcl_bn_job"""
    _fields_ = [("x", _vp), ("res", _vp), ("dy", _vp), ("ymask", _vp), ("gamma", _vp), ("beta", _vp), ("part", _vp),
                ("part_all", _vp), ("mean", _vp), ("invstd", _vp), ("running_mean", _vp), ("running_var", _vp),
                ("batches_tracked", _vp), ("pivot", _vp), ("y", _vp), ("relu_mask", _vp), ("dx", _vp), ("dres", _vp),
                ("dbeta", _vp), ("dgamma", _vp), ("amax", _vp),
                ("N", _i), ("C", _i), ("HW", _i), ("relu_mode", _i), ("eps", ctypes.c_float), ("momentum", ctypes.c_float)]


MAX_JOBS = 4
TILE_P = 2          # rows per wave of the merged convolution tiles: (3, 2) packs best (conv_multi_time.py: 130 us vs 156 for (3, 4))


def _p(t):
    return None if t is None else t.data_ptr()


def conv_multi_ok(convs, xs):
    """Every job is a plain stride-1 3x3 DirectConv2d on the (3, P) interleaved tile: Cin % 16 == 0, Cout % 96 == 0."""
    from .ops import DirectConv2d
    if not (1 < len(convs) <= MAX_JOBS):
        return False
    for c, x in zip(convs, xs):
        if not (isinstance(c, DirectConv2d) and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.bias is None
                and c.eligible(x) and c.in_channels % 16 == 0 and c.out_channels % 96 == 0):
            return False
    return True


def _launch_convs(xs, wps, wamaxs, couts, outs, addends=None):
    n = len(xs)
    arr = (ConvJob * n)()
    keep = []
    for k in range(n):
        x = xs[k]
        xa = _amax.amax_of(x)
        keep.append(xa)
        N, C, H, W = x.shape
        arr[k] = ConvJob(x.data_ptr(), wps[k].data_ptr(), xa.data_ptr(), wamaxs[k].data_ptr(),
                         _p(addends[k]) if addends else None, None, outs[k].data_ptr(), N, C, couts[k], H, W, xa.numel(), TILE_P, 0)
    _lib.check(_lib.lib().dcl_conv3x3_f16x3_multi(ctypes.addressof(arr), n, _lib.stream_ptr(xs[0].device)),
               "dcl_conv3x3_f16x3_multi")
    return outs


class _ConvGroupFn(torch.autograd.Function):
    """The 3x3 / stride-1 convolutions of several branches (independent inputs and weights) as one autograd node: forward =
    one merged launch, backward = one merged data-gradient launch (residual gradients from the GradTokens added in its
    epilogue) + the weight gradients (csrc/dcl_wgrad3x3d.hip; each fills the chip by itself)."""

    @staticmethod
    def forward(ctx, mods, toks, *tensors):
        xs, ws = tensors[0::2], tensors[1::2]
        wamaxs, wps = [], []
        for m in mods:
            wa, wp, _ = m.packed_weights()
            wamaxs.append(wa)
            wps.append(wp)
        outs = [torch.empty((x.shape[0], w.shape[0], x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
                for x, w in zip(xs, ws)]
        _launch_convs(xs, wps, wamaxs, [w.shape[0] for w in ws], outs)
        ctx.save_for_backward(*tensors)
        ctx.mods, ctx.toks = mods, toks
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        from .ops import conv3x3_wgrad, conv3x3_wgrad_supported
        tensors = ctx.saved_tensors
        xs, ws = tensors[0::2], tensors[1::2]
        n = len(xs)
        gys = [g.contiguous() for g in gys]
        grads = [None] * (2 * n)
        want_x = [ctx.needs_input_grad[2 + 2 * k] for k in range(n)]
        if any(want_x):
            idx = [k for k in range(n) if want_x[k]]
            wamaxs, wpts, addends, gxs = [], [], [], []
            for k in idx:
                wa, _, wpt = ctx.mods[k].packed_weights()
                wamaxs.append(wa)
                wpts.append(wpt)
                tok = ctx.toks[k]
                ad = None
                if tok is not None and tok.dres is not None:
                    ad, tok.dres = tok.dres, None           # gradient of the residual branch, fused into the epilogue
                addends.append(ad)
                gxs.append(torch.empty_like(xs[k]))
            if len(idx) > 1:
                _launch_convs([gys[k] for k in idx], wpts, wamaxs, [ws[k].shape[1] for k in idx], gxs, addends)
            else:
                from .ops import conv3x3_launch
                k = idx[0]
                conv3x3_launch(gys[k], wpts[0], ws[k].shape[1], _amax.amax_of(gys[k]), wamaxs[0], gxs[0], addend=addends[0])
            for k, gx in zip(idx, gxs):
                grads[2 * k] = gx
        for k in range(n):
            if ctx.needs_input_grad[3 + 2 * k]:
                if conv3x3_wgrad_supported(xs[k], ws[k].shape[0], 1):
                    grads[2 * k + 1] = conv3x3_wgrad(xs[k], gys[k], 1)
                else:
                    grads[2 * k + 1] = torch.ops.aten.convolution_backward(gys[k], xs[k], ws[k], None, [1, 1], [1, 1], [1, 1],
                                                                           False, [0, 0], 1, [False, True, False])[1]
        return (None, None, *grads)


def conv3x3_group(convs, xs, tokens=None):
    """``[conv_k(x_k)]`` for independent DirectConv2d layers in one launch (see ``conv_multi_ok``)."""
    tokens = tuple(tokens) if tokens else (None,) * len(convs)
    tensors = []
    for c, x in zip(convs, xs):
        tensors += [x, c.weight]
    return list(_ConvGroupFn.apply(tuple(convs), tokens, *tensors))


class _BNMergedFn(torch.autograd.Function):
    """Training-mode batch norm (+ residual) (+ ReLU) of several independent FusedBatchNorm2d layers with ONE launch per kernel
    stage (statistics, apply; backward: reduce, apply) -- and, as SyncBatchNorm on several ranks, one stacked all-reduce per
    direction (the partial sums of all members are slices of one buffer).  Same arithmetic as ``_FusedBNFunction``."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        L = _lib.lib()
        n = len(meta)
        xs, ress, ws, bs = tensors[0::4], tensors[1::4], tensors[2::4], tensors[3::4]
        dev = xs[0].device
        st = _lib.stream_ptr(dev)
        world = _world() if any(m['sync'] for m in meta) else 1
        if world > 1:
            _check_equal_batch(xs[0].shape[0], dev)
        relu = meta[0]['relu']
        sizes = [x.shape[1] * L.dcl_bn_num_slices(x.shape[0], x.shape[1]) * 2 for x in xs]
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        csum = [0]
        for x in xs:
            csum.append(csum[-1] + 3 * x.shape[1])
        stack = torch.empty((offs[-1],), dtype=torch.float32, device=dev)
        small = torch.empty((csum[-1],), dtype=torch.float32, device=dev)         # per member [mean C | invstd C | pivot C]
        arr = (BnJob * n)()
        ys, masks, saved = [], [], []
        for k, m in enumerate(meta):
            x, res = xs[k], ress[k]
            N, C, H, W = x.shape
            HW = H * W
            y = torch.empty_like(x)
            need_y = relu and res is not None
            mask = torch.empty(N * C * HW // 64, dtype=torch.int64, device=dev) \
                if need_y and HW % 256 == 0 and _PACKED_RELU_MASK else None
            wsb = small[csum[k]:csum[k + 1]]
            j = BnJob()
            j.x, j.res, j.gamma, j.beta = x.data_ptr(), _p(res), _p(ws[k]), _p(bs[k])
            j.part = stack[offs[k]:].data_ptr()
            j.mean, j.invstd, j.pivot = wsb.data_ptr(), wsb[C:].data_ptr(), wsb[2 * C:].data_ptr()
            j.running_mean, j.running_var, j.batches_tracked = m['running_mean'].data_ptr(), m['running_var'].data_ptr(), \
                m['nbt'].data_ptr()
            j.y, j.relu_mask, j.amax = y.data_ptr(), _p(mask), _p(m['amax'])
            j.N, j.C, j.HW, j.eps, j.momentum = N, C, HW, m['eps'], m['momentum']
            arr[k] = j
            ys.append(y)
            masks.append(mask is not None)
            saved += [x, (mask if mask is not None else y) if need_y else None, ws[k], bs[k]]
        _lib.check(L.dcl_bn_stats_part_multi(ctypes.addressof(arr), n, st), "dcl_bn_stats_part_multi")
        if world > 1:
            _all_reduce_async(stack).wait()
        _lib.check(L.dcl_bn_apply_fused_multi(ctypes.addressof(arr), n, world, 1 if relu else 0, st), "dcl_bn_apply_fused_multi")
        ctx.save_for_backward(*saved, small)
        ctx.meta, ctx.masks, ctx.world, ctx.csum = meta, masks, world, csum
        ctx.has_res = [r is not None for r in ress]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        L = _lib.lib()
        meta = ctx.meta
        sv = ctx.saved_tensors
        small = sv[-1]
        n = len(meta)
        dev = small.device
        st = _lib.stream_ptr(dev)
        relu = meta[0]['relu']
        xs = [sv[4 * k] for k in range(n)]
        sizes = [x.shape[1] * L.dcl_bn_num_slices(x.shape[0], x.shape[1]) * 2 for x in xs]
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        local = torch.empty((offs[-1],), dtype=torch.float32, device=dev)
        dys = [dy.contiguous() for dy in dys]
        arr = (BnJob * n)()
        outs = []
        for k, m in enumerate(meta):
            x, ym, w, b = sv[4 * k:4 * k + 4]
            N, C, H, W = x.shape
            wsb = small[ctx.csum[k]:ctx.csum[k + 1]]
            tok = m.get('token')
            dx = torch.empty_like(x)
            want_res = ctx.has_res[k] and (ctx.needs_input_grad[1 + 4 * k + 1] or tok is not None)
            dres = torch.empty_like(x) if want_res else None
            dgamma = torch.empty((C,), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1 + 4 * k + 2] else None
            dbeta = torch.empty((C,), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1 + 4 * k + 3] else None
            amax = _amax.zeros(_amax.SLOTS, dev) if m['amax'] is not None else None
            j = BnJob()
            j.x, j.dy, j.ymask, j.gamma, j.beta = x.data_ptr(), dys[k].data_ptr(), _p(ym), _p(w), _p(b)
            j.part = local[offs[k]:].data_ptr()
            j.mean, j.invstd = wsb.data_ptr(), wsb[C:].data_ptr()
            j.dx, j.dres, j.dbeta, j.dgamma, j.amax = dx.data_ptr(), _p(dres), _p(dbeta), _p(dgamma), _p(amax)
            j.N, j.C, j.HW = N, C, H * W
            j.relu_mode = (2 if ctx.masks[k] else 1) if relu else 0
            arr[k] = j
            outs.append((dx, dres, dgamma, dbeta, amax, tok))
        _lib.check(L.dcl_bn_bwd_reduce_part_multi(ctypes.addressof(arr), n, st), "dcl_bn_bwd_reduce_part_multi")
        total = None
        if ctx.world > 1:
            # dx needs the sums over ALL ranks; dbeta / dgamma stay this rank's sums (DDP averages parameter gradients)
            total = local.clone()
            _all_reduce_async(total).wait()
            for k in range(n):
                arr[k].part_all = total[offs[k]:].data_ptr()
        _lib.check(L.dcl_bn_bwd_apply_fused_multi(ctypes.addressof(arr), n, ctx.world, st), "dcl_bn_bwd_apply_fused_multi")
        grads = []
        for dx, dres, dgamma, dbeta, amax, tok in outs:
            if amax is not None:
                _amax.tag(dx, amax)
            if tok is not None:
                tok.dres, dres = dres, None
            grads += [dx, dres, dgamma, dbeta]
        return (None, *grads)


def bn_merged_ok(bns, xs, residuals=None):
    from .fused_bn import FusedBatchNorm2d
    residuals = residuals or [None] * len(bns)
    return (1 < len(bns) <= MAX_JOBS and all(isinstance(bn, FusedBatchNorm2d) and bn._fusable(x, r)
                                             for bn, x, r in zip(bns, xs, residuals))
            and len({r is None for r in residuals}) == 1)


def bn_act_merged(bns, xs, residuals=None, relu=True, tokens=None):
    """``[bn_act(bn_k, x_k, residual_k, relu, token_k)]`` for independent FusedBatchNorm2d layers, one launch per kernel stage."""
    n = len(bns)
    residuals = residuals or [None] * n
    tokens = tokens or [None] * n
    meta, tensors = [], []
    for k, bn in enumerate(bns):
        amax = _amax.zeros(_amax.SLOTS, xs[k].device) if bn.emit_amax else None
        meta.append(dict(sync=bool(bn.sync), running_mean=bn.running_mean, running_var=bn.running_var, nbt=bn.num_batches_tracked,
                         eps=float(bn.eps), momentum=float(bn.momentum), relu=bool(relu), amax=amax,
                         token=tokens[k] if residuals[k] is not None else None))
        tensors += [xs[k], residuals[k], bn.weight, bn.bias]
    ys = _BNMergedFn.apply(meta, *tensors)
    return [(_amax.tag(y, meta[k]['amax']) if meta[k]['amax'] is not None else y) for k, y in enumerate(ys)]
