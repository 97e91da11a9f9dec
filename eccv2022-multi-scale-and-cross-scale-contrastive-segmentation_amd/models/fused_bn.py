"""Fused training-mode BatchNorm2d (+ residual add) (+ ReLU) on the HIP kernels of csrc/dcl_bn.hip.

``FusedBatchNorm2d`` IS an ``nn.BatchNorm2d`` (same parameters, buffers and state_dict keys, so
reference checkpoints load); called as ``bn(x)`` it behaves exactly like one.  Called as
``bn(x, residual=r, relu=True)`` it additionally folds the residual add and the ReLU that follow it in
the reference's blocks (models/HRNet.py:77-93, 118-137; Projector / transition / fuse layers) into the
same two HBM passes, forward and backward.  With ``sync=True`` the per-channel sums are all-reduced
across ranks between the statistics and the apply kernel (SyncBatchNorm semantics over RCCL).

The fused path runs for CUDA / float32 / contiguous NCHW tensors in training mode; evaluation mode and
CPU tensors use PyTorch's own batch_norm (running statistics, no batch reduction)."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from . import amax as _amax


def _stream():
    return _lib.stream_ptr()


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


_EQUAL_BATCH_CHECKED = set()
from ..debug import cfg as _dbg      # noqa: E402
_PACKED_RELU_MASK = _dbg.packed_relu_mask       # A/B switch for the tuning tools


def _check_equal_batch(n, device):
    """SyncBatchNorm path: the statistics are all-reduced as per-slice partial sums whose layout (and the element
    count N * H * W * world) assumes the SAME per-rank batch on every rank -- true for DistributedSampler with
    drop_last=True, which the managers use.  Verified once per batch size (one tiny all-reduce + host read) instead
    of being assumed: unequal batches would give wrong statistics or mismatched collective sizes."""
    if n in _EQUAL_BATCH_CHECKED:
        return
    import torch.distributed as dist
    t = torch.tensor([n, -n], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lo_hi = t.tolist()
    if lo_hi[0] != -lo_hi[1]:
        raise RuntimeError(f"FusedBatchNorm2d(sync): per-rank batch sizes differ across ranks (max {lo_hi[0]}, min "
                           f"{-lo_hi[1]}); use a sampler that gives every rank the same batch (drop_last=True)")
    _EQUAL_BATCH_CHECKED.add(n)


class _FusedBNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, weight, bias, running_mean, running_var, nbt, eps, momentum, relu, sync, amax,
                token=None, xmask=False, pre=None):
        """``pre`` (a list, filled with [sc, sh]): deferred mode -- statistics and their finalisation only; the output is an alias
        of x that the caller marks with amax.PreAct (relu, no residual)."""
        ctx.token = token
        ctx.xmask = bool(xmask)     # x is a ReLU output whose backward this norm's backward performs (relu_then_bn)
        L = _lib.lib()
        N, C, H, W = x.shape
        HW = H * W
        dev = x.device
        st = _stream()
        world = _world() if sync else 1
        count = float(N * HW * world)
        exch = None
        ns = L.dcl_bn_num_slices(N, C)
        # one workspace: [part C*ns*2 | mean C | invstd C | pivot C]; the per-slice partial sums are combined in the
        # prologue of the apply kernel (no combine / finalize launches); SyncBatchNorm = all-reduce of `part` in
        # between.  The sums are shifted by the running mean (identical on every rank) against cancellation.
        ws = torch.empty((C * ns * 2 + 3 * C,), dtype=torch.float32, device=dev)
        part, mean, invstd = ws[:C * ns * 2], ws[C * ns * 2:C * ns * 2 + C], ws[C * ns * 2 + C:C * ns * 2 + 2 * C]
        pivot = ws[C * ns * 2 + 2 * C:]
        if pre is not None:
            assert relu and res is None and not xmask and amax is not None
            # [mm C*ns*2 | sc C | sh C]: the slices' extrema of x; the map the consumer applies
            ws2 = torch.empty((C * ns * 2 + 2 * C,), dtype=torch.float32, device=dev)
            mm, sc, sh = ws2[:C * ns * 2], ws2[C * ns * 2:C * ns * 2 + C], ws2[C * ns * 2 + C:]
            if world == 1:
                # one launch: the last workgroup of a channel's statistics finalises the channel (csrc/dcl_bn.hip k_bn_stats_pre)
                _lib.check(L.dcl_bn_stats_pre(_lib.ptr(x), N, C, HW, _lib.ptr(part), _lib.ptr(mm), _lib.ptr(_amax.zeros(C, dev)),
                                              count, eps, momentum, _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(mean),
                                              _lib.ptr(invstd), _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(nbt),
                                              _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(amax), st), "dcl_bn_stats_pre")
            else:
                _lib.check(L.dcl_bn_stats_minmax_part(_lib.ptr(x), N, C, HW, _lib.ptr(part), _lib.ptr(mm), _lib.ptr(running_mean),
                                                      _lib.ptr(pivot), st), "dcl_bn_stats_minmax_part")
                _check_equal_batch(N, dev)
                exch = _all_reduce_async(part)
                exch.wait()
                _lib.check(L.dcl_bn_finalize_pre(_lib.ptr(part), _lib.ptr(mm), ns, count, eps, momentum, _lib.ptr(weight),
                                                 _lib.ptr(bias), C, _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(running_mean),
                                                 _lib.ptr(running_var), _lib.ptr(nbt), _lib.ptr(pivot), _lib.ptr(sc), _lib.ptr(sh),
                                                 _lib.ptr(amax), st), "dcl_bn_finalize_pre")
            pre += [sc, sh]
            ctx.save_for_backward(x, None, weight, bias, mean, invstd)
            ctx.packed_mask = False
            ctx.relu, ctx.world, ctx.count = True, world, count
            ctx.has_res = False
            ctx.emit_amax = True
            return x.detach()           # same storage: the consumer maps it on the fly (amax.PreAct)
        _lib.check(L.dcl_bn_stats_part(_lib.ptr(x), N, C, HW, _lib.ptr(part), _lib.ptr(running_mean), _lib.ptr(pivot),
                                       st), "dcl_bn_stats_part")
        if world > 1:
            _check_equal_batch(N, dev)
            exch = _all_reduce_async(part)
        y = torch.empty_like(x)
        # The backward needs y only for the ReLU mask.  Without a residual it recomputes y > 0 from x (one tensor less
        # to read, twice); with one, the apply kernel packs the sign bits (1/32 of y) and the backward reads those.
        need_y = relu and res is not None
        mask = torch.empty(N * C * HW // 64, dtype=torch.int64, device=dev) if need_y and HW % 256 == 0 and _PACKED_RELU_MASK else None
        if exch is not None:
            exch.wait()                 # stream-side wait, right before the consumer of the sums
        _lib.check(L.dcl_bn_apply_parts(_lib.ptr(x), _lib.ptr(res), _lib.ptr(part), ns, count, eps, momentum,
                                        _lib.ptr(weight), _lib.ptr(bias), N, C, HW, 1 if relu else 0, _lib.ptr(y),
                                        _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(running_mean),
                                        _lib.ptr(running_var), _lib.ptr(nbt), _lib.ptr(amax), _lib.ptr(pivot),
                                        _lib.ptr(mask), st),
                   "dcl_bn_apply_parts")
        ctx.save_for_backward(x, (mask if mask is not None else y) if need_y else None, weight, bias, mean, invstd)
        ctx.packed_mask = mask is not None
        ctx.relu, ctx.world, ctx.count = relu, world, count
        ctx.has_res = res is not None
        ctx.emit_amax = amax is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, y, weight, bias, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        HW = H * W
        dev = x.device
        dy = dy.contiguous()
        ns = L.dcl_bn_num_slices(N, C)
        part = torch.empty((C * ns * 2,), dtype=torch.float32, device=dev)
        st = _stream()
        relu = (2 if ctx.packed_mask else 1) if ctx.relu else 0         # 2: `y` is the packed sign mask
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev) if ctx.needs_input_grad[3] else None
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        _lib.check(L.dcl_bn_bwd_reduce_part(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), _lib.ptr(mean),
                                            _lib.ptr(invstd), _lib.ptr(weight), _lib.ptr(bias), N, C, HW, relu,
                                            _lib.ptr(part), st), "dcl_bn_bwd_reduce_part")
        # dx needs the sums over ALL ranks; dbeta / dgamma stay this rank's sums (DDP averages parameter gradients)
        part_all, exch = part, None
        if ctx.world > 1:
            part_all = part.clone()
            exch = _all_reduce_async(part_all)
        dx = torch.empty_like(x)
        want_res = ctx.has_res and (ctx.needs_input_grad[1] or ctx.token is not None)
        dres = torch.empty_like(x) if want_res else None
        # partial max|dx| (64 slots) for the consumer (the data / weight gradient of the convolution in front of this norm)
        amax = _amax.zeros(_amax.SLOTS, dev) if ctx.emit_amax else None
        if exch is not None:
            exch.wait()
        _lib.check(L.dcl_bn_bwd_apply_fused(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), _lib.ptr(mean),
                                            _lib.ptr(invstd), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(part_all),
                                            _lib.ptr(part), ctx.count, N, C, HW, relu + (4 if ctx.xmask else 0), _lib.ptr(dx),
                                            _lib.ptr(dres), _lib.ptr(dbeta), _lib.ptr(dgamma), _lib.ptr(amax), st),
                   "dcl_bn_bwd_apply_fused")
        if amax is not None:
            _amax.tag(dx, amax)
        if ctx.token is not None:
            # the residual's gradient travels through the token to the convolution that shares the input
            ctx.token.dres, dres = dres, None
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None


# SyncBatchNorm exchanges issued by this process, and how many of them made the HOST wait (tests / tools read it): on RCCL the
# wait of an all-reduce orders the calling STREAM behind the collective and returns at once -- 0 host waits; gloo (the CPU
# stand-in of the two-rank tests) completes on the host
COLLECTIVES = {"count": 0, "host_waits": 0}
DEFERRED = {"count": 0}           # norms whose output was not written (defer=True): tests / tools read it
FORCE_GROUP = False               # tests: take the grouped schedule on ONE rank too (the exchange itself is skipped)


class _Exchange:
    """An all-reduce in flight (reference semantics: nn.SyncBatchNorm over the process group, BaseManager.py:447-455).  Issued
    with ``async_op=True`` right behind the kernel that produced the partial sums: RCCL runs it on its own stream; ``wait()``
    -- called right before the kernel that consumes the sums, after the host-side preparation of that launch -- makes the
    calling stream wait for it with a stream event.  The host never blocks on RCCL."""
    __slots__ = ("work", "host")

    def __init__(self, t):
        import torch.distributed as dist
        COLLECTIVES["count"] += 1
        self.host = dist.get_backend() != "nccl"
        self.work = dist.all_reduce(t, async_op=True)

    def wait(self):
        if self.work is not None:
            if self.host:
                COLLECTIVES["host_waits"] += 1
            self.work.wait()
            self.work = None


def _all_reduce_async(t):
    return _Exchange(t)


def _all_reduce(t):
    _Exchange(t).wait()


class _FusedBNGroupFunction(torch.autograd.Function):
    """SyncBatchNorm of SEVERAL independent norm layers (the branches of an exchange module at one block depth) with ONE
    stacked statistics exchange per direction: every member's per-slice partial sums land in one buffer, one all-reduce
    moves it, every member's apply kernel reads its slice.  Same kernels and arithmetic as ``_FusedBNFunction``; member
    k's kernels run on ``meta[k]['stream']`` (its branch's HIP stream), the collective on the calling stream, with the
    joins / forks in between.  A single autograd node: in the backward it runs once all members' gradients are there."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        L = _lib.lib()
        nm = len(meta)
        xs, ress, ws, bs = tensors[0::4], tensors[1::4], tensors[2::4], tensors[3::4]
        dev = xs[0].device
        main = torch.cuda.current_stream(dev)
        world = _world() if any(m['sync'] for m in meta) else 1
        if world > 1:
            _check_equal_batch(xs[0].shape[0], dev)
        sizes = [x.shape[1] * L.dcl_bn_num_slices(x.shape[0], x.shape[1]) * 2 for x in xs]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        stack = torch.empty((offs[-1],), dtype=torch.float32, device=dev)
        saved, ys, masks = [], [], []
        for k, m in enumerate(meta):
            st = m['stream'] or main
            x = xs[k]
            N, C, H, W = x.shape
            with torch.cuda.stream(st):
                if st is not main:
                    stack.record_stream(st)
                wsb = torch.empty((3 * C,), dtype=torch.float32, device=dev)
                m['ws'] = wsb
                if m.get('defer'):
                    # the member's output is not written (see _FusedBNFunction, ``pre``): statistics with the slices' extrema
                    ns = L.dcl_bn_num_slices(N, C)
                    m['ws2'] = torch.empty((C * ns * 2 + 2 * C,), dtype=torch.float32, device=dev)
                    _lib.check(L.dcl_bn_stats_minmax_part(_lib.ptr(x), N, C, H * W, _lib.ptr(stack[offs[k]:]), _lib.ptr(m['ws2']),
                                                          _lib.ptr(m['running_mean']), _lib.ptr(wsb[2 * C:]), _lib.stream_ptr(dev)),
                               "dcl_bn_stats_minmax_part")
                else:
                    _lib.check(L.dcl_bn_stats_part(_lib.ptr(x), N, C, H * W, _lib.ptr(stack[offs[k]:]),
                                                   _lib.ptr(m['running_mean']), _lib.ptr(wsb[2 * C:]), _lib.stream_ptr(dev)),
                               "dcl_bn_stats_part")
        for m in meta:
            if m['stream'] is not None and m['stream'] is not main:
                main.wait_stream(m['stream'])
        if world > 1:
            _all_reduce(stack)
        for k, m in enumerate(meta):
            st = m['stream'] or main
            x, res = xs[k], ress[k]
            N, C, H, W = x.shape
            HW = H * W
            relu = m['relu']
            if st is not main:
                st.wait_stream(main)
            if m.get('defer'):
                with torch.cuda.stream(st):
                    wsb, ws2 = m['ws'], m['ws2']
                    ns = L.dcl_bn_num_slices(N, C)
                    sc, sh = ws2[C * ns * 2:C * ns * 2 + C], ws2[C * ns * 2 + C:]
                    _lib.check(L.dcl_bn_finalize_pre(_lib.ptr(stack[offs[k]:]), _lib.ptr(ws2), ns, float(N * HW * world), m['eps'],
                                                     m['momentum'], _lib.ptr(ws[k]), _lib.ptr(bs[k]), C, _lib.ptr(wsb[:C]),
                                                     _lib.ptr(wsb[C:2 * C]), _lib.ptr(m['running_mean']), _lib.ptr(m['running_var']),
                                                     _lib.ptr(m['nbt']), _lib.ptr(wsb[2 * C:]), _lib.ptr(sc), _lib.ptr(sh),
                                                     _lib.ptr(m['amax']), _lib.stream_ptr(dev)), "dcl_bn_finalize_pre")
                    m['pre'] = (sc, sh)
                ys.append(x.detach())
                saved += [x, None, ws[k], bs[k], wsb]
                masks.append(False)
                continue
            with torch.cuda.stream(st):
                wsb = m['ws']
                y = torch.empty_like(x)
                need_y = relu and res is not None
                mask = torch.empty(N * C * HW // 64, dtype=torch.int64, device=dev) \
                    if need_y and HW % 256 == 0 and _PACKED_RELU_MASK else None
                _lib.check(L.dcl_bn_apply_fused(_lib.ptr(x), _lib.ptr(res), _lib.ptr(stack[offs[k]:]),
                                                float(N * HW * world), m['eps'], m['momentum'], _lib.ptr(ws[k]),
                                                _lib.ptr(bs[k]), N, C, HW, 1 if relu else 0, _lib.ptr(y),
                                                _lib.ptr(wsb[:C]), _lib.ptr(wsb[C:2 * C]), _lib.ptr(m['running_mean']),
                                                _lib.ptr(m['running_var']), _lib.ptr(m['nbt']), _lib.ptr(m['amax']),
                                                _lib.ptr(wsb[2 * C:]), _lib.ptr(mask), _lib.stream_ptr(dev)),
                           "dcl_bn_apply_fused")
            ys.append(y)
            saved += [x, (mask if mask is not None else y) if need_y else None, ws[k], bs[k], wsb]
            masks.append(mask is not None)
        ctx.save_for_backward(*saved)
        ctx.meta, ctx.masks, ctx.world = meta, masks, world
        ctx.has_res = [r is not None for r in ress]
        ctx.keep = stack
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        L = _lib.lib()
        meta = ctx.meta
        sv = ctx.saved_tensors
        dev = sv[0].device
        main = torch.cuda.current_stream(dev)
        nm = len(meta)
        sizes = [sv[5 * k].shape[1] * L.dcl_bn_num_slices(sv[5 * k].shape[0], sv[5 * k].shape[1]) * 2 for k in range(nm)]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        local = torch.empty((offs[-1],), dtype=torch.float32, device=dev)
        dys = [dy.contiguous() for dy in dys]
        relus = []
        for k, m in enumerate(meta):
            st = m['stream'] or main
            x, ym, w, b, wsb = sv[5 * k:5 * k + 5]
            N, C, H, W = x.shape
            relu = (2 if ctx.masks[k] else 1) if m['relu'] else 0
            relus.append(relu)
            if st is not main:
                st.wait_stream(main)               # the engine has made `main` wait for the producers of dys
                local.record_stream(st)
                dys[k].record_stream(st)
            with torch.cuda.stream(st):
                _lib.check(L.dcl_bn_bwd_reduce_part(_lib.ptr(dys[k]), _lib.ptr(x), _lib.ptr(ym), _lib.ptr(wsb[:C]),
                                                    _lib.ptr(wsb[C:2 * C]), _lib.ptr(w), _lib.ptr(b), N, C, H * W, relu,
                                                    _lib.ptr(local[offs[k]:]), _lib.stream_ptr(dev)),
                           "dcl_bn_bwd_reduce_part")
        for m in meta:
            if m['stream'] is not None and m['stream'] is not main:
                main.wait_stream(m['stream'])
        total = local
        if ctx.world > 1:
            total = local.clone()
            _all_reduce(total)
        grads = []
        for k, m in enumerate(meta):
            st = m['stream'] or main
            x, ym, w, b, wsb = sv[5 * k:5 * k + 5]
            N, C, H, W = x.shape
            tok = m.get('token')
            if st is not main:
                st.wait_stream(main)
                total.record_stream(st)
            with torch.cuda.stream(st):
                dx = torch.empty_like(x)
                want_res = ctx.has_res[k] and (ctx.needs_input_grad[1 + 4 * k + 1] or tok is not None)
                dres = torch.empty_like(x) if want_res else None
                dgamma = torch.empty((C,), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1 + 4 * k + 2] else None
                dbeta = torch.empty((C,), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1 + 4 * k + 3] else None
                amax = _amax.zeros(_amax.SLOTS, dev) if m['amax'] is not None else None
                _lib.check(L.dcl_bn_bwd_apply_fused(_lib.ptr(dys[k]), _lib.ptr(x), _lib.ptr(ym), _lib.ptr(wsb[:C]),
                                                    _lib.ptr(wsb[C:2 * C]), _lib.ptr(w), _lib.ptr(b),
                                                    _lib.ptr(total[offs[k]:]), _lib.ptr(local[offs[k]:]),
                                                    float(N * H * W * ctx.world), N, C, H * W, relus[k], _lib.ptr(dx),
                                                    _lib.ptr(dres), _lib.ptr(dbeta), _lib.ptr(dgamma), _lib.ptr(amax),
                                                    _lib.stream_ptr(dev)), "dcl_bn_bwd_apply_fused")
                if amax is not None:
                    _amax.tag(dx, amax)
                if tok is not None:
                    tok.dres, dres = dres, None
            if st is not main:                      # the node's results belong to `main` as far as the engine knows
                main.wait_stream(st)
                for t in (dx, dres, dgamma, dbeta, amax):
                    if t is not None:
                        t.record_stream(main)
            grads += [dx, dres, dgamma, dbeta]
        return (None, *grads)


def bn_act_group(bns, xs, residuals=None, relu=True, tokens=None, streams=None, defers=None):
    """``[bn_act(bn_k, x_k, residual_k, relu, token_k)]`` for independent FusedBatchNorm2d layers in SyncBatchNorm mode
    with one stacked statistics exchange per direction instead of one per layer (see _FusedBNGroupFunction); member k's
    kernels run on ``streams[k]`` (None: the current stream), its input must have been produced there.  ``defers[k]``: member k's
    output is not written (FusedBatchNorm2d.forward, ``defer``)."""
    n = len(bns)
    defers = defers or [False] * n
    residuals = residuals or [None] * n
    tokens = tokens or [None] * n
    streams = streams or [None] * n
    meta, tensors = [], []
    for k, bn in enumerate(bns):
        amax = None
        if bn.emit_amax or defers[k]:
            st = streams[k]
            if st is not None:
                with torch.cuda.stream(st):
                    amax = _amax.zeros(_amax.SLOTS, xs[k].device)
            else:
                amax = _amax.zeros(_amax.SLOTS, xs[k].device)
        meta.append(dict(stream=streams[k], sync=bool(bn.sync), running_mean=bn.running_mean, running_var=bn.running_var,
                         nbt=bn.num_batches_tracked, eps=float(bn.eps), momentum=float(bn.momentum), relu=bool(relu),
                         amax=amax, token=tokens[k] if residuals[k] is not None else None,
                         defer=bool(defers[k]) and relu and residuals[k] is None))
        tensors += [xs[k], residuals[k], bn.weight, bn.bias]
    ys = _FusedBNGroupFunction.apply(meta, *tensors)
    out = []
    for k, y in enumerate(ys):
        if meta[k]['amax'] is not None:
            _amax.tag(y, meta[k]['amax'])
        if meta[k].get('pre') is not None:
            DEFERRED["count"] += 1
            y._dcl_pre = _amax.PreAct(meta[k]['pre'][0], meta[k]['pre'][1], meta[k]['amax'], y._version)
        out.append(y)
    return out


def can_group_static(bns):
    """The part of ``can_group`` that depends on the MODEL and the process group only (identical on every rank): the decision
    which collective schedule a module takes must not depend on per-rank tensor properties -- ranks that disagree would
    issue different all-reduce sequences and hang."""
    if len(bns) < 2 or not _dbg.coalesced_sync_bn:
        return False
    # (the conditions of FusedBatchNorm2d._fusable that depend on the MODEL only -- affine, running statistics, a fixed
    # momentum, no autocast -- belong here too: a model that fails them takes the per-norm schedule on every rank instead of
    # raising "not a contiguous float32 CUDA tensor" from the stacked one)
    if not all(isinstance(bn, FusedBatchNorm2d) and bn.training and bn.affine and bn.track_running_stats
               and bn.momentum is not None for bn in bns) or torch.is_autocast_enabled():
        return False
    return FORCE_GROUP or (_world() >= 2 and all(bn.sync for bn in bns))


def can_group(bns, xs, residuals=None):
    """True when ``bn_act_group`` applies: several fused norms in SyncBatchNorm mode on more than one rank."""
    if len(bns) < 2 or not _dbg.coalesced_sync_bn:
        return False
    if not all(isinstance(bn, FusedBatchNorm2d) for bn in bns):
        return False
    if not FORCE_GROUP and (_world() < 2 or not all(bn.sync for bn in bns)):
        return False
    residuals = residuals or [None] * len(bns)
    return all(bn._fusable(x, r) for bn, x, r in zip(bns, xs, residuals))


class FusedBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d with an optional fused (residual add, ReLU) epilogue; see module docstring."""
    sync = False
    emit_amax = True

    def _fusable(self, x, residual):
        return (self.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and x.is_contiguous() and self.affine and self.track_running_stats
                and self.momentum is not None and not torch.is_autocast_enabled()
                and (residual is None or (residual.shape == x.shape and residual.is_contiguous()
                                          and residual.dtype == torch.float32)))

    def forward(self, x, residual=None, relu=False, grad_token=None, input_relu=False, defer=False):
        """``input_relu`` (relu_then_bn only): x is the output of a ReLU whose backward is left to this norm's backward kernel.
        ``defer`` (relu, no residual; the caller has asked the consuming convolution, DirectConv2d.fuses_input_norm): the
        normalised tensor is not written -- the result aliases x and carries the map as an amax.PreAct mark."""
        if defer:
            if not (relu and residual is None and not input_relu and self._fusable(x, None)):
                raise RuntimeError("FusedBatchNorm2d(defer=True): needs relu, no residual and an input of the fused path")
            amax = _amax.zeros(_amax.SLOTS, x.device)
            DEFERRED["count"] += 1
            pre = []
            y = _FusedBNFunction.apply(x, None, self.weight, self.bias, self.running_mean, self.running_var,
                                       self.num_batches_tracked, float(self.eps), float(self.momentum), True, bool(self.sync),
                                       amax, None, False, pre)
            _amax.tag(y, amax)
            y._dcl_pre = _amax.PreAct(pre[0], pre[1], amax, y._version)
            return y
        if self._fusable(x, residual):
            # partial max|y| side output (64 slots) for the f16x3 convolutions that consume y (models/amax.py)
            amax = _amax.zeros(_amax.SLOTS, x.device) if self.emit_amax else None
            y = _FusedBNFunction.apply(x, residual, self.weight, self.bias, self.running_mean,
                                       self.running_var, self.num_batches_tracked, float(self.eps),
                                       float(self.momentum), bool(relu), bool(self.sync), amax,
                                       grad_token if residual is not None else None, bool(input_relu))
            return _amax.tag(y, amax) if amax is not None else y
        assert not input_relu, "input_relu is only valid on the fused path (relu_then_bn checks it)"
        if self.sync and self.training and _world() > 1:
            # convert_sync_batchnorm only flips the flag of a FusedBatchNorm2d: nn.BatchNorm2d.forward below would
            # silently normalise with THIS rank's statistics
            raise RuntimeError("FusedBatchNorm2d(sync=True): input not supported by the fused SyncBatchNorm path "
                               f"(shape {tuple(x.shape)}, dtype {x.dtype}, device {x.device}, contiguous "
                               f"{x.is_contiguous()}); it needs contiguous float32 NCHW CUDA tensors outside autocast")
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y, inplace=True) if relu else y


class _ReluMaskedDownstream(torch.autograd.Function):
    """In-place ReLU whose backward is the IDENTITY: only for relu_then_bn, where the norm behind it zeroes its input gradient
    wherever this output is 0 (dcl_bn_bwd_apply_fused, relu + 4) -- the three-tensor threshold_backward pass disappears."""

    @staticmethod
    def forward(ctx, t):
        ctx.mark_dirty(t)
        return t.relu_()

    @staticmethod
    def backward(ctx, g):
        return g


def relu_then_bn(bn, t):
    """bn(relu_(t)): the order of the projection heads' hidden layers (reference models/Projector.py:46-51: conv -> ReLU -> BN).
    On the fused training path of a single norm the ReLU's backward rides in the norm's backward kernel; else the two modules run
    as they are.  ``t`` is overwritten (as nn.ReLU(inplace=True) does)."""
    if (isinstance(bn, FusedBatchNorm2d) and bn._fusable(t, None) and torch.is_grad_enabled() and t.requires_grad
            and _dbg.relu_then_bn and not (FORCE_GROUP)):
        return bn(_ReluMaskedDownstream.apply(t), input_relu=True)
    return bn(F.relu(t, inplace=True))


def bn_act(bn, x, residual=None, relu=True, grad_token=None, defer=False):
    """norm (+ residual) (+ ReLU) for any norm layer; one fused call when ``bn`` supports it.  ``grad_token``
    (models/ops.py GradToken): hand the residual's gradient to the convolution that shares the residual tensor.
    ``defer``: see FusedBatchNorm2d.forward."""
    if isinstance(bn, FusedBatchNorm2d):
        return bn(x, residual=residual, relu=relu, grad_token=grad_token, defer=defer)
    assert not defer
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y, inplace=True) if relu else y


def convert_sync_batchnorm(module: nn.Module) -> nn.Module:
    """SyncBatchNorm conversion that keeps ``FusedBatchNorm2d`` modules (it switches their ``sync`` flag
    on: statistics are all-reduced inside the fused op) and converts every other BatchNorm through
    torch.nn.SyncBatchNorm.convert_sync_batchnorm (reference: BaseManager.py:450-451)."""
    if isinstance(module, FusedBatchNorm2d):
        module.sync = True
        return module
    if isinstance(module, nn.modules.batchnorm._BatchNorm):
        return nn.SyncBatchNorm.convert_sync_batchnorm(module)
    for name, child in module.named_children():
        new = convert_sync_batchnorm(child)
        if new is not child:
            setattr(module, name, new)
    return module
