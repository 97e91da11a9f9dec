"""Device orchestration of the dense contrastive loss: one ``torch.autograd.Function`` that runs the
whole multi-scale / cross-scale loss (forward and backward) through libdcl_hip.so.

torch is used for device memory (caching allocator), the current HIP stream and autograd
plumbing only; every compute step is a C-ABI call into the HIP library (include/dcl_hip.h).
There is no fallback path: CPU tensors or a missing library raise.

Step -> kernel -> reference lines replaced
  K1 dcl_label_hist            F.interpolate nearest + one-hot counts (DenseContrastiveLossV2.py:205, :100-103)
  (host) build_host_plan       pair selection, V, torch.randperm per pair (:106-110, :64-84, :121)
  K2 dcl_rank_select           nonzero + perm[:V] indexing (:119-122)
  K3 dcl_gather_normalize      features[b,:,idx], F.normalize, transpose/view (:123, :138-149)
  K4 dcl_infonce_fwd           matmul/div, masks, get_loss / InfoNce_loss (:150-192; ms:84-161)
  K5 dcl_infonce_prep_stats/bwd  autograd of the above
  K6 dcl_normalize_bwd_scatter   autograd of normalize + index (T x IndexBackward in the reference)

Since round 6 one module per stage: engine_state (records, helpers), engine_plan (labels -> plan -> sampled pixels),
engine_banks (banks, shared negative bank); this module holds the InfoNCE terms, the autograd node and the stream-K bookkeeping
and re-exports the others, so ``engine.X`` keeps working.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..utils import printlog

from .. import _lib
from ..models import amax as _amax
from .plan import HostPlan, build_host_plan, positive_ranges
from .engine_state import *  # noqa: F401,F403
from .engine_plan import *  # noqa: F401,F403
from .engine_banks import *  # noqa: F401,F403


KEEP_POSITIVES_MAX_SPAN = 2048     # floats per row of the kept-positives buffer (N1pad x span x 4 B: 80 MB at N1 = 9 804)


def _keep_positives(t) -> bool:
    """Single-segment term whose widest positive range is small enough to keep the positives' similarities in memory: the forward
    is then ONE sweep over the bank (dcl_infonce_zsweep_keep + dcl_infonce_pos_finish) instead of the Z and POS sweeps.  A few
    classes with thousands of views each (blocky labels) keep the two-sweep form: the buffer would approach the N x N matrix."""
    from ..debug import cfg as _dbg
    # (max_span is the host plan's figure for the rank-local ranges: only the default own segment qualifies)
    return (_dbg.keep_positives and len(t.segs) == 1 and t.segs[0].rng_lo is t.rng_lo and t.pcount is None
            and 0 < t.max_span <= KEEP_POSITIVES_MAX_SPAN)


def run_forward_terms(st: StepState):
    """K4 for every term over its contrast segments; raw term losses land in st.loss_buf[idx]."""
    L = _lib.lib()
    stream = _stream_ptr()
    dev = st.scales[0].bank.device
    st.loss_buf = torch.empty((len(st.terms),), dtype=torch.float32, device=dev)
    for idx, t in enumerate(st.terms):
        A = st.scales[t.a]
        N1, V1 = A.plan.N, A.plan.V
        N1pad = _npad(N1)
        zs_total = sum(sg.nsplit for sg in t.segs)
        t.nsplit = zs_total
        zpart = torch.empty((zs_total * N1pad,), dtype=torch.float32, device=dev)
        t.Z = torch.empty((N1pad,), dtype=torch.float32, device=dev)
        t.W = torch.empty((N1pad,), dtype=torch.float32, device=dev)
        rowloss = torch.empty((N1pad,), dtype=torch.float32, device=dev)
        inv_tau = 1.0 / t.tau
        if _keep_positives(t):
            # one pass over the bank: the Z sweep keeps the positives' similarities, a row-wise kernel finishes them
            sg = t.segs[0]
            ld = (t.max_span + 3) & ~3
            spos = torch.empty((N1pad, ld), dtype=torch.float32, device=dev)
            ah = A.bank_h if sg.bank_h is not None else None
            _lib.check(L.dcl_infonce_zsweep_keep(_lib.ptr(A.bank), N1, V1, _lib.ptr(sg.bank), sg.N, _lib.ptr(sg.rng_lo),
                                                 _lib.ptr(sg.rng_hi), inv_tau, sg.nsplit, _lib.ptr(zpart), _lib.ptr(ah),
                                                 _lib.ptr(sg.bank_h), _lib.ptr(spos), ld, stream), "dcl_infonce_zsweep_keep")
            _lib.check(L.dcl_infonce_pos_finish(_lib.ptr(spos), ld, N1, V1, _lib.ptr(sg.rng_lo), _lib.ptr(sg.rng_hi), inv_tau,
                                                1 if (t.intra and sg.own) else 0, 1 if sg.bank_h is not None else 0,
                                                _lib.ptr(zpart), zs_total, _lib.ptr(t.Z), _lib.ptr(rowloss), _lib.ptr(t.W),
                                                stream), "dcl_infonce_pos_finish")
            _lib.check(L.dcl_infonce_loss(_lib.ptr(rowloss), _lib.ptr(sg.rng_lo), _lib.ptr(sg.rng_hi),
                                          _lib.ptr(t.pcount), N1, V1, 1 if t.intra else 0,
                                          _lib.ptr(st.loss_buf[idx:]), stream), "dcl_infonce_loss")
            st.keepalive += [zpart, rowloss, spos]
            continue
        off = 0
        for sg in t.segs:
            ah = A.bank_h if sg.bank_h is not None else None
            _lib.check(L.dcl_infonce_zsweep(_lib.ptr(A.bank), N1, V1, _lib.ptr(sg.bank), sg.N,
                                            _lib.ptr(sg.rng_lo), _lib.ptr(sg.rng_hi), inv_tau, sg.nsplit,
                                            _lib.ptr(zpart[off * N1pad:]), _lib.ptr(ah), _lib.ptr(sg.bank_h),
                                            stream), "dcl_infonce_zsweep")
            off += sg.nsplit
        for k, sg in enumerate(t.segs):
            _lib.check(L.dcl_infonce_possweep(_lib.ptr(A.bank), N1, V1, _lib.ptr(sg.bank), sg.N,
                                              _lib.ptr(sg.rng_lo), _lib.ptr(sg.rng_hi), inv_tau,
                                              1 if (t.intra and sg.own) else 0, _lib.ptr(zpart), zs_total,
                                              0 if k == 0 else 1, _lib.ptr(t.Z), _lib.ptr(rowloss),
                                              _lib.ptr(t.W),
                                              _lib.ptr(A.bank_h if sg.bank_h is not None else None),
                                              _lib.ptr(sg.bank_h), stream), "dcl_infonce_possweep")
        first = t.segs[0]
        _lib.check(L.dcl_infonce_loss(_lib.ptr(rowloss), _lib.ptr(first.rng_lo), _lib.ptr(first.rng_hi),
                                      _lib.ptr(t.pcount), N1, V1, 1 if t.intra else 0,
                                      _lib.ptr(st.loss_buf[idx:]), stream), "dcl_infonce_loss")
        st.keepalive += [zpart, rowloss]


class DenseContrastFunction(torch.autograd.Function):
    """loss_terms = f(features...) with a hand-written backward (K5/K6)."""

    @staticmethod
    def forward(ctx, cfg: EngineConfig, label: torch.Tensor, holder: dict, *feats):
        with_cross = bool(cfg.cross_scale_contrast) and len(feats) > 1
        f16x3 = cfg.mfma == "f16x3"
        peers = holder.get("emulated_peers")
        shared = peers is None and cfg.global_negatives and _dist_world() > 1
        if shared:
            # planning can fail on ONE rank (no qualifying pair on its shard): agree before any collective
            err = None
            try:
                st = plan_and_sample(cfg, label, feats, with_cross, staged=holder.get("staged"))
            except Exception as e:  # noqa: BLE001
                err = e
            agree_or_raise(err, feats[0].device)
            gather = _BankGather(st, cfg.max_features_total, f16x3)
            build_banks(st, feats, f16x3=f16x3, gather=gather)     # gathers issued per scale, behind its K3
            rank, pb, pbh, lay = gather.finish()                   # waited for just before the first sweep
            attach_global_segments(st, rank, pb, lay, peer_banks_h=pbh)
        else:
            st = holder.get("st")               # planned ahead: the features are LazyProjection rows of this plan
            if st is None:
                st = plan_and_sample(cfg, label, feats, with_cross, staged=holder.get("staged"))
            build_banks(st, feats, f16x3=f16x3)
            if peers is not None:                   # single-process emulation of other ranks (tests)
                rank, peer_banks, peer_layouts = peers[:3]
                peer_banks_h = peers[3] if len(peers) > 3 else None
                peer_layouts = list(peer_layouts)
                peer_layouts[rank] = [class_layout(sc.plan) for sc in st.scales]
                attach_global_segments(st, rank, peer_banks, peer_layouts, peer_banks_h=peer_banks_h)
            else:
                _own_segments(st)
        run_forward_terms(st)
        ctx.st = st
        ctx.feats_meta = [(tuple(f.shape), tuple(f.stride()), f.dtype) for f in feats]
        holder["state"] = st
        return st.loss_buf.clone()

    @staticmethod
    def backward(ctx, grad_terms):
        st = ctx.st
        need = list(ctx.needs_input_grad[3:])
        grads = _backward_with_term_grads(st, grad_terms, ctx.feats_meta, need)
        return (None, None, None, *grads)


_SK_WS = {}


class StreamKTimeout(RuntimeError):
    """A stream-K backward launch gave up waiting for a partial tile (see dcl_infonce_bwd_streamk): its gradient was invalid."""


def _streamk_workspace(dev, G: int):
    """(partial-tile workspace f32 [G, 128, 256], flags int32 [G + 1]) of the stream-K backward for the CURRENT stream of
    ``dev``.  One persistent pair per (device, stream) serves every term of every step: the launches of one stream run one
    after the other and a flag is only valid for the launch whose number it carries (nothing has to be reset).  flags[0] is
    the error word (hand-overs that timed out; a FIXED word, whatever the grid size of the launch that counts into it): a copy
    of it travels to a pinned host word after every backward pass (``_streamk_note_launches``).  ``streamk_check`` -- the
    managers' optimizer pre-step hook -- is what acts on it.  On ONE rank the word is also looked at, without waiting, when the
    workspace is asked for the next time (for callers without the hook): a non-zero value switches stream-K off for the
    process and raises, because the gradients of that earlier step were wrong.  On several ranks that second look is left out:
    it depends on this rank's timing, and a rank that raises out of its backward pass while its peers go on leaves them in
    a collective without a partner -- there the all-rank maximum and the pre-step hook decide for everybody."""
    key = (dev.index, _lib.stream_ptr(dev))
    got = _SK_WS.get(key)
    if got is not None and _dist_world() == 1:
        ev, host = got[2], got[3]
        if ev is not None and ev.query() and int(host[0]) != 0:
            _lib.lib().dcl_infonce_set_streamk(0)
            _SK_WS.pop(key, None)
            raise StreamKTimeout(
                f"stream-K backward: {int(host[0])} hand-over(s) between persistent workgroups timed out in an earlier step "
                "(a contributor workgroup never became resident: CU mask, another persistent kernel, several ranks on one "
                "device?).  That step's feature gradients were invalid.  Stream-K is now OFF for this process "
                "(dcl_infonce_set_streamk(0)): later steps use the column-split backward; set DCL_SWEEP_STREAMK=0 to start that way.")
    if got is None or got[1].numel() < G + 1:
        got = [torch.empty((G, _lib.ROW_TILE, _lib.CP), dtype=torch.float32, device=dev),
               torch.zeros((G + 1,), dtype=torch.int32, device=dev), None if got is None else got[2],
               torch.zeros((1,), dtype=torch.int32).pin_memory() if got is None else got[3], G]
        _SK_WS[key] = got
    return got[0], got[1]


def _streamk_note_launches(dev):
    """After the stream-K launches of a backward pass: asynchronous copy of the error word to the host + an event.  On several
    ranks the word is first all-reduced (MAX, 4 bytes, asynchronous): one rank's invalid gradient reaches every rank through the
    gradient all-reduce, so every rank has to see the error (``streamk_check`` then raises on all of them, not on one whose
    peers would hang in their next collective)."""
    key = (dev.index, _lib.stream_ptr(dev))
    got = _SK_WS.get(key)
    world = _dist_world()
    if got is None:
        if world == 1:
            return
        # several ranks: EVERY rank takes part in the exchange of the word in EVERY backward pass, also one that launched no
        # stream-K kernel (switched off on it, no term with a gradient) -- a collective that only some ranks issue pairs up
        # with the peers' next gradient bucket.  Its word is zero.
        got = [None, torch.zeros((1,), dtype=torch.int32, device=dev), None, torch.zeros((1,), dtype=torch.int32).pin_memory(), 0]
        _SK_WS[key] = got
    word = got[1][0:1]
    if world > 1:
        import torch.distributed as dist
        word = word.clone()
        work = dist.all_reduce(word, op=dist.ReduceOp.MAX, async_op=True)
        work.wait()                             # (orders the current stream behind the collective; does not block the host on RCCL)
    got[3].copy_(word, non_blocking=True)
    ev = got[2] or torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    got[2] = ev


def streamk_check(wait: bool = True):
    """Raise ``StreamKTimeout`` if a stream-K backward launch of THIS backward pass (or an earlier one) reported a timed-out
    hand-over -- to be called after ``loss.backward()`` and BEFORE the gradients are consumed: the managers register it as a
    pre-step hook of their optimizer (BaseManager.load_optimiser), so an invalid gradient is never applied.  ``wait``: wait for
    the event behind the error word's copy; that copy sits right behind the loss's sweeps at the very START of the backward
    pass, so by the time the host has enqueued the rest of the backward it has normally completed (no stall in a GPU-bound
    step).  The error switches stream-K off for the process (column-split ``dcl_infonce_bwd`` from then on), so repeating the
    step is safe.  On several ranks the word is the maximum over the ranks: all of them raise."""
    for key, got in list(_SK_WS.items()):
        ev, host = got[2], got[3]
        if ev is None:
            continue
        if wait:
            ev.synchronize()
        elif not ev.query():
            continue
        n = int(host[0])
        if n != 0:
            _lib.lib().dcl_infonce_set_streamk(0)
            _SK_WS.pop(key, None)
            raise StreamKTimeout(
                f"stream-K backward: {n} hand-over(s) between persistent workgroups timed out (a contributor workgroup never "
                "became resident: CU mask, another persistent kernel, several ranks on one device?).  The feature gradients of "
                "this backward pass are invalid and must not be applied.  Stream-K is now OFF for this process "
                "(dcl_infonce_set_streamk(0)): repeat the step -- it uses the column-split backward; set DCL_SWEEP_STREAMK=0 to "
                "start that way.")


def _backward_with_term_grads(st: StepState, grad_terms: torch.Tensor, feats_meta, need):
    """The Function outputs the RAW per-term losses; the module forms the weighted sum with torch
    ops, so autograd hands us d total / d term_i = weight_i * upstream.  Each term's kernels read
    their own element of that vector as the device-side scale."""
    L = _lib.lib()
    stream = _stream_ptr()
    dev = st.scales[0].bank.device
    g = grad_terms.detach().to(device=dev, dtype=torch.float32).contiguous()
    slabs: List[List[torch.Tensor]] = [[] for _ in st.scales]
    used_streamk = False
    for idx, t in enumerate(st.terms):
        A, B = st.scales[t.a], st.scales[t.b]
        N1, N2 = A.plan.N, B.plan.N
        N1pad, N2pad = _npad(N1), _npad(N2)
        want_a = need[t.a]
        want_b = (not t.intra) and need[t.b] and not t.detach_b
        if not (want_a or want_b):
            continue
        inv_tau = 1.0 / t.tau
        first = t.segs[0]
        stat = torch.empty((N1pad + 1, 4), dtype=torch.float32, device=dev)  # + bound row (dcl_hip.h)
        _lib.check(L.dcl_infonce_prep_stats(_lib.ptr(t.Z), _lib.ptr(t.W), _lib.ptr(first.rng_lo),
                                            _lib.ptr(first.rng_hi), _lib.ptr(t.pcount), N1, A.plan.V,
                                            1 if t.intra else 0, 1.0, inv_tau, _lib.ptr(g[idx:]),
                                            _lib.ptr(stat), stream), "dcl_infonce_prep_stats")
        if want_a:
            for sg in t.segs:
                sym = t.intra and sg.own            # own rows are anchors AND contrast columns
                G = int(L.dcl_infonce_bwd_streamk_workgroups(N1, sg.N)) if sg.bank_h is not None else 0
                if G > 0:
                    # f16x3: stream-K partition, ONE finished [N1pad, 256] tile array per launch instead of nsplit slabs
                    nsl = int(L.dcl_infonce_bwd_streamk_slabs(N1, sg.N))           # one finished slab per column slice
                    dout = torch.empty((nsl, N1pad, _lib.CP), dtype=torch.float32, device=dev)
                    ws, flags = _streamk_workspace(dev, G)
                    used_streamk = True
                    _lib.check(L.dcl_infonce_bwd_streamk(_lib.ptr(A.bank), N1, A.plan.V, _lib.ptr(sg.bank), sg.N,
                                                         _lib.ptr(sg.rng_lo), _lib.ptr(sg.rng_hi), inv_tau,
                                                         1 if sym else 0, 1, 1 if sym else 0, _lib.ptr(stat),
                                                         _lib.ptr(stat) if sym else None, _lib.ptr(dout), _lib.ptr(ws),
                                                         _lib.ptr(flags), _lib.ptr(A.bank_h), _lib.ptr(sg.bank_h), stream),
                               "dcl_infonce_bwd_streamk")
                    slabs[t.a] += [dout[i] for i in range(nsl)]
                    continue
                dpart = torch.empty((sg.nsplit, N1pad, _lib.CP), dtype=torch.float32, device=dev)
                _lib.check(L.dcl_infonce_bwd(_lib.ptr(A.bank), N1, A.plan.V, _lib.ptr(sg.bank), sg.N,
                                             _lib.ptr(sg.rng_lo), _lib.ptr(sg.rng_hi), inv_tau,
                                             1 if sym else 0, 1, 1 if sym else 0, _lib.ptr(stat),
                                             _lib.ptr(stat) if sym else None, sg.nsplit,
                                             _lib.ptr(dpart),
                                             _lib.ptr(A.bank_h if sg.bank_h is not None else None),
                                             _lib.ptr(sg.bank_h), stream), "dcl_infonce_bwd")
                slabs[t.a] += [dpart[i] for i in range(sg.nsplit)]
        if want_b:
            # G^T F1 restricted to this rank's rows of bank b (they are columns of the own segment)
            G = int(L.dcl_infonce_bwd_streamk_workgroups(N2, N1)) if (B.bank_h is not None and A.bank_h is not None) else 0
            if G > 0:
                nsl = int(L.dcl_infonce_bwd_streamk_slabs(N2, N1))
                dout = torch.empty((nsl, N2pad, _lib.CP), dtype=torch.float32, device=dev)
                ws, flags = _streamk_workspace(dev, G)
                used_streamk = True
                _lib.check(L.dcl_infonce_bwd_streamk(_lib.ptr(B.bank), N2, B.plan.V, _lib.ptr(A.bank), N1,
                                                     _lib.ptr(t.rev_lo), _lib.ptr(t.rev_hi), inv_tau, 0, 0, 1, None,
                                                     _lib.ptr(stat), _lib.ptr(dout), _lib.ptr(ws), _lib.ptr(flags),
                                                     _lib.ptr(B.bank_h), _lib.ptr(A.bank_h), stream),
                           "dcl_infonce_bwd_streamk")
                slabs[t.b] += [dout[i] for i in range(nsl)]
                continue
            ns = int(L.dcl_suggest_nsplit(N2, N1))
            dpart = torch.empty((ns, N2pad, _lib.CP), dtype=torch.float32, device=dev)
            _lib.check(L.dcl_infonce_bwd(_lib.ptr(B.bank), N2, B.plan.V, _lib.ptr(A.bank), N1,
                                         _lib.ptr(t.rev_lo), _lib.ptr(t.rev_hi), inv_tau, 0, 0, 1,
                                         None, _lib.ptr(stat), ns, _lib.ptr(dpart), _lib.ptr(B.bank_h),
                                         _lib.ptr(A.bank_h if B.bank_h is not None else None), stream),
                       "dcl_infonce_bwd")
            slabs[t.b] += [dpart[i] for i in range(ns)]
    if used_streamk or _dist_world() > 1:
        _streamk_note_launches(dev)
    grads = []
    for s, sc in enumerate(st.scales):
        if not need[s]:
            grads.append(None)
            continue
        shape, strides_full, _ = feats_meta[s]
        dfeat = torch.empty_strided(shape, strides_full, dtype=torch.float32, device=dev).zero_()
        if len(slabs[s]) > _lib.MAX_SLABS:
            raise RuntimeError(f"{len(slabs[s])} gradient slabs exceed DCL_MAX_SLABS")
        if slabs[s]:
            arr = (ctypes.c_void_p * len(slabs[s]))(*[x.data_ptr() for x in slabs[s]])
            sn, scs, sp = sc.strides
            p = sc.plan
            am = _amax.zeros(_amax.SLOTS, dev)
            _lib.check(L.dcl_normalize_bwd_scatter(arr, len(slabs[s]), _lib.ptr(sc.bank),
                                                   _lib.ptr(sc.nrm), _lib.ptr(_kernel_pix(sc)),
                                                   _lib.ptr(sc.pair_b), _lib.ptr(sc.slot_pair), p.T,
                                                   p.V, sc.C, _lib.ptr(dfeat), sn, scs, sp, _lib.ptr(am), stream),
                       "dcl_normalize_bwd_scatter")
            _amax.tag(dfeat, am)                # absmax of the (otherwise zero) map: no dcl_absmax pass in the projector
        grads.append(dfeat)
    return grads


def dense_contrast_terms(cfg: EngineConfig, label: torch.Tensor, feats: Sequence[torch.Tensor],
                         staged: Optional[StagedLabels] = None, emulated_peers=None):
    """Returns (term_losses f32 [n_terms] with grad, StepState).  Term order: intra scale 0..S-1,
    then cross (0, S-1), then cross (0, S-2) if S > 2.  ``staged``: result of an earlier
    ``stage_labels`` call on the same label tensor (ignored if stale)."""
    from ..models.Projector import LazyProjection
    holder = {"staged": staged, "emulated_peers": emulated_peers}
    feats = [f if f.dtype == torch.float32 else f.float() for f in feats]
    if any(isinstance(f, LazyProjection) for f in feats):
        if emulated_peers is not None or (cfg.global_negatives and _dist_world() > 1):
            # the shared negative bank plans behind an all-rank agreement inside the autograd node: take the maps
            feats = [f.materialize() if isinstance(f, LazyProjection) else f for f in feats]
        else:
            # plan first (the sampled pixels depend on the labels only), then evaluate the heads' last layer on exactly
            # the sampled pixels; the autograd node receives [T * V, C] rows instead of [n, C, h, w] maps
            with_cross = bool(cfg.cross_scale_contrast) and len(feats) > 1
            st = plan_and_sample(cfg, label, feats, with_cross, staged=staged)
            feats = [f.rows(sc.pair_b, sc.pix) if isinstance(f, LazyProjection) else f for f, sc in zip(feats, st.scales)]
            holder["st"] = st
    out = DenseContrastFunction.apply(cfg, label, holder, *feats)
    return out, holder["state"]
