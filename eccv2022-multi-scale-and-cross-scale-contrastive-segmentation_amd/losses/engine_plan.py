"""Dense contrastive loss engine, stage 1: label staging (K1 dcl_label_hist), the host sampling plan and the sampled pixel indices
(K2 dcl_rank_select) -- reference losses/DenseContrastiveLossV2.py:86-125 (sample_anchors_fast), :64-84 (_select_views_per_class),
:194-206 (label down-sampling)."""
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


class StagedLabels:
    """Result of the label stage (K1 on every scale + D2H of the histogram), possibly produced ahead
    of time on a side stream while the model forward runs (DenseContrastiveLossV2_ms.prepare)."""

    def __init__(self):
        self.key = None                 # (data_ptr, shape, version) of the label tensor it was made from
        self.geoms = None               # [(scale, h, w)] per feature map
        self.lbl_s: List[torch.Tensor] = []
        self.seg_hists: List[torch.Tensor] = []
        self.counts = None              # device int32 [S, n, K]
        self.counts_host = None         # pinned int32 [S, n, K]
        self.event = None               # recorded after the D2H copy
        self.stream = None              # stream the stage ran on
        self.label = None               # keeps the (int64, contiguous) label alive


def _label_key(label: torch.Tensor):
    return (label.data_ptr(), tuple(label.shape), label.dtype, label._version)


def _canon_label(label: torch.Tensor, dev):
    if label.device != dev:
        label = label.to(dev)
    if label.dtype != torch.int64:
        label = label.long()
    return label.contiguous()


def feature_geometry(label_shape, feats: Sequence[torch.Tensor]):
    """[(scale, h, w)] with scale = W_label // W_feat (DenseContrastiveLossV2.py:46) and the checks the
    reference leaves to an IndexError."""
    n, H, W = label_shape
    geoms = []
    for s, f in enumerate(feats):
        if f.dim() != 4 or f.shape[0] != n:
            raise RuntimeError(f"features[{s}] must be [n, C, h, w] with n={n}, got {tuple(f.shape)}")
        scale = int(W // f.shape[-1])
        if scale < 1:
            raise RuntimeError(f"features[{s}] is wider than the label map")
        h, w = H // scale, W // scale
        if (h, w) != (f.shape[2], f.shape[3]):
            raise RuntimeError(
                f"features[{s}] is {f.shape[2]}x{f.shape[3]} but the label map down-sampled by "
                f"{scale} is {h}x{w}; the reference indexes features with label-grid positions "
                "(DenseContrastiveLossV2.py:97,123), so the two grids must coincide")
        geoms.append((scale, h, w))
    return geoms


def stage_labels(K: int, label: torch.Tensor, geoms, side_stream=None, ready_event=None) -> StagedLabels:
    """K1 for every scale + asynchronous D2H of the [S, n, K] histogram into pinned memory.
    With ``side_stream`` the work is enqueued there, so it overlaps whatever the caller enqueues next on the
    current stream.  It starts after ``ready_event`` (an event recorded once the label tensor is complete, e.g.
    right after its H2D copy) or, without one, after everything already queued on the current stream -- the
    event form lets the label stage (and the host-side plan that waits for it) run while the GPU is still busy
    with the PREVIOUS step, so the host never waits for the device inside a training step."""
    L = _lib.lib()
    if not 0 < K <= _lib.MAX_CLASSES:
        raise RuntimeError(f"num_all_classes={K} outside the supported range [1, 255]")
    dev = label.device
    st = StagedLabels()
    st.key = _label_key(label)
    label = _canon_label(label, dev)
    st.label, st.geoms = label, list(geoms)
    n, H, W = label.shape
    S = len(geoms)
    cur = torch.cuda.current_stream()
    run = side_stream if side_stream is not None else cur
    if side_stream is not None:
        if ready_event is not None:
            side_stream.wait_event(ready_event)     # the label is complete once this event has fired
        else:
            side_stream.wait_stream(cur)            # label is produced on the current stream
        label.record_stream(side_stream)
    with torch.cuda.stream(run):
        stream = ctypes.c_void_p(run.cuda_stream)
        st.counts = torch.zeros((S, n, K), dtype=torch.int32, device=dev)
        for s, (scale, h, w) in enumerate(geoms):
            nseg = (h * w + _lib.SEG - 1) // _lib.SEG
            lbl_s = torch.empty((n, h * w), dtype=torch.uint8, device=dev)
            seg_hist = torch.empty((n, nseg, K), dtype=torch.int32, device=dev)
            _lib.check(L.dcl_label_hist(_lib.ptr(label), n, H, W, scale, K, _lib.ptr(lbl_s),
                                        _lib.ptr(seg_hist), _lib.ptr(st.counts[s]), stream),
                       "dcl_label_hist")
            st.lbl_s.append(lbl_s)
            st.seg_hists.append(seg_hist)
        st.counts_host = _COUNTS_RING.get(S * n * K).view(S, n, K)
        st.counts_host.copy_(st.counts, non_blocking=True)
        st.event = torch.cuda.Event()
        st.event.record(run)
    st.stream = run
    return st


def _plan_terms_and_sample(cfg: EngineConfig, staged: StagedLabels, with_cross: bool, dev) -> StepState:
    """Everything of the sampling stage that needs the LABELS only: waits for the staged histograms, builds the
    host plans (this is where the reference's RNG draws happen, in its order: scale 0 pairs ..., scale 1 pairs ...),
    uploads the plan pack and runs K2 for every scale -- all on the CURRENT stream.  Feature geometry (channel
    count, strides) is bound later by ``_bind_features``."""
    L = _lib.lib()
    K = staged.counts.shape[-1]
    n = staged.counts.shape[1]
    S = len(staged.geoms)
    st = StepState()
    stream = _stream_ptr()
    for s in range(S):
        _, h, w = staged.geoms[s]
        st.scales.append(_Scale(plan=None, h=h, w=w, C=0, strides=None, lbl_s=staged.lbl_s[s]))
    seg_hists = staged.seg_hists
    staged.event.synchronize()                      # the one host wait of the loss (K1 + 960-B D2H)
    counts_host = staged.counts_host.numpy()
    cur = torch.cuda.current_stream()
    if staged.stream is not cur:
        cur.wait_event(staged.event)
        for t in staged.lbl_s + staged.seg_hists:
            t.record_stream(cur)
    st.keepalive.append(staged)

    # ---- host: plans in scale order (this is the RNG consumption order of the reference)
    for s in range(S):
        st.scales[s].plan = build_host_plan(counts_host[s], cfg.min_views_per_class,
                                            cfg.max_views_per_class, cfg.max_features_total)

    # ---- term list (DenseContrastiveLossV2_ms.py:51-80)
    weights = list(cfg.weights)
    for s in range(S):
        st.terms.append(_Term(a=s, b=s, intra=True, tau=cfg.temperature, weight=float(weights[s])))
    if with_cross:
        assert S > 1
        st.terms.append(_Term(a=0, b=S - 1, intra=False, tau=cfg.cross_scale_temperature,
                              weight=float(cfg.w_high_low), detach_b=cfg.detach_deepest))
        if S > 2:
            st.terms.append(_Term(a=0, b=S - 2, intra=False, tau=cfg.cross_scale_temperature,
                                  weight=float(cfg.w_high_mid), detach_b=cfg.detach_deepest))

    # ---- one upload pack: per scale [pair_b | pair_k | slot_pair | sel], per term [lo | hi (| rev)]
    chunks, where = [], []

    def add(arr):
        arr = np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)
        off = sum(c.size for c in chunks)
        chunks.append(arr)
        where.append((off, arr.size))
        return len(where) - 1

    scale_slots = []
    for sc in st.scales:
        p = sc.plan
        scale_slots.append((add(p.pair_b), add(p.pair_k), add(p.slot_pair), add(p.sel)))
    term_slots = []
    for t in st.terms:
        pa, pb = st.scales[t.a].plan, st.scales[t.b].plan
        lo, hi = positive_ranges(pa, pb)
        ids = [add(lo), add(hi)]
        t.max_span = int((hi - lo).max()) if len(lo) else 0      # widest positive range of an anchor slot (host plan)
        if not t.intra:
            rlo, rhi = positive_ranges(pb, pa)
            ids += [add(rlo), add(rhi)]
        term_slots.append(ids)
    total = sum(c.size for c in chunks)
    pack_host = _PACK_RING.get(total)
    np.concatenate(chunks, out=pack_host.numpy())
    pack = pack_host.to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _PACK_RING.release_after(ev)
    st.keepalive += [pack_host, pack]

    def view(idx):
        off, size = where[idx]
        return pack[off:off + size]

    # ---- K2
    for s, sc in enumerate(st.scales):
        p = sc.plan
        ib, ik, isp, isel = scale_slots[s]
        sc.pair_b, sc.pair_k, sc.slot_pair, sc.sel = view(ib), view(ik), view(isp), view(isel)
        sc.pix = torch.empty((p.T, p.V), dtype=torch.int32, device=dev)
        _lib.check(L.dcl_rank_select(_lib.ptr(sc.lbl_s), _lib.ptr(seg_hists[s]), n, sc.h * sc.w, K,
                                     _lib.ptr(sc.pair_b), _lib.ptr(sc.pair_k), p.T, p.V,
                                     _lib.ptr(sc.sel), _lib.ptr(sc.pix), stream), "dcl_rank_select")
    for t, ids in zip(st.terms, term_slots):
        t.rng_lo, t.rng_hi = view(ids[0]), view(ids[1])
        if not t.intra:
            t.rev_lo, t.rev_hi = view(ids[2]), view(ids[3])
    st.keepalive += seg_hists
    st.pack = pack
    return st


class PreSampled:
    """A StepState whose label-only half (plans, RNG draws, K2) was computed ahead of the model forward on a side
    stream (DenseContrastiveLossV2_ms.prepare); ``event`` marks its completion on that stream."""

    def __init__(self, st, key, geoms, cfg_key, stream, event):
        self.st, self.key, self.geoms, self.cfg_key, self.stream, self.event = st, key, geoms, cfg_key, stream, event


class PreSampleFailed:
    """prepare() ran the sampling plan ahead of the model forward and it raised (its RNG draws, if any, are spent).
    The error is re-raised by ``plan_and_sample`` -- inside DenseContrastFunction.forward, i.e. behind
    ``agree_or_raise`` when the shared negative bank is on, so that every rank leaves the step together."""

    def __init__(self, error: BaseException, label: torch.Tensor):
        self.error, self.key = error, _label_key(label)


def _cfg_key(cfg: EngineConfig, with_cross: bool):
    return (cfg.num_all_classes, cfg.min_views_per_class, cfg.max_views_per_class, cfg.max_features_total,
            tuple(cfg.weights), bool(with_cross), cfg.temperature, cfg.cross_scale_temperature, bool(cfg.detach_deepest),
            cfg.w_high_low, cfg.w_high_mid)


def presample(cfg: EngineConfig, label: torch.Tensor, geoms, with_cross: bool, side_stream,
              ready_event=None) -> PreSampled:
    """Label stage + host plans + K2 on ``side_stream``, before the model forward is enqueued: the host builds the
    sampling plan (its ~1 ms and the wait for the 960-byte histogram) while the GPU is still busy with the previous
    step, instead of after the forward with the GPU idle.  Consumes the CPU RNG exactly like the in-forward path
    (same draws, same order); nothing else in a training step draws from the CPU generator."""
    key = _label_key(label)
    staged = stage_labels(cfg.num_all_classes, label, geoms, side_stream=side_stream, ready_event=ready_event)
    with torch.cuda.stream(side_stream):
        st = _plan_terms_and_sample(cfg, staged, with_cross, staged.label.device)
        ev = torch.cuda.Event()
        ev.record(side_stream)
    return PreSampled(st, key, list(geoms), _cfg_key(cfg, with_cross), side_stream, ev)


_ROW_INDEX = {}


def _row_index(n: int, dev) -> torch.Tensor:
    """int32 [>= n] 0, 1, 2, ...: the 'pixel' table of a feature tensor that already holds one row per bank slot."""
    key = str(dev)
    t = _ROW_INDEX.get(key)
    if t is None or t.numel() < n:
        t = _ROW_INDEX[key] = torch.arange(max(n, 1 << 16), dtype=torch.int32, device=dev)
    return t


def _kernel_pix(sc: _Scale) -> torch.Tensor:
    return _row_index(sc.plan.T * sc.plan.V, sc.pix.device) if sc.rows else sc.pix


def _bind_features(st: StepState, feats: Sequence[torch.Tensor]):
    from ..models.Projector import LazyProjection
    for s, (sc, f) in enumerate(zip(st.scales, feats)):
        if f.dtype != torch.float32:
            raise RuntimeError(f"features[{s}] must be float32, got {f.dtype}")
        C = f.shape[1]
        if C > _lib.CP:
            raise RuntimeError(f"embedding width {C} > {_lib.CP} is not supported by the sweep kernels")
        if isinstance(f, LazyProjection):
            # one row per (pair, view) slot, row t * V + v: the kernels address it as a one-image map of T * V 'pixels'
            # with pixel stride C (NHWC) through the identity pixel table
            sc.C, sc.strides, sc.rows = C, (0, 1, C), True
            continue
        sc.rows = False
        strides = _feature_strides(f)
        if strides is None:
            raise RuntimeError(f"features[{s}] has a non-collapsible (h, w) layout; call .contiguous()")
        sc.C, sc.strides = C, strides


def plan_and_sample(cfg: EngineConfig, label: torch.Tensor, feats: Sequence[torch.Tensor],
                    with_cross: bool, staged=None) -> StepState:
    """label stage (or a pre-staged / pre-sampled one) -> host plan -> K2 for every scale; builds the term list with
    its positive ranges.  ``staged``: a ``StagedLabels`` (label stage done ahead) or a ``PreSampled`` (plans and K2
    done ahead as well); either is ignored when it was made from another label tensor / geometry / configuration."""
    dev = feats[0].device
    if dev.type != "cuda":
        raise RuntimeError("mscs_amd dense contrastive loss runs on the MI355X only: features are on "
                           f"{dev}; there is no CPU fallback")
    n, H, W = label.shape
    K = cfg.num_all_classes
    geoms = feature_geometry((n, H, W), feats)
    if isinstance(staged, PreSampleFailed):
        if staged.key == _label_key(label):
            raise staged.error                       # planning of THIS step already failed in prepare()
        printlog(f'dense contrastive loss: a planning error parked by prepare() belongs to another label tensor and is '
                 f'dropped (its randperm draws are spent): {staged.error}')
        staged = None
    if isinstance(staged, PreSampled):
        pre, staged = staged, None
        if pre.key == _label_key(label) and pre.geoms == geoms and pre.cfg_key == _cfg_key(cfg, with_cross):
            st = pre.st
            cur = torch.cuda.current_stream()
            if pre.stream is not cur:
                cur.wait_event(pre.event)
                for sc in st.scales:
                    sc.pix.record_stream(cur)
                st.pack.record_stream(cur)
            _bind_features(st, feats)
            return st
        # stale: fall through to the in-forward path.  NOTE the pre-sampling already consumed its RNG draws.
    if staged is not None and (staged.key != _label_key(label) or staged.geoms != geoms
                               or staged.counts.shape[-1] != K):
        staged = None                               # stale: made from another label / geometry
    if staged is None:
        staged = stage_labels(K, label.to(dev) if label.device != dev else label, geoms)
    st = _plan_terms_and_sample(cfg, staged, with_cross, dev)
    _bind_features(st, feats)
    return st

__all__ = [_n for _n in dir() if not _n.startswith('__')]      # private helpers too: the stage modules share them
