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


@dataclass
class EngineConfig:
    num_all_classes: int
    temperature: float
    min_views_per_class: int = 5
    max_views_per_class: int = 2500
    max_features_total: int = 10000
    weights: Sequence[float] = (1.0,)
    cross_scale_contrast: bool = False
    cross_scale_temperature: float = 0.1
    detach_deepest: bool = False
    w_high_low: float = 1.0
    w_high_mid: float = 1.0
    # extension (not in the reference, SURVEY.md section 8 row e): contrast against the all-gathered
    # banks of every rank instead of the rank-local bank only.  Off = reference semantics.
    global_negatives: bool = False
    # similarity-product arithmetic of the sweep kernels: "f32" = v_mfma_f32_32x32x2_f32 (exact fp32),
    # "f16x3" = three f16 MFMA passes on (hi, lo)-split operands, fp32-equivalent accuracy (csrc/dcl_sweep.hip)
    mfma: str = "f16x3"


@dataclass
class _Scale:
    plan: HostPlan
    h: int
    w: int
    C: int
    strides: Tuple[int, int, int]        # (stride_n, stride_c, stride_p) of the feature tensor
    pair_b: torch.Tensor = None          # device int32 views into the upload pack
    pair_k: torch.Tensor = None
    slot_pair: torch.Tensor = None
    sel: torch.Tensor = None
    pix: torch.Tensor = None             # int32 [T, V]
    rows: bool = False                   # features arrive as [T * V, C] rows (models.Projector.LazyProjection.rows)
    bank: torch.Tensor = None            # f32 [Npad, 256]
    bank_h: torch.Tensor = None          # f16 [Npad, 512] = (hi | lo) halves of bank * 2^10 (f16x3 mode)
    nrm: torch.Tensor = None             # f32 [Npad]
    lbl_s: torch.Tensor = None


@dataclass
class _Term:
    """One InfoNCE evaluation: anchors = bank a, contrast = bank b."""
    a: int
    b: int
    intra: bool
    tau: float
    weight: float
    detach_b: bool = False
    rng_lo: torch.Tensor = None          # [T_a] positive ranges of a's slots in bank b
    rng_hi: torch.Tensor = None
    rev_lo: torch.Tensor = None          # cross only: [T_b] positive ranges of b's slots in bank a
    rev_hi: torch.Tensor = None
    max_span: int = 0                    # max over the anchor slots of rng_hi - rng_lo (host side, from the plan)
    Z: torch.Tensor = None
    W: torch.Tensor = None
    nsplit: int = 1
    segs: list = None                    # contrast-bank segments (own bank first unless gathered)
    pcount: torch.Tensor = None          # int32 [T_a] positives per anchor slot over ALL segments


@dataclass
class _Seg:
    """One segment of a term's contrast bank: the rank-local bank or one remote rank's bank."""
    bank: torch.Tensor                   # f32 [>= N rows, 256]
    N: int
    rng_lo: torch.Tensor                 # int32 [T_a] positive range of each anchor slot in this segment
    rng_hi: torch.Tensor
    own: bool                            # rows of this segment are this rank's own bank rows
    nsplit: int = 1
    bank_h: torch.Tensor = None          # f16x3 copy of ``bank`` (own segments only; None -> f32 product)


class StepState:
    """Everything the backward needs (and what tests inspect): plans, banks, row statistics."""

    def __init__(self):
        self.scales: List[_Scale] = []
        self.terms: List[_Term] = []
        self.loss_buf: Optional[torch.Tensor] = None     # f32 [n_terms] raw (unweighted) term losses
        self.pack: Optional[torch.Tensor] = None         # device copy of the plan upload pack
        self.keepalive: list = []


class _PinnedRing:
    """Persistent pinned host staging buffers.  Allocating pinned memory per step (hipHostMalloc) costs
    tens of milliseconds whenever the host allocator cannot recycle a block that is still in flight,
    so the loss keeps a small ring of grow-only buffers instead.  A slot is reused every ``depth``
    uses; ``release_after(event)`` ties the slot handed out last to an event (recorded after the
    asynchronous copy that reads it), and ``get`` waits for that event before it hands the slot out
    again -- the host may run several steps ahead of the GPU (nothing else in a training step makes
    it wait), so "it was three steps ago" is not a guarantee that the copy has happened."""

    def __init__(self, dtype, depth=3):
        self.dtype, self.depth = dtype, depth
        self.slots = [None] * depth
        self.events = [None] * depth
        self.i = 0

    def get(self, numel: int) -> torch.Tensor:
        self.i = (self.i + 1) % self.depth
        if self.events[self.i] is not None:
            self.events[self.i].synchronize()
            self.events[self.i] = None
        buf = self.slots[self.i]
        if buf is None or buf.numel() < numel:
            buf = torch.empty((max(numel, 1) * 3 // 2 + 64,), dtype=self.dtype, pin_memory=True)
            self.slots[self.i] = buf
        return buf[:numel]

    def release_after(self, event):
        self.events[self.i] = event


_PACK_RING = _PinnedRing(torch.int32)
_COUNTS_RING = _PinnedRing(torch.int32)


def _dist_world() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _stream_ptr():
    return _lib.stream_ptr()


def _feature_strides(f: torch.Tensor):
    """(stride_n, stride_c, stride_p) if the (h, w) plane can be walked with one pixel stride."""
    n, C, h, w = f.shape
    sn, sc, sh, sw = f.stride()
    if h == 1 or sh == w * sw:
        return sn, sc, sw
    return None


def _npad(N: int) -> int:
    return (N + _lib.ROW_TILE - 1) // _lib.ROW_TILE * _lib.ROW_TILE


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


def build_banks(st: StepState, feats: Sequence[torch.Tensor], f16x3: bool = False, gather=None):
    """K3 for every scale (optionally also the f16x3 copy of each bank).  ``gather``: a ``_BankGather`` -- the
    all-gather of a scale's bank is issued right behind its K3, so that it travels over xGMI while the next scale's
    bank is gathered from HBM (K3) instead of after all of them."""
    L = _lib.lib()
    stream = _stream_ptr()
    for sc, f in zip(st.scales, feats):
        p = sc.plan
        Npad = _npad(p.N)
        sc.bank = torch.empty((Npad, _lib.CP), dtype=torch.float32, device=f.device)
        sc.nrm = torch.empty((Npad,), dtype=torch.float32, device=f.device)
        sc.bank_h = torch.empty((Npad, 2 * _lib.CP), dtype=torch.float16, device=f.device) if f16x3 else None
        sn, scs, sp = sc.strides
        _lib.check(L.dcl_gather_normalize(_lib.ptr(f), sn, scs, sp, sc.C, _lib.ptr(_kernel_pix(sc)),
                                          _lib.ptr(sc.pair_b), _lib.ptr(sc.slot_pair), p.T, p.V,
                                          _lib.ptr(sc.bank), _lib.ptr(sc.nrm), _lib.ptr(sc.bank_h), stream),
                   "dcl_gather_normalize")
        if gather is not None:
            gather.issue(sc)


def _own_segments(st: StepState):
    """Default (reference) contrast banks: every term contrasts against the rank-local bank only."""
    L = _lib.lib()
    for t in st.terms:
        B = st.scales[t.b]
        t.segs = [_Seg(bank=B.bank, N=B.plan.N, rng_lo=t.rng_lo, rng_hi=t.rng_hi, own=True,
                       nsplit=int(L.dcl_suggest_nsplit(st.scales[t.a].plan.N, B.plan.N)), bank_h=B.bank_h)]
        t.pcount = None


def class_layout(plan: HostPlan) -> np.ndarray:
    """[V, pairs of class 0, ..., pairs of class K-1]: all a peer needs to address a class-sorted bank."""
    return np.concatenate([[plan.V], plan.cls_hi - plan.cls_lo]).astype(np.int32)


def attach_global_segments(st: StepState, rank: int, peer_banks, peer_layouts, peer_banks_h=None):
    """Replace every term's contrast bank by the concatenation of all ranks' banks.

    peer_banks[q][s]: f32 [>= N_q, 256] bank of rank q at scale s, and / or peer_banks_h[q][s]: its (hi | lo) half
    rows f16 [>= N_q, 512] (entry ``rank`` is ignored: the local bank is used; with only the half rows the peers'
    segments run on the f16x3 path like the local one); peer_layouts[q][s]: ``class_layout`` of that bank (host
    int32 [K + 1]).
    Positives of a local anchor = rows of its class in EVERY segment (minus itself), negatives = all
    other rows of every segment; gradients flow to the local bank only (all_gather has no gradient,
    the convention of the reference's unused concat_all_gather, utils/distributed.py:50-55)."""
    L = _lib.lib()
    dev = st.scales[0].bank.device
    world = len(peer_layouts)
    chunks, where = [], []

    def add(arr):
        arr = np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)
        where.append((sum(c.size for c in chunks), arr.size))
        chunks.append(arr)
        return len(where) - 1

    todo = []
    for t in st.terms:
        pa = st.scales[t.a].plan
        cls = pa.pair_k[pa.slot_pair]
        total_rows = 0
        pc = np.zeros(pa.T, dtype=np.int64)
        seg_ids = []
        for q in range(world):
            lay = np.asarray(peer_layouts[q][t.b])
            Vq, per_cls = int(lay[0]), lay[1:].astype(np.int64)
            hi = np.cumsum(per_cls)
            lo = hi - per_cls
            Nq = int(per_cls.sum()) * Vq
            total_rows += Nq
            pc += per_cls[cls] * Vq
            seg_ids.append((q, Nq, add(lo[cls] * Vq), add(hi[cls] * Vq)))
        if t.intra:
            pc -= 1                                   # the anchor itself is not its own positive
        todo.append((t, seg_ids, add(pc), total_rows))
    pack_host = _PACK_RING.get(sum(c.size for c in chunks))
    np.concatenate(chunks, out=pack_host.numpy())
    pack = pack_host.to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _PACK_RING.release_after(ev)
    st.keepalive += [pack_host, pack]

    def view(idx):
        off, size = where[idx]
        return pack[off:off + size]

    for t, seg_ids, pc_id, total_rows in todo:
        N1 = st.scales[t.a].plan.N
        per_seg = max(1, int(L.dcl_suggest_nsplit(N1, total_rows)) // world)
        t.segs = []
        for q, Nq, lo_id, hi_id in seg_ids:
            if Nq == 0:
                continue
            own = q == rank
            if own:
                bank, bank_h = st.scales[t.b].bank, st.scales[t.b].bank_h
            else:
                bank_h = peer_banks_h[q][t.b] if peer_banks_h is not None else None
                # the f32 rows of a peer are only read by the f32 kernels; in f16x3 mode the local bank stands in as
                # the (unused) f32 argument of the C ABI
                bank = peer_banks[q][t.b] if peer_banks is not None else st.scales[t.b].bank
            t.segs.append(_Seg(bank=bank, N=Nq, rng_lo=view(lo_id), rng_hi=view(hi_id), own=own,
                               nsplit=min(per_seg, max(1, (Nq + 31) // 32)), bank_h=bank_h))
        t.pcount = view(pc_id)


class _BankGather:
    """RCCL all-gather of every scale's bank (padded to a fixed row count) and class layout for the shared negative
    bank.  What travels is the representation the sweep kernels read: in ``f16x3`` mode the (hi | lo) half rows
    (``bank_h``, so that peers' segments run on the f16 matrix pipe like the local one), else the f32 rows -- 1 KiB per
    row either way (<= 10.24 MB per rank and scale).  ``issue(scale)`` is called right behind the scale's K3 (async
    collective on RCCL's own stream, ordered after the producer through the work object); ``finish()`` waits for all of
    them just before the first sweep and returns (rank, peer_banks, peer_banks_h, peer_layouts)."""

    def __init__(self, st: StepState, max_features_total: int, f16x3: bool, group=None):
        import torch.distributed as dist
        self.dist, self.group, self.st, self.f16x3 = dist, group, st, f16x3
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        sc0 = st.scales[0]
        dev = (sc0.pix if sc0.pix is not None else sc0.bank).device
        self.cap = _npad(max(max_features_total, max(sc.plan.N for sc in st.scales)))
        lay = torch.from_numpy(np.stack([class_layout(sc.plan) for sc in st.scales])).to(dev)
        # outputs are laid out [world * rows, ...] (concatenation along dim 0) and viewed per rank afterwards
        self.lay_all = torch.empty((self.world * lay.shape[0], lay.shape[1]), dtype=torch.int32, device=dev)
        self.work = [dist.all_gather_into_tensor(self.lay_all, lay, group=group, async_op=True)]
        self.gathered = []
        st.keepalive.append(lay)

    def issue(self, sc: _Scale):
        src = sc.bank_h if self.f16x3 else sc.bank
        width = src.shape[1]
        if src.shape[0] != self.cap:
            pad = torch.zeros((self.cap, width), dtype=src.dtype, device=src.device)
            pad[:src.shape[0]] = src
            src = pad
        out = torch.empty((self.world * self.cap, width), dtype=src.dtype, device=src.device)
        self.work.append(self.dist.all_gather_into_tensor(out, src, group=self.group, async_op=True))
        self.gathered.append(out.view(self.world, self.cap, width))
        self.st.keepalive.append(src)

    def finish(self):
        for w in self.work:
            w.wait()
        S = len(self.st.scales)
        layouts = self.lay_all.view(self.world, S, -1).cpu().numpy()   # [world, S, K + 1]: the one extra host sync
        peer = [[self.gathered[s][q] for s in range(S)] for q in range(self.world)]
        peer_layouts = [[layouts[q, s] for s in range(S)] for q in range(self.world)]
        self.st.keepalive += self.gathered
        if self.f16x3:
            return self.rank, None, peer, peer_layouts
        return self.rank, peer, None, peer_layouts


def gather_peer_banks(st: StepState, max_features_total: int, group=None, f16x3: bool = False):
    """All scales at once (banks already built): (rank, peer_banks, peer_banks_h, peer_layouts)."""
    g = _BankGather(st, max_features_total, f16x3, group)
    for sc in st.scales:
        g.issue(sc)
    return g.finish()


def agree_or_raise(error: Optional[BaseException], device, group=None):
    """Shared-negative-bank mode: every rank is about to enter collectives; if ANY rank failed while planning
    (e.g. no (image, class) pair with min_views pixels on its shard), all ranks must raise instead of some of them
    hanging in the all-gather.  One 4-byte all-reduce(MAX)."""
    import torch.distributed as dist
    flag = torch.tensor([1 if error is not None else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if error is not None:
        raise error
    if int(flag.item()):
        raise RuntimeError("another rank failed while planning the contrastive loss (shared negative bank): "
                           "aborting this step on every rank")


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
