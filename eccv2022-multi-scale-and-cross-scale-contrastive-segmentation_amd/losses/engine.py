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

from .. import _lib
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
    bank: torch.Tensor = None            # f32 [Npad, 256]
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
    Z: torch.Tensor = None
    W: torch.Tensor = None
    nsplit: int = 1


class StepState:
    """Everything the backward needs (and what tests inspect): plans, banks, row statistics."""

    def __init__(self):
        self.scales: List[_Scale] = []
        self.terms: List[_Term] = []
        self.loss_buf: Optional[torch.Tensor] = None     # f32 [n_terms] raw (unweighted) term losses
        self.keepalive: list = []


class _PinnedRing:
    """Persistent pinned host staging buffers.  Allocating pinned memory per step (hipHostMalloc) costs
    tens of milliseconds whenever the host allocator cannot recycle a block that is still in flight,
    so the loss keeps a small ring of grow-only buffers instead.  A slot is reused every ``depth``
    uses; every use is followed (stream-ordered) by the next step's label stage, whose completion the
    host waits for before it writes the following slot."""

    def __init__(self, dtype, depth=3):
        self.dtype, self.depth = dtype, depth
        self.slots = [None] * depth
        self.i = 0

    def get(self, numel: int) -> torch.Tensor:
        self.i = (self.i + 1) % self.depth
        buf = self.slots[self.i]
        if buf is None or buf.numel() < numel:
            buf = torch.empty((max(numel, 1) * 3 // 2 + 64,), dtype=self.dtype, pin_memory=True)
            self.slots[self.i] = buf
        return buf[:numel]


_PACK_RING = _PinnedRing(torch.int32)
_COUNTS_RING = _PinnedRing(torch.int32)


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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


def stage_labels(K: int, label: torch.Tensor, geoms, side_stream=None) -> StagedLabels:
    """K1 for every scale + asynchronous D2H of the [S, n, K] histogram into pinned memory.
    With ``side_stream`` the work is enqueued there (after everything already queued on the current
    stream), so it overlaps whatever the caller enqueues next on the current stream."""
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
        side_stream.wait_stream(cur)                # label is produced on the current stream
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


def plan_and_sample(cfg: EngineConfig, label: torch.Tensor, feats: Sequence[torch.Tensor],
                    with_cross: bool, staged: Optional[StagedLabels] = None) -> StepState:
    """label stage (or a pre-staged one) -> host plan -> K2 for every scale; builds the term list with
    its positive ranges."""
    L = _lib.lib()
    dev = feats[0].device
    if dev.type != "cuda":
        raise RuntimeError("mscs_amd dense contrastive loss runs on the MI355X only: features are on "
                           f"{dev}; there is no CPU fallback")
    n, H, W = label.shape
    K = cfg.num_all_classes
    S = len(feats)
    geoms = feature_geometry((n, H, W), feats)
    if staged is not None and (staged.key != _label_key(label) or staged.geoms != geoms
                               or staged.counts.shape[-1] != K):
        staged = None                               # stale: made from another label / geometry
    if staged is None:
        staged = stage_labels(K, label.to(dev) if label.device != dev else label, geoms)
    st = StepState()
    stream = _stream_ptr()
    for s, f in enumerate(feats):
        if f.dtype != torch.float32:
            raise RuntimeError(f"features[{s}] must be float32, got {f.dtype}")
        C = f.shape[1]
        if C > _lib.CP:
            raise RuntimeError(f"embedding width {C} > {_lib.CP} is not supported by the sweep kernels")
        strides = _feature_strides(f)
        if strides is None:
            raise RuntimeError(f"features[{s}] has a non-collapsible (h, w) layout; call .contiguous()")
        _, h, w = geoms[s]
        st.scales.append(_Scale(plan=None, h=h, w=w, C=C, strides=strides, lbl_s=staged.lbl_s[s]))
    seg_hists = staged.seg_hists
    staged.event.synchronize()                      # the one host wait of the forward (K1 + 960-B D2H)
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
        if not t.intra:
            rlo, rhi = positive_ranges(pb, pa)
            ids += [add(rlo), add(rhi)]
        term_slots.append(ids)
    total = sum(c.size for c in chunks)
    pack_host = _PACK_RING.get(total)
    np.concatenate(chunks, out=pack_host.numpy())
    pack = pack_host.to(dev, non_blocking=True)
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
    return st


def build_banks(st: StepState, feats: Sequence[torch.Tensor]):
    """K3 for every scale."""
    L = _lib.lib()
    stream = _stream_ptr()
    for sc, f in zip(st.scales, feats):
        p = sc.plan
        Npad = _npad(p.N)
        sc.bank = torch.empty((Npad, _lib.CP), dtype=torch.float32, device=f.device)
        sc.nrm = torch.empty((Npad,), dtype=torch.float32, device=f.device)
        sn, scs, sp = sc.strides
        _lib.check(L.dcl_gather_normalize(_lib.ptr(f), sn, scs, sp, sc.C, _lib.ptr(sc.pix),
                                          _lib.ptr(sc.pair_b), _lib.ptr(sc.slot_pair), p.T, p.V,
                                          _lib.ptr(sc.bank), _lib.ptr(sc.nrm), stream),
                   "dcl_gather_normalize")


def run_forward_terms(st: StepState):
    """K4 for every term; raw term losses land in st.loss_buf[idx]."""
    L = _lib.lib()
    stream = _stream_ptr()
    dev = st.scales[0].bank.device
    st.loss_buf = torch.empty((len(st.terms),), dtype=torch.float32, device=dev)
    for idx, t in enumerate(st.terms):
        A, B = st.scales[t.a], st.scales[t.b]
        N1, N2 = A.plan.N, B.plan.N
        N1pad = _npad(N1)
        t.nsplit = int(L.dcl_suggest_nsplit(N1, N2))
        zpart = torch.empty((t.nsplit * N1pad,), dtype=torch.float32, device=dev)
        t.Z = torch.empty((N1pad,), dtype=torch.float32, device=dev)
        t.W = torch.empty((N1pad,), dtype=torch.float32, device=dev)
        rowloss = torch.empty((N1pad,), dtype=torch.float32, device=dev)
        _lib.check(L.dcl_infonce_fwd(_lib.ptr(A.bank), N1, A.plan.V, _lib.ptr(B.bank), N2,
                                     _lib.ptr(t.rng_lo), _lib.ptr(t.rng_hi), 1.0 / t.tau,
                                     1 if t.intra else 0, t.nsplit, _lib.ptr(zpart), _lib.ptr(t.Z),
                                     _lib.ptr(rowloss), _lib.ptr(t.W), _lib.ptr(st.loss_buf[idx:]),
                                     stream), "dcl_infonce_fwd")
        st.keepalive += [zpart, rowloss]


class DenseContrastFunction(torch.autograd.Function):
    """loss_terms = f(features...) with a hand-written backward (K5/K6)."""

    @staticmethod
    def forward(ctx, cfg: EngineConfig, label: torch.Tensor, holder: dict, *feats):
        with_cross = bool(cfg.cross_scale_contrast) and len(feats) > 1
        st = plan_and_sample(cfg, label, feats, with_cross, staged=holder.get("staged"))
        build_banks(st, feats)
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


def _backward_with_term_grads(st: StepState, grad_terms: torch.Tensor, feats_meta, need):
    """The Function outputs the RAW per-term losses; the module forms the weighted sum with torch
    ops, so autograd hands us d total / d term_i = weight_i * upstream.  Each term's kernels read
    their own element of that vector as the device-side scale."""
    L = _lib.lib()
    stream = _stream_ptr()
    dev = st.scales[0].bank.device
    g = grad_terms.detach().to(device=dev, dtype=torch.float32).contiguous()
    slabs: List[List[torch.Tensor]] = [[] for _ in st.scales]
    for idx, t in enumerate(st.terms):
        A, B = st.scales[t.a], st.scales[t.b]
        N1, N2 = A.plan.N, B.plan.N
        N1pad, N2pad = _npad(N1), _npad(N2)
        want_a = need[t.a]
        want_b = (not t.intra) and need[t.b] and not t.detach_b
        if not (want_a or want_b):
            continue
        stat = torch.empty((N1pad, 4), dtype=torch.float32, device=dev)
        _lib.check(L.dcl_infonce_prep_stats(_lib.ptr(t.Z), _lib.ptr(t.W), _lib.ptr(t.rng_lo),
                                            _lib.ptr(t.rng_hi), N1, A.plan.V, 1 if t.intra else 0,
                                            1.0, 1.0 / t.tau, _lib.ptr(g[idx:]), _lib.ptr(stat),
                                            stream), "dcl_infonce_prep_stats")
        if want_a:
            ns = int(L.dcl_suggest_nsplit(N1, N2))
            dpart = torch.empty((ns, N1pad, _lib.CP), dtype=torch.float32, device=dev)
            _lib.check(L.dcl_infonce_bwd(_lib.ptr(A.bank), N1, A.plan.V, _lib.ptr(B.bank), N2,
                                         _lib.ptr(t.rng_lo), _lib.ptr(t.rng_hi), 1.0 / t.tau,
                                         1 if t.intra else 0, 1, 1 if t.intra else 0,
                                         _lib.ptr(stat), _lib.ptr(stat) if t.intra else None, ns,
                                         _lib.ptr(dpart), stream), "dcl_infonce_bwd")
            slabs[t.a] += [dpart[i] for i in range(ns)]
        if want_b:
            ns = int(L.dcl_suggest_nsplit(N2, N1))
            dpart = torch.empty((ns, N2pad, _lib.CP), dtype=torch.float32, device=dev)
            _lib.check(L.dcl_infonce_bwd(_lib.ptr(B.bank), N2, B.plan.V, _lib.ptr(A.bank), N1,
                                         _lib.ptr(t.rev_lo), _lib.ptr(t.rev_hi), 1.0 / t.tau, 0, 0, 1,
                                         None, _lib.ptr(stat), ns, _lib.ptr(dpart), stream),
                       "dcl_infonce_bwd")
            slabs[t.b] += [dpart[i] for i in range(ns)]
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
            _lib.check(L.dcl_normalize_bwd_scatter(arr, len(slabs[s]), _lib.ptr(sc.bank),
                                                   _lib.ptr(sc.nrm), _lib.ptr(sc.pix),
                                                   _lib.ptr(sc.pair_b), _lib.ptr(sc.slot_pair), p.T,
                                                   p.V, sc.C, _lib.ptr(dfeat), sn, scs, sp, stream),
                       "dcl_normalize_bwd_scatter")
        grads.append(dfeat)
    return grads


def dense_contrast_terms(cfg: EngineConfig, label: torch.Tensor, feats: Sequence[torch.Tensor],
                         staged: Optional[StagedLabels] = None):
    """Returns (term_losses f32 [n_terms] with grad, StepState).  Term order: intra scale 0..S-1,
    then cross (0, S-1), then cross (0, S-2) if S > 2.  ``staged``: result of an earlier
    ``stage_labels`` call on the same label tensor (ignored if stale)."""
    holder = {"staged": staged}
    feats = [f if f.dtype == torch.float32 else f.float() for f in feats]
    out = DenseContrastFunction.apply(cfg, label, holder, *feats)
    return out, holder["state"]
