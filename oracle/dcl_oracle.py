"""oracle/dcl_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy + the plain-C sampling half in ``sampling_oracle.c``) of
the multi-scale / cross-scale dense pixel-contrastive loss of
RViMLab/ECCV2022-multi-scale-and-cross-scale-contrastive-segmentation.  Each
function cites the reference lines it follows (paths relative to
``/root/reference``).

Pinning: ``tests/test_oracle.py`` checks this file against the golden vectors
under ``tests/golden/`` which were produced by *running the reference itself*
on CPU in the build container (``tools/gen_golden.py``), bit-exact for sampled
pixel indices and to fp32 round-off for losses / gradients.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``mscs_amd``) never does: it fails
loudly if its HIP library is missing instead of falling back to this code.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsampling_oracle.so")
_lib = None


def build_c_oracle(force: bool = False) -> str:
    """gcc-compile ``sampling_oracle.c`` into ``oracle/_build/``."""
    src = os.path.join(_HERE, "sampling_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src, "-lm"])
    return _SO


def _c():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build_c_oracle())
        i64p = ctypes.POINTER(ctypes.c_int64)
        vp = ctypes.c_void_p
        lib.orc_mt_sizeof.restype = ctypes.c_size_t
        lib.orc_mt_seed.argtypes = [vp, ctypes.c_uint64]
        lib.orc_mt_next.argtypes = [vp]
        lib.orc_mt_next.restype = ctypes.c_uint32
        lib.orc_randperm.argtypes = [vp, ctypes.c_int64, i64p]
        lib.orc_downsample_labels.argtypes = [i64p] + [ctypes.c_int64] * 4 + [i64p]
        lib.orc_class_counts.argtypes = [i64p] + [ctypes.c_int64] * 3 + [i64p]
        lib.orc_select_pairs.argtypes = [i64p] + [ctypes.c_int64] * 3 + [i64p] * 3
        lib.orc_select_pairs.restype = ctypes.c_int64
        lib.orc_select_views.argtypes = [i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.POINTER(ctypes.c_int)]
        lib.orc_select_views.restype = ctypes.c_int64
        lib.orc_sample_pixels.argtypes = [vp, i64p, ctypes.c_int64, i64p, i64p, ctypes.c_int64,
                                          ctypes.c_int64, i64p]
        lib.orc_sample_pixels_with_perms.argtypes = [i64p, ctypes.c_int64, i64p, i64p,
                                                     ctypes.c_int64, ctypes.c_int64, i64p, i64p, i64p]
        _lib = lib
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


class MT19937:
    """torch's CPU default generator (at::mt19937) restated; see sampling_oracle.c."""

    def __init__(self, seed: int):
        self._buf = ctypes.create_string_buffer(_c().orc_mt_sizeof())
        _c().orc_mt_seed(self._buf, ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF))

    def next_u32(self) -> int:
        return int(_c().orc_mt_next(self._buf))

    def randperm(self, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.int64)
        _c().orc_randperm(self._buf, n, _p(out))
        return out


# --------------------------------------------------------------------------
# integer half: labels -> sampling plan   (losses/DenseContrastiveLossV2.py)
# --------------------------------------------------------------------------
def downsample_labels(label: np.ndarray, scale: int) -> np.ndarray:
    """get_dist_and_classes, DenseContrastiveLossV2.py:194-206 (nearest, (H//s, W//s))."""
    label = np.ascontiguousarray(label, dtype=np.int64)
    n, H, W = label.shape
    out = np.empty((n, H // scale, W // scale), dtype=np.int64)
    _c().orc_downsample_labels(_p(label), n, H, W, scale, _p(out))
    return out


def class_counts(lbl_s: np.ndarray, K: int) -> np.ndarray:
    """DenseContrastiveLossV2.py:100-103."""
    n = lbl_s.shape[0]
    flat = np.ascontiguousarray(lbl_s.reshape(n, -1), dtype=np.int64)
    counts = np.empty((n, K), dtype=np.int64)
    _c().orc_class_counts(_p(flat), n, flat.shape[1], K, _p(counts))
    return counts


@dataclass
class Plan:
    """Sampling plan of one scale (what sample_anchors_fast decides, :86-125)."""
    scale: int
    h: int
    w: int
    pair_b: np.ndarray          # [T] image index, reference order
    pair_k: np.ndarray          # [T] class id
    pair_cnt: np.ndarray        # [T] pixels of that class in that image
    V: int
    log_this_step: bool
    pix: np.ndarray             # [T, V] flat pixel index into h*w, reference order

    @property
    def T(self) -> int:
        return int(self.pair_b.shape[0])

    @property
    def N(self) -> int:
        return self.T * self.V


def make_plan(label: np.ndarray, scale: int, K: int, min_views: int, max_views: int,
              max_total: int, rng: Optional[MT19937] = None,
              randperm: Optional[Callable[[int], np.ndarray]] = None) -> Plan:
    """sample_anchors_fast, DenseContrastiveLossV2.py:86-125, index part only.

    Draws come either from ``rng`` (the C restatement of torch's generator) or from a
    ``randperm(n)`` callable (e.g. ``lambda n: torch.randperm(n).numpy()``)."""
    lbl_s = downsample_labels(label, scale)
    n, h, w = lbl_s.shape
    counts = class_counts(lbl_s, K)
    pb = np.empty(n * K, dtype=np.int64)
    pk = np.empty(n * K, dtype=np.int64)
    pc = np.empty(n * K, dtype=np.int64)
    T = int(_c().orc_select_pairs(_p(counts), n, K, min_views, _p(pb), _p(pk), _p(pc)))
    if T == 0:
        # torch.min of an empty tensor raises in the reference (:110)
        raise RuntimeError("no (image, class) pair has >= min_views_per_class pixels")
    pb, pk, pc = pb[:T].copy(), pk[:T].copy(), pc[:T].copy()
    flag = ctypes.c_int(0)
    V = int(_c().orc_select_views(_p(pc), T, max_views, max_total, ctypes.byref(flag)))
    flat = np.ascontiguousarray(lbl_s.reshape(n, -1))
    pix = np.empty((T, V), dtype=np.int64)
    if randperm is None:
        assert rng is not None, "need rng or randperm"
        _c().orc_sample_pixels(rng._buf, _p(flat), h * w, _p(pb), _p(pk), T, V, _p(pix))
    else:
        perms = [np.asarray(randperm(int(c)), dtype=np.int64) for c in pc]
        off = np.zeros(T, dtype=np.int64)
        off[1:] = np.cumsum([len(p) for p in perms])[:-1]
        cat = np.ascontiguousarray(np.concatenate(perms)) if perms else np.zeros(0, np.int64)
        _c().orc_sample_pixels_with_perms(_p(flat), h * w, _p(pb), _p(pk), T, V, _p(cat), _p(off),
                                          _p(pix))
    return Plan(scale, h, w, pb, pk, pc, V, bool(flag.value), pix)


# --------------------------------------------------------------------------
# floating-point half
# --------------------------------------------------------------------------
def gather_bank(features: np.ndarray, plan: Plan) -> np.ndarray:
    """sampled_features[i] = features[b, :, idx]  (:123) -> X [T, C, V]."""
    n, C = features.shape[:2]
    f = features.reshape(n, C, -1)
    X = np.empty((plan.T, C, plan.V), dtype=features.dtype)
    for t in range(plan.T):
        X[t] = f[plan.pair_b[t]][:, plan.pix[t]]
    return X


def normalize_bank(X: np.ndarray, eps: float = 1e-12):
    """F.normalize(p=2, dim=1) + transpose + view  (:138-149) -> (Fhat [N, C], nrm [N])."""
    T, C, V = X.shape
    nrm = np.sqrt((X.astype(np.float64) ** 2).sum(axis=1)).astype(X.dtype)   # [T, V]
    den = np.maximum(nrm, X.dtype.type(eps))
    Fh = (X / den[:, None, :]).transpose(0, 2, 1).reshape(T * V, C)
    return np.ascontiguousarray(Fh), nrm.reshape(-1)


def normalize_backward(X: np.ndarray, dFh: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """VJP of F.normalize(dim=1) (clamp_min(norm, eps) denominator) -> dX [T, C, V]."""
    T, C, V = X.shape
    x = X.transpose(0, 2, 1).reshape(T * V, C)
    nrm = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(X.dtype)
    den = np.maximum(nrm, X.dtype.type(eps))
    fh = x / den
    inner = (fh * dFh).sum(axis=1, keepdims=True)
    dx = np.where(nrm > eps, (dFh - fh * inner) / den, dFh / den)
    return np.ascontiguousarray(dx.reshape(T, V, C).transpose(0, 2, 1))


def _row_labels(lab: np.ndarray, V: int) -> np.ndarray:
    """labels.view(-1,1).repeat(1,V).view(-1)  (:141-143)."""
    return np.repeat(np.asarray(lab), V)


def intra_loss(Fh: np.ndarray, lab_rows: np.ndarray, tau: float, want_grad: bool = True):
    """contrastive_loss / get_masks2 / get_loss, DenseContrastiveLossV2.py:127-192.

    Returns (loss, dFh).  No max-shift in the softmax, exactly like the reference."""
    dt = Fh.dtype.type
    N = Fh.shape[0]
    S = (Fh @ Fh.T) / dt(tau)
    same = lab_rows[:, None] == lab_rows[None, :]
    neg = (~same).astype(Fh.dtype)
    pos = same.astype(Fh.dtype)
    np.fill_diagonal(pos, 0)
    E = np.exp(S)
    Z = (E * neg).sum(axis=1, keepdims=True)
    logp = S - np.log(E + Z)
    P = pos.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        mlpp = (pos * logp).sum(axis=1) / P
    loss = -mlpp.mean()
    if not want_grad:
        return loss, None
    inv = 1.0 / (E + Z)
    W = (pos * inv).sum(axis=1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        coef = (1.0 / (N * P))[:, None]
    G = coef * (-pos * Z * inv + neg * E * W)
    dFh = ((G + G.T) @ Fh) / dt(tau)
    return loss, dFh.astype(Fh.dtype)


def cross_loss(F1: np.ndarray, l1: np.ndarray, F2: np.ndarray, l2: np.ndarray, tau: float,
               want_grad: bool = True):
    """ms contrastive_loss / get_masks2 / InfoNce_loss, DenseContrastiveLossV2_ms.py:84-161.

    Returns (loss, dF1, dF2)."""
    dt = F1.dtype.type
    N1 = F1.shape[0]
    S = (F1 @ F2.T) / dt(tau)
    pos = (l1[:, None] == l2[None, :]).astype(F1.dtype)
    neg = 1 - pos
    E = np.exp(S)
    Z = (E * neg).sum(axis=1, keepdims=True)
    logp = S - np.log(E + Z)
    P = pos.sum(axis=1)
    norm = np.where(P > 0, P, 1).astype(F1.dtype)
    loss = -((pos * logp).sum(axis=1) / norm).mean()
    if not want_grad:
        return loss, None, None
    inv = 1.0 / (E + Z)
    W = (pos * inv).sum(axis=1, keepdims=True)
    coef = (1.0 / (N1 * norm))[:, None]
    G = coef * (-pos * Z * inv + neg * E * W)
    dF1 = (G @ F2) / dt(tau)
    dF2 = (G.T @ F1) / dt(tau)
    return loss, dF1.astype(F1.dtype), dF2.astype(F1.dtype)


def scatter_grad(dX: np.ndarray, plan: Plan, shape) -> np.ndarray:
    """Adjoint of the gather at :123 -> dense dfeat [n, C, h, w]."""
    n, C, h, w = shape
    d = np.zeros((n, C, h * w), dtype=dX.dtype)
    for t in range(plan.T):
        d[plan.pair_b[t]][:, plan.pix[t]] += dX[t]
    return d.reshape(n, C, h, w)


@dataclass
class LossConfig:
    """Keys of the flat ``loss`` block consumed by DCV2 / DCV2_ms."""
    num_all_classes: int
    temperature: float = 0.5                    # DenseContrastiveLossV2.py:19
    min_views_per_class: int = 5                # :21
    max_views_per_class: int = 2500             # :27
    max_features_total: int = 10000             # :28
    scales: int = 2                             # ms:21
    weights: Optional[Sequence[float]] = None   # ms:22
    cross_scale_contrast: bool = False          # ms:27
    has_cross_scale_temperature_key: bool = False  # ms:28 quirk -> constant 0.1
    detach_deepest: bool = False                # ms:29
    w_high_low: float = 1.0                     # ms:30
    w_high_mid: float = 1.0                     # ms:31

    @property
    def cross_scale_temperature(self) -> float:
        return 0.1 if self.has_cross_scale_temperature_key else self.temperature


@dataclass
class Result:
    loss: float
    ms_losses: List[float]
    cs_losses: List[float]
    plans: List[Plan]
    grads: List[np.ndarray] = field(default_factory=list)
    log_this_step: List[bool] = field(default_factory=list)


def dcv2_single(label: np.ndarray, features: np.ndarray, cfg: LossConfig,
                rng: Optional[MT19937] = None, randperm=None, want_grad: bool = True,
                dtype=np.float64):
    """DenseContrastiveLossV2.forward (+ backward), DenseContrastiveLossV2.py:44-62.

    Returns (loss, plan, dfeat)."""
    scale = int(label.shape[-1] // features.shape[-1])
    plan = make_plan(label, scale, cfg.num_all_classes, cfg.min_views_per_class,
                     cfg.max_views_per_class, cfg.max_features_total, rng=rng, randperm=randperm)
    feats = features.astype(dtype)
    X = gather_bank(feats, plan)
    Fh, _ = normalize_bank(X)
    rows = _row_labels(plan.pair_k, plan.V)
    loss, dFh = intra_loss(Fh, rows, cfg.temperature, want_grad)
    if not want_grad:
        return float(loss), plan, None
    dX = normalize_backward(X, dFh)
    return float(loss), plan, scatter_grad(dX, plan, feats.shape)


def dcv2_ms(label: np.ndarray, features: Sequence[np.ndarray], cfg: LossConfig,
            rng: Optional[MT19937] = None, randperm=None, want_grad: bool = True,
            dtype=np.float64) -> Result:
    """DenseContrastiveLossV2_ms.forward (+ analytic backward), DenseContrastiveLossV2_ms.py:44-82."""
    S = cfg.scales
    weights = list(cfg.weights) if cfg.weights is not None else [1.0] * S
    assert len(weights) == S
    plans, Xs, Fhs, rows, dFhs, ms = [], [], [], [], [], []
    total = 0.0
    for s in range(S):
        f = features[s].astype(dtype)
        scale = int(label.shape[-1] // f.shape[-1])
        plan = make_plan(label, scale, cfg.num_all_classes, cfg.min_views_per_class,
                         cfg.max_views_per_class, cfg.max_features_total, rng=rng,
                         randperm=randperm)
        X = gather_bank(f, plan)
        Fh, _ = normalize_bank(X)
        r = _row_labels(plan.pair_k, plan.V)
        loss_s, dFh = intra_loss(Fh, r, cfg.temperature, want_grad)
        total += weights[s] * loss_s
        ms.append(float(loss_s))
        plans.append(plan); Xs.append(X); Fhs.append(Fh); rows.append(r)
        dFhs.append(weights[s] * dFh if want_grad else None)
    cs = []
    if cfg.cross_scale_contrast:
        assert S > 1
        tc = cfg.cross_scale_temperature
        terms = [(S - 1, cfg.w_high_low)]
        if S > 2:
            terms.append((S - 2, cfg.w_high_mid))
        for idx, (k, wgt) in enumerate(terms):
            l, d1, d2 = cross_loss(Fhs[0], rows[0], Fhs[k], rows[k], tc, want_grad)
            total += wgt * l
            # ms:66-70: the first term is logged only when the deep bank is NOT detached
            if idx == 1 or not cfg.detach_deepest:
                cs.append(float(l))
            if want_grad:
                dFhs[0] = dFhs[0] + wgt * d1
                if not cfg.detach_deepest:
                    dFhs[k] = dFhs[k] + wgt * d2
    grads = []
    if want_grad:
        for s in range(S):
            dX = normalize_backward(Xs[s], dFhs[s])
            grads.append(scatter_grad(dX, plans[s], features[s].shape))
    return Result(float(total), ms, cs, plans, grads, [p.log_this_step for p in plans])


# --------------------------------------------------------------------------
# Extension without a reference oracle (SURVEY.md section 8 row e): shared negative bank.
# Pinned by (1) world_size = 1 == the reference exactly (all tests above) and (2) this single-process
# emulation: the reference-style InfoNCE with self-mask evaluated on the CONCATENATION of the per-rank
# banks, anchors = the local rows, gradient w.r.t. the local features only (remote banks constant).
# --------------------------------------------------------------------------
def _global_term(Fa, ra, banks_b, rows_b, rank, tau, intra):
    """Anchors Fa (local) vs concat(banks_b).  Returns (loss, dFa, dFb_local)."""
    dt = Fa.dtype.type
    Fcat = np.concatenate(banks_b, axis=0)
    rcat = np.concatenate(rows_b, axis=0)
    off = int(sum(b.shape[0] for b in banks_b[:rank]))
    nb = banks_b[rank].shape[0]
    N1 = Fa.shape[0]
    S = (Fa @ Fcat.T) / dt(tau)
    same = ra[:, None] == rcat[None, :]
    pos = same.astype(Fa.dtype)
    neg = (~same).astype(Fa.dtype)
    if intra:
        pos[np.arange(N1), off + np.arange(N1)] = 0          # self
    E = np.exp(S)
    Z = (E * neg).sum(1, keepdims=True)
    logp = S - np.log(E + Z)
    P = pos.sum(1)
    Pn = P if intra else np.where(P > 0, P, 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        loss = -((pos * logp).sum(1) / Pn).mean()
        inv = 1.0 / (E + Z)
        W = (pos * inv).sum(1, keepdims=True)
        G = (1.0 / (N1 * Pn))[:, None] * (-pos * Z * inv + neg * E * W)
    dFa = (G @ Fcat) / dt(tau)
    Gown = G[:, off:off + nb]
    if intra:
        dFa = dFa + (Gown.T @ Fa) / dt(tau)
        return float(loss), dFa, None
    return float(loss), dFa, (Gown.T @ Fa) / dt(tau)


def dcv2_ms_global(labels, features, cfg: LossConfig, seeds, rank: int, dtype=np.float64) -> Result:
    """Loss and gradients of ``rank`` when every term contrasts against the banks of all ranks.
    labels[q], features[q][s]: inputs of rank q; seeds[q]: that rank's torch seed."""
    world = len(labels)
    S = cfg.scales
    weights = list(cfg.weights) if cfg.weights is not None else [1.0] * S
    plans, Xs, Fhs, rows = [], [], [], []
    for q in range(world):
        rng = MT19937(seeds[q])
        pq, xq, fq, rq = [], [], [], []
        for s in range(S):
            f = features[q][s].astype(dtype)
            scale = int(labels[q].shape[-1] // f.shape[-1])
            plan = make_plan(labels[q], scale, cfg.num_all_classes, cfg.min_views_per_class,
                             cfg.max_views_per_class, cfg.max_features_total, rng=rng)
            X = gather_bank(f, plan)
            Fh, _ = normalize_bank(X)
            pq.append(plan); xq.append(X); fq.append(Fh); rq.append(_row_labels(plan.pair_k, plan.V))
        plans.append(pq); Xs.append(xq); Fhs.append(fq); rows.append(rq)
    total, ms, cs = 0.0, [], []
    dF = [np.zeros_like(Fhs[rank][s]) for s in range(S)]
    for s in range(S):
        l, da, _ = _global_term(Fhs[rank][s], rows[rank][s], [Fhs[q][s] for q in range(world)],
                                [rows[q][s] for q in range(world)], rank, cfg.temperature, True)
        total += weights[s] * l
        ms.append(l)
        dF[s] += weights[s] * da
    if cfg.cross_scale_contrast:
        terms = [(S - 1, cfg.w_high_low)] + ([(S - 2, cfg.w_high_mid)] if S > 2 else [])
        for k, wgt in terms:
            l, da, db = _global_term(Fhs[rank][0], rows[rank][0], [Fhs[q][k] for q in range(world)],
                                     [rows[q][k] for q in range(world)], rank,
                                     cfg.cross_scale_temperature, False)
            total += wgt * l
            cs.append(l)
            dF[0] += wgt * da
            if not cfg.detach_deepest:
                dF[k] += wgt * db
    grads = [scatter_grad(normalize_backward(Xs[rank][s], dF[s]), plans[rank][s], features[rank][s].shape)
             for s in range(S)]
    return Result(float(total), ms, cs, plans[rank], grads, [p.log_this_step for p in plans[rank]])


# ---- per-step metrics -------------------------------------------------------------

def confusion_matrix(logits: np.ndarray, target: np.ndarray, with_ignore: bool) -> np.ndarray:
    """t_get_confusion_matrix (utils/torch_utils.py:157-183), restated with integer arithmetic:
    p = argmax over C (first maximal index, NaN maximal -- torch.argmax); one_hot(target, C + 1)[:, :-1] when the
    experiment has an ignore class (:172-175) else one_hot(target, C); cm = one_hot(p)^T @ one_hot(t) -> int32
    [C, C], rows = predicted, columns = target.  Targets outside the one-hot range raise, like F.one_hot."""
    n, C = logits.shape[0], logits.shape[1]
    flat = np.moveaxis(logits.reshape(n, C, -1), 1, 0).reshape(C, -1)          # :163-164
    nan = np.isnan(flat)
    key = np.where(nan, np.inf, flat)
    pred = np.argmax(key, axis=0)                                               # first maximum; NaN first among NaNs
    t = target.reshape(-1).astype(np.int64)
    cols = C + 1 if with_ignore else C
    if t.size and (t.min() < 0 or t.max() >= cols):
        raise RuntimeError("Class values must be smaller than num_classes.")
    cm = np.bincount(pred * cols + t, minlength=C * cols).reshape(C, cols)[:, :C]
    return cm.astype(np.int32)


def pixel_accuracy(cm: np.ndarray):
    """t_get_pixel_accuracy (utils/torch_utils.py:201-213), fp32."""
    diag = np.diag(cm).astype(np.float32)
    acc = diag.sum(dtype=np.float32) / np.float32(cm.sum())
    rows = cm.sum(1).astype(np.float32)
    rows[rows == 0] = 1
    return np.float32(acc), np.float32(np.mean(diag / rows, dtype=np.float32))


def mean_iou(cm: np.ndarray) -> np.float32:
    """t_get_miou over all classes (utils/torch_utils.py:253-283): NaN IoU -> 0 and kept in the mean."""
    diag = np.diag(cm).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        iou = diag / (cm.sum(0).astype(np.float32) + cm.sum(1).astype(np.float32) - diag)
    iou[np.isnan(iou)] = 0
    return np.float32(iou.mean(dtype=np.float32))
