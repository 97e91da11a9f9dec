"""Host half of the sampling step: from the per-image class histogram to a sampling plan.

Pure host logic (numpy + ``torch.randperm`` on the CPU default generator) so that the sampled
pixel indices are bit-identical to the reference under a fixed seed: the reference draws one
``torch.randperm(count)`` per (image, class) pair, pairs in row-major (image, class) order, scales
in order (losses/DenseContrastiveLossV2.py:106-107, 117-122; DenseContrastiveLossV2_ms.py:51-60).
The device only compacts / selects / gathers (csrc/dcl_sampling.hip, dcl_bank.hip).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch


@dataclass
class HostPlan:
    """Sampling plan of one scale."""
    K: int
    T: int
    V: int
    pair_b: np.ndarray       # int32 [T]  image index, reference (row-major) order
    pair_k: np.ndarray       # int32 [T]  class id
    pair_cnt: np.ndarray     # int32 [T]  pixels of that class in that image
    sel: np.ndarray          # int32 [T, V]  first V entries of randperm(pair_cnt[t])
    slot_pair: np.ndarray    # int32 [T]  bank slot u holds pair slot_pair[u] (class-major order)
    cls_lo: np.ndarray       # int32 [K]  bank slots [cls_lo[c], cls_hi[c]) hold class c
    cls_hi: np.ndarray       # int32 [K]
    log_this_step: bool

    @property
    def N(self) -> int:
        return self.T * self.V



class NoQualifyingPair(RuntimeError):
    """No (image, class) pair reaches ``min_views_per_class`` at a scale: the one planning error that depends on the DATA
    of a step (the reference dies in torch.min() of an empty tensor, losses/DenseContrastiveLossV2.py:110).  Its own type so
    that ``DenseContrastiveLossV2_ms.prepare`` can defer exactly this error to forward() and let every other one surface."""


def select_views_per_class(min_views: int, total_cls: int, max_views_per_class: int,
                           max_features_total: int):
    """``_select_views_per_class`` (DenseContrastiveLossV2.py:64-84) -> (V, log_this_step)."""
    log = False
    if max_views_per_class == 1:
        v = min_views                       # "no capping" sentinel of the reference
    else:
        v = min(min_views, max_views_per_class)
        if v == max_views_per_class:
            log = True
    if v * total_cls > max_features_total:
        v = max_features_total // total_cls
        log = True
    return int(v), log


def _native_randperm_select(cnt: np.ndarray, V: int) -> np.ndarray:
    """First V entries of torch.randperm(cnt[t]) for every pair, drawn from torch's global CPU
    generator by the native MT19937 stepper (csrc/dcl_host_rng.cpp): same values, same final
    generator state as T separate torch.randperm calls."""
    import ctypes
    from .. import _lib
    L = _lib.lib()
    T = int(cnt.shape[0])
    counts32 = np.ascontiguousarray(cnt, dtype=np.int32)
    sel = np.empty((T, V), dtype=np.int32)
    state = torch.get_rng_state()
    _lib.check(L.dcl_host_randperm_select(ctypes.c_void_p(state.data_ptr()), state.numel(),
                                          counts32.ctypes.data_as(ctypes.c_void_p), T, V,
                                          sel.ctypes.data_as(ctypes.c_void_p)),
               "dcl_host_randperm_select")
    torch.set_rng_state(state)
    return sel


def build_host_plan(counts: np.ndarray, min_views_per_class: int, max_views_per_class: int,
                    max_features_total: int,
                    randperm: Optional[Callable[[int], torch.Tensor]] = None,
                    native_rng: bool = True) -> HostPlan:
    """counts: int [n, K] per-image class histogram of the down-sampled label map.

    Follows sample_anchors_fast (DenseContrastiveLossV2.py:100-124): pairs = where(counts[:, :-1]
    >= min_views) in row-major order (the last class column is ALWAYS dropped), V from the minimum
    count over pairs, one randperm per pair in pair order.  Draws: ``randperm`` callable if given,
    else the native stepper of torch's global generator (``native_rng``), else torch.randperm."""
    counts = np.asarray(counts)
    n, K = counts.shape
    pb, pk = np.nonzero(counts[:, :-1] >= min_views_per_class)      # row-major, like torch.where
    T = int(pb.shape[0])
    if T == 0:
        # the reference dies in torch.min() of an empty tensor (:110); say why instead
        raise NoQualifyingPair(
            "DenseContrastiveLoss: no (image, class) pair has >= min_views_per_class="
            f"{min_views_per_class} pixels at this scale (reference: RuntimeError in torch.min, "
            "losses/DenseContrastiveLossV2.py:110)")
    cnt = counts[pb, pk].astype(np.int64)
    V, log = select_views_per_class(int(cnt.min()), T, max_views_per_class, max_features_total)
    if randperm is None and native_rng:
        sel = _native_randperm_select(cnt, V)
    else:
        draw = randperm or torch.randperm
        sel = np.empty((T, V), dtype=np.int32)
        for t in range(T):
            sel[t] = draw(int(cnt[t]))[:V].numpy()
    pk32 = pk.astype(np.int32)
    slot_pair = np.argsort(pk32, kind="stable").astype(np.int32)     # class-major bank order
    per_cls = np.bincount(pk32, minlength=K).astype(np.int32)
    cls_hi = np.cumsum(per_cls).astype(np.int32)
    cls_lo = (cls_hi - per_cls).astype(np.int32)
    return HostPlan(K=K, T=T, V=V, pair_b=pb.astype(np.int32), pair_k=pk32,
                    pair_cnt=cnt.astype(np.int32), sel=sel, slot_pair=slot_pair, cls_lo=cls_lo,
                    cls_hi=cls_hi, log_this_step=log)


def positive_ranges(anchor: HostPlan, contrast: HostPlan):
    """For every bank slot of ``anchor``: the contiguous ROW range [lo, hi) of ``contrast``'s bank
    holding the same class (empty range = class absent there).  int32 [T_anchor] each."""
    cls = anchor.pair_k[anchor.slot_pair]
    lo = (contrast.cls_lo[cls] * contrast.V).astype(np.int32)
    hi = (contrast.cls_hi[cls] * contrast.V).astype(np.int32)
    return lo, hi
