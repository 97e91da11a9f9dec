"""Dense contrastive loss engine, shared state: configuration, per-scale / per-term records of a step, pinned staging rings,
small helpers.  Stage modules: engine_plan (labels -> sampling plan -> sampled pixels), engine_banks (feature banks, the shared
negative bank of several ranks), engine (InfoNCE terms forward / backward, the autograd node).  Reference: losses/
DenseContrastiveLossV2.py:44-192, DenseContrastiveLossV2_ms.py:44-161."""
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

__all__ = [_n for _n in dir() if not _n.startswith('__')]      # private helpers too: the stage modules share them
