"""Dense contrastive loss engine, stage 2: the L2-normalised feature banks (K3 dcl_gather_normalize; reference losses/
DenseContrastiveLossV2.py:123, :138-149) and, on several ranks, the all-gathered shared negative bank (BASELINE north_star;
reference utils/distributed.py:50-55 concat_all_gather has no gradient either)."""
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

__all__ = [_n for _n in dir() if not _n.startswith('__')]      # private helpers too: the stage modules share them
