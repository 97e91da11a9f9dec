"""oracle/eager_torch.py -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

Eager-PyTorch restatement of the dense contrastive loss that deliberately KEEPS THE REFERENCE'S
OP STRUCTURE: a Python loop over (image, class) pairs with tensor-scalar indexing, ``nonzero`` and a
host ``torch.randperm`` per pair, a materialised N x N similarity matrix with explicit float masks,
and autograd for the backward (one index-backward per pair).  It exists for two measurements
(BASELINE.md section 3):
  * ``cpu_baseline`` in bench.py -- this code on the GPU box's host cores (kind "port");
  * the "reference PyTorch-eager GPU step" comparator -- this code on the MI355X with stock ops.
It is validated against the golden vectors in tests/test_oracle.py.  The product never imports it.

Reference lines restated: losses/DenseContrastiveLossV2.py:44-206,
losses/DenseContrastiveLossV2_ms.py:44-161.
"""
import torch
import torch.nn.functional as F


def _views_per_class(min_views, total, max_views, max_total):
    """DenseContrastiveLossV2.py:64-84."""
    v = min_views if max_views == 1 else min(min_views, max_views)
    if v * total > max_total:
        v = max_total // total
    return v


def sample_bank(label, feats, K, min_views, max_views, max_total):
    """DenseContrastiveLossV2.py:44-48, 86-125, 194-206 -> (X [T, C, V], classes [T])."""
    n, H, W = label.shape
    scale = int(W // feats.shape[-1])
    with torch.no_grad():
        small = F.interpolate(label.unsqueeze(1).float(), (H // scale, W // scale), mode="nearest").long()
    c = feats.shape[1]
    flat_f = feats.view(n, c, -1)
    flat_l = small.view(n, -1)
    ids = torch.arange(K, device=flat_l.device)
    onehot = flat_l.unsqueeze(-1) == ids.view(1, 1, -1)
    counts = onehot.sum(1)
    img_idx, cls_idx = torch.where(counts[:, :-1] >= min_views)
    smallest = torch.min(counts[img_idx, cls_idx])
    total = cls_idx.shape[0]
    V = _views_per_class(int(smallest.item()), total, max_views, max_total)
    bank = torch.zeros((total, c, V), dtype=torch.float, device=feats.device)
    classes = torch.zeros(total, dtype=torch.float, device=feats.device)
    for t in range(total):
        where = onehot[img_idx[t], :, cls_idx[t]].nonzero().squeeze()
        perm = torch.randperm(where.shape[0]).to(where.device)
        chosen = where[perm[:V]]
        bank[t] = flat_f[img_idx[t], :, chosen]
        classes[t] = cls_idx[t]
    return bank, classes


def _flatten(bank, classes):
    f = F.normalize(bank, p=2, dim=1).transpose(1, 2)
    T, V, c = f.shape
    rows = classes.contiguous().view(-1, 1).repeat(1, V).view(-1, 1)
    return f.contiguous().view(-1, c), rows


def intra_loss(bank, classes, tau):
    """DenseContrastiveLossV2.py:127-192 (dense N x N, no max-shift)."""
    f, rows = _flatten(bank, classes)
    N = f.shape[0]
    same = torch.eq(rows, rows.t()).float()
    neg = 1 - same
    keep = torch.ones_like(same).scatter_(1, torch.arange(N, device=f.device).view(-1, 1), 0)
    pos = same * keep
    logits = torch.matmul(f, f.t()) / tau
    neg_sum = (torch.exp(logits) * neg).sum(1, keepdim=True)
    log_prob = logits - torch.log(torch.exp(logits) + neg_sum)
    return -((pos * log_prob).sum(1) / pos.sum(1)).mean()


def cross_loss(bank1, classes1, bank2, classes2, tau):
    """DenseContrastiveLossV2_ms.py:84-161."""
    f1, r1 = _flatten(bank1, classes1)
    f2, r2 = _flatten(bank2, classes2)
    pos = torch.eq(r1, r2.t()).float()
    neg = 1 - pos
    logits = torch.matmul(f1, f2.t()) / tau
    neg_sum = (torch.exp(logits) * neg).sum(1, keepdim=True)
    log_prob = logits - torch.log(torch.exp(logits) + neg_sum)
    cnt = pos.sum(1)
    norm = torch.where(cnt > 0, cnt, torch.ones_like(cnt))
    return -((pos * log_prob).sum(1) / norm).mean()


def dcv2_ms(label, feats, K, tau, weights, cross=False, cross_tau=None, min_views=5, max_views=2500,
            max_total=10000, detach_deepest=False, w_high_low=1.0, w_high_mid=1.0):
    """DenseContrastiveLossV2_ms.py:44-82 -> (total, ms_losses, cs_losses)."""
    S = len(weights)
    total = torch.tensor(0.0, device=feats[0].device)
    banks, ms, cs = [], [], []
    for s in range(S):
        bank, classes = sample_bank(label, feats[s], K, min_views, max_views, max_total)
        l = intra_loss(bank, classes, tau)
        total = total + weights[s] * l
        ms.append(l.detach())
        banks.append((bank, classes))
    if cross:
        ct = tau if cross_tau is None else cross_tau
        deep = banks[-1][0].detach() if detach_deepest else banks[-1][0]
        l = cross_loss(banks[0][0], banks[0][1], deep, banks[-1][1], ct)
        if not detach_deepest:
            cs.append(l.detach())
        total = total + w_high_low * l
        if S > 2:
            mid = banks[-2][0].detach() if detach_deepest else banks[-2][0]
            l2 = cross_loss(banks[0][0], banks[0][1], mid, banks[-2][1], ct)
            total = total + w_high_mid * l2
            cs.append(l2.detach())
    return total, ms, cs
