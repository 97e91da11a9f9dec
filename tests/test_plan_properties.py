"""Property-based tests (hypothesis) of the host plan: structure of the sampling plan for arbitrary
class histograms, agreement with the oracle's C restatement, and positive-range bookkeeping."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

import mscs_amd  # noqa: F401
from mscs_amd.losses.plan import build_host_plan, positive_ranges, select_views_per_class


@st.composite
def histograms(draw):
    n = draw(st.integers(1, 4))
    K = draw(st.integers(2, 24))
    counts = draw(st.lists(st.integers(0, 400), min_size=n * K, max_size=n * K))
    c = np.array(counts, dtype=np.int64).reshape(n, K)
    c[draw(st.integers(0, n - 1)), draw(st.integers(0, K - 2))] = draw(st.integers(5, 400))   # one pair qualifies
    return c


@settings(max_examples=60, deadline=None)
@given(histograms(), st.integers(1, 50), st.integers(20, 3000), st.integers(0, 2 ** 31 - 1))
def test_plan_structure(counts, max_views, max_total, seed):
    torch.manual_seed(seed)
    p = build_host_plan(counts, 5, max_views, max_total)
    n, K = counts.shape
    # pairs: exactly the (image, class < K-1) cells with >= 5 pixels, row-major
    want = [(b, k) for b in range(n) for k in range(K - 1) if counts[b, k] >= 5]
    assert list(zip(p.pair_b.tolist(), p.pair_k.tolist())) == want
    m = min(counts[b, k] for b, k in want)
    v, _ = select_views_per_class(int(m), len(want), max_views, max_total)
    assert p.V == v and p.sel.shape == (p.T, p.V)
    if p.V > 0:
        for t in range(p.T):                       # distinct ranks below the pair's pixel count
            row = p.sel[t]
            assert len(set(row.tolist())) == p.V and row.min() >= 0 and row.max() < p.pair_cnt[t]
    cls = p.pair_k[p.slot_pair]
    assert np.all(np.diff(cls) >= 0) and sorted(p.slot_pair.tolist()) == list(range(p.T))
    lo, hi = positive_ranges(p, p)
    for u in range(p.T):                           # a slot lies inside its own positive range
        assert lo[u] <= u * p.V < hi[u] or p.V == 0
        assert hi[u] - lo[u] == np.sum(cls == cls[u]) * p.V


@settings(max_examples=25, deadline=None)
@given(histograms(), st.integers(0, 2 ** 31 - 1))
def test_native_and_python_draws_agree(counts, seed):
    torch.manual_seed(seed)
    a = build_host_plan(counts, 5, 2500, 10000, native_rng=True)
    ra = torch.rand(3)
    torch.manual_seed(seed)
    b = build_host_plan(counts, 5, 2500, 10000, native_rng=False)
    rb = torch.rand(3)
    assert np.array_equal(a.sel, b.sel) and torch.equal(ra, rb)
