"""CPU tests of the host half: C-ABI surface, plan construction vs the oracle, positive ranges."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_names, load_golden, num_classes_for

import mscs_amd  # noqa: F401
from mscs_amd import _lib
from mscs_amd.losses.plan import build_host_plan, positive_ranges, select_views_per_class


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dcl_hip.h")).read()
    return sorted(set(re.findall(r"\b(dcl_[a-z0-9_]+)\s*\(", hdr)))


def test_library_builds_and_exports_every_declared_symbol():
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 11
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/dcl_hip.h but not exported"
    # and the ctypes table binds exactly the declared entry points (dcl_last_error, dcl_last_kernel: char * results, bound separately)
    assert set(_lib.SIGNATURES) | {"dcl_last_error", "dcl_last_kernel"} == set(names)


def test_suggest_nsplit_is_sane():
    L = _lib.lib()
    for n1, n2 in [(9804, 9804), (2736, 9804), (50, 9900), (128, 128), (1, 1)]:
        s = L.dcl_suggest_nsplit(n1, n2)
        assert 1 <= s <= 32
        assert s <= max(1, (n2 + 31) // 32)
    assert L.dcl_version() >= 1


@pytest.mark.parametrize("name", golden_names(["G1_", "G2", "G3", "G4", "G9"]))
def test_host_plan_matches_reference_fixture(oracle, name):
    """counts (from the oracle's label path) -> build_host_plan with torch.randperm -> the pixel
    list implied by (sel, ascending positions) equals the reference's sampled indices bit for bit."""
    g = load_golden(name)
    c = g["config"]
    K = num_classes_for(c)
    S = c.get("scales", 1) if name.startswith(("G2", "G4", "G9")) else 1
    lab = g["label"].astype(np.int64)
    torch.manual_seed(int(g["seed"]))
    for s in range(S):
        feat = g[f"feat{s}"]
        scale = lab.shape[-1] // feat.shape[-1]
        lbl_s = oracle.downsample_labels(lab, scale)
        counts = oracle.class_counts(lbl_s, K)
        plan = build_host_plan(counts, c.get("min_views_per_class", 5),
                               c.get("max_views_per_class", 2500), c.get("max_features_total", 10000))
        assert plan.V == int(g[f"s{s}_V"])
        np.testing.assert_array_equal(plan.pair_b, g[f"s{s}_pair_b"])
        np.testing.assert_array_equal(plan.pair_k, g[f"s{s}_pair_k"])
        assert plan.log_this_step == bool(g[f"s{s}_log_this_step"])
        flat = lbl_s.reshape(lbl_s.shape[0], -1)
        pix = np.stack([np.flatnonzero(flat[b] == k)[plan.sel[t]]
                        for t, (b, k) in enumerate(zip(plan.pair_b, plan.pair_k))])
        np.testing.assert_array_equal(pix, g[f"s{s}_pix"])
        # class-major slot order is a stable sort of the pair list by class
        cls = plan.pair_k[plan.slot_pair]
        assert np.all(np.diff(cls) >= 0)
        for cc in np.unique(cls):
            assert np.all(np.diff(plan.slot_pair[cls == cc]) > 0)
            assert plan.cls_hi[cc] - plan.cls_lo[cc] == np.sum(cls == cc)


def test_select_views_rule():
    assert select_views_per_class(43, 228, 2500, 10000) == (43, False)
    assert select_views_per_class(3000, 2, 2500, 10000) == (2500, True)
    assert select_views_per_class(3000, 2, 1, 10000) == (3000, False)
    assert select_views_per_class(3000, 5, 1, 10000) == (2000, True)
    assert select_views_per_class(50, 228, 2500, 10000) == (43, True)


def test_positive_ranges_cover_same_class_rows():
    rng = np.random.RandomState(0)
    ca = rng.randint(0, 40, size=(3, 8)); ca[:, -1] = 100
    cb = rng.randint(0, 40, size=(3, 8)); cb[:, -1] = 100
    pa = build_host_plan(ca, 5, 2500, 10000)
    pb = build_host_plan(cb, 5, 2500, 10000)
    lo, hi = positive_ranges(pa, pb)
    rows_b = np.repeat(pb.pair_k[pb.slot_pair], pb.V)
    for u in range(pa.T):
        cls = pa.pair_k[pa.slot_pair[u]]
        expect = np.flatnonzero(rows_b == cls)
        if expect.size == 0:
            assert lo[u] == hi[u]
        else:
            assert (lo[u], hi[u]) == (expect[0], expect[-1] + 1)
            assert np.array_equal(expect, np.arange(lo[u], hi[u]))


def test_no_pairs_raises_clear_error():
    counts = np.zeros((2, 20), dtype=np.int32)
    counts[:, 19] = 500
    with pytest.raises(RuntimeError, match="min_views_per_class"):
        build_host_plan(counts, 5, 2500, 10000)


def test_loss_modules_fail_loudly_without_gpu():
    from mscs_amd.losses import DenseContrastiveLossV2
    m = DenseContrastiveLossV2({"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 16, 16, dtype=torch.long), torch.randn(1, 8, 4, 4))


@pytest.mark.parametrize("seed", [0, 7, 2 ** 31 + 5])
def test_native_rng_equals_torch_randperm_and_keeps_stream(seed):
    """csrc/dcl_host_rng.cpp: same sel as T torch.randperm calls AND the same generator state after."""
    rs = np.random.RandomState(seed % 1000)
    counts = rs.randint(0, 3000, size=(4, 20)).astype(np.int64)
    counts[0, 3] = 5                                   # the minimum -> V = 5
    torch.manual_seed(seed)
    torch.rand(7)                                      # arbitrary stream position (mid-block)
    st = torch.get_rng_state()
    a = build_host_plan(counts, 5, 2500, 10000, native_rng=False)
    after_a = torch.rand(5)
    torch.set_rng_state(st)
    b = build_host_plan(counts, 5, 2500, 10000, native_rng=True)
    after_b = torch.rand(5)
    np.testing.assert_array_equal(a.sel, b.sel)
    assert torch.equal(after_a, after_b)
    # crossing several MT19937 block boundaries (624 draws per block)
    torch.set_rng_state(st)
    big = np.zeros((1, 3), dtype=np.int64); big[0, 0] = 40000; big[0, 1] = 1300
    c = build_host_plan(big, 5, 2500, 10000, native_rng=False)
    torch.set_rng_state(st)
    d = build_host_plan(big, 5, 2500, 10000, native_rng=True)
    np.testing.assert_array_equal(c.sel, d.sel)
    # many views per pair (V = 300: the sparse-permutation path with a large table, lists on both sides of its 1024 threshold)
    torch.set_rng_state(st)
    wide = rs.randint(600, 3000, size=(2, 9)).astype(np.int64)
    wide[1, 2], wide[0, 5] = 1024, 1025
    e = build_host_plan(wide, 5, 300, 10000, native_rng=False)
    after_e = torch.rand(3)
    torch.set_rng_state(st)
    f = build_host_plan(wide, 5, 300, 10000, native_rng=True)
    assert e.V == f.V == 300
    np.testing.assert_array_equal(e.sel, f.sel)
    assert torch.equal(after_e, torch.rand(3))


def test_tapup_support_query_needs_no_gpu():
    """dcl_tapup_supported: pure host arithmetic (LDS tile sizes of the tap gather, forward and both backward forms) -- the
    benchmark's head shapes fit, an absurdly wide map does not (the caller then convolves the materialised concatenation)."""
    from mscs_amd import _lib
    L = _lib.lib()
    for ac in (0, 1):
        assert L.dcl_tapup_supported(32, 64, 16, 32, 128, 256, ac) == 1          # HRNet-W48 head, 512 x 1024 input
        assert L.dcl_tapup_supported(16, 16, 32, 32, 128, 128, ac) == 1          # UPerNet fusion, 512 x 512
        assert L.dcl_tapup_supported(20, 20, 0, 0, 160, 160, ac) == 1            # Swin-L at 640 x 640, one source
        assert L.dcl_tapup_supported(64, 16384, 0, 0, 256, 65536, ac) == 0
    assert L.dcl_tapup_supported(0, 4, 0, 0, 8, 8, 0) == 0
