"""Per-step metrics: oracle and host path against fixtures produced by the reference's
own t_get_confusion_matrix / t_get_pixel_accuracy / t_get_mean_iou (tests/golden/G12_metrics.npz,
tools/gen_golden_metrics.py); the HIP kernel (through the C ABI) against fixture and oracle, bit-exact."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

import mscs_amd  # noqa: F401
from mscs_amd.utils.metrics import (t_get_confusion_matrix, t_get_mean_iou, t_get_pixel_accuracy, out_of_range,
                                    t_metrics_from_confusion_matrix)

CASES = ["cts", "cts_ties", "ade", "cadis_noignore"]


def _case(name):
    z = np.load(os.path.join(GOLDEN, "G12_metrics.npz"))
    return {k[len(name) + 2:]: z[k] for k in z.files if k.startswith(name + "__")}


@pytest.mark.parametrize("name", CASES)
def test_oracle_metrics_match_reference(oracle, name):
    g = _case(name)
    with_ignore = name != "cadis_noignore"
    cm = oracle.confusion_matrix(g["logits"], g["target"], with_ignore)
    np.testing.assert_array_equal(cm, g["cm"])
    pa, pac = oracle.pixel_accuracy(cm)
    np.testing.assert_allclose([pa, pac, oracle.mean_iou(cm)], [g["pa"], g["pac"], g["miou"]], rtol=1e-6)


def _check_host(g, dev):
    ds, exp = str(g["dataset"]), int(g["experiment"])
    logits, target = torch.from_numpy(g["logits"]).to(dev), torch.from_numpy(g["target"]).to(dev)
    cm = t_get_confusion_matrix(logits, target, ds)
    assert cm.dtype == torch.int32
    np.testing.assert_array_equal(cm.cpu().numpy(), g["cm"])                     # bit-exact
    for tdt in (torch.int64, torch.uint8):
        np.testing.assert_array_equal(t_get_confusion_matrix(logits, target.to(tdt), ds).cpu().numpy(), g["cm"])
    cm2 = t_get_confusion_matrix(logits.flip(0), target, ds, existing_matrix=cm.clone())
    np.testing.assert_array_equal(cm2.cpu().numpy(), g["cm_accumulated"])
    pa, pac = t_get_pixel_accuracy(cm)
    np.testing.assert_allclose([pa.item(), pac.item()], [g["pa"], g["pac"]], rtol=1e-6)
    np.testing.assert_allclose(t_get_mean_iou(cm, exp, ds)["mean_iou"].item(), g["miou"], rtol=1e-6)
    np.testing.assert_allclose(t_get_mean_iou(cm).item(), g["miou"], rtol=1e-6)
    fused = t_metrics_from_confusion_matrix(cm)           # one HIP launch on the GPU, the torch path on the CPU
    np.testing.assert_allclose([v.item() for v in fused], [g["pa"], g["pac"], g["miou"]], rtol=2e-6)


@pytest.mark.parametrize("name", CASES)
def test_host_metrics_match_reference(name):
    _check_host(_case(name), torch.device("cpu"))


def test_cpu_out_of_range_target_raises_like_one_hot():
    with pytest.raises(RuntimeError, match="smaller than num_classes"):
        t_get_confusion_matrix(torch.randn(1, 19, 4, 4), torch.full((1, 4, 4), 20), "CITYSCAPES")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_confusion_matrix_matches_reference(name):
    _check_host(_case(name), torch.device("cuda:0"))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,K", [((12, 19, 512, 1024), 20), ((16, 150, 128, 128), 151), ((3, 19, 33, 47), 20),
                                     ((2, 200, 16, 24), 200)])
def test_hip_confusion_matrix_at_size_vs_oracle(oracle, shape, K):
    """Benchmark shape (12 x 19 x 512 x 1024), the ADE20K matrix (LDS histogram of 150 x 151), a ragged plane
    (HW % 4 != 0) and a matrix beyond the LDS budget (global atomics); NaNs and ties planted."""
    dev = torch.device("cuda:0")
    n, C, H, W = shape
    gen = torch.Generator().manual_seed(1)
    logits = torch.randn(n, C, H, W, generator=gen)
    logits[0, :, :4, :8] = 0.25                              # ties across all classes -> class 0
    logits[0, 3, 5, :8] = float("nan")                       # NaN wins
    logits[0, 5, 5, :4] = float("nan")                       # first NaN wins
    target = torch.randint(0, K, (n, H, W), generator=gen)
    ds = "CITYSCAPES" if C == 19 else ("ADE20K" if C == 150 else "CADIS")
    with_ignore = K == C + 1
    if not with_ignore:
        from mscs_amd.utils.datasets_info import register_dataset
        register_dataset("SYN200", [f"c{i}" for i in range(C)], ignore=False)
        ds = "SYN200"
    ref = oracle.confusion_matrix(logits.numpy(), target.numpy(), with_ignore)
    before = int(out_of_range(dev).item())
    cm = t_get_confusion_matrix(logits.to(dev), target.to(dev), ds)
    np.testing.assert_array_equal(cm.cpu().numpy(), ref)
    assert int(cm.sum().item()) == int((target < C).sum().item())           # every non-ignored pixel counted once
    assert int(out_of_range(dev).item()) == before
    bad = target.clone()
    bad[0, 0, :7] = K + 3
    t_get_confusion_matrix(logits.to(dev), bad.to(dev), ds)
    assert int(out_of_range(dev).item()) == before + 7


@pytest.mark.gpu
@pytest.mark.parametrize("total,C,K,tdtype", [(12 * 512 * 1024, 19, 20, torch.int64), (1000003, 19, 20, torch.uint8),
                                               (16 * 128 * 128 + 2, 150, 151, torch.int32), (5, 19, 19, torch.int64),
                                               (4096 + 1, 200, 200, torch.int64)])
def test_hip_confusion_matrix_from_argmax_map(total, C, K, tdtype):
    """dcl_confusion_matrix_pred (the histogram of (arg-max map, target) that the fused up-sampling + CE path leaves to the
    metrics tail; 512 workgroups, four pixels per thread and trip): exact counts against numpy for the benchmark size, totals
    that are not multiples of four, the three target types, the LDS and the global-atomic histogram, out-of-range targets
    counted and skipped."""
    from mscs_amd import _lib
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(total % 1000)
    pred = torch.randint(0, C, (total,), generator=gen).to(torch.uint8)
    hi = 256 if tdtype == torch.uint8 else K + 5
    target = torch.randint(0, min(hi, K + 5), (total,), generator=gen).to(tdtype)
    ref = np.zeros((C, K), dtype=np.int64)
    tn, pn = target.numpy().astype(np.int64), pred.numpy().astype(np.int64)
    ok = tn < K
    np.add.at(ref, (pn[ok], tn[ok]), 1)
    cm = torch.zeros((C, K), dtype=torch.int32, device=dev)
    oob = torch.zeros(1, dtype=torch.int32, device=dev)
    p, t = pred.to(dev), target.to(dev)
    _lib.check(_lib.lib().dcl_confusion_matrix_pred(_lib.ptr(p), total, _lib.ptr(t), t.element_size(), C, K, _lib.ptr(cm),
                                                    _lib.ptr(oob), _lib.stream_ptr(dev)), "dcl_confusion_matrix_pred")
    np.testing.assert_array_equal(cm.cpu().numpy(), ref)
    assert int(oob.item()) == int((~ok).sum())
