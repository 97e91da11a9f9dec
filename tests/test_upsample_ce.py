"""Fused bilinear up-sampling + weighted cross-entropy through the C ABI against
F.interpolate + nn.CrossEntropyLoss in float64: loss, gradient w.r.t. the low-resolution logits, arg-max map."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,C,h,w,H,W,align,weighted", [
    (2, 19, 16, 32, 64, 128, True, True), (2, 19, 16, 32, 64, 128, False, True), (1, 150, 8, 8, 32, 32, False, False),
    (2, 7, 9, 13, 36, 52, True, True), (3, 19, 5, 6, 20, 24, False, False), (12, 19, 128, 256, 512, 1024, True, True),
    (2, 150, 128, 128, 512, 512, False, False)])
def test_upsample_ce_matches_interpolate_plus_cross_entropy(n, C, h, w, H, W, align, weighted):
    from mscs_amd.models.ops import UpsampledLogits
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n * 1000 + C + h)
    z = torch.randn(n, C, h, w, generator=gen) * 3
    target = torch.randint(0, C + 1, (n, H, W), generator=gen)                 # class C = ignore id
    weight = (torch.rand(C, generator=gen) + 0.5) if weighted else None
    big = n * C * H * W > 5e7
    dt = torch.float32 if big else torch.float64                                # the full-size reference in fp32 on the GPU
    zr = z.to(dev, dt).requires_grad_(True)
    full = F.interpolate(zr, size=(H, W), mode="bilinear", align_corners=align)
    ref = F.cross_entropy(full, target.to(dev), weight=None if weight is None else weight.to(dev, dt), ignore_index=C)
    ref.backward()
    zh = z.to(dev).requires_grad_(True)
    lz = UpsampledLogits(zh, (H, W), align)
    loss = lz.cross_entropy(target.to(dev), weight=None if weight is None else weight.to(dev), ignore_index=C)
    (loss * 1.7).backward()                                                     # non-trivial upstream gradient
    tol = 2e-5 if big else 2e-6
    assert abs(loss.item() - ref.item()) <= tol * abs(ref.item())
    g, r = zh.grad.double() / 1.7, zr.grad.double()
    assert (g - r).abs().max().item() <= (2e-4 if big else 1e-5) * r.abs().max().item()
    # arg-max map: equal to torch's wherever the top two logits are not within round-off of each other
    top2 = full.detach().float().topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(lz.pred.long()[clear], full.detach().argmax(1)[clear])
    assert lz.shape == (n, C, H, W)
    assert (lz.materialize().detach().double() - full.detach().double()).abs().max().item() <= 1e-5 * full.abs().max().item()


def test_upsample_ce_all_ignored_is_nan_like_torch():
    from mscs_amd.models.ops import UpsampledLogits
    dev = torch.device("cuda:0")
    z = torch.randn(1, 5, 4, 4, device=dev, requires_grad=True)
    t = torch.full((1, 8, 8), 5, device=dev)
    loss = UpsampledLogits(z, (8, 8), True).cross_entropy(t, ignore_index=5)
    ref = F.cross_entropy(F.interpolate(z, size=(8, 8), mode="bilinear", align_corners=True), t, ignore_index=5)
    assert torch.isnan(loss) and torch.isnan(ref)


def test_lazy_logits_training_step_equals_materialised_step():
    """HRNetManager step with graph.lazy_logits (fused up-sampling + CE, confusion matrix from the arg-max map)
    against the same step with materialised logits: loss, loss_vals, parameter gradients, confusion matrix."""
    import mscs_amd  # noqa: F401
    from mscs_amd.managers import HRNetManager
    from mscs_amd.utils import set_verbosity
    from mscs_amd.utils.metrics import t_get_confusion_matrix
    set_verbosity(40)
    dev = torch.device("cuda:0")
    res = {}
    for lazy in (False, True):
        cfg = {"name": "t", "mode": "training", "manager": "HRNet", "cuda": True, "parallel": False, "seed": 3,
               "graph": {"model": "HRNet", "backbone": "hrnet18", "sync_bn": False, "pretrained": False,
                         "align_corners": True, "lazy_logits": lazy,
                         "ms_projector": {"mlp": [[1, -1, 1]], "scales": 2, "d": 64, "use_bn": True}},
               "data": {"dataset": "CITYSCAPES", "experiment": 1, "batch_size": 2, "synthetic": True,
                        "synthetic_length": 4, "transform_values": {"crop_shape": [64, 128]}},
               "loss": {"name": "LossWrapper", "losses": {"CrossEntropyLoss": 1, "DenseContrastiveLossV2_ms": 0.1},
                        "temperature": 0.1, "scales": 2, "weights": [1.0, 0.5], "cross_scale_contrast": True,
                        "min_views_per_class": 2, "max_features_total": 600},
               "train": {"learning_rate": 0.01, "lr_fct": "polynomial", "optim": "SGD", "lr_batchwise": True, "epochs": 1}}
        mgr = HRNetManager(cfg, autostart=False)
        mgr.setup()
        mgr.model.train()
        gen = torch.Generator().manual_seed(0)
        img = torch.randn(2, 3, 64, 128, generator=gen).to(dev)
        lbl = torch.randint(0, 20, (2, 64, 128), generator=gen).to(dev)
        torch.manual_seed(5)
        ret = mgr.forward_step(img, lbl)
        ret["loss"].backward()
        cm = t_get_confusion_matrix(ret["output"], lbl, "CITYSCAPES")
        res[lazy] = (ret["loss"].item(), {k: float(v) for k, v in mgr.loss.loss_vals.items()}, cm.cpu(),
                     {k: p.grad.clone() for k, p in mgr.model.named_parameters() if p.grad is not None},
                     type(ret["output"]).__name__)
    a, b = res[False], res[True]
    assert a[4] == "Tensor" and b[4] == "UpsampledLogits"
    assert abs(a[0] - b[0]) <= 1e-5 * abs(a[0])
    for k in a[1]:
        assert abs(a[1][k] - b[1][k]) <= 1e-5 * max(abs(a[1][k]), 1e-6), k
    assert (a[2] - b[2]).abs().sum().item() <= 2            # ties within round-off may flip a pixel's arg-max
    assert a[3].keys() == b[3].keys()
    gmax = max(g.abs().max().item() for g in a[3].values())
    for k in a[3]:
        assert (a[3][k] - b[3][k]).abs().max().item() <= 2e-3 * max(a[3][k].abs().max().item(), 1e-5 * gmax), k


def test_lazy_logits_upernet_two_scale_step_equals_materialised_step():
    """OCRNetManager step of UPerNet + Swin-T with TwoScaleLoss (auxiliary + final cross-entropy) and graph.lazy_logits:
    both heads' logits stay at low resolution and go through the fused up-sampling + CE kernel; loss, loss_vals and
    parameter gradients equal the step with materialised logits (eval-mode dropout so that both runs see one network)."""
    import mscs_amd  # noqa: F401
    from mscs_amd.managers import OCRNetManager
    from mscs_amd.utils import set_verbosity
    set_verbosity(40)
    dev = torch.device("cuda:0")
    res = {}
    for lazy in (False, True):
        cfg = {"name": "t4", "mode": "training", "manager": "OCRNet", "cuda": True, "parallel": False, "seed": 3,
               "graph": {"model": "UPerNet", "backbone": "swinT", "sync_bn": False, "out_stride": 32, "pretrained": False,
                         "align_corners": False, "aux_head": {"in_index": 3, "dropout_rate": 0.0}, "dropout_rate": 0.0,
                         "drop_path_rate": 0.0, "fpn_channels": 64, "lazy_logits": lazy,
                         "ms_projector": {"mlp": [[1, -1, 1]], "scales": 4, "d": 64, "use_bn": True, "position": "fpn"}},
               "data": {"dataset": "ADE20K", "experiment": 1, "batch_size": 2, "synthetic": True,
                        "synthetic_length": 4, "transform_values": {"crop_shape": [128, 128]}},
               "loss": {"name": "LossWrapper", "temperature": 0.1, "scales": 4, "weights": [1.0, 0.7, 0.4, 0.1],
                        "cross_scale_contrast": True, "min_views_per_class": 2, "max_features_total": 600,
                        "interm": {"name": "CrossEntropyLoss", "args": [], "weight": 0.4},
                        "final": {"name": "CrossEntropyLoss", "args": [], "weight": 1.0},
                        "losses": {"TwoScaleLoss": 1.0, "DenseContrastiveLossV2_ms": 0.1}},
               "train": {"learning_rate": 0.01, "lr_fct": "polynomial", "optim": "SGD", "lr_batchwise": True, "epochs": 1}}
        mgr = OCRNetManager(cfg, autostart=False)
        mgr.setup()
        mgr.model.train()
        gen = torch.Generator().manual_seed(0)
        img = torch.randn(2, 3, 128, 128, generator=gen).to(dev)
        lbl = torch.randint(0, 151, (2, 128, 128), generator=gen).to(dev)
        torch.manual_seed(5)
        ret = mgr.forward_step(img, lbl)
        ret["loss"].backward()
        res[lazy] = (ret["loss"].item(), {k: float(v) for k, v in mgr.loss.loss_vals.items()},
                     {k: p.grad.clone() for k, p in mgr.model.named_parameters() if p.grad is not None},
                     type(ret["output"]).__name__, type(ret["interm_output"]).__name__)
    a, b = res[False], res[True]
    assert a[3] == a[4] == "Tensor" and b[3] == b[4] == "UpsampledLogits"
    assert abs(a[0] - b[0]) <= 1e-5 * abs(a[0])
    for k in a[1]:
        assert abs(a[1][k] - b[1][k]) <= 1e-5 * max(abs(a[1][k]), 1e-6), k
    assert a[2].keys() == b[2].keys()
    gmax = max(g.abs().max().item() for g in a[2].values())
    for k in a[2]:
        assert (a[2][k] - b[2][k]).abs().max().item() <= 2e-3 * max(a[2][k].abs().max().item(), 1e-5 * gmax), k
