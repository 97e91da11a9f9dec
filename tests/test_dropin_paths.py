"""Round-4 parity additions (-m gpu, through the package's reference-shaped classes):

* BASELINE.json configs[0] end to end: HRNetManager, `hrnet18` backbone, SINGLE `projector`, bare DenseContrastiveLossV2
  next to CrossEntropyLoss, 2 x 256 x 256, 3 classes + ignore -- one training step, the contrastive term against
  ``oracle.dcv2_single`` on the model's own embeddings (reference: losses/LossWrapper.py:66-72, models/HRNet.py:587-590,
  losses/DenseContrastiveLossV2.py:44-62; BASELINE.md: the "CPU" wording of configs[0] is superseded -- the loss has no
  CPU path by design, the plumbing of that config runs here on the GPU).
* the projection heads' pixel-major (channels-last strided) embedding maps against the plain convolution
  (models/Projector.py:56-63), forward, gradients, and through the loss.
* the `sampled_features` bank of a bare DenseContrastiveLossV2 with cross_scale_contrast (:58-61): backward through
  ``dcl_scatter_raw`` against torch indexing; a LazyProjection input gives the materialised map's results.
* every direct-convolution shape of the benchmark step AT ITS SIZE against float64 (forward, data gradient, weight
  gradient), bar 5e-6 of max: a missing wait state between an inline-asm split and an MFMA shows as a 1e-4 error in one of
  ~1e5 results, invisible in small tests (was tools/probes/at_size_accuracy.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import mscs_amd  # noqa: F401
    from mscs_amd import _lib
    _lib.lib()
    from mscs_amd.utils import set_verbosity
    set_verbosity(40)
    return torch.device("cuda:0")


def _config0(lazy=None):
    graph = {"model": "HRNet", "backbone": "hrnet18", "sync_bn": False, "pretrained": False, "align_corners": True,
             "projector": {"mlp": [[1, -1, 1]], "d": 64, "use_bn": True}}
    if lazy is not None:
        graph["lazy_projector"] = lazy
        graph["lazy_logits"] = lazy
    return {"name": "cfg0", "mode": "training", "manager": "HRNet", "cuda": True, "parallel": False, "seed": 0,
            "graph": graph,
            "data": {"dataset": "CITYSCAPES", "experiment": 1, "batch_size": 2, "synthetic": True, "synthetic_length": 4,
                     "transform_values": {"crop_shape": [256, 256]}},
            "loss": {"name": "LossWrapper", "losses": {"CrossEntropyLoss": 1, "DenseContrastiveLossV2": 0.1},
                     "temperature": 0.1, "min_views_per_class": 5, "max_views_per_class": 2500,
                     "max_features_total": 10000, "label_scaling_mode": "nn"},
            "train": {"learning_rate": 0.01, "lr_fct": "polynomial", "optim": "SGD", "lr_batchwise": True, "epochs": 1,
                      "momentum": 0.9, "weight_decay": 0.0005}}


@pytest.mark.parametrize("lazy", [None, False])
def test_baseline_config0_hrnet18_single_projector_dcv2_manager_step(dev, lazy):
    """One HRNetManager training step of BASELINE configs[0]; `lazy=None`: the config as a reference user writes it (the
    manager picks the fused consumers), False: the reference's tensors (full logits, embedding map)."""
    from mscs_amd.managers import HRNetManager
    from oracle import dcl_oracle as orc
    mgr = HRNetManager(_config0(lazy), autostart=False)
    mgr.setup()
    mgr.model.train()
    dc = mgr.loss.loss_classes["DenseContrastiveLossV2"]
    dc.num_all_classes, dc.ignore_class = 4, 3          # 3 real classes + ignore, as DenseContrastiveLossV2.py:238-239 does
    gen = torch.Generator().manual_seed(0)
    img = torch.randn(2, 3, 256, 256, generator=gen).to(dev)
    lbl_cpu = torch.randint(0, 4, (2, 256, 256), generator=gen)
    lbl = lbl_cpu.to(dev)
    mgr.optimiser.zero_grad(set_to_none=True)
    torch.manual_seed(11)
    ret = mgr.forward_step(img, lbl)
    assert sorted(mgr.loss.loss_vals) == ["CrossEntropyLoss", "DenseContrastiveLossV2"]
    feats = ret["feats"]
    emb = feats.materialize() if hasattr(feats, "materialize") else feats
    assert tuple(emb.shape) == (2, 64, 64, 64)
    if lazy is None:
        assert type(ret["output"]).__name__ == "UpsampledLogits" and type(feats).__name__ == "LazyProjection"
    else:
        assert isinstance(ret["output"], torch.Tensor) and tuple(ret["output"].shape) == (2, 19, 256, 256)
        assert isinstance(feats, torch.Tensor)      # (270 input channels: not a GEMM shape, the plain convolution's NCHW map;
        # the pixel-major form of the W48 heads is checked in test_projector_pixel_major_maps_equal_the_plain_convolution)
    st = dc.last_state
    sc = st.scales[0]
    # iid labels at 2 x 256 x 256, K = 4 -> T = 6 pairs, N pinned below the 10 000 cap
    assert sc.plan.T == 6 and sc.plan.N <= 10000
    ocfg = orc.LossConfig(num_all_classes=4, temperature=0.1)
    loss_ref, plan_ref, _ = orc.dcv2_single(lbl_cpu.numpy(), emb.detach().float().cpu().numpy(), ocfg,
                                            rng=orc.MT19937(11), want_grad=False)
    assert np.array_equal(sc.pix.cpu().numpy(), plan_ref.pix), "sampled pixels differ from the reference order"
    got = float(mgr.loss.loss_vals["DenseContrastiveLossV2"]) / 0.1
    # lazy: the loss saw the embeddings of LazyProjection.rows (the 1x1 convolution on the sampled pixels, fp32 addmm), the
    # oracle those of materialize() (the full-map convolution kernel): two fp32 evaluations of the same map, 1e-6 apart
    assert abs(got - loss_ref) <= (5e-5 if lazy is None else 1e-5) * abs(loss_ref), (got, loss_ref)
    ret["loss"].backward()
    mgr.optimiser.step()
    mgr.scheduler.step()
    grads = [p.grad for p in mgr.model.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(g).all() for g in grads)
    assert all(torch.isfinite(p).all() for p in mgr.model.parameters())
    # the contrastive term reaches the single projector and, through the concatenated map, the backbone
    pg = [p.grad for n, p in mgr.model.named_parameters() if n.startswith("projector_model") and p.grad is not None]
    assert pg and all(g.abs().max().item() > 0 for g in pg)


def test_config0_loss_gradient_matches_oracle_on_the_models_embeddings(dev):
    """The bare DenseContrastiveLossV2 on a config-0-shaped embedding map (2 x 64 x 64 x 64, 3 classes + ignore): loss and
    the gradient with respect to the map against the oracle in float64 (1e-5 / 1e-4 of max), NCHW and channels-last maps."""
    from mscs_amd.losses import DenseContrastiveLossV2
    from oracle import dcl_oracle as orc
    gen = torch.Generator().manual_seed(3)
    lbl = torch.randint(0, 4, (2, 256, 256), generator=gen)
    emb = torch.randn(2, 64, 64, 64, generator=gen)
    mod = DenseContrastiveLossV2({"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1})
    mod.num_all_classes, mod.ignore_class = 4, 3
    ref_loss, plan, ref_grad = orc.dcv2_single(lbl.numpy(), emb.numpy(), orc.LossConfig(num_all_classes=4, temperature=0.1),
                                               rng=orc.MT19937(5))
    for fmt in (torch.contiguous_format, torch.channels_last):
        f = emb.to(dev).contiguous(memory_format=fmt).requires_grad_(True)
        torch.manual_seed(5)
        loss = mod(lbl.to(dev), f)
        loss.backward()
        assert np.array_equal(mod.last_state.scales[0].pix.cpu().numpy(), plan.pix)
        assert abs(loss.item() - ref_loss) <= 1e-5 * abs(ref_loss)
        g = f.grad.cpu().numpy()
        assert np.abs(g - ref_grad).max() <= 1e-4 * np.abs(ref_grad).max()


def test_projector_pixel_major_maps_equal_the_plain_convolution(dev):
    """Projector head (conv1x1 -> ReLU -> BN -> conv1x1 + bias) in training mode: the pixel-major output of
    models/ops._Conv1x1ToNHWC against the module's plain path -- same shape, values to 2e-6 of max, channels-last strides;
    input / weight / bias gradients under a random upstream gradient to 1e-5 of max; and through DenseContrastiveLossV2_ms."""
    from mscs_amd.models.Projector import Projector
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    from mscs_amd.models.ops import use_direct_conv1x1
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    gen = torch.Generator().manual_seed(1)
    xs = [torch.randn(2, c, 64 // (1 << i), 128 // (1 << i), generator=gen).to(dev) for i, c in enumerate((48, 96))]
    label = torch.randint(0, 20, (2, 256, 512), generator=gen).to(dev)
    res = {}
    for nhwc in (False, True):
        torch.manual_seed(0)
        proj = Projector({"mlp": [[1, -1, 1]], "d": 256, "c_in": [48, 96], "use_bn": True}).to(dev).train()
        for m in proj.modules():
            if type(m) is torch.nn.BatchNorm2d:
                m.__class__ = FusedBatchNorm2d
        use_direct_conv1x1(proj)
        proj.nhwc = nhwc
        ins = [x.clone().requires_grad_(True) for x in xs]
        outs = proj(ins)
        up = [torch.randn(o.shape, generator=torch.Generator().manual_seed(7 + i)).to(dev) for i, o in enumerate(outs)]
        if nhwc:
            assert all(o.stride(1) == 1 and tuple(o.shape) == (2, 256, x.shape[2], x.shape[3]) for o, x in zip(outs, xs))
        torch.autograd.backward(outs, up, retain_graph=True)
        g1 = [i.grad.clone() for i in ins] + [p.grad.clone() for p in proj.parameters()]
        for i in ins:
            i.grad = None
        proj.zero_grad()
        mod = DenseContrastiveLossV2_ms({"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 2,
                                         "weights": [1.0, 0.5], "cross_scale_contrast": True})
        torch.manual_seed(4)
        loss = mod(label, outs)
        loss.backward()
        g2 = [i.grad.clone() for i in ins] + [p.grad.clone() for p in proj.parameters()]
        res[nhwc] = ([o.detach().clone() for o in outs], g1, loss.item(), g2)
    a, b = res[False], res[True]
    for oa, ob in zip(a[0], b[0]):
        assert (oa - ob).abs().max().item() <= 2e-6 * oa.abs().max().item()
    for ga, gb in zip(a[1], b[1]):
        assert (ga - gb).abs().max().item() <= 1e-5 * max(ga.abs().max().item(), 1e-20)
    assert abs(a[2] - b[2]) <= 1e-6 * abs(a[2])
    for ga, gb in zip(a[3], b[3]):
        assert (ga - gb).abs().max().item() <= 2e-4 * max(ga.abs().max().item(), 1e-20)


def test_bare_dcv2_cross_scale_bank_and_its_backward(dev):
    """DenseContrastiveLossV2 with cross_scale_contrast returns (loss, sampled_features [T, C, V], sampled_labels, flag)
    (reference :58-61): the bank equals torch indexing of the map at the sampled pixels, its backward (dcl_scatter_raw) equals
    autograd's index backward, for NCHW and channels-last maps; a LazyProjection input gives the materialised map's results."""
    from mscs_amd.losses import DenseContrastiveLossV2
    from mscs_amd.models.Projector import LazyProjection
    gen = torch.Generator().manual_seed(9)
    n, C, h, w = 2, 32, 32, 64
    label = torch.randint(0, 20, (n, 4 * h, 4 * w), generator=gen).to(dev)
    base = torch.randn(n, C, h, w, generator=gen).to(dev)
    mod = DenseContrastiveLossV2({"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "cross_scale_contrast": True,
                                  "max_features_total": 900})
    outs = {}
    for fmt in (torch.contiguous_format, torch.channels_last):
        f = base.contiguous(memory_format=fmt).clone().requires_grad_(True)
        torch.manual_seed(2)
        loss, bank, labels, flag = mod(label, f)
        sc = mod.last_state.scales[0]
        T, V = sc.plan.T, sc.plan.V
        assert tuple(bank.shape) == (T, C, V) and tuple(labels.shape) == (T,) and flag is False
        pix, b = sc.pix.long(), sc.pair_b.long()
        fr = base.clone().requires_grad_(True)
        want = fr.reshape(n, C, -1)[b[:, None].expand(T, V), :, pix].permute(0, 2, 1)
        assert torch.equal(bank, want)
        up = torch.randn(bank.shape, generator=torch.Generator().manual_seed(3)).to(dev)
        (bank * up).sum().backward()
        (want * up).sum().backward()
        assert torch.equal(f.grad, fr.grad)
        outs[fmt] = loss.item()
    conv = torch.nn.Conv2d(16, C, 1).to(dev)
    hidden = torch.randn(n, 16, h, w, generator=gen).to(dev)
    res = []
    for lazy in (False, True):
        hd = hidden.clone().requires_grad_(True)
        conv.zero_grad()
        feat = LazyProjection(hd, conv) if lazy else conv(hd)
        torch.manual_seed(2)
        loss, bank, labels, _ = mod(label, feat)
        (loss + bank.square().mean()).backward()
        res.append((loss.item(), bank.detach().clone(), hd.grad.clone(), conv.weight.grad.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0])
    assert (res[0][1] - res[1][1]).abs().max().item() <= 1e-6 * res[0][1].abs().max().item()
    for k in (2, 3):
        assert (res[0][k] - res[1][k]).abs().max().item() <= 1e-5 * res[0][k].abs().max().item()


# (n, Cin, Cout, H, W, stride): every direct 3x3 shape of the HRNet-W48 benchmark step, at its size
AT_SIZE_SHAPES = [(12, 48, 48, 128, 256, 1), (12, 96, 96, 64, 128, 1), (12, 192, 192, 32, 64, 1), (12, 384, 384, 16, 32, 1),
                  (12, 144, 720, 128, 256, 1), (12, 64, 64, 128, 256, 1), (12, 48, 96, 128, 256, 2), (12, 64, 64, 256, 512, 2),
                  (12, 96, 192, 64, 128, 2), (12, 192, 384, 32, 64, 2)]


@pytest.mark.parametrize("shape", AT_SIZE_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_direct_convolutions_at_benchmark_size_against_fp64(dev, shape):
    """Forward, data gradient and weight gradient of the direct split-f16 kernels on the benchmark's own tensor sizes
    against float64 (image by image, to bound memory): 5e-6 of max each."""
    from mscs_amd.models import ops
    n, ci, co, h, w, st = shape
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, ci, h, w, device=dev, generator=gen).relu_()
    wt = torch.randn(co, ci, 3, 3, device=dev, generator=gen) * 0.1
    ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
    gy = torch.randn(n, co, ho, wo, device=dev, generator=gen) * 1e-3
    conv = torch.nn.Conv2d(ci, co, 3, st, 1, bias=False).to(dev)
    conv.weight.data.copy_(wt)
    ops.use_direct_conv3x3(conv)
    xi = x.clone().requires_grad_(True)
    y = conv(xi)
    y.backward(gy)
    gw64 = torch.zeros(co, ci, 3, 3, dtype=torch.float64, device=dev)
    ey = egx = ymax = gxmax = 0.0
    w64 = wt.double()
    for b in range(n):
        x64 = x[b:b + 1].double()
        g64 = gy[b:b + 1].double()
        y64 = F.conv2d(x64, w64, None, st, 1)
        gx64 = torch.nn.grad.conv2d_input(x64.shape, w64, g64, st, 1)
        gw64 += torch.nn.grad.conv2d_weight(x64, w64.shape, g64, st, 1)
        ey = max(ey, (y[b:b + 1].double() - y64).abs().max().item())
        ymax = max(ymax, y64.abs().max().item())
        egx = max(egx, (xi.grad[b:b + 1].double() - gx64).abs().max().item())
        gxmax = max(gxmax, gx64.abs().max().item())
    ew = (conv.weight.grad.double() - gw64).abs().max().item() / gw64.abs().max().item()
    assert ey <= 5e-6 * ymax, ("forward", ey / ymax)
    assert egx <= 5e-6 * gxmax, ("data gradient", egx / gxmax)
    assert ew <= 5e-6, ("weight gradient", ew)


def test_head_split_falls_back_when_the_tap_gather_does_not_fit(dev, monkeypatch):
    """conv3x3_over_upsampled asks dcl_tapup_supported before it commits to the split form; when the answer is no, the
    materialised concatenation is convolved -- same result (here forced on a small case)."""
    from mscs_amd import _lib
    from mscs_amd.models import ops
    gen = torch.Generator().manual_seed(0)
    ts = [torch.randn(2, c, 64 // s, 128 // s, generator=gen).to(dev).requires_grad_(True) for c, s in ((16, 1), (16, 2), (32, 4), (32, 8))]
    wt = (torch.randn(48, 96, 3, 3, generator=gen) * 0.05).to(dev).requires_grad_(True)
    bias = torch.randn(48, generator=gen).to(dev).requires_grad_(True)
    up = torch.randn(2, 48, 64, 128, generator=gen).to(dev)
    res = []
    for force in (False, True):
        L = _lib.lib()
        if force:
            monkeypatch.setattr(L, "dcl_tapup_supported", lambda *a: 0, raising=False)
        for t in ts + [wt, bias]:
            t.grad = None
        y = ops.conv3x3_over_upsampled(ts, True, wt, bias)
        y.backward(up)
        res.append([y.detach().clone()] + [t.grad.clone() for t in ts + [wt, bias]])
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item()


# (n, Cin, Cout, H, W): shapes whose automatic tile is one of the BasicBlock tiles (the epilogue-statistics kernels): the four
# branch shapes at a reduced batch, ragged rows / columns (H, W not multiples of the tile) and a ragged last channel tile
@pytest.mark.parametrize("shape", [(2, 3, 64, 64, 128), (1, 3, 64, 33, 47), (2, 1, 16, 17, 40), (1, 4, 70, 20, 64), (12, 3, 64, 512, 1024)],
                         ids=lambda s: "x".join(map(str, s)))
def test_stem_small_cin_stride2_convolution_against_fp64(dev, shape):
    """dcl_conv3x3_s2_smallcin (the stem's conv1 on the image, reference models/HRNet.py:404-405: 3 -> 64, stride 2): plain fp32
    FMAs against float64, ragged sizes, 1 .. 4 input channels, more than 64 output channels, with and without bias: 2e-6 of
    max; and that DirectConv2d takes this kernel for such a layer (forward), with the library's weight gradient behind it."""
    from mscs_amd import _lib
    from mscs_amd.models import ops
    n, ci, co, h, w = shape
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, ci, h, w, device=dev, generator=gen)
    wt = torch.randn(co, ci, 3, 3, device=dev, generator=gen) * 0.2
    bias = torch.randn(co, device=dev, generator=gen)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    for b in (None, bias):
        y = torch.full((n, co, ho, wo), float("nan"), device=dev)
        _lib.check(_lib.lib().dcl_conv3x3_s2_smallcin(_lib.ptr(x), n, ci, h, w, _lib.ptr(wt), co, _lib.ptr(b), _lib.ptr(y),
                                                      _lib.stream_ptr(dev)), "smallcin")
        worst, top = 0.0, 0.0
        for i in range(n):
            ref = F.conv2d(x[i:i + 1].double(), wt.double(), None if b is None else b.double(), 2, 1)
            worst = max(worst, (y[i:i + 1].double() - ref).abs().max().item())
            top = max(top, ref.abs().max().item())
        assert worst <= 2e-6 * top, (worst, top)
    if n <= 2:
        conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=True).to(dev)
        ops.use_direct_conv3x3(conv)
        conv.weight.data.copy_(wt)
        conv.bias.data.copy_(bias)
        from mscs_amd.utils.kernel_timer import KernelTimer
        with KernelTimer(["dcl_conv3x3_s2_smallcin", "dcl_conv3x3_f16x3"]) as kt:
            out = conv(x)
            out.square().mean().backward()
            torch.cuda.synchronize()
        assert [c[0] for c in kt.calls] == ["dcl_conv3x3_s2_smallcin"]
        w64 = wt.double().requires_grad_(True)
        F.conv2d(x.double(), w64, bias.double(), 2, 1).square().mean().backward()
        assert (conv.weight.grad.double() - w64.grad).abs().max().item() <= 2e-5 * w64.grad.abs().max().item()
