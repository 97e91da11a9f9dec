"""GPU parity tests: the HIP path (through the C ABI of libdcl_hip.so) against
  * the golden vectors produced by running the reference (tests/golden),
  * the oracle on seeded inputs,
  * size-independent properties at BASELINE config-2 size.
Tolerances (fp32 path): loss rtol 1e-5; gradients atol 1e-4 * max|grad| (summation order differs);
sampled pixel indices bit-exact."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, num_classes_for

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5
GRAD_ATOL_REL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import mscs_amd  # noqa: F401
    from mscs_amd import _lib
    _lib.lib()                                   # fails loudly if libdcl_hip.so is missing
    return torch.device("cuda:0")


def _module_cfg(c):
    cfg = {k: v for k, v in c.items() if not k.startswith("_override_")}
    return cfg


def _apply_overrides(mod, c):
    for k in ("num_all_classes", "ignore_class"):
        if "_override_" + k in c:
            setattr(mod, k, c["_override_" + k])
            for s in range(getattr(mod, "scales", 0)):
                setattr(getattr(mod, f"DCV2_scale{s}"), k, c["_override_" + k])


def _check_plans(g, st):
    for s, sc in enumerate(st.scales):
        assert sc.plan.V == int(g[f"s{s}_V"])
        np.testing.assert_array_equal(sc.plan.pair_b, g[f"s{s}_pair_b"])
        np.testing.assert_array_equal(sc.plan.pair_k, g[f"s{s}_pair_k"])
        np.testing.assert_array_equal(sc.pix.cpu().numpy(), g[f"s{s}_pix"])      # bit-exact


def _check_grad(got, ref):
    scale = max(float(np.abs(ref).max()), 1e-30)
    np.testing.assert_allclose(got, ref, atol=GRAD_ATOL_REL * scale, rtol=1e-3)


@pytest.mark.parametrize("name", golden_names(["G1_", "G1b_", "G3"]))
def test_single_scale_vs_reference(dev, name):
    from mscs_amd.losses import DenseContrastiveLossV2
    g = load_golden(name)
    c = g["config"]
    mod = DenseContrastiveLossV2(_module_cfg(c))
    _apply_overrides(mod, c)
    feat = torch.from_numpy(g["feat0"]).to(dev).requires_grad_(True)
    label = torch.from_numpy(g["label"].astype(np.int64)).to(dev)
    torch.manual_seed(int(g["seed"]))
    out = mod(label, feat)
    loss = out[0] if isinstance(out, tuple) else out
    loss.backward()
    _check_plans(g, mod.last_state)
    assert mod.log_this_step == bool(g["s0_log_this_step"])
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=LOSS_RTOL)
    _check_grad(feat.grad.cpu().numpy(), g["s0_grad"])
    if isinstance(out, tuple):
        np.testing.assert_array_equal(out[1].detach().cpu().numpy(), g["sampled_features"])
        np.testing.assert_array_equal(out[2].cpu().numpy(), g["sampled_labels"])
        assert out[3] is False


@pytest.mark.parametrize("mfma", ["f16x3", "f32"])
@pytest.mark.parametrize("name", golden_names(["G2", "G4", "G5", "G9"]))
def test_multi_scale_vs_reference(dev, name, mfma, monkeypatch):
    """Both similarity-product arithmetics (split-f16 MFMA and exact-f32 MFMA) against the reference."""
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    monkeypatch.delenv("DCL_MFMA", raising=False)
    g = load_golden(name)
    c = g["config"]
    mod = DenseContrastiveLossV2_ms(dict(_module_cfg(c), mfma_mode=mfma))
    assert mod.DCV2_scale0.mfma_mode == mfma
    _apply_overrides(mod, c)
    S = mod.scales
    feats = [torch.from_numpy(g[f"feat{s}"]).to(dev).requires_grad_(True) for s in range(S)]
    label = torch.from_numpy(g["label"].astype(np.int64)).to(dev)
    torch.manual_seed(int(g["seed"]))
    loss = mod(label, feats)
    loss.backward()
    _check_plans(g, mod.last_state)
    np.testing.assert_allclose([x.item() for x in mod.ms_losses], g["ms_losses"], rtol=LOSS_RTOL)
    assert len(mod.cs_losses) == len(g["cs_losses"])
    np.testing.assert_allclose([x.item() for x in mod.cs_losses], g["cs_losses"], rtol=LOSS_RTOL)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=LOSS_RTOL)
    for s in range(S):
        got = feats[s].grad
        got = np.zeros_like(g[f"s{s}_grad"]) if got is None else got.cpu().numpy()
        _check_grad(got, g[f"s{s}_grad"])


def test_losswrapper_vs_reference(dev):
    from mscs_amd.losses import LossWrapper
    g = load_golden("G6_losswrapper")
    c = dict(g["config"], device=str(dev))
    lw = LossWrapper(c)
    logits = torch.from_numpy(g["logits_f16"].astype(np.float32)).to(dev).requires_grad_(True)
    feats = [torch.from_numpy(g[f"feat{s}"]).to(dev).requires_grad_(True) for s in range(2)]
    label = torch.from_numpy(g["label"].astype(np.int64)).to(dev)
    torch.manual_seed(int(g["seed"]))
    total = lw(logits, label, deep_features=feats)
    total.backward()
    np.testing.assert_allclose(total.item(), g["total"], rtol=LOSS_RTOL)
    keys = sorted(k[5:] for k in g if k.startswith("val__"))
    assert sorted(lw.loss_vals.keys()) == keys
    for k in keys:
        np.testing.assert_allclose(float(lw.loss_vals[k]), g["val__" + k], rtol=LOSS_RTOL)
    _check_grad(logits.grad[:, :, ::4, ::4].cpu().numpy(), g["logits_grad_sample"])
    for s in range(2):
        _check_grad(feats[s].grad.cpu().numpy(), g[f"s{s}_grad"])
    # TwoScaleLoss through the wrapper
    c2 = dict(json.loads(str(g["twoscale_config_json"])), device=str(dev))
    lw2 = LossWrapper(c2)
    interm = torch.from_numpy(g["interm_f16"].astype(np.float32)).to(dev)
    t2 = lw2(logits.detach(), label, interm_prediction=interm)
    np.testing.assert_allclose(t2.item(), g["twoscale_total"], rtol=LOSS_RTOL)


def _random_case(seed, n, H, W, K, C, strides, classes=None):
    gen = torch.Generator().manual_seed(seed)
    if classes is None:
        label = torch.randint(0, K, (n, H, W), generator=gen)
    else:
        idx = torch.randint(0, len(classes), (n, H // 8, W // 8), generator=gen)
        label = torch.tensor(classes)[idx].repeat_interleave(8, 1).repeat_interleave(8, 2)
    feats = [torch.randn(n, C, H // s, W // s, generator=gen) for s in strides]
    return label, feats


@pytest.mark.parametrize("channels_last", [False, True])
def test_vs_oracle_c256_three_scales(dev, oracle, channels_last):
    """C = 256 (the production width), 3 scales + cross-scale, against the fp64 oracle."""
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    label, feats = _random_case(5, 2, 128, 256, 20, 256, (4, 8, 16), classes=[0, 3, 5, 7, 11, 19])
    cfg = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 3,
           "weights": [1.0, 0.7, 0.4], "cross_scale_contrast": True, "max_features_total": 3000}
    mod = DenseContrastiveLossV2_ms(cfg)
    ocfg = oracle.LossConfig(num_all_classes=20, temperature=0.1, max_features_total=3000, scales=3,
                             weights=[1.0, 0.7, 0.4], cross_scale_contrast=True)
    res = oracle.dcv2_ms(label.numpy(), [f.numpy() for f in feats], ocfg, rng=oracle.MT19937(77))
    dfeats = []
    for f in feats:
        f = f.to(dev)
        if channels_last:
            f = f.contiguous(memory_format=torch.channels_last)
        dfeats.append(f.requires_grad_(True))
    torch.manual_seed(77)
    loss = mod(label.to(dev), dfeats)
    loss.backward()
    for s, sc in enumerate(mod.last_state.scales):
        np.testing.assert_array_equal(sc.pix.cpu().numpy(), res.plans[s].pix)
    np.testing.assert_allclose(loss.item(), res.loss, rtol=LOSS_RTOL)
    np.testing.assert_allclose([x.item() for x in mod.ms_losses], res.ms_losses, rtol=LOSS_RTOL)
    np.testing.assert_allclose([x.item() for x in mod.cs_losses], res.cs_losses, rtol=LOSS_RTOL)
    for s in range(3):
        _check_grad(dfeats[s].grad.cpu().numpy(), res.grads[s])
        if channels_last:
            assert dfeats[s].grad.is_contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("prepared", [False, True])
def test_lazy_projection_equals_materialised_projection(dev, prepared):
    """models.Projector with ``lazy = True`` hands the loss LazyProjection objects: the heads' last 1x1 convolution is
    evaluated on the sampled pixels only ([T * V, c] x [c, d] instead of the [n, d, h, w] map).  Same sampling plan (same RNG
    draws), same loss and the same gradients of the projector's parameters and of the backbone features as the
    materialised maps (3 scales + cross-scale, 256-d embedding), to fp32 round-off; with and without the plan made ahead by
    prepare(); evaluation mode returns the maps."""
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    from mscs_amd.models.Projector import LazyProjection, Projector
    label, xs = _random_case(21, 2, 128, 256, 20, 0, (4, 8, 16), classes=[0, 3, 5, 7, 11, 19])
    chans = (48, 96, 192)
    gen = torch.Generator().manual_seed(5)
    xs = [torch.randn(2, c, 128 // s, 256 // s, generator=gen) for c, s in zip(chans, (4, 8, 16))]
    cfg = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 3,
           "weights": [1.0, 0.7, 0.4], "cross_scale_contrast": True, "max_features_total": 3000}
    torch.manual_seed(3)
    proj = Projector({"mlp": [[1, -1, 1]], "d": 256, "use_bn": True, "c_in": list(chans)}).to(dev).train()
    results = {}
    for lazy in (False, True):
        proj.lazy = lazy
        proj.zero_grad(set_to_none=True)
        mod = DenseContrastiveLossV2_ms(cfg)
        ins = [x.to(dev).requires_grad_(True) for x in xs]
        lbl = label.to(dev)
        for rep in range(2 if prepared else 1):          # prepare() needs the geometry of one earlier forward
            torch.manual_seed(77)
            if prepared and rep == 1:
                assert mod.prepare(lbl)
            feats = proj(ins)
            assert all(isinstance(f, LazyProjection) == lazy for f in feats)
            loss = mod(lbl, feats)
            if rep == (1 if prepared else 0):
                loss.backward()
        results[lazy] = (loss.item(), [i.grad.clone() for i in ins], [p.grad.clone() for p in proj.parameters()],
                         [sc.pix.clone() for sc in mod.last_state.scales])
    (l0, gi0, gp0, px0), (l1, gi1, gp1, px1) = results[False], results[True]
    for a, b in zip(px0, px1):
        assert torch.equal(a, b)
    assert abs(l0 - l1) <= 2e-6 * abs(l0)
    for a, b in zip(gi0 + gp0, gi1 + gp1):
        assert ((a - b).abs().max() / a.abs().max().clamp_min(1e-30)).item() < 2e-5
    proj.eval()
    with torch.no_grad():
        assert all(torch.is_tensor(f) for f in proj([x.to(dev) for x in xs]))


def test_matrix_adaptive_avg_pool_matches_aten(dev):
    """UPerNet's pyramid pooling as one matrix product (models/UPerNet._MatrixAdaptiveAvgPool2d): forward and input gradient
    against nn.AdaptiveAvgPool2d in float64 (1e-6 of max), bins that overlap (20 -> 6, 20 -> 3), divide evenly (20 -> 2) and
    are coarser than the input (5 -> 6); bitwise reproducible (ATen's backward uses float atomics)."""
    from mscs_amd.models.UPerNet import _MatrixAdaptiveAvgPool2d
    torch.manual_seed(4)
    for (h, w, s) in [(20, 20, 1), (20, 20, 2), (20, 20, 3), (20, 20, 6), (16, 16, 6), (7, 9, 3), (5, 5, 6)]:
        x = torch.randn(3, 24, h, w, device=dev, requires_grad=True)
        gy = torch.randn(3, 24, s, s, device=dev)
        y = _MatrixAdaptiveAvgPool2d(s)(x)
        y.backward(gy)
        x64 = x.detach().double().requires_grad_(True)
        y64 = torch.nn.functional.adaptive_avg_pool2d(x64, s)
        y64.backward(gy.double())
        assert ((y.double() - y64).abs().max() / y64.abs().max()).item() < 1e-6, (h, w, s)
        assert ((x.grad.double() - x64.grad).abs().max() / x64.grad.abs().max()).item() < 1e-6, (h, w, s)
        g1 = x.grad.clone()
        x.grad = None
        _MatrixAdaptiveAvgPool2d(s)(x).backward(gy)
        assert torch.equal(g1, x.grad)


def test_deterministic_bitwise(dev):
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    label, feats = _random_case(9, 2, 128, 256, 20, 64, (4, 8))
    cfg = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 2,
           "weights": [1.0, 0.5], "cross_scale_contrast": True}
    mod = DenseContrastiveLossV2_ms(cfg)
    outs = []
    for _ in range(2):
        fs = [f.to(dev).requires_grad_(True) for f in feats]
        torch.manual_seed(1)
        loss = mod(label.to(dev), fs)
        loss.backward()
        outs.append((loss.item(), [f.grad.clone() for f in fs]))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


def test_streamk_error_word_is_seen_before_the_gradients_are_applied(dev):
    """a timed-out stream-K hand-over must be detected in the SAME step, before optimizer.step().  The error word
    (flags[0], counted up by the kernel, never reset) is planted by hand; the backward pass copies it to the host right behind
    its sweeps, `streamk_check()` -- the managers' optimizer pre-step hook -- raises StreamKTimeout and switches the process to
    the column-split backward, whose gradients equal the stream-K ones (2e-6 of max); a clean pass raises nothing."""
    from mscs_amd import _lib
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    from mscs_amd.losses import engine
    L = _lib.lib()
    label, feats = _random_case(33, 3, 128, 256, 20, 64, (4, 8))
    cfg = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 2, "weights": [1.0, 0.5],
           "cross_scale_contrast": True}
    mod = DenseContrastiveLossV2_ms(cfg)
    if (mod.DCV2_scale0.mfma_mode or "f16x3") != "f16x3":
        pytest.skip("stream-K is the f16x3 backward")

    def run():
        fs = [f.to(dev).requires_grad_(True) for f in feats]
        torch.manual_seed(1)
        mod(label.to(dev), fs).backward()
        return [f.grad.clone() for f in fs]
    L.dcl_infonce_set_streamk(1)
    try:
        good = run()
        engine.streamk_check()                                  # clean pass: nothing to report
        assert engine._SK_WS, "the stream-K backward did not run (workspace missing)"
        for got in engine._SK_WS.values():
            got[1][0:1].fill_(3)                                # "three hand-overs timed out"
        run()
        with pytest.raises(engine.StreamKTimeout):
            engine.streamk_check()
        assert not engine._SK_WS                               # workspace dropped, stream-K off: the repeated step is column-split
        again = run()
        engine.streamk_check()
        assert not engine._SK_WS
        for a, b in zip(good, again):
            assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()
        # the hook the managers install: optimizer.step() raises BEFORE touching the parameters
        L.dcl_infonce_set_streamk(1)
        w = torch.nn.Parameter(torch.ones(4, device=dev))
        opt = torch.optim.SGD([w], lr=1.0)
        opt.register_step_pre_hook(lambda *_a, **_k: engine.streamk_check())
        run()
        for got in engine._SK_WS.values():
            got[1][0:1].fill_(1)
        run()
        w.grad = torch.ones_like(w)
        with pytest.raises(engine.StreamKTimeout):
            opt.step()
        assert torch.equal(w.detach(), torch.ones(4, device=dev))
    finally:
        L.dcl_infonce_set_streamk(1)
        engine._SK_WS.clear()


@pytest.mark.parametrize("n,H,W,cap", [(2, 128, 256, 10000), (3, 64, 128, 700), (12, 256, 512, 10000), (2, 128, 128, 90)])
def test_streamk_backward_equals_column_split_backward(dev, n, H, W, cap):
    """dcl_infonce_bwd_streamk (persistent workgroups over the (row block, chunk) sequence, finished tiles) against
    dcl_infonce_bwd (one slab per column split, summed here) on the banks of a real step: intra-scale (H = G + G^T),
    cross-scale dF1 (rows) and dF2 (columns, rectangular, other bank's statistics); ragged row blocks, fewer units than
    workgroups, ranges that span several row blocks.  Same products, different summation order: 2e-6 of max.  Bitwise
    reproducible on a flags buffer that is never reset (a flag is valid for the launch whose number it carries), also
    when it starts out with garbage from an aborted launch; no hand-over timed out (error word flags[0] stays 0)."""
    from mscs_amd import _lib
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    L = _lib.lib()
    label, feats = _random_case(21, n, H, W, 20, 64, (4, 16))
    cfg = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 2, "weights": [1.0, 0.5],
           "cross_scale_contrast": True, "max_features_total": cap, "mfma_mode": "f16x3"}
    mod = DenseContrastiveLossV2_ms(cfg)
    fs = [f.to(dev).requires_grad_(True) for f in feats]
    torch.manual_seed(2)
    mod(label.to(dev), fs).backward()
    st = mod.last_state
    p = _lib.ptr
    stream = _lib.stream_ptr(dev)
    for t in st.terms:
        A, B = st.scales[t.a], st.scales[t.b]
        N1, N2 = A.plan.N, B.plan.N
        N1pad, N2pad = A.bank.shape[0], B.bank.shape[0]
        stat = torch.empty((N1pad + 1, 4), device=dev)
        _lib.check(L.dcl_infonce_prep_stats(p(t.Z), p(t.W), p(t.rng_lo), p(t.rng_hi), None, N1, A.plan.V,
                                            1 if t.intra else 0, 1.0, 1.0 / t.tau, None, p(stat), stream), "prep")
        cases = [(A, B, N1, N2, N1pad, t.rng_lo, t.rng_hi, 1 if t.intra else 0, 1, 1 if t.intra else 0,
                  stat, stat if t.intra else None)]
        if not t.intra:
            cases.append((B, A, N2, N1, N2pad, t.rev_lo, t.rev_hi, 0, 0, 1, None, stat))
        for (X, Y, n1, n2, n1pad, lo, hi, intra, use_row, use_col, rstat, cstat) in cases:
            ns = int(L.dcl_suggest_nsplit(n1, n2))
            dpart = torch.empty((ns, n1pad, 256), device=dev)
            _lib.check(L.dcl_infonce_bwd(p(X.bank), n1, X.plan.V, p(Y.bank), n2, p(lo), p(hi), 1.0 / t.tau, intra,
                                         use_row, use_col, p(rstat), p(cstat), ns, p(dpart), p(X.bank_h), p(Y.bank_h),
                                         stream), "bwd")
            want = dpart.sum(0)
            G = int(L.dcl_infonce_bwd_streamk_workgroups(n1, n2))
            assert 0 < G <= 256
            nsl = int(L.dcl_infonce_bwd_streamk_slabs(n1, n2))       # column slices (4 at the benchmark size, 1 on short banks)
            assert nsl in (1, 4, 8)
            ws = torch.full((G, 128, 256), float("nan"), device=dev)
            flags = torch.zeros(G + 1, dtype=torch.int32, device=dev)
            outs = []
            for k in range(3):
                if k == 1:                 # what an aborted launch leaves behind: every "tile present" mark of the last launch
                    flags[1:] = flags[1:].max()
                    ws.fill_(float("nan"))
                dout = torch.full((nsl, n1pad, 256), float("nan"), device=dev)
                _lib.check(L.dcl_infonce_bwd_streamk(p(X.bank), n1, X.plan.V, p(Y.bank), n2, p(lo), p(hi), 1.0 / t.tau,
                                                     intra, use_row, use_col, p(rstat), p(cstat), p(dout), p(ws), p(flags),
                                                     p(X.bank_h), p(Y.bank_h), stream), "bwd_streamk")
                outs.append(dout.sum(0) if nsl > 1 else dout[0])
                assert torch.isfinite(dout[:, :n1]).all()
                assert int(flags[0].item()) == 0
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
            scale = want[:n1].abs().max().item()
            assert torch.isfinite(outs[0][:n1]).all()
            assert (outs[0][:n1] - want[:n1]).abs().max().item() <= 2e-6 * scale, (t.a, t.b, n1, n2, G)


@pytest.mark.parametrize("mfma", ["f16x3", "f32"])
def test_properties_at_baseline_config2_size(dev, mfma, monkeypatch):
    """BASELINE config 2 (n=12, 512x1024, K=20, C=256, 3 scales + cross-scale), size-independent checks:
    sampled pixels have the pair's class and are unique; the gradient lives only on sampled pixels
    and is orthogonal to the feature vector there (VJP of the L2 normalisation); scaling the features
    leaves the loss unchanged; the loss equals a dense fp32 torch evaluation of the same banks."""
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    gen = torch.Generator().manual_seed(0)
    n, H, W, K, C = 12, 512, 1024, 20, 256
    label = torch.randint(0, K, (n, H, W), generator=gen).to(dev)
    feats = [torch.randn(n, C, H // s, W // s, generator=gen).to(dev).requires_grad_(True)
             for s in (4, 8, 16)]
    monkeypatch.delenv("DCL_MFMA", raising=False)
    cfg = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 3,
           "weights": [1.0, 0.7, 0.4], "cross_scale_contrast": True, "mfma_mode": mfma}
    mod = DenseContrastiveLossV2_ms(cfg)
    torch.manual_seed(0)
    loss = mod(label, feats)
    loss.backward()
    st = mod.last_state
    assert [(sc.plan.T, sc.plan.V) for sc in st.scales] == [(228, 43)] * 3
    for s, sc in enumerate(st.scales):
        stride = 4 * 2 ** s
        lbl_s = label[:, ::stride, ::stride].reshape(n, -1)
        pix = sc.pix.long()
        b = sc.pair_b.long()[:, None].expand_as(pix)
        assert torch.equal(lbl_s[b, pix], sc.pair_k.long()[:, None].expand_as(pix))
        key = (b * lbl_s.shape[1] + pix).flatten()
        assert key.unique().numel() == key.numel()
        g = feats[s].grad.reshape(n, C, -1)
        nz = (g.abs().sum(1) > 0)
        touched = torch.zeros_like(nz)
        touched[b.flatten(), pix.flatten()] = True
        assert torch.equal(nz & ~touched, torch.zeros_like(nz))
        x = feats[s].detach().reshape(n, C, -1)[b.flatten(), :, pix.flatten()]
        gx = g[b.flatten(), :, pix.flatten()]
        cos = (x * gx).sum(1).abs() / (x.norm(dim=1) * gx.norm(dim=1) + 1e-30)
        assert cos.max().item() < 1e-3
    # dense torch fp32 evaluation of the same banks (reference formulas on the device)
    total = 0.0
    for t, term in enumerate(st.terms):
        A, B = st.scales[term.a], st.scales[term.b]
        Fa, Fb = A.bank[:A.plan.N], B.bank[:B.plan.N]
        ca = torch.from_numpy(np.repeat(A.plan.pair_k[A.plan.slot_pair], A.plan.V)).to(dev)
        cb = torch.from_numpy(np.repeat(B.plan.pair_k[B.plan.slot_pair], B.plan.V)).to(dev)
        Smat = (Fa @ Fb.T) / term.tau
        pos = (ca[:, None] == cb[None, :]).float()
        neg = 1 - pos
        if term.intra:
            pos.fill_diagonal_(0)
        E = torch.exp(Smat)
        Z = (E * neg).sum(1, keepdim=True)
        logp = Smat - torch.log(E + Z)
        P = pos.sum(1)
        Pn = P if term.intra else torch.where(P > 0, P, torch.ones_like(P))
        ref = -((pos * logp).sum(1) / Pn).mean()
        np.testing.assert_allclose(st.loss_buf[t].item(), ref.item(), rtol=LOSS_RTOL)
        total += term.weight * ref.item()
    np.testing.assert_allclose(loss.item(), total, rtol=LOSS_RTOL)
    # scale invariance (features are L2-normalised before use)
    feats2 = [(f.detach() * 3.0) for f in feats]
    torch.manual_seed(0)
    loss2 = mod(label, feats2)
    np.testing.assert_allclose(loss2.item(), loss.item(), rtol=1e-5)


def test_c_abi_direct_infonce_cross(dev, oracle):
    """dcl_infonce_fwd / _prep_stats / _bwd called through ctypes on hand-made banks (ragged sizes,
    a class missing from the contrast bank) against the oracle's cross_loss."""
    from mscs_amd import _lib
    import ctypes
    L = _lib.lib()
    rs = np.random.RandomState(3)
    V1, V2 = 7, 5
    cls1 = np.array([0, 0, 2, 5, 5, 5, 9])          # slots, class-sorted
    cls2 = np.array([0, 2, 2, 9, 9, 11])            # class 5 absent -> rows with P = 0
    N1, N2 = len(cls1) * V1, len(cls2) * V2
    F1 = rs.randn(N1, 256).astype(np.float32); F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
    F2 = rs.randn(N2, 256).astype(np.float32); F2 /= np.linalg.norm(F2, axis=1, keepdims=True)
    r1, r2 = np.repeat(cls1, V1), np.repeat(cls2, V2)
    tau = 0.1
    ref_loss, d1, d2 = oracle.cross_loss(F1.astype(np.float64), r1, F2.astype(np.float64), r2, tau)

    def ranges(ca, cb, Vb):
        lo = np.array([np.flatnonzero(cb == c)[0] * Vb if (cb == c).any() else 0 for c in ca], np.int32)
        hi = np.array([(np.flatnonzero(cb == c)[-1] + 1) * Vb if (cb == c).any() else 0 for c in ca], np.int32)
        return lo, hi
    lo, hi = ranges(cls1, cls2, V2)
    rlo, rhi = ranges(cls2, cls1, V1)

    def bank(F):
        pad = (-F.shape[0]) % 128
        return torch.from_numpy(np.concatenate([F, np.zeros((pad, 256), np.float32)])).to(dev)
    A, B = bank(F1), bank(F2)
    t = lambda a: torch.from_numpy(a).to(dev)
    lo_d, hi_d, rlo_d, rhi_d = t(lo), t(hi), t(rlo), t(rhi)
    ns = 2
    N1pad, N2pad = A.shape[0], B.shape[0]
    zpart = torch.empty(ns * N1pad, device=dev); Z = torch.empty(N1pad, device=dev)
    rl = torch.empty(N1pad, device=dev); Wt = torch.empty(N1pad, device=dev); loss = torch.empty(1, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = _lib.ptr
    _lib.check(L.dcl_infonce_fwd(p(A), N1, V1, p(B), N2, p(lo_d), p(hi_d), 1 / tau, 0, ns, p(zpart), p(Z),
                                 p(rl), p(Wt), p(loss), st), "fwd")
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=LOSS_RTOL)
    stat = torch.empty(N1pad + 1, 4, device=dev)
    _lib.check(L.dcl_infonce_prep_stats(p(Z), p(Wt), p(lo_d), p(hi_d), None, N1, V1, 0, 1.0, 1 / tau, None,
                                        p(stat), st), "prep")
    dp1 = torch.empty(ns, N1pad, 256, device=dev)
    _lib.check(L.dcl_infonce_bwd(p(A), N1, V1, p(B), N2, p(lo_d), p(hi_d), 1 / tau, 0, 1, 0, p(stat), None,
                                 ns, p(dp1), None, None, st), "bwd1")
    dp2 = torch.empty(ns, N2pad, 256, device=dev)
    _lib.check(L.dcl_infonce_bwd(p(B), N2, V2, p(A), N1, p(rlo_d), p(rhi_d), 1 / tau, 0, 0, 1, None, p(stat),
                                 ns, p(dp2), None, None, st), "bwd2")
    _check_grad(dp1.sum(0)[:N1].cpu().numpy(), d1)
    _check_grad(dp2.sum(0)[:N2].cpu().numpy(), d2)


@pytest.mark.parametrize("f16x3", [False, True])
@pytest.mark.parametrize("intra", [0, 1])
def test_one_sweep_forward_equals_two_sweep_forward(dev, f16x3, intra):
    """dcl_infonce_zsweep_keep + dcl_infonce_pos_finish (one pass over the bank; the positives' similarities kept) against
    dcl_infonce_zsweep + dcl_infonce_possweep on ragged class-sorted banks (a class absent from the contrast bank, ranges that
    cross chunk and column-split boundaries, N not a multiple of anything): the same zpart and Z bit for bit, rowloss / W to
    fp32 summation order."""
    from mscs_amd import _lib
    import ctypes
    L = _lib.lib()
    rs = np.random.RandomState(5)
    V = 9
    cls1 = np.sort(rs.randint(0, 9, size=61))
    cls2 = cls1 if intra else np.sort(rs.choice([0, 1, 2, 3, 5, 6, 7, 8], size=47))      # class 4 absent from the contrast bank
    N1, N2 = len(cls1) * V, len(cls2) * V
    F1 = rs.randn(N1, 256).astype(np.float32); F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
    F2 = F1 if intra else rs.randn(N2, 256).astype(np.float32)
    F2 = F2 / np.linalg.norm(F2, axis=1, keepdims=True)
    lo = np.array([np.flatnonzero(cls2 == c)[0] * V if (cls2 == c).any() else 0 for c in cls1], np.int32)
    hi = np.array([(np.flatnonzero(cls2 == c)[-1] + 1) * V if (cls2 == c).any() else 0 for c in cls1], np.int32)

    def bank(F):
        pad = (-F.shape[0]) % 128
        return torch.from_numpy(np.concatenate([F, np.zeros((pad, 256), np.float32)])).to(dev)
    A, B = bank(F1), bank(F2)
    Ah = Bh = None
    if f16x3:
        def halves(b):
            x = b.double() * 1024.0
            h = x.to(torch.float16)
            l = (x - h.double()).to(torch.float16)
            return torch.cat([h, l], 1).contiguous()
        Ah, Bh = halves(A), halves(B)
    lo_d, hi_d = torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev)
    N1pad = A.shape[0]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = _lib.ptr
    tau = 0.1
    for ns in (1, 3):
        zp0 = torch.empty(ns * N1pad, device=dev); zp1 = torch.empty(ns * N1pad, device=dev)
        Z0, rl0, W0 = (torch.empty(N1pad, device=dev) for _ in range(3))
        Z1, rl1, W1 = (torch.empty(N1pad, device=dev) for _ in range(3))
        _lib.check(L.dcl_infonce_zsweep(p(A), N1, V, p(B), N2, p(lo_d), p(hi_d), 1 / tau, ns, p(zp0), p(Ah), p(Bh), st), "z")
        _lib.check(L.dcl_infonce_possweep(p(A), N1, V, p(B), N2, p(lo_d), p(hi_d), 1 / tau, intra, p(zp0), ns, 0, p(Z0), p(rl0),
                                          p(W0), p(Ah), p(Bh), st), "pos")
        ld = (int((hi - lo).max()) + 3) & ~3
        spos = torch.full((N1pad, ld), float("nan"), device=dev)
        _lib.check(L.dcl_infonce_zsweep_keep(p(A), N1, V, p(B), N2, p(lo_d), p(hi_d), 1 / tau, ns, p(zp1), p(Ah), p(Bh),
                                             p(spos), ld, st), "zkeep")
        _lib.check(L.dcl_infonce_pos_finish(p(spos), ld, N1, V, p(lo_d), p(hi_d), 1 / tau, intra, int(f16x3), p(zp1), ns,
                                            p(Z1), p(rl1), p(W1), st), "finish")
        assert torch.equal(zp0, zp1) and torch.equal(Z0, Z1)
        # every positive was written exactly where the finish kernel reads it, nothing else was touched
        span = torch.from_numpy(np.repeat(hi - lo, V)).to(dev)
        cols = torch.arange(ld, device=dev).view(1, -1)
        assert torch.equal(~torch.isnan(spos[:N1]), cols < span.view(-1, 1)) and torch.isnan(spos[N1:]).all()
        assert torch.allclose(rl0[:N1], rl1[:N1], rtol=2e-6, atol=2e-5) and torch.allclose(W0[:N1], W1[:N1], rtol=2e-6, atol=1e-9)
        assert (rl1[N1:] == 0).all() and (W1[N1:] == 0).all()


def test_keep_positives_switch_gives_the_same_loss_and_gradients(dev):
    """The loss module with the one-sweep forward (default) and with the two-sweep forward (debug.cfg.keep_positives = False):
    same sampling, loss and feature gradients to round-off, on a three-scale cross-scale configuration."""
    import mscs_amd  # noqa: F401
    from mscs_amd.debug import cfg as dbg
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    from mscs_amd.losses import engine
    conf = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "min_views_per_class": 5, "max_views_per_class": 60,
            "max_features_total": 3000, "scales": 3, "weights": [1.0, 0.7, 0.4], "cross_scale_contrast": True}
    g = torch.Generator().manual_seed(2)
    label = torch.randint(0, 19, (3, 96, 192), generator=g).to(dev)
    feats = [torch.randn(3, 64, 96 // s, 192 // s, generator=g).to(dev) for s in (1, 2, 4)]
    outs = {}
    keep = dbg.keep_positives
    try:
        for flag in (True, False):
            dbg.keep_positives = flag
            mod = DenseContrastiveLossV2_ms(dict(conf)).to(dev)
            fs = [f.clone().requires_grad_(True) for f in feats]
            torch.manual_seed(9)
            loss = mod(label, fs)
            loss.backward()
            used = [engine._keep_positives(t) for t in mod.last_state.terms] if hasattr(mod, "last_state") else None
            outs[flag] = (loss.detach(), [f.grad for f in fs], used)
    finally:
        dbg.keep_positives = keep
    assert outs[True][2] and all(outs[True][2]) and not any(outs[False][2])
    assert torch.allclose(outs[True][0], outs[False][0], rtol=2e-6)
    for a, b in zip(outs[True][1], outs[False][1]):
        assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item() + 1e-12


def test_k1_k2_direct_odd_sizes(dev, oracle):
    """K1/K2 through ctypes at sizes that are not multiples of the stride or the segment length."""
    from mscs_amd import _lib
    import ctypes
    L = _lib.lib()
    rs = np.random.RandomState(11)
    n, H, W, K, scale = 3, 101, 203, 7, 3
    label = rs.randint(0, K + 2, size=(n, H, W)).astype(np.int64)      # ids >= K are out of range
    label[0, :5, :5] = -1
    lbl_ref = oracle.downsample_labels(label, scale)
    h, w = lbl_ref.shape[1:]
    counts_ref = oracle.class_counts(lbl_ref, K)
    nseg = (h * w + 255) // 256
    lab_d = torch.from_numpy(label).to(dev)
    lbl_s = torch.empty(n, h * w, dtype=torch.uint8, device=dev)
    seg = torch.empty(n, nseg, K, dtype=torch.int32, device=dev)
    counts = torch.zeros(n, K, dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = _lib.ptr
    _lib.check(L.dcl_label_hist(p(lab_d), n, H, W, scale, K, p(lbl_s), p(seg), p(counts), st), "k1")
    np.testing.assert_array_equal(counts.cpu().numpy(), counts_ref)
    exp = lbl_ref.reshape(n, -1).copy(); exp[(exp < 0) | (exp >= K)] = 255
    np.testing.assert_array_equal(lbl_s.cpu().numpy(), exp.astype(np.uint8))
    np.testing.assert_array_equal(seg.sum(1).cpu().numpy(), counts_ref)
    # every rank of a few pairs
    pairs = [(0, 1), (2, 6), (1, 0)]
    V = int(min(counts_ref[b, k] for b, k in pairs))
    sel = np.stack([rs.permutation(counts_ref[b, k])[:V] for b, k in pairs]).astype(np.int32)
    pix = torch.empty(len(pairs), V, dtype=torch.int32, device=dev)
    pb = torch.tensor([b for b, _ in pairs], dtype=torch.int32, device=dev)
    pk = torch.tensor([k for _, k in pairs], dtype=torch.int32, device=dev)
    _lib.check(L.dcl_rank_select(p(lbl_s), p(seg), n, h * w, K, p(pb), p(pk), len(pairs), V,
                                 p(torch.from_numpy(sel).to(dev)), p(pix), st), "k2")
    flat = lbl_ref.reshape(n, -1)
    want = np.stack([np.flatnonzero(flat[b] == k)[sel[t]] for t, (b, k) in enumerate(pairs)])
    np.testing.assert_array_equal(pix.cpu().numpy(), want)


@pytest.mark.parametrize("halves", [False, True])
def test_global_negative_bank_two_virtual_ranks(dev, oracle, halves):
    """Extension (no reference oracle, SURVEY section 8 row e): every term contrasts against the banks of
    all ranks.  Two virtual ranks on one GPU (the peer's banks are injected where the RCCL all-gather
    would deliver them) against the oracle's single-process emulation on the concatenated banks.
    ``halves``: the peer's banks arrive as (hi | lo) f16 rows -- what the f16x3 mode gathers -- so the peer segments
    run on the f16 matrix pipe too; otherwise as f32 rows (peer segments on the f32 kernels)."""
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    from mscs_amd.losses.engine import class_layout
    cfg = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 2, "weights": [1.0, 0.6],
           "cross_scale_contrast": True, "max_features_total": 1500, "global_negatives": True}
    data = [_random_case(21 + q, 2, 64, 128, 20, 32, (4, 8), classes=[[0, 3, 5, 7], [3, 5, 9, 19]][q])
            for q in range(2)]
    seeds = [100, 200]
    ocfg = oracle.LossConfig(num_all_classes=20, temperature=0.1, max_features_total=1500, scales=2,
                             weights=[1.0, 0.6], cross_scale_contrast=True)
    # virtual rank 1 first: its banks are what the all-gather would hand to rank 0
    mods = [DenseContrastiveLossV2_ms(cfg) for _ in range(2)]
    torch.manual_seed(seeds[1])
    f1 = [f.to(dev) for f in data[1][1]]
    mods[1](data[1][0].to(dev), f1)
    st1 = mods[1].last_state
    peer_banks = [None, [sc.bank for sc in st1.scales]]
    peer_layouts = [None, [class_layout(sc.plan) for sc in st1.scales]]
    if halves:
        assert all(sc.bank_h is not None for sc in st1.scales)
        mods[0]._emulated_peers = (0, None, peer_layouts, [None, [sc.bank_h for sc in st1.scales]])
    else:
        mods[0]._emulated_peers = (0, peer_banks, peer_layouts)
    f0 = [f.to(dev).requires_grad_(True) for f in data[0][1]]
    torch.manual_seed(seeds[0])
    loss = mods[0](data[0][0].to(dev), f0)
    loss.backward()
    ref = oracle.dcv2_ms_global([d[0].numpy() for d in data], [[f.numpy() for f in d[1]] for d in data],
                                ocfg, seeds, rank=0)
    np.testing.assert_allclose(loss.item(), ref.loss, rtol=LOSS_RTOL)
    np.testing.assert_allclose([x.item() for x in mods[0].ms_losses], ref.ms_losses, rtol=LOSS_RTOL)
    np.testing.assert_allclose([x.item() for x in mods[0].cs_losses], ref.cs_losses, rtol=LOSS_RTOL)
    for s in range(2):
        _check_grad(f0[s].grad.cpu().numpy(), ref.grads[s])
    # class 19 / 9 exist on rank 1 only, class 0 / 7 on rank 0 only: segments with empty positive ranges
    assert len(mods[0].last_state.terms[0].segs) == 2


def test_upernet_swin_training_step_with_twoscale_and_contrastive_loss(dev):
    """BASELINE configs[3] plumbing at toy size: UPerNet + Swin-T through OCRNetManager.forward_step with
    TwoScaleLoss (aux + main CE) and the HIP DCV2_ms on the four FPN projector maps; one SGD step."""
    from mscs_amd.managers import OCRNetManager
    from mscs_amd.utils import set_verbosity
    set_verbosity(40)
    cfg = {"name": "t", "mode": "training", "manager": "OCRNet", "cuda": True, "seed": 1,
           "graph": {"model": "UPerNet", "backbone": "swinT", "sync_bn": False, "pretrained": False,
                     "align_corners": False, "aux_head": {"in_index": 3, "dropout_rate": 0.1}, "dropout_rate": 0.1,
                     "ms_projector": {"mlp": [[1, -1, 1]], "scales": 4, "d": 256, "use_bn": True, "position": "fpn"}},
           "data": {"dataset": "ADE20K", "experiment": 1, "batch_size": 2, "synthetic": True, "synthetic_length": 2,
                    "synthetic_mode": "blocky", "transform_values": {"crop_shape": [128, 128]}},
           "loss": {"name": "LossWrapper", "losses": {"TwoScaleLoss": 1, "DenseContrastiveLossV2_ms": 0.1},
                    "interm": {"name": "CrossEntropyLoss", "weight": 0.4, "args": []},
                    "final": {"name": "CrossEntropyLoss", "weight": 1.0, "args": []},
                    "temperature": 0.1, "scales": 4, "weights": [1, 0.7, 0.4, 0.1], "cross_scale_contrast": True,
                    "min_views_per_class": 2, "max_views_per_class": 2500, "max_features_total": 2000},
           "train": {"learning_rate": 1e-4, "lr_fct": "polynomial", "optim": "AdamW", "lr_batchwise": True,
                     "epochs": 1, "weight_decay": 0.01}}
    mgr = OCRNetManager(cfg)
    mgr.model.train()
    img, _, _ = next(iter(mgr.data_loaders["train_loader"]))
    # four classes in 32x32 blocks: every scale down to stride 32 keeps >= 2 pixels for some (image, class)
    # pairs (V >= 2, so every anchor has a positive; V = 1 would give the reference's 0/0 = NaN)
    g = torch.Generator().manual_seed(4)
    lbl = torch.randint(0, 4, (2, 4, 4), generator=g).repeat_interleave(32, 1).repeat_interleave(32, 2).int()
    before = torch.cat([p.detach().flatten() for p in mgr.model.parameters()]).clone()
    mgr.optimiser.zero_grad()
    ret = mgr.forward_step(img.to(dev), lbl.to(dev))
    assert ret["interm_output"].shape == ret["output"].shape == (2, 150, 128, 128)
    assert [tuple(f.shape) for f in ret["feats"]] == [(2, 256, 32, 32), (2, 256, 16, 16), (2, 256, 8, 8), (2, 256, 4, 4)]
    ret["loss"].backward()
    mgr.optimiser.step()
    assert torch.isfinite(ret["loss"]).item()
    vals = mgr.loss.loss_vals
    assert {"TwoScaleLoss", "DenseContrastiveLossV2_ms", "DenseContrastiveLossV2_ms_ms0", "DenseContrastiveLossV2_ms_ms3",
            "DenseContrastiveLossV2_ms_cs0", "DenseContrastiveLossV2_ms_cs1"} <= set(vals)
    after = torch.cat([p.detach().flatten() for p in mgr.model.parameters()])
    assert not torch.equal(before, after)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mgr.model.parameters() if p.requires_grad)


def test_single_view_gives_nan_like_the_reference(dev, oracle):
    """V = 1 and a class present in one image only -> that anchor has no positive: the reference divides 0/0
    (DenseContrastiveLossV2.py:188) and returns NaN; so do the oracle and the HIP path (no silent clamp)."""
    from mscs_amd.losses import DenseContrastiveLossV2
    label = torch.zeros(1, 16, 16, dtype=torch.long)
    label[0, :8] = 1
    label[0, 0, 0] = 2                      # one pixel of class 2 -> min count 1 -> V = 1
    feat = torch.randn(1, 8, 16, 16, generator=torch.Generator().manual_seed(0))
    cfg = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "min_views_per_class": 1}
    mod = DenseContrastiveLossV2(cfg)
    torch.manual_seed(0)
    loss = mod(label.to(dev), feat.to(dev))
    ocfg = oracle.LossConfig(num_all_classes=20, temperature=0.1, min_views_per_class=1, scales=1)
    ref, plan, _ = oracle.dcv2_single(label.numpy(), feat.numpy(), ocfg, rng=oracle.MT19937(0), want_grad=False)
    assert plan.V == 1 and np.isnan(ref)
    assert torch.isnan(loss).item()


def test_ade20k_class_count_and_mixed_input_dtypes(dev):
    """K = 151 (ADE20K): 600 (image, class) pairs, V cut by max_features_total; int32 labels; the loss equals a
    dense fp32 torch evaluation of the same banks; features arriving as bf16 are promoted."""
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    gen = torch.Generator().manual_seed(2)
    n, H, W, K = 4, 256, 256, 151
    label = torch.randint(0, K, (n, H, W), generator=gen, dtype=torch.int32).to(dev)
    feats = [torch.randn(n, 256, H // s, W // s, generator=gen).to(dev).requires_grad_(True) for s in (4, 8)]
    cfg = {"dataset": "ADE20K", "experiment": 1, "temperature": 0.1, "scales": 2, "weights": [1.0, 0.7],
           "cross_scale_contrast": True}
    mod = DenseContrastiveLossV2_ms(cfg)
    torch.manual_seed(0)
    loss = mod(label, feats)
    loss.backward()
    st = mod.last_state
    p0 = st.scales[0].plan
    assert p0.T == 600 and p0.V == min(int(p0.pair_cnt.min()), 10000 // 600)
    total = 0.0
    for t, term in enumerate(st.terms):
        A, B = st.scales[term.a], st.scales[term.b]
        Fa, Fb = A.bank[:A.plan.N], B.bank[:B.plan.N]
        ca = torch.from_numpy(np.repeat(A.plan.pair_k[A.plan.slot_pair], A.plan.V)).to(dev)
        cb = torch.from_numpy(np.repeat(B.plan.pair_k[B.plan.slot_pair], B.plan.V)).to(dev)
        Smat = (Fa @ Fb.T) / term.tau
        pos = (ca[:, None] == cb[None, :]).float()
        neg = 1 - pos
        if term.intra:
            pos.fill_diagonal_(0)
        E = torch.exp(Smat)
        Z = (E * neg).sum(1, keepdim=True)
        P = pos.sum(1)
        Pn = P if term.intra else torch.where(P > 0, P, torch.ones_like(P))
        ref = -((pos * (Smat - torch.log(E + Z))).sum(1) / Pn).mean()
        np.testing.assert_allclose(st.loss_buf[t].item(), ref.item(), rtol=LOSS_RTOL)
        total += term.weight * ref.item()
    np.testing.assert_allclose(loss.item(), total, rtol=LOSS_RTOL)
    torch.manual_seed(0)
    loss_bf16 = mod(label, [f.detach().bfloat16() for f in feats])
    assert torch.isfinite(loss_bf16).item() and abs(loss_bf16.item() - loss.item()) < 0.05 * abs(loss.item())
