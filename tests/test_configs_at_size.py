"""BASELINE.json configs 2-5 at their FULL loss workload on the MI355X: the sampling plan
must give the (T, V) the reference produced in the survey probe, and loss AND feature gradients must match a
dense fp32 torch evaluation of the reference formulas (losses/DenseContrastiveLossV2.py:127-192,
DenseContrastiveLossV2_ms.py:84-161) built from the SAME sampled pixels through torch indexing + autograd --
i.e. everything after the (separately bit-exact-tested) sampling is checked end to end at size, including the
gather, the L2 normalisation and the scatter of the gradient.

Tolerances: loss rtol 1e-5, gradients 1e-4 of max|grad| (summation order differs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5
GRAD_ATOL_REL = 1e-4

# (id, dataset, K, n, H, W, expected (T, V) per stride 4 / 8 / 16 / 32): SURVEY.md Appendix C, iid labels seed 0
CASES = {
    "cfg2_hrnet_cts_n12": ("CITYSCAPES", 20, 12, 512, 1024, [(228, 43), (228, 43), (228, 43), (228, 12)]),
    "cfg3_hrnet_cts_4gpu_n3": ("CITYSCAPES", 20, 3, 512, 1024, [(57, 175), (57, 175), (57, 81), (57, 12)]),
    "cfg4_swinT_ade_8gpu_n2": ("ADE20K", 151, 2, 512, 512, [(300, 33), (300, 15), (239, 5), (10, 5)]),
    "cfg4p_swinT_ade_1gpu_n16": ("ADE20K", 151, 16, 512, 512, [(2400, 4), (2400, 4), (1929, 5), (69, 5)]),
    "cfg5_swinL_ade_8gpu_n2_640": ("ADE20K", 151, 2, 640, 640, [(300, 33), (300, 27), (294, 5), (38, 5)]),
}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import mscs_amd  # noqa: F401
    from mscs_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _dense_term(Fa, ca, Fb, cb, tau, intra):
    """One InfoNCE term exactly as the reference writes it (masks as float matrices, no max-shift)."""
    S = (Fa @ Fb.T) / tau
    pos = (ca[:, None] == cb[None, :]).to(Fa.dtype)
    neg = 1 - pos
    if intra:
        pos = pos * (1 - torch.eye(Fa.shape[0], device=Fa.device, dtype=Fa.dtype))
    E = torch.exp(S)
    Z = (E * neg).sum(1, keepdim=True)
    logp = S - torch.log(E + Z)
    P = pos.sum(1)
    Pn = P if intra else torch.where(P > 0, P, torch.ones_like(P))
    return -((pos * logp).sum(1) / Pn).mean()


def _run_case(dev, case, mfma, S, cross, weights, ref_dtype=torch.float32):
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    dataset, K, n, H, W, expect = CASES[case]
    C = 256
    gen = torch.Generator().manual_seed(0)
    label = torch.randint(0, K, (n, H, W), generator=gen).to(dev)
    strides = [4 << s for s in range(S)]
    feats = [torch.randn(n, C, H // s, W // s, generator=gen).to(dev).requires_grad_(True) for s in strides]
    cfg = {"dataset": dataset, "experiment": 1, "temperature": 0.1, "scales": S, "weights": weights,
           "cross_scale_contrast": cross, "mfma_mode": mfma}
    mod = DenseContrastiveLossV2_ms(cfg)
    assert mod.DCV2_scale0.num_all_classes == K
    torch.manual_seed(0)
    loss = mod(label, feats)
    loss.backward()
    st = mod.last_state
    assert [(sc.plan.T, sc.plan.V) for sc in st.scales] == expect[:S]

    # ---- dense fp32 evaluation from the same pixels, through torch indexing and autograd
    ref_feats = [f.detach().to(ref_dtype).requires_grad_(True) for f in feats]
    banks, classes = [], []
    for s, sc in enumerate(st.scales):
        stride = strides[s]
        lbl_s = label[:, ::stride, ::stride].reshape(n, -1)
        pix = sc.pix.long()
        b = sc.pair_b.long()[:, None].expand_as(pix)
        assert torch.equal(lbl_s[b, pix], sc.pair_k.long()[:, None].expand_as(pix))       # right class
        key = (b * lbl_s.shape[1] + pix).flatten()
        assert key.unique().numel() == key.numel()                                         # no pixel twice
        X = ref_feats[s].reshape(n, C, -1)[b.flatten(), :, pix.flatten()]                  # [N, C], (t, v) order
        banks.append(F.normalize(X, p=2, dim=1))
        classes.append(sc.pair_k.long()[:, None].expand_as(pix).flatten())
    total = 0.0
    for t, term in enumerate(st.terms):
        Fb = banks[term.b].detach() if term.detach_b else banks[term.b]
        ref = _dense_term(banks[term.a], classes[term.a], Fb, classes[term.b], term.tau, term.intra)
        np.testing.assert_allclose(st.loss_buf[t].item(), ref.item(), rtol=LOSS_RTOL, err_msg=f"term {t}")
        total = total + term.weight * ref
    np.testing.assert_allclose(loss.item(), total.item(), rtol=LOSS_RTOL)
    total.backward()
    for s in range(S):
        g, r = feats[s].grad, ref_feats[s].grad
        scale = r.abs().max().item()
        if scale == 0:                          # a scale no weighted term touches (cross-scale-only weights)
            assert g.abs().max().item() == 0
            continue
        err = (g.to(r.dtype) - r).abs().max().item()
        assert err <= GRAD_ATOL_REL * scale, (case, s, err, scale)
        assert torch.equal(g != 0, r != 0) or ((g != 0) & (r == 0)).sum().item() == 0     # support = sampled pixels
    return mod


@pytest.mark.parametrize("mfma", ["f16x3", "f32"])
@pytest.mark.parametrize("case", list(CASES))
def test_config_at_size_four_scales_with_cross(dev, case, mfma):
    """4 scales + both cross-scale terms: the shipped json's shape (weights [1, .7, .4, .1])."""
    _run_case(dev, case, mfma, 4, True, [1.0, 0.7, 0.4, 0.1])


def test_config2_three_scales(dev):
    """BASELINE configs[1] as worded ("3 scales"): the workload bench.py times."""
    _run_case(dev, "cfg2_hrnet_cts_n12", "f16x3", 3, True, [1.0, 0.7, 0.4])


def test_config2_three_scales_against_fp64(dev):
    """The same workload against the dense evaluation in FLOAT64 (N x N = 770 MB per matrix at N = 9 804): a comparator
    that shares no fp32 GEMM rounding with anything -- the fp32 variants above run the reference formulas through the
    library's fp32 GEMM.  Same tolerances (they are the kernels' bars, not the comparator's)."""
    _run_case(dev, "cfg2_hrnet_cts_n12", "f16x3", 3, True, [1.0, 0.7, 0.4], ref_dtype=torch.float64)


def test_config5_cross_scale_only_weights(dev):
    """Config 5 = "cross-scale contrastive": same graph with the intra-scale weights at zero still has to produce the
    cross terms' gradients on scale 0 and on the coarse scales."""
    mod = _run_case(dev, "cfg5_swinL_ade_8gpu_n2_640", "f16x3", 4, True, [0.0, 0.0, 0.0, 0.0])
    assert len(mod.cs_losses) == 2 and all(torch.isfinite(x) for x in mod.cs_losses)


def test_single_pass_f16_similarity_misses_the_gradient_tolerance(dev):
    """Why config 5's "fp16 MFMA similarity" runs as split-f16 (three passes) and not one f16 pass: rounding the
    normalised banks to f16 once perturbs the logits by ~1e-3 at tau = 0.1 and per-element gradients by more than the
    1e-4-of-max tolerance this suite holds the kernels to, while the f16x3 kernels meet it (measured here, not
    argued)."""
    from mscs_amd.losses import DenseContrastiveLossV2_ms
    dataset, K, n, H, W, _ = CASES["cfg5_swinL_ade_8gpu_n2_640"]
    gen = torch.Generator().manual_seed(0)
    label = torch.randint(0, K, (n, H, W), generator=gen).to(dev)
    feat = torch.randn(n, 256, H // 4, W // 4, generator=gen).to(dev).requires_grad_(True)
    mod = DenseContrastiveLossV2_ms({"dataset": dataset, "experiment": 1, "temperature": 0.1, "scales": 1,
                                     "weights": [1.0], "cross_scale_contrast": False})
    torch.manual_seed(0)
    mod(label, [feat]).backward()
    sc = mod.last_state.scales[0]
    N = sc.plan.N
    cls = torch.from_numpy(np.repeat(sc.plan.pair_k[sc.plan.slot_pair], sc.plan.V)).to(dev)

    def grad_of(bank):
        b = bank.detach().clone().requires_grad_(True)
        _dense_term(b, cls, b, cls, 0.1, True).backward()
        return b.grad

    exact = grad_of(sc.bank[:N].double()).float()
    single = grad_of(sc.bank[:N].half().float())            # one f16 rounding of the operands, f32 accumulation
    scale = exact.abs().max().item()
    err_single = (single - exact).abs().max().item() / scale
    assert err_single > GRAD_ATOL_REL, err_single           # fails the bar ...
    # ... which the shipped f16x3 kernel meets on the same banks (dF before the normalisation VJP is not exposed, so
    # compare through the scatter: gradient w.r.t. the features, dense fp64 reference through autograd)
    ref_feat = feat.detach().double().requires_grad_(True)
    pix = sc.pix.long()
    b = sc.pair_b.long()[:, None].expand_as(pix)
    X = F.normalize(ref_feat.reshape(n, 256, -1)[b.flatten(), :, pix.flatten()], dim=1)
    c2 = sc.pair_k.long()[:, None].expand_as(pix).flatten()
    _dense_term(X, c2, X, c2, 0.1, True).backward()
    r = ref_feat.grad.float()
    assert (feat.grad - r).abs().max().item() <= GRAD_ATOL_REL * r.abs().max().item()
