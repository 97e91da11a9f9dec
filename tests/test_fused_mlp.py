"""GPU tests of the GEMM's fused epilogues (csrc/dcl_gemm.hip, dcl_gemm_f16x3_ep) and of what the Swin port builds on them:
fc1 + GELU, fc2's data gradient x gelu'(h), shortcut + drop_path(branch) in the projection / fc2 epilogue
(reference models/Swin.py:62-76 Mlp, :318-321 the two residual sums of a block)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gelu64(v):
    return 0.5 * v * (1.0 + torch.erf(v * 0.7071067811865476))


def _dgelu64(v):
    return 0.5 * (1.0 + torch.erf(v * 0.7071067811865476)) + v * torch.exp(-0.5 * v * v) * 0.3989422804014327


@pytest.mark.parametrize("M,K,N", [(6400, 384, 1536), (2080, 96, 384), (25600, 192, 768), (1056, 1536, 6144)])
def test_gemm_epilogues_against_fp64(M, K, N):
    """The three epilogues of dcl_gemm_f16x3_ep against float64 on Swin Mlp shapes (interior and ragged tiles: 2080 and 1056 rows are
    not multiples of any workgroup tile): values to 3e-6 of the result's maximum, the emitted absmax exact, bitwise reproducible."""
    import mscs_amd  # noqa: F401
    from mscs_amd.models import amax as am, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    v64 = x.double() @ w.double().t() + b.double()
    # ep 1: pre-activation and GELU in one launch
    for rep in range(2):
        h, a = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        ca = am.zeros(1, dev)
        ops.gemm_f16x3_ep(x, w, True, M, N, K, h, am.amax_of(x), am.amax_of(w), 1, bias=b, c_amax=ca, out2=a)
        if rep == 0:
            h0, a0 = h.clone(), a.clone()
    assert torch.equal(h, h0) and torch.equal(a, a0)
    den = v64.abs().max()
    assert ((h.double() - v64).abs().max() / den).item() < 3e-6
    assert ((a.double() - _gelu64(v64)).abs().max() / den).item() < 3e-6
    assert ca.item() == h.abs().max().item()
    # ep 2: data gradient times gelu'(aux)
    gy = torch.randn(M, N, generator=g).to(dev)            # dy of fc2's output side: here [M, N] . W [N, K] -> [M, K]
    hh = (torch.randn(M, K, generator=g) * 1.5).to(dev)
    out = torch.empty(M, K, device=dev)
    cg = am.zeros(1, dev)
    ops.gemm_f16x3_ep(gy, w, False, M, K, N, out, am.amax_of(gy), am.amax_of(w), 2, c_amax=cg, aux=hh)
    want = (gy.double() @ w.double()) * _dgelu64(hh.double())
    assert ((out.double() - want).abs().max() / want.abs().max()).item() < 3e-6
    assert cg.item() == out.abs().max().item()
    # ep 3: shortcut + per-sample factor * (product + bias)
    B = 4 if M % 4 == 0 else 1
    sc = torch.tensor([0.0, 1.0 / 0.7, 1.0 / 0.7, 0.0][:B]).to(dev)
    short = torch.randn(M, N, generator=g).to(dev)
    for scale in (None, sc):
        y = torch.empty(M, N, device=dev)
        cy = am.zeros(1, dev)
        ops.gemm_f16x3_ep(x, w, True, M, N, K, y, am.amax_of(x), am.amax_of(w), 3, bias=b, c_amax=cy, aux=short, rowscale=scale,
                          rows_per_scale=M // B)
        f = 1.0 if scale is None else sc.double().repeat_interleave(M // B).view(-1, 1)
        want = short.double() + f * v64
        assert ((y.double() - want).abs().max() / want.abs().max()).item() < 3e-6
        assert cy.item() == y.abs().max().item()
        if scale is not None:
            assert torch.equal(y[:M // B], short[:M // B])          # a dropped sample is the shortcut, bit for bit


def test_gemm_epilogue_argument_errors():
    import mscs_amd  # noqa: F401
    from mscs_amd.models import amax as am, ops
    dev = torch.device("cuda:0")
    x, w = torch.randn(1024, 64, device=dev), torch.randn(96, 64, device=dev)
    out = torch.empty(1024, 96, device=dev)
    with pytest.raises(RuntimeError, match="second output"):
        ops.gemm_f16x3_ep(x, w, True, 1024, 96, 64, out, am.amax_of(x), am.amax_of(w), 1)
    with pytest.raises(RuntimeError, match="second input"):
        ops.gemm_f16x3_ep(x, w, True, 1024, 96, 64, out, am.amax_of(x), am.amax_of(w), 3)
    with pytest.raises(RuntimeError, match="k-major"):
        ops.gemm_f16x3_ep(x, w, False, 1024, 96, 64, out, am.amax_of(x), am.amax_of(w), 3, aux=out)


@pytest.mark.parametrize("residual,drop", [(False, 0.0), (True, 0.0), (True, 0.3)])
def test_fused_mlp_matches_unfused_and_fp64(residual, drop):
    """Mlp (+ residual sum, + DropPath) through ops._FusedMlp against the unfused chain TokenLinear -> GELU -> TokenLinear (->
    addcmul) under the same mask draw, and both against float64: output, input / shortcut gradients, all four parameter gradients."""
    import mscs_amd  # noqa: F401
    from mscs_amd.models import ops, ops_linear
    from mscs_amd.models.Swin import DropPath, Mlp
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, L, C = 4, 3200, 192
    mlp = Mlp(C, 4 * C).to(dev).train()
    dp = DropPath(drop).train() if drop else torch.nn.Identity()
    x = torch.randn(B, L, C, device=dev)
    s = torch.randn(B, L, C, device=dev)
    gy = torch.randn(B, L, C, device=dev)
    res = {}
    for fused in (True, False):
        ops_linear.FUSED_MLP = fused
        try:
            mlp.zero_grad()
            xi, si = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
            torch.manual_seed(11)
            assert mlp._fusable(xi) == fused
            y = mlp.add_to(si, xi, dp) if residual else mlp(xi)
            y.backward(gy)
            res[fused] = [y.detach(), xi.grad] + ([si.grad] if residual else []) + [p.grad.clone() for p in mlp.parameters()]
        finally:
            ops_linear.FUSED_MLP = True
    # float64 (same mask: the factors are recovered from the unfused run's shortcut-free part when DropPath is active)
    w1, b1, w2, b2 = [p.detach().double() for p in (mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias)]
    x64 = x.double().requires_grad_(True)
    s64 = s.double().requires_grad_(True)
    h = x64 @ w1.t() + b1
    br = _gelu64(h) @ w2.t() + b2
    if residual:
        torch.manual_seed(11)
        f = torch.ones(B, 1, 1, dtype=torch.float64, device=dev)
        if drop:
            f = (x.new_empty((B, 1, 1)).bernoulli_(1.0 - drop) / (1.0 - drop)).double()
        y64 = s64 + f * br
    else:
        y64 = br
    gs = torch.autograd.grad(y64, [x64, s64] if residual else [x64], gy.double(), retain_graph=True)
    names = ["y", "dx"] + (["dshortcut"] if residual else []) + ["dw1", "db1", "dw2", "db2"]
    for i, n in enumerate(names):
        a, b = res[True][i], res[False][i]
        den = b.abs().max().item() + 1e-12
        assert (a - b).abs().max().item() <= 2e-5 * den, (n, (a - b).abs().max().item() / den)
    assert ((res[True][0].double() - y64).abs().max() / y64.abs().max()).item() < 3e-6
    assert ((res[True][1].double() - gs[0]).abs().max() / gs[0].abs().max()).item() < 5e-6
    if residual:
        assert torch.equal(res[True][2], gy)                 # the shortcut's gradient is the incoming one, untouched
        if drop:                                             # a dropped sample sends nothing into the branch
            assert torch.equal(res[True][1].view(B, -1).abs().amax(1) == 0, f.view(-1) == 0)


def test_swin_block_fused_epilogues_match_unfused_chain():
    """A whole SwinTransformerBlock (HIP attention) with DropPath active: the block with the residual sums in the projection / fc2
    epilogues and GELU inside the GEMMs against the same block on the unfused chain, same mask draws: output, input gradient, every
    parameter gradient."""
    import mscs_amd  # noqa: F401
    from mscs_amd.models import ops, ops_linear
    from mscs_amd.models.Swin import BasicLayer, SwinTransformerBlock
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    blk = SwinTransformerBlock(dim=96, num_heads=3, window_size=7, shift_size=3, drop_path=0.25).to(dev).train()
    H, W = 40, 40
    blk.H, blk.W = H, W
    x = torch.randn(4, H * W, 96, device=dev)
    mask = BasicLayer(dim=96, depth=2, num_heads=3)._shift_mask(H, W, dev)
    res = {}
    for fused in (True, False):
        ops_linear.FUSED_MLP = fused
        try:
            blk.zero_grad()
            xi = x.clone().requires_grad_(True)
            torch.manual_seed(21)
            y = blk(xi, mask)
            (y * torch.cos(torch.arange(y.numel(), device=dev).view_as(y) * 0.37)).sum().backward()
            res[fused] = (y.detach(), xi.grad, {k: p.grad.clone() for k, p in blk.named_parameters()})
        finally:
            ops_linear.FUSED_MLP = True
    ya, ga, pa = res[True]
    yb, gb, pb = res[False]
    assert (ya - yb).abs().max().item() <= 2e-5 * yb.abs().max().item()
    assert (ga - gb).abs().max().item() <= 1e-4 * gb.abs().max().item()
    for k in pa:
        assert (pa[k] - pb[k]).abs().max().item() <= 1e-4 * max(pb[k].abs().max().item(), 1e-6), k


def test_dropout2d_folded_into_the_classifier_matches_the_two_modules():
    """ops.dropout2d_conv1x1 (channel factors in per-sample weights) against conv(F.dropout2d(x)) under the same generator state:
    the same channels dropped, output and all gradients to fp32 round-off; eval mode and p = 0 run the modules as they are."""
    import mscs_amd  # noqa: F401
    from mscs_amd.models import ops, ops_linear
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    drop = torch.nn.Dropout2d(0.1).train()
    conv = torch.nn.Conv2d(512, 150, 1).to(dev)
    x = torch.randn(4, 512, 40, 48, device=dev)
    gy = torch.randn(4, 150, 40, 48, device=dev)
    res = []
    for folded in (True, False):
        conv.zero_grad()
        xi = x.clone().requires_grad_(True)
        torch.manual_seed(77)
        y = ops.dropout2d_conv1x1(xi, drop, conv) if folded else conv(drop(xi))
        y.backward(gy)
        res.append((y.detach(), xi.grad, conv.weight.grad.clone(), conv.bias.grad.clone()))
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    dropped = (res[0][1].abs().amax((2, 3)) == 0)
    assert dropped.any() and torch.equal(dropped, res[1][1].abs().amax((2, 3)) == 0)       # whole channels, the same ones
    drop.eval()
    assert torch.equal(ops.dropout2d_conv1x1(x, drop, conv), conv(x))


@pytest.mark.parametrize("B,H,W,C", [(4, 40, 48, 192), (2, 32, 32, 96), (3, 20, 24, 384)])
def test_lateral_convolution_reads_tokens(B, H, W, C):
    """UPerNet's lateral block (conv1x1 -> BN -> ReLU) on a backbone level kept token-major (models/Swin.TokenMap -> ops.
    conv1x1_from_tokens) against the same block on the reference's NCHW copy, and the 1x1 convolution alone against float64:
    output, token gradient (token-major, contiguous), weight and norm-parameter gradients."""
    import mscs_amd  # noqa: F401
    from mscs_amd.models import ops, ops_linear
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    from mscs_amd.models.Swin import TokenMap
    from mscs_amd.models.UPerNet import FPN, _conv1x1_bn_relu
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    block = ops.use_gemm_conv1x1(_conv1x1_bn_relu(C, 512, FusedBatchNorm2d)).to(dev).train()
    tok = torch.randn(B, H * W, C, device=dev)
    gy = torch.randn(B, 512, H, W, device=dev)
    assert ops.conv1x1_from_tokens_ok(tok, block[0], H, W)
    res = []
    for tokens in (True, False):
        block.zero_grad()
        ti = tok.clone().requires_grad_(True)
        tm = TokenMap(ti, H, W)
        y = FPN._lateral(None, block, tm if tokens else tm.nchw())
        y.backward(gy)
        assert ti.grad.is_contiguous()
        res.append((y.detach(), ti.grad) + tuple(p.grad.clone() for p in block.parameters()))
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 3e-5 * b.abs().max().item() + 1e-9
    # the convolution alone against float64
    ti = tok.clone().requires_grad_(True)
    w = block[0].weight
    w.grad = None
    y = ops.conv1x1_from_tokens(ti, block[0], H, W)
    y.backward(gy)
    w64 = w.detach().double().view(512, C)
    y64 = torch.einsum("oc,bpc->bop", w64, tok.double()).view(B, 512, H, W)
    gt64 = torch.einsum("bop,oc->bpc", gy.double().view(B, 512, -1), w64)
    gw64 = torch.einsum("bop,bpc->oc", gy.double().view(B, 512, -1), tok.double())
    for got, want in ((y, y64), (ti.grad, gt64), (w.grad.view(512, C), gw64)):
        assert ((got.double() - want).abs().max() / want.abs().max()).item() < 3e-6


@pytest.mark.parametrize("M,N,K", [(4 * 1600, 768, 3072), (8 * 800, 192, 192), (4 * 6400, 384, 96)])
def test_gemm_with_a_per_token_factor_on_the_gradient(M, N, K):
    """dcl_gemm_f16x3_ascaled behind linear_dgrad_f16x3 / linear_wgrad_f16x3: dy's rows times a per-sample factor (DropPath's mask /
    keep, zeros included) inside the data-gradient and the weight-gradient GEMM (k-split slabs, bias gradient as scaled row sums)
    against float64 on the explicitly scaled dy, and bitwise against ... itself (deterministic)."""
    import mscs_amd  # noqa: F401
    from mscs_amd.models import ops, ops_linear
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N)
    B = 4 if M % 1600 == 0 and M // 4 % 32 == 0 else 8
    gy = torch.randn(M, N, generator=g).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    sc = torch.tensor(([0.0, 1 / 0.7, 1 / 0.7, 0.0] * 2)[:B]).to(dev)
    grp = M // B
    gs64 = gy.double() * sc.double().repeat_interleave(grp).view(-1, 1)
    for rep in range(2):
        gx = ops.linear_dgrad_f16x3(gy, w, sc, grp)
        gw, gb = ops.linear_wgrad_f16x3(gy, x, want_bias=True, scale=sc, group=grp)
        if rep == 0:
            keep = (gx.clone(), gw.clone(), gb.clone())
    assert all(torch.equal(a, b) for a, b in zip(keep, (gx, gw, gb)))
    for got, want in ((gx, gs64 @ w.double()), (gw, gs64.t() @ x.double()), (gb, gs64.sum(0))):
        assert ((got.double() - want).abs().max() / want.abs().max()).item() < 3e-6
    assert (gx[:grp] == 0).all()                             # a dropped sample's rows: exact zeros


@pytest.mark.parametrize("shape", [(4, 512, 40, 48), (3, 96, 17, 23), (2, 48, 128, 256)])
def test_relu_then_bn_matches_the_two_modules(shape):
    """fused_bn.relu_then_bn (conv output -> in-place ReLU -> FusedBatchNorm2d with the ReLU's backward inside the norm's backward
    kernel: dcl_bn_bwd_apply_fused, relu + 4) against ReLU + the same norm run as separate autograd nodes, and against
    nn.BatchNorm2d in float64: output, input gradient (zero exactly where the input was <= 0), dgamma, dbeta."""
    import mscs_amd  # noqa: F401
    from mscs_amd.debug import cfg as dbg
    from mscs_amd.models.fused_bn import FusedBatchNorm2d, relu_then_bn
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    C = shape[1]
    bn = FusedBatchNorm2d(C).to(dev).train()
    torch.nn.init.normal_(bn.weight, 1.0, 0.3)
    torch.nn.init.normal_(bn.bias, 0.0, 0.3)
    x = torch.randn(*shape, device=dev)
    gy = torch.randn(*shape, device=dev)
    res = []
    keep = dbg.relu_then_bn
    try:
        for fused in (True, False):
            dbg.relu_then_bn = fused
            bn.zero_grad()
            bn.running_mean.zero_(); bn.running_var.fill_(1.0)
            xi = x.clone().requires_grad_(True)
            y = relu_then_bn(bn, xi * 1.0)
            y.backward(gy)
            res.append((y.detach(), xi.grad, bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_var.clone()))
    finally:
        dbg.relu_then_bn = keep
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item() + 1e-9
    assert torch.equal(res[0][1] == 0, x <= 0) or ((res[0][1] == 0) & (x > 0)).sum().item() < 4     # (a true zero gradient is possible)
    ref = torch.nn.BatchNorm2d(C).to(dev).double().train()
    ref.load_state_dict({k: v.double() for k, v in bn.state_dict().items() if "running" not in k and "num" not in k}, strict=False)
    x64 = x.double().requires_grad_(True)
    y64 = ref(torch.relu(x64))
    y64.backward(gy.double())
    for got, want in ((res[0][0], y64), (res[0][1], x64.grad), (res[0][2], ref.weight.grad), (res[0][3], ref.bias.grad)):
        assert ((got.double() - want).abs().max() / want.abs().max()).item() < 2e-5
