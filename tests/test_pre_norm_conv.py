"""conv1 -> bn1 -> relu -> conv2 without the normalised tensor (reference models/HRNet.py:77-93
BasicBlock, :117-137 Bottleneck): bn1's apply pass folded into the operand staging of conv2's forward (csrc/dcl_conv3x3_pre.hip) and
weight-gradient kernels (csrc/dcl_wgrad3x3d.hip, PRE forms), the map and the operand scale coming from dcl_bn_stats_minmax_part /
dcl_bn_finalize_pre.  The fused path must be BITWISE the path that writes the tensor (same fma, same operand scale, same tiles), and
that path is held to float64 by tests/test_model_ops_parity.py and tests/test_models.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _norm_inputs(n, c, h, w, dev, seed):
    g = torch.Generator().manual_seed(seed)
    z = (torch.randn(n, c, h, w, generator=g) * 1.7 + 0.3).to(dev)
    gamma = (torch.rand(c, generator=g) + 0.5)
    gamma[::5] *= -1.0                      # negative scales: the channel's MINIMUM maps to the largest output
    beta = torch.randn(c, generator=g) * 0.5
    rmean = torch.randn(c, generator=g) * 0.1
    rvar = torch.rand(c, generator=g) + 0.5
    return z, gamma.to(dev), beta.to(dev), rmean.to(dev), rvar.to(dev)


def _written(L, P, z, gamma, beta, rmean, rvar, eps=1e-5, mom=0.1):
    """the norm as the product ran it until round 5: statistics, then the apply kernel writes y and the absmax slots"""
    from mscs_amd.models import amax as A
    n, c, h, w = z.shape
    ns = L.dcl_bn_num_slices(n, c)
    part = torch.empty(c * ns * 2, device=z.device)
    mean, invstd, pivot = (torch.empty(c, device=z.device) for _ in range(3))
    rm, rv = rmean.clone(), rvar.clone()
    nbt = torch.zeros(1, dtype=torch.int64, device=z.device)
    amax = A.zeros(A.SLOTS, z.device)
    st = P.stream_ptr(z.device)
    P.check(L.dcl_bn_stats_part(P.ptr(z), n, c, h * w, P.ptr(part), P.ptr(rm), P.ptr(pivot), st), "stats")
    y = torch.empty_like(z)
    P.check(L.dcl_bn_apply_parts(P.ptr(z), None, P.ptr(part), ns, float(n * h * w), eps, mom, P.ptr(gamma), P.ptr(beta), n, c, h * w,
                                 1, P.ptr(y), P.ptr(mean), P.ptr(invstd), P.ptr(rm), P.ptr(rv), P.ptr(nbt), P.ptr(amax), P.ptr(pivot),
                                 None, st), "apply")
    return y, amax, mean, invstd, rm, rv, nbt


def _deferred(L, P, z, gamma, beta, rmean, rvar, eps=1e-5, mom=0.1):
    from mscs_amd.models import amax as A
    n, c, h, w = z.shape
    ns = L.dcl_bn_num_slices(n, c)
    part, mm = torch.empty(c * ns * 2, device=z.device), torch.empty(c * ns * 2, device=z.device)
    mean, invstd, pivot, sc, sh = (torch.empty(c, device=z.device) for _ in range(5))
    rm, rv = rmean.clone(), rvar.clone()
    nbt = torch.zeros(1, dtype=torch.int64, device=z.device)
    amax = A.zeros(A.SLOTS, z.device)
    st = P.stream_ptr(z.device)
    P.check(L.dcl_bn_stats_minmax_part(P.ptr(z), n, c, h * w, P.ptr(part), P.ptr(mm), P.ptr(rm), P.ptr(pivot), st), "stats_mm")
    P.check(L.dcl_bn_finalize_pre(P.ptr(part), P.ptr(mm), ns, float(n * h * w), eps, mom, P.ptr(gamma), P.ptr(beta), c, P.ptr(mean),
                                  P.ptr(invstd), P.ptr(rm), P.ptr(rv), P.ptr(nbt), P.ptr(pivot), P.ptr(sc), P.ptr(sh), P.ptr(amax),
                                  st), "finalize_pre")
    return sc, sh, amax, mean, invstd, rm, rv, nbt, mm, ns


# (N, C, H, W): the four branch widths of HRNet-W48 and layer 1's 64 channels on small maps; ragged tile edges; a plane that is
# not a multiple of four pixels (scalar tail of the statistics kernel)
_SHAPES = [(2, 48, 32, 64), (2, 96, 16, 32), (1, 192, 8, 32), (2, 384, 8, 16), (3, 64, 20, 40), (2, 32, 7, 24), (1, 16, 5, 9)]


@pytest.mark.parametrize("shape", _SHAPES)
def test_finalize_pre_reproduces_the_apply_kernel(dev, shape):
    """mean / invstd / running statistics / counter bitwise those of dcl_bn_apply_parts; relu(fma(z, sc, sh)) bitwise its output;
    the absmax derived from the channel extrema equal to the maximum over the written tensor; the extrema exact."""
    from mscs_amd import _lib as P
    L = P.lib()
    z, gamma, beta, rmean, rvar = _norm_inputs(*shape, dev, seed=sum(shape))
    y, amax_w, *stats_w = _written(L, P, z, gamma, beta, rmean, rvar)
    sc, sh, amax_d, *rest = _deferred(L, P, z, gamma, beta, rmean, rvar)
    stats_d, mm, ns = rest[:5], rest[5], rest[6]
    for a, b in zip(stats_w, stats_d):
        assert torch.equal(a, b)
    c = shape[1]
    y2 = torch.addcmul(sh.view(1, c, 1, 1).double(), z.double(), sc.view(1, c, 1, 1).double()).float().relu_()     # one rounding: an fma
    assert torch.equal(y, y2)
    assert amax_w.max().item() == amax_d.max().item() == y.max().item()
    mm = mm.view(c, ns, 2)
    assert torch.equal(mm[:, :, 0].min(1).values, z.amin((0, 2, 3))) and torch.equal(mm[:, :, 1].max(1).values, z.amax((0, 2, 3)))


@pytest.mark.parametrize("shape", _SHAPES + [(12, 48, 128, 256), (12, 384, 16, 32)])
def test_one_launch_statistics_and_finalisation_equal_the_two_calls(dev, shape):
    """dcl_bn_stats_pre (the last workgroup of a channel finalises it) against dcl_bn_stats_minmax_part + dcl_bn_finalize_pre: every
    output bitwise, twenty times in a row (the hand-over between workgroups is the part that could race)."""
    from mscs_amd import _lib as P
    from mscs_amd.models import amax as A
    L = P.lib()
    n, c, h, w = shape
    z, gamma, beta, rmean, rvar = _norm_inputs(*shape, dev, seed=3 + sum(shape))
    sc, sh, amax_d, mean, invstd, rm, rv, nbt, mm, ns = _deferred(L, P, z, gamma, beta, rmean, rvar)
    st = P.stream_ptr(dev)
    for _ in range(20):
        part2, mm2 = torch.empty(c * ns * 2, device=dev), torch.empty(c * ns * 2, device=dev)
        mean2, invstd2, sc2, sh2 = (torch.full((c,), float("nan"), device=dev) for _ in range(4))
        rm2, rv2 = rmean.clone(), rvar.clone()
        nbt2 = torch.zeros(1, dtype=torch.int64, device=dev)
        amax2, tickets = A.zeros(A.SLOTS, dev), torch.zeros(c, dtype=torch.int32, device=dev)
        P.check(L.dcl_bn_stats_pre(P.ptr(z), n, c, h * w, P.ptr(part2), P.ptr(mm2), P.ptr(tickets), float(n * h * w), 1e-5, 0.1,
                                   P.ptr(gamma), P.ptr(beta), P.ptr(mean2), P.ptr(invstd2), P.ptr(rm2), P.ptr(rv2), P.ptr(nbt2),
                                   P.ptr(sc2), P.ptr(sh2), P.ptr(amax2), st), "stats_pre")
        for a, b in ((sc, sc2), (sh, sh2), (mean, mean2), (invstd, invstd2), (rm, rm2), (rv, rv2), (nbt, nbt2), (mm, mm2)):
            assert torch.equal(a, b)
        assert amax_d.max().item() == amax2.max().item()
        assert (tickets == ns).all()


def _conv_pre(L, P, z, wp, co, amax, wamax, sc, sh, stride=1, r=0, p=0, bias=None):
    n, ci, h, w = z.shape
    out = torch.full((n, co, (h - 1) // stride + 1, (w - 1) // stride + 1), float("nan"), device=z.device)
    rc = L.dcl_conv3x3_pre_f16x3(P.ptr(z), n, ci, h, w, P.ptr(wp), co, P.ptr(amax), amax.numel(), P.ptr(wamax), P.ptr(sc), P.ptr(sh),
                                 P.ptr(bias), P.ptr(out), stride, r, p, P.stream_ptr(z.device))
    return rc, out


@pytest.mark.parametrize("shape", [s for s in _SHAPES if s[1] % 16 == 0])
def test_pre_convolution_is_bitwise_the_convolution_of_the_written_tensor(dev, shape):
    """dcl_conv3x3_pre_f16x3 on (z, sc, sh) == dcl_conv3x3_f16x3 on the tensor the apply kernel wrote, for the automatic tile and
    every tile that has the form, stride 1 and 2, with and without a bias; image borders are where a wrong order of map and
    padding would show (relu(sh) instead of 0)."""
    from mscs_amd import _lib as P
    from mscs_amd.models import ops
    from mscs_amd.models.amax import amax_of
    L = P.lib()
    n, c, h, w = shape
    z, gamma, beta, rmean, rvar = _norm_inputs(*shape, dev, seed=7 + sum(shape))
    y, amax_w, *_ = _written(L, P, z, gamma, beta, rmean, rvar)
    sc, sh, amax_d, *_ = _deferred(L, P, z, gamma, beta, rmean, rvar)
    for co in (c, 96):
        wt = torch.randn(co, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5
        bias = torch.randn(co, device=dev)
        wamax = amax_of(wt)
        wp = ops.conv3x3_pack(wt, wamax)
        for stride in (1, 2):
            assert L.dcl_conv3x3_pre_supported(n, c, co, h, w, stride) == 1, (shape, co, stride)
            oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
            seen = 0
            for (r, p) in [(0, 0)] + [(r, p) for r in (1, 2, 3) for p in ((1, 2, 4) if stride == 1 else (1,))]:
                for b in (None, bias):
                    rc, got = _conv_pre(L, P, z, wp, co, amax_d, wamax, sc, sh, stride, r, p, b)
                    if rc != 0:
                        assert (r, p) != (0, 0), "the automatic tile must have the form"
                        continue
                    want = torch.full((n, co, oh, ow), float("nan"), device=dev)
                    ops.conv3x3_launch(y, wp, co, amax_w, wamax, want, r, p, stride=stride, bias=b)
                    assert torch.equal(got, want), (shape, co, stride, r, p, b is not None)
                    seen += 1
            assert seen >= 8


def test_pre_convolution_with_its_tables_on_either_side_of_a_2_gib_boundary(dev):
    """The kernel builds the 64-bit table address from two 32-bit scalar halves: a set bit 31 in the LOW word must not leak into the
    high word (it did once -- a sign extension that faulted or not depending on where the allocator had put the norm's workspace)."""
    from mscs_amd import _lib as P
    from mscs_amd.models import ops
    from mscs_amd.models.amax import amax_of
    L = P.lib()
    shape = (2, 64, 16, 32)
    n, c, h, w = shape
    z, gamma, beta, rmean, rvar = _norm_inputs(*shape, dev, seed=99)
    sc, sh, amax_d, *_ = _deferred(L, P, z, gamma, beta, rmean, rvar)
    wt = torch.randn(c, c, 3, 3, device=dev) * 0.05
    wamax = amax_of(wt)
    wp = ops.conv3x3_pack(wt, wamax)
    rc, want = _conv_pre(L, P, z, wp, c, amax_d, wamax, sc, sh)
    assert rc == 0
    big = torch.empty(3 << 28, dtype=torch.float32, device=dev)             # 3 GiB: holds addresses with bit 31 set and clear
    base = big.data_ptr()
    seen = set()
    for off in range(0, big.numel() - 2 * c, 1 << 26):                       # every 256 MiB
        bit = ((base + 4 * off) >> 31) & 1
        if bit in seen:
            continue
        seen.add(bit)
        t_sc, t_sh = big[off:off + c], big[off + c:off + 2 * c]
        t_sc.copy_(sc)
        t_sh.copy_(sh)
        rc, got = _conv_pre(L, P, z, wp, c, amax_d, wamax, t_sc, t_sh)
        assert rc == 0 and torch.equal(got, want), hex(base + 4 * off)
    assert seen == {0, 1}


@pytest.mark.parametrize("shape", [(2, 48, 32, 64), (2, 96, 16, 32), (1, 192, 8, 32), (12, 384, 16, 32), (3, 64, 20, 40), (2, 32, 7, 24),
                                   (2, 64, 21, 48), (1, 48, 9, 16)])
def test_pre_weight_gradient_is_bitwise_the_weight_gradient_on_the_written_tensor(dev, shape):
    """dcl_wgrad3x3_pre_f16x3 on (z, sc, sh) == dcl_wgrad3x3_f16x3 on the written tensor (workgroup form, wave form at 192 tile pairs)."""
    from mscs_amd import _lib as P
    from mscs_amd.models import ops
    from mscs_amd.models.amax import tag
    L = P.lib()
    n, c, h, w = shape
    z, gamma, beta, rmean, rvar = _norm_inputs(*shape, dev, seed=11 + sum(shape))
    y, amax_w, *_ = _written(L, P, z, gamma, beta, rmean, rvar)
    sc, sh, amax_d, *_ = _deferred(L, P, z, gamma, beta, rmean, rvar)
    gy = torch.randn(n, c, h, w, device=dev) * 1e-4
    assert L.dcl_wgrad3x3_pre_supported(n, c, c, h, w, 1) == 1
    tag(y, amax_w)
    want = ops.conv3x3_wgrad(y, gy)
    got = ops.conv3x3_wgrad_pre(z, gy, sc, sh, amax_d)
    assert torch.equal(got, want)
    if w % 16 == 0:
        # stride 2 (csrc/dcl_wgrad3x3_s2.hip: the LDS-DMA form for one ci tile per wave, the direct-load form for 64 channels)
        for co in (c, 2 * c):
            gy2 = torch.randn(n, co, (h - 1) // 2 + 1, w // 2, device=dev) * 1e-4
            assert L.dcl_wgrad3x3_pre_supported(n, c, co, h, w, 2) == 1
            want = ops.conv3x3_wgrad(y, gy2, stride=2)
            got = ops.conv3x3_wgrad_pre(z, gy2, sc, sh, amax_d, stride=2)
            assert torch.equal(got, want), (shape, co)


def _block_run(dev, block_cls, cin, planes, shape, fuse):
    from mscs_amd.debug import cfg
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    from mscs_amd.models.ops import use_direct_conv3x3, use_direct_conv1x1
    old = cfg.fuse_bn_apply
    cfg.fuse_bn_apply = fuse
    try:
        torch.manual_seed(5)
        blk = block_cls(cin, planes, norm_layer=FusedBatchNorm2d).to(dev)
        use_direct_conv3x3(blk)
        use_direct_conv1x1(blk)
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, FusedBatchNorm2d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.uniform_(-0.3, 0.3)
        x = torch.randn(shape, device=dev).relu_().requires_grad_(True)
        y = blk(x)
        (y * torch.randn_like(y)).sum().backward()
        out = [y.detach(), x.grad] + [p.grad for p in blk.parameters()] + [b.clone() for b in blk.buffers()]
        return out
    finally:
        cfg.fuse_bn_apply = old


@pytest.mark.parametrize("case", [("BasicBlock", 48, 48, (2, 48, 32, 64)), ("BasicBlock", 96, 96, (2, 96, 16, 32)),
                                  ("BasicBlock", 384, 384, (2, 384, 8, 16)), ("Bottleneck", 256, 64, (2, 256, 16, 32))])
def test_residual_block_with_deferred_norm_is_bitwise_the_block_that_writes_it(dev, case):
    """A BasicBlock / Bottleneck with bn1 deferred into conv2 (default) against the same block with DCL_FUSE_BN_APPLY off: output,
    input gradient, every parameter gradient and every buffer (running statistics, counters) bitwise equal -- and the deferred path
    really ran (fused_bn.DEFERRED counts the norms that did not write their output)."""
    import importlib
    from mscs_amd.models import fused_bn
    H = importlib.import_module("mscs_amd.models.HRNet")
    name, cin, planes, shape = case
    n0 = fused_bn.DEFERRED["count"]
    a = _block_run(dev, getattr(H, name), cin, planes, shape, True)
    assert fused_bn.DEFERRED["count"] == n0 + 1
    b = _block_run(dev, getattr(H, name), cin, planes, shape, False)
    assert fused_bn.DEFERRED["count"] == n0 + 1
    assert len(a) == len(b)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def _module_run(dev, fuse):
    """One HighResolutionModule (three branches: two-step stride-2 chains in its fuse layers) and the stem of an HRNet, forward + backward."""
    import importlib
    from mscs_amd.debug import cfg
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    from mscs_amd.models.ops import use_direct_conv3x3, use_direct_conv1x1
    H = importlib.import_module("mscs_amd.models.HRNet")
    old = cfg.fuse_bn_apply
    cfg.fuse_bn_apply = fuse
    try:
        torch.manual_seed(9)
        mod = H.HighResolutionModule(3, H.BasicBlock, [1, 1, 1], [48, 96, 192], [48, 96, 192], "SUM", norm_layer=FusedBatchNorm2d).to(dev)
        stem = torch.nn.ModuleDict({"conv1": torch.nn.Conv2d(3, 64, 3, 2, 1, bias=False), "bn1": FusedBatchNorm2d(64),
                                    "conv2": torch.nn.Conv2d(64, 64, 3, 2, 1, bias=False), "bn2": FusedBatchNorm2d(64)}).to(dev)
        for m in (mod, stem):
            use_direct_conv3x3(m)
            use_direct_conv1x1(m)
        xs = [torch.randn(2, c, 64 >> k, 128 >> k, device=dev).relu_().requires_grad_(True) for k, c in enumerate((48, 96, 192))]
        ys = mod(list(xs))
        img = torch.randn(2, 3, 64, 128, device=dev)
        t = stem["conv1"](img)
        t = H.bn_act(stem["bn1"], t, defer=H._defers(stem["bn1"], t, stem["conv2"]))
        t = H.bn_act(stem["bn2"], stem["conv2"](t))
        (sum((y * torch.randn_like(y)).sum() for y in ys) + (t * torch.randn_like(t)).sum()).backward()
        torch.cuda.synchronize()
        return [y.detach() for y in ys] + [t.detach()] + [x.grad for x in xs] + [p.grad for m in (mod, stem) for p in m.parameters()] + \
               [b.clone() for m in (mod, stem) for b in m.buffers()]
    finally:
        cfg.fuse_bn_apply = old


def test_exchange_module_and_stem_with_deferred_norms_are_bitwise_the_ones_that_write_them(dev):
    """The stride-2 consumers: the two-step down-sampling chains of a three-branch exchange module's fuse layers (reference
    models/HRNet.py:236-258) and the stem's conv1 -> bn1 -> relu -> conv2 (:333-338), next to the module's BasicBlocks."""
    from mscs_amd.models import fused_bn
    n0 = fused_bn.DEFERRED["count"]
    a = _module_run(dev, True)
    assert fused_bn.DEFERRED["count"] == n0 + 3 + 1 + 1          # three BasicBlocks, the chain 0 -> 2, the stem
    b = _module_run(dev, False)
    assert len(a) == len(b)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_a_deferred_norm_output_is_refused_by_everything_but_its_consumer(dev):
    """The alias tensor of a deferred norm holds the norm's INPUT: a convolution that does not apply the map must raise, never read it."""
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    from mscs_amd.models.ops import DirectConv2d
    import copy
    bn = FusedBatchNorm2d(32).to(dev)
    bn_w = copy.deepcopy(bn)                # the same layer, writing its output (same running statistics = same pivot of the sums)
    z = torch.randn(2, 32, 8, 16, device=dev, requires_grad=True)
    y = bn(z, relu=True, defer=True)
    assert y.data_ptr() == z.data_ptr()
    conv1 = DirectConv2d(32, 32, 1, bias=False).to(dev)            # 1x1: no PRE form
    with pytest.raises(RuntimeError):
        conv1(y)
    with pytest.raises(RuntimeError):
        bn(z, relu=False, defer=True)
    conv3 = DirectConv2d(32, 32, 3, padding=1, bias=False).to(dev)
    assert conv3.fuses_input_norm(z)
    out = conv3(y)
    ref = conv3(bn_w(z, relu=True))
    assert torch.equal(out, ref)
    for a, b in zip(bn.buffers(), bn_w.buffers()):
        assert torch.equal(a, b)
