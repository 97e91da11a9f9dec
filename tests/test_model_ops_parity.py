"""GPU parity tests of the model-side operators: fused BatchNorm, bilinear resizes, the direct
split-f16 3x3 / 1x1 convolutions (forward, data gradient, weight gradient; every tile and variant), the head convolution over
up-sampled maps, the split-f16 GEMM, stream / schedule equivalences of the HRNet exchange modules -- through the C ABI of libdcl_hip.so
against float64 evaluations of the same operator (tolerances written per test) or bitwise against another formulation of this
package.  Split off tests/test_hip_parity.py (the loss) in round 6; the deferred norms have their own file (test_pre_norm_conv.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import mscs_amd  # noqa: F401
    from mscs_amd import _lib
    _lib.lib()                                   # fails loudly if libdcl_hip.so is missing
    return torch.device("cuda:0")


@pytest.mark.parametrize("relu,use_res,shape", [(True, True, (3, 48, 33, 47)), (True, False, (4, 18, 16, 24)),
                                                (False, False, (2, 720, 8, 12)), (False, True, (5, 7, 5, 3))])
def test_fused_batchnorm_matches_fp64_reference(dev, relu, use_res, shape):
    """csrc/dcl_bn.hip through FusedBatchNorm2d against nn.BatchNorm2d (+ add) (+ ReLU) evaluated on the
    CPU in float64.  (Not against the GPU library: PyTorch-ROCm 2.10's MIOpen batch-norm BACKWARD returns
    wrong dx / dweight when H*W is not a multiple of 4 -- 3e-3 / 6.4 absolute error at 3x48x33x47,
    tools/debug_bn.py -- while the fused kernels stay at 4e-7.)"""
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    torch.manual_seed(3)
    C = shape[1]
    ref = torch.nn.BatchNorm2d(C, momentum=0.1).double()
    fus = FusedBatchNorm2d(C, momentum=0.1).to(dev)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.uniform_(-0.5, 0.5)
        ref.running_mean.normal_(); ref.running_var.uniform_(0.5, 2.0)
    fus.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in ref.state_dict().items()})
    x = (torch.randn(shape) * 2 + 0.7)
    r = torch.randn(shape) if use_res else None
    gy = torch.randn(shape)
    xr = x.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if use_res else None
    y = ref(xr)
    if use_res:
        y = y + rr
    if relu:
        y = torch.relu(y)
    y.backward(gy.double())
    want = (y.detach(), xr.grad, rr.grad if use_res else None, ref.weight.grad, ref.bias.grad,
            ref.running_mean, ref.running_var, ref.num_batches_tracked)
    xf = x.to(dev).requires_grad_(True)
    rf = r.to(dev).requires_grad_(True) if use_res else None
    yf = fus(xf, residual=rf, relu=relu)
    yf.backward(gy.to(dev))
    got = (yf.detach(), xf.grad, rf.grad if use_res else None, fus.weight.grad, fus.bias.grad,
           fus.running_mean, fus.running_var, fus.num_batches_tracked)
    for a, b in zip(want, got):
        if a is None:
            continue
        scale = max(a.abs().max().item(), 1e-6)
        assert (a.double() - b.double().cpu()).abs().max().item() <= 2e-5 * scale + 1e-6, (relu, use_res, shape)
    fus.eval(); ref.eval()
    assert torch.allclose(fus(x.to(dev)).cpu().double(), ref(x.double()), atol=1e-5)


def test_fused_batchnorm_large_mean_small_std_at_benchmark_plane_size(dev):
    """|mean| >> std over 393k elements per channel (12 x 48 x 128 x 256, mean 50, std 0.1): plain f32 sums of x and x^2
    lose the variance to cancellation (E[x^2] - mean^2 = 2500.01 - 2500); the statistics kernel shifts its sums by the
    running mean.  Checked against float64 after the running mean has moved to the data (second step), and -- looser --
    on the very first step, where the pivot is still 0."""
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    torch.manual_seed(0)
    shape = (12, 48, 128, 256)
    x = (torch.randn(shape, device=dev) * 0.1 + 50.0)
    xd = x.double()
    mean = xd.mean((0, 2, 3))
    var = xd.var((0, 2, 3), unbiased=False)
    bn = FusedBatchNorm2d(48, momentum=1.0).to(dev)         # momentum 1: running mean = the batch mean after one step
    y1 = bn(x)
    want = (xd - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + bn.eps)
    err_first = (y1.double() - want).abs().max().item()
    y2 = bn(x)                                              # pivot = running mean ~ 50 now
    err = (y2.double() - want).abs().max().item()
    assert err <= 2e-3, err                                 # x itself carries 50 * 2^-24 = 3e-6 of rounding, / std 0.1
    assert err_first <= 0.5, err_first                      # un-shifted first step: finite, within the f32 sum error
    xg = x.clone().requires_grad_(True)
    bn2 = FusedBatchNorm2d(48, momentum=1.0).to(dev)
    with torch.no_grad():
        bn2.running_mean.copy_(mean.float())
    gy = torch.randn(shape, device=dev)
    bn2(xg).backward(gy)
    xr = xd.clone().requires_grad_(True)
    ref = torch.nn.functional.batch_norm(xr, None, None, torch.ones(48, device=dev, dtype=torch.float64),
                                         torch.zeros(48, device=dev, dtype=torch.float64), True, 0.0, bn.eps)
    ref.backward(gy.double())
    rel = (xg.grad.double() - xr.grad).abs().max().item() / xr.grad.abs().max().item()
    assert rel <= 2e-3, rel


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("shape,size", [((2, 5, 16, 32), (64, 128)), ((3, 7, 9, 13), (36, 52)), ((2, 3, 16, 32), (128, 256)),
                                        ((1, 4, 7, 5), (56, 44)), ((2, 2, 128, 256), (512, 1024)),
                                        ((1, 4, 8, 8), (30, 33)), ((2, 3, 1, 5), (4, 20)),
                                        ((2, 19, 32, 64), (128, 256))])
def test_upsample_bilinear_matches_torch(dev, align, shape, size):
    """csrc/dcl_resize.hip against F.interpolate(mode='bilinear') forward and backward (fp32, 1e-5)."""
    from mscs_amd.models.ops import upsample_bilinear
    torch.manual_seed(1)
    x = torch.randn(shape, device=dev)
    gy = torch.randn(shape[:2] + size, device=dev)
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    add_a = torch.randn_like(gy).requires_grad_(True)
    add_b = add_a.detach().clone().requires_grad_(True)
    ya = add_a + torch.nn.functional.interpolate(xa, size=size, mode="bilinear", align_corners=align)
    yb = upsample_bilinear(xb, size, align, add=add_b)
    ya.backward(gy)
    yb.backward(gy)
    assert (ya - yb).abs().max().item() <= 1e-5 * max(1.0, ya.abs().max().item())
    assert (xa.grad - xb.grad).abs().max().item() <= 1e-4 * max(1.0, xa.grad.abs().max().item())
    assert torch.equal(add_a.grad, add_b.grad)
    yc = upsample_bilinear(x, size, align)                       # without addend
    assert (yc - (ya - add_a).detach()).abs().max().item() <= 1e-5 * max(1.0, yc.abs().max().item())


def _conv_ref64(x, w, gy):
    x64 = x.double().cpu().requires_grad_(True)
    w64 = w.double().cpu().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, padding=1)
    y64.backward(gy.double().cpu())
    return y64.detach(), x64.grad, w64.grad


# (N, Cin, Cout, H, W): HRNet-W48 branch shapes in small, ragged tile edges (H % 16, W % 32 != 0), channel counts
# that are not multiples of 16 / 32 (padded weight tiles, ragged last octet), a 1-pixel-high image, hrnet18's C = 18
_DIRECT_SHAPES = [(2, 48, 48, 32, 64), (1, 96, 96, 16, 32), (3, 16, 32, 7, 40), (2, 40, 24, 19, 33),
                  (1, 18, 18, 9, 16), (2, 64, 64, 1, 8), (1, 192, 192, 8, 8), (2, 256, 48, 12, 24)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", _DIRECT_SHAPES)
def test_direct_conv3x3_forward_dgrad_match_fp64(dev, shape):
    """csrc/dcl_conv3x3.hip through the C-ABI (pack + convolution, forward and transposed weights) against a
    float64 convolution; tolerance 3e-6 of the output's max (fp32-equivalent: the f32 MIOpen convolution sits at
    3e-7 .. 1.2e-6 on the same inputs), every workgroup tile shape."""
    from mscs_amd.models import ops
    from mscs_amd.models.amax import amax_of
    n, ci, co, h, w = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(n, ci, h, w, device=dev).relu_() * 2.5
    wt = torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (9 * ci)) ** 0.5
    gy = torch.randn(n, co, h, w, device=dev) * 3e-5                   # gradients are small: exercises the scaling
    y64, gx64, _ = _conv_ref64(x, wt, gy)
    y = ops.conv3x3_direct(x, wt)
    gx = ops.conv3x3_direct(gy, wt, transposed=True)
    assert ((y.double().cpu() - y64).abs().max() / y64.abs().max()).item() < 3e-6
    assert ((gx.double().cpu() - gx64).abs().max() / gx64.abs().max()).item() < 3e-6
    # every tile configuration computes the same convolution (different accumulation order inside the MFMA only)
    wamax, xamax = amax_of(wt), amax_of(x)
    wp = ops.conv3x3_pack(wt, wamax)
    for r in (1, 2, 3):
        for p in (1, 2, 4):
            out = torch.full_like(y, float("nan"))
            ops.conv3x3_launch(x, wp, co, xamax, wamax, out, r, p)
            assert ((out.double().cpu() - y64).abs().max() / y64.abs().max()).item() < 3e-6, (r, p)


@pytest.mark.gpu
@pytest.mark.parametrize("interleave", [1, 0, 2])
def test_direct_conv3x3_every_tile_matches_fp64(dev, interleave):
    """Every (channel tiles, pixel tiles) instantiation of the stride-1 kernel -- with the staging interleaved among the
    MFMAs (k_conv3x3_il, the default for Cin % 16 == 0: one and two register sets, micro-operations placed by MFMA index)
    and in fenced blocks (k_conv3x3), and with the (2, 2) tile's waves split 2 x 2 over rows and channel tiles (mode 2) -- on shapes with ragged tiles, one to five K chunks and images smaller than a tile,
    against float64 (3e-6 of max); forward and, through the transposed fragments, the data gradient."""
    import torch.nn.functional as F
    from mscs_amd import _lib
    from mscs_amd.models import ops
    from mscs_amd.models.amax import amax_of
    L = _lib.lib()
    torch.manual_seed(11)
    try:
        L.dcl_conv3x3_set_interleave(interleave)
        for (n, ci, co, h, w) in [(2, 16, 32, 9, 40), (1, 48, 96, 20, 33), (2, 80, 48, 5, 7), (1, 32, 64, 33, 64)]:
            x = torch.randn(n, ci, h, w, device=dev).relu_()
            wt = torch.randn(co, ci, 3, 3, device=dev) * 0.1
            ref = F.conv2d(x.double(), wt.double(), padding=1)
            sx, sw = amax_of(x), amax_of(wt)
            wp = ops.conv3x3_pack(wt, sw)
            for r in (1, 2, 3):
                for p in (1, 2, 4):
                    out = torch.full((n, co, h, w), float("nan"), device=dev)
                    ops.conv3x3_launch(x, wp, co, sx, sw, out, r, p)
                    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
                    assert err < 3e-6, (interleave, n, ci, co, h, w, r, p, err)
            gy = torch.randn(n, co, h, w, device=dev) * 1e-3
            gref = F.conv_transpose2d(gy.double(), wt.double(), padding=1)
            gx = ops.conv3x3_direct(gy, wt, transposed=True)
            assert ((gx.double() - gref).abs().max() / gref.abs().max()).item() < 3e-6
    finally:
        L.dcl_conv3x3_set_interleave(2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [s for s in _DIRECT_SHAPES if s[1] % 16 == 0 and s[2] % 16 == 0 and s[4] % 8 == 0])
def test_direct_conv3x3_wgrad_matches_fp64(dev, shape):
    """csrc/dcl_wgrad3x3.hip against the float64 weight gradient (3e-6 of max), and bitwise run-to-run
    reproducibility (slab reduction in fixed order, no float atomics)."""
    from mscs_amd.models import ops
    n, ci, co, h, w = shape
    torch.manual_seed(sum(shape) + 1)
    x = torch.randn(n, ci, h, w, device=dev).relu_() * 2.5
    wt = torch.randn(co, ci, 3, 3, device=dev)
    gy = torch.randn(n, co, h, w, device=dev) * 3e-5
    _, _, gw64 = _conv_ref64(x, wt, gy)
    gw = ops.conv3x3_wgrad(x, gy)
    assert ((gw.double().cpu() - gw64).abs().max() / gw64.abs().max()).item() < 3e-6
    assert torch.equal(gw, ops.conv3x3_wgrad(x, gy))


@pytest.mark.gpu
def test_direct_conv_module_autograd_and_absmax_tags(dev):
    """DirectConv2d inside conv -> fused BN -> conv: gradients match the nn.Conv2d / nn.BatchNorm2d graph in
    float64; the BN outputs carry absmax tags that bound the tensors they describe, and a tag is dropped once its
    tensor is modified in place."""
    from mscs_amd.models import ops
    from mscs_amd.models.amax import amax_of
    from mscs_amd.models.fused_bn import FusedBatchNorm2d, bn_act
    torch.manual_seed(3)
    c1, c2 = torch.nn.Conv2d(32, 48, 3, padding=1, bias=False).to(dev), torch.nn.Conv2d(48, 32, 3, padding=1, bias=False).to(dev)
    bn = FusedBatchNorm2d(48).to(dev)
    ops.use_direct_conv3x3(c1), ops.use_direct_conv3x3(c2)
    assert isinstance(c1, ops.DirectConv2d) and isinstance(c2, ops.DirectConv2d)
    x = torch.randn(3, 32, 12, 40, device=dev, requires_grad=True)
    mid = bn_act(bn, c1(x))
    tag = mid._dcl_amax
    assert tag[1].numel() == 64 and abs(tag[1].max().item() - mid.abs().max().item()) < 1e-6
    out = c2(mid)
    out.square().mean().backward()
    # float64 reference of the same graph
    r1, r2 = torch.nn.Conv2d(32, 48, 3, padding=1, bias=False).double(), torch.nn.Conv2d(48, 32, 3, padding=1, bias=False).double()
    rb = torch.nn.BatchNorm2d(48).double()
    r1.weight.data.copy_(c1.weight.detach().cpu()); r2.weight.data.copy_(c2.weight.detach().cpu())
    x64 = x.detach().double().cpu().requires_grad_(True)
    o64 = r2(torch.relu(rb(r1(x64))))
    o64.square().mean().backward()
    for got, want in ((out, o64), (x.grad, x64.grad), (c1.weight.grad, r1.weight.grad), (c2.weight.grad, r2.weight.grad)):
        assert ((got.detach().double().cpu() - want.detach()).abs().max() / want.detach().abs().max()).item() < 2e-5
    # stale tags are ignored
    t = torch.randn(1, 16, 8, 8, device=dev)
    a0 = amax_of(t)
    t.mul_(4.0)
    a1 = amax_of(t)
    assert a1 is not a0 and abs(a1.max().item() - t.abs().max().item()) < 1e-6


@pytest.mark.gpu
def test_basicblock_residual_gradient_fused_into_dgrad(dev):
    """BasicBlock with the direct convolutions: the residual's gradient reaches the block input through the
    GradToken (added in the epilogue of conv1's data-gradient kernel); input and parameter gradients match the
    plain nn.Conv2d / nn.BatchNorm2d block in float64."""
    from mscs_amd.models import ops
    from mscs_amd.models.HRNet import BasicBlock
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    torch.manual_seed(11)
    blk = BasicBlock(32, 32, norm_layer=FusedBatchNorm2d).to(dev).train()
    ops.use_direct_conv3x3(blk)
    ref = BasicBlock(32, 32, norm_layer=torch.nn.BatchNorm2d).double().train()
    ref.load_state_dict({k: v.double().cpu() for k, v in blk.state_dict().items()})
    x = torch.randn(2, 32, 10, 24, device=dev)
    xin = (x * 1.0).requires_grad_(True)            # non-leaf input like inside the network
    xin.retain_grad()
    y = blk(xin)
    y.square().mean().backward()
    x64 = x.double().cpu().requires_grad_(True)
    y64 = ref(x64)
    y64.square().mean().backward()
    assert ((y.detach().double().cpu() - y64.detach()).abs().max() / y64.detach().abs().max()).item() < 1e-5
    assert ((xin.grad.double().cpu() - x64.grad).abs().max() / x64.grad.abs().max()).item() < 2e-5
    for (n1, p1), (n2, p2) in zip(blk.named_parameters(), ref.named_parameters()):
        assert ((p1.grad.double().cpu() - p2.grad).abs().max() / (p2.grad.abs().max() + 1e-30)).item() < 5e-5, n1


@pytest.mark.gpu
def test_branch_streams_are_bitwise_equivalent(dev):
    """HighResolutionModule branches on one HIP stream each (default) against single-stream execution.  The branches
    consist of our deterministic kernels only (direct convolutions, fused BN), so outputs, input gradients and
    parameter gradients must agree BITWISE; a missing stream dependency or allocator hazard would show up here."""
    import importlib
    hm = importlib.import_module("mscs_amd.models.HRNet")
    graph = {"backbone": "hrnet48", "pretrained": False, "dataset": "CITYSCAPES", "align_corners": True}
    torch.manual_seed(7)
    mod = hm.HRNet(graph, 1).backbone.stage4[1].to(dev).train()
    assert mod.num_branches == 4
    xs0 = [torch.randn(3, 48 * 2 ** i, 64 // 2 ** i, 96 // 2 ** i, device=dev) for i in range(4)]

    def run(flag):
        hm._BRANCH_STREAMS = flag
        mod.zero_grad(set_to_none=True)
        state = {k: v.clone() for k, v in mod.state_dict().items()}
        xs = [(x * 1.0).requires_grad_(True) for x in xs0]
        for x in xs:
            x.retain_grad()
        outs = mod._run_branches(list(xs))
        sum(o.square().mean() for o in outs).backward()
        torch.cuda.synchronize()
        res = ([o.detach().clone() for o in outs], [x.grad.clone() for x in xs],
               {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None})
        mod.load_state_dict(state)
        return res
    try:
        ref = run(False)
        for trial in range(3):
            got = run(True)
            for a, b in zip(got[0] + got[1], ref[0] + ref[1]):
                assert torch.equal(a, b)
            assert got[2].keys() == ref[2].keys() and len(ref[2]) == 4 * 4 * 6
            for n in ref[2]:
                assert torch.equal(got[2][n], ref[2][n]), n
    finally:
        hm._BRANCH_STREAMS = True




@pytest.mark.gpu
@pytest.mark.parametrize("where", ["start", "end"])
def test_kernels_stay_inside_their_tensors(dev, where):
    """Out-of-bounds guard (no GPU sanitizer on this pool): operands are placed at the very start / very end of a
    fresh 64 MiB device allocation, so a read before the first or past the last element of a tensor leaves the
    allocation (and faults when the neighbourhood is unmapped, which is how the weight-gradient halo bug was found).
    Ragged shapes exercise every clamped-address path of the convolution, BN and resize kernels."""
    from mscs_amd.models import ops
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    torch.cuda.empty_cache()

    def place(t):
        buf = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device=dev)       # its own 64 MiB segment
        n = t.numel()
        view = buf[:n] if where == "start" else buf[buf.numel() - n:]
        view.copy_(t.reshape(-1))
        return view.view(t.shape)

    torch.manual_seed(2)
    for (n, ci, co, h, w) in [(1, 16, 16, 1, 8), (2, 40, 24, 19, 40), (1, 48, 48, 5, 8), (2, 32, 64, 3, 24)]:
        x = place(torch.randn(n, ci, h, w, device=dev))
        wt = place(torch.randn(co, ci, 3, 3, device=dev))
        gy = place(torch.randn(n, co, h, w, device=dev))
        y = ops.conv3x3_direct(x, wt)
        gx = ops.conv3x3_direct(gy, wt, transposed=True)
        ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1)
        assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6
        assert torch.isfinite(gx).all()
        if ci % 16 == 0 and co % 16 == 0:
            gw = ops.conv3x3_wgrad(x, gy)
            gw_ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), wt.double(), None, [1, 1], [1, 1],
                                                         [1, 1], False, [0, 0], 1, [False, True, False])[1]
            assert ((gw.double() - gw_ref).abs().max() / gw_ref.abs().max()).item() < 3e-6
        bn = FusedBatchNorm2d(ci).to(dev).train()
        xb = place(torch.randn(n, ci, h, w, device=dev)).requires_grad_(True)
        res = place(torch.randn(n, ci, h, w, device=dev))
        out = bn(xb, residual=res, relu=True)
        out.backward(place(torch.randn(n, ci, h, w, device=dev)))
        assert torch.isfinite(out).all() and torch.isfinite(xb.grad).all()
        up_in = place(torch.randn(n, ci, h, w, device=dev)).requires_grad_(True)
        up = ops.upsample_bilinear(up_in, (2 * h + 1, 3 * w), True)
        up.backward(place(torch.randn_like(up)))
        assert torch.isfinite(up).all() and torch.isfinite(up_in.grad).all()
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_grouped_syncbn_schedule_is_bitwise_the_free_running_one(dev):
    """The depth-major schedule of an exchange module's branches (one stacked SyncBatchNorm statistics exchange per block
    depth and direction, models/fused_bn.bn_act_group) runs the same kernels on the same data as the free-running
    per-branch schedule: outputs, input gradients and parameter gradients bitwise equal (one rank: the exchange itself is
    skipped; two ranks: tests/test_gpu_multiproc.py)."""
    import importlib
    from mscs_amd.models import fused_bn
    from mscs_amd.models.ops import use_direct_conv1x1, use_direct_conv3x3
    hm = importlib.import_module("mscs_amd.models.HRNet")
    torch.manual_seed(5)
    ch = [48, 96, 192, 384]
    mod = hm.HighResolutionModule(4, hm.BasicBlock, [4] * 4, ch, ch, 'SUM', True, norm_layer=fused_bn.FusedBatchNorm2d)
    use_direct_conv3x3(mod)
    use_direct_conv1x1(mod)
    mod.to(dev).train()
    state = {k: v.clone() for k, v in mod.state_dict().items()}
    xs0 = [torch.randn(2, c, 64 >> i, 96 >> i, device=dev) for i, c in enumerate(ch)]
    res = []
    for grouped in (False, True):
        mod.load_state_dict(state)
        mod.zero_grad(set_to_none=True)
        xs = [x.clone().requires_grad_(True) for x in xs0]
        fused_bn.FORCE_GROUP = grouped
        try:
            assert mod._groupable(xs) == grouped
            outs = mod(list(xs))
            sum((o * torch.cos(torch.arange(o.numel(), device=dev).view(o.shape) * 0.37)).mean() for o in outs).backward()
        finally:
            fused_bn.FORCE_GROUP = False
        torch.cuda.synchronize()
        res.append(([o.detach().clone() for o in outs], [x.grad.clone() for x in xs],
                    [p.grad.clone() for p in mod.parameters()], [b.clone() for b in mod.buffers()]))
    for a, b in zip(res[0], res[1]):
        for t, u in zip(a, b):
            assert torch.equal(t, u)


@pytest.mark.gpu
def test_fuse_layer_streams_match_single_stream(dev):
    """A whole HighResolutionModule (branches + fuse layers, one stream per branch / per fused output) against its
    single-stream execution, to within 10x the measured run-to-run noise of the single-stream run -- which is ZERO now that every
    kernel of the module is deterministic: the two schedules agree bitwise."""
    import importlib
    hm = importlib.import_module("mscs_amd.models.HRNet")
    graph = {"backbone": "hrnet48", "pretrained": False, "dataset": "CITYSCAPES", "align_corners": True}
    torch.manual_seed(9)
    mod = hm.HRNet(graph, 1).backbone.stage4[0].to(dev).train()
    xs0 = [torch.randn(4, 48 * 2 ** i, 64 // 2 ** i, 128 // 2 ** i, device=dev) for i in range(4)]

    def run(flag):
        hm._BRANCH_STREAMS = flag
        mod.zero_grad(set_to_none=True)
        state = {k: v.clone() for k, v in mod.state_dict().items()}
        xs = [(x * 1.0).requires_grad_(True) for x in xs0]
        for x in xs:
            x.retain_grad()
        outs = mod(list(xs))
        sum(o.square().mean() for o in outs).backward()
        torch.cuda.synchronize()
        res = ([o.detach().clone() for o in outs], [x.grad.clone() for x in xs],
               {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None})
        mod.load_state_dict(state)
        return res
    def dist(a, b):
        return (a - b).abs().max().item() / (b.abs().max().item() + 1e-20)
    try:
        ref = run(False)
        ref2 = run(False)
        # run-to-run noise of the library kernels in the fuse layers (0 when they happen to be reproducible)
        noise_f = max([dist(a, b) for a, b in zip(ref2[0], ref[0])] + [0.0])
        noise_g = max([dist(a, b) for a, b in zip(ref2[1], ref[1])] + [dist(ref2[2][n], ref[2][n]) for n in ref[2]]
                      + [0.0])
        # (no floors under the noise since round 5: with every kernel of the module deterministic the two schedules must agree
        # BITWISE -- the former floors of 1e-6 / 1e-5 were wide enough to hide the packed-FP32 fault of DESIGN.md section 7)
        for trial in range(2):
            got = run(True)
            for a, b in zip(got[0], ref[0]):
                assert dist(a, b) <= 10 * noise_f, (dist(a, b), noise_f)
            for a, b in zip(got[1], ref[1]):
                assert dist(a, b) <= 10 * noise_g, (dist(a, b), noise_g)
            for n in ref[2]:
                assert dist(got[2][n], ref[2][n]) <= 10 * noise_g, (n, dist(got[2][n], ref[2][n]), noise_g)
    finally:
        hm._BRANCH_STREAMS = True


@pytest.mark.gpu
def test_stage_without_joins_between_modules_matches_joined(dev):
    """A whole stage (three HighResolutionModules in sequence): with fused output i left on stream i for the next
    module's branch i (no join / fork between modules, models/HRNet.py `_DEFER_JOIN`) against the joined schedule, to
    within the run-to-run noise of the joined one -- repeated, so that a missing dependency would show."""
    import importlib
    hm = importlib.import_module("mscs_amd.models.HRNet")
    graph = {"backbone": "hrnet48", "pretrained": False, "dataset": "CITYSCAPES", "align_corners": True}
    torch.manual_seed(10)
    stage = hm.HRNet(graph, 1).backbone.stage4.to(dev).train()
    assert [m.join_output for m in stage] == [False, False, True]
    xs0 = [torch.randn(3, 48 * 2 ** i, 64 // 2 ** i, 96 // 2 ** i, device=dev) for i in range(4)]

    def run(flag):
        hm._DEFER_JOIN = flag
        stage.zero_grad(set_to_none=True)
        state = {k: v.clone() for k, v in stage.state_dict().items()}
        xs = [(x * 1.0).requires_grad_(True) for x in xs0]
        outs = stage(list(xs))
        sum(o.square().mean() for o in outs).backward()
        torch.cuda.synchronize()
        res = ([o.detach().clone() for o in outs], [x.grad.clone() for x in xs],
               {n: p.grad.clone() for n, p in stage.named_parameters() if p.grad is not None})
        stage.load_state_dict(state)
        return res
    def dist(a, b):
        return (a - b).abs().max().item() / (b.abs().max().item() + 1e-20)
    try:
        ref, ref2 = run(False), run(False)
        noise = max([dist(a, b) for a, b in zip(ref2[0] + ref2[1], ref[0] + ref[1])]
                    + [dist(ref2[2][n], ref[2][n]) for n in ref[2]] + [0.0])
        for trial in range(3):
            got = run(True)
            for a, b in zip(got[0] + got[1], ref[0] + ref[1]):
                assert dist(a, b) <= 10 * noise, (dist(a, b), noise)
            for n in ref[2]:
                assert dist(got[2][n], ref[2][n]) <= 10 * noise, (n, dist(got[2][n], ref[2][n]), noise)
    finally:
        hm._DEFER_JOIN = True


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 48, 96, 32, 64), (1, 16, 32, 7, 40), (2, 32, 16, 9, 8), (1, 64, 64, 16, 24),
                                   (2, 3, 64, 12, 16), (1, 16, 32, 7, 48), (2, 32, 48, 9, 32), (1, 48, 16, 5, 16),
                                   (3, 96, 32, 6, 80)])
def test_direct_conv3x3_stride2_matches_fp64(dev, shape):
    """Stride-2 convolution on the direct kernels: forward (stride-2 tile), data gradient (stride-1 kernel over the
    zero-inserted gradient) and weight gradient (GEMM over the output pixels, csrc/dcl_wgrad3x3_s2.hip, when W % 16 == 0;
    else the stride-1 kernel on a zero-inserted dy) against float64, odd sizes included."""
    from mscs_amd.models import ops
    n, ci, co, h, w = shape
    torch.manual_seed(sum(shape) + 5)
    x = torch.randn(n, ci, h, w, device=dev).relu_() * 2.0
    wt = torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (9 * ci)) ** 0.5
    x64 = x.double().cpu().requires_grad_(True)
    w64 = wt.double().cpu().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, stride=2, padding=1)
    gy = torch.randn(y64.shape, device=dev) * 1e-4
    y64.backward(gy.double().cpu())
    y = ops.conv3x3_direct(x, wt, stride=2)
    assert y.shape == y64.shape
    assert ((y.double().cpu() - y64.detach()).abs().max() / y64.detach().abs().max()).item() < 3e-6
    gx = ops.conv3x3_direct(gy, wt, transposed=True, stride=2, out_hw=(h, w))
    assert ((gx.double().cpu() - x64.grad).abs().max() / x64.grad.abs().max()).item() < 3e-6
    if ci % 16 == 0 and co % 16 == 0 and w % 8 == 0:
        gw = ops.conv3x3_wgrad(x, gy, stride=2)
        assert ((gw.double().cpu() - w64.grad).abs().max() / w64.grad.abs().max()).item() < 3e-6
    # the module path (autograd) agrees as well
    conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=False).to(dev)
    conv.weight.data.copy_(wt)
    ops.use_direct_conv3x3(conv)
    assert isinstance(conv, ops.DirectConv2d)
    xi = (x * 1.0).requires_grad_(True)
    xi.retain_grad()
    conv(xi).backward(gy)
    assert ((xi.grad.double().cpu() - x64.grad).abs().max() / x64.grad.abs().max()).item() < 3e-6
    assert ((conv.weight.grad.double().cpu() - w64.grad).abs().max() / w64.grad.abs().max()).item() < 3e-6


@pytest.mark.gpu
def test_direct_conv_with_bias_matches_fp64(dev):
    """DirectConv2d with a bias (HRNet's head convolution 720 -> 720 has one): y, dx, dW, db against float64."""
    from mscs_amd.models import ops
    torch.manual_seed(21)
    conv = torch.nn.Conv2d(48, 80, 3, padding=1).to(dev)
    ops.use_direct_conv3x3(conv)
    assert isinstance(conv, ops.DirectConv2d) and conv.bias is not None
    x = (torch.randn(2, 48, 20, 40, device=dev) * 2 + 0.3).requires_grad_(True)
    gy = torch.randn(2, 80, 20, 40, device=dev) * 1e-3
    y = conv(x)
    y.backward(gy)
    ref = torch.nn.Conv2d(48, 80, 3, padding=1).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in conv.state_dict().items()})
    x64 = x.detach().double().cpu().requires_grad_(True)
    y64 = ref(x64)
    y64.backward(gy.double().cpu())
    for got, want in ((y, y64), (x.grad, x64.grad), (conv.weight.grad, ref.weight.grad), (conv.bias.grad, ref.bias.grad)):
        assert ((got.detach().double().cpu() - want.detach()).abs().max() / want.detach().abs().max()).item() < 3e-6


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["hrnet", "upernet"])
def test_models_with_direct_kernels_match_library_kernels(dev, family):
    """Whole-model check of the fp32-equivalence claim: the same weights through the direct split-f16 convolution
    kernels and through the library (MIOpen f32) give the same logits in eval mode (1e-4 of max, the tolerance of
    the reference goldens in tests/test_models.py) and the same training-mode logits (batch statistics)."""
    import mscs_amd.models as M
    torch.manual_seed(4)
    if family == "hrnet":
        g = {"backbone": "hrnet48", "pretrained": False, "dataset": "CITYSCAPES", "align_corners": True}
        a = M.HRNet(dict(g), 1).to(dev)
        b = M.HRNet(dict(g, branch_conv="library", head_conv="library", fused_bn=False, conv1x1="library"), 1).to(dev)
    else:
        g = {"backbone": "swinT", "pretrained": False, "dataset": "ADE20K", "align_corners": False,
             "fpn_channels": 128}
        a = M.UPerNet(dict(g), 1).to(dev)
        b = M.UPerNet(dict(g, direct_conv=False), 1).to(dev)
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 3, 128, 256, device=dev)
    # (Swin's stochastic depth makes two training-mode forwards incomparable: eval only for UPerNet)
    for mode in (("eval", "train") if family == "hrnet" else ("eval",)):
        getattr(a, mode)(), getattr(b, mode)()
        with torch.no_grad():
            ya, yb = a(x), b(x)
        ya = ya[0] if isinstance(ya, (tuple, list)) else ya
        yb = yb[0] if isinstance(yb, (tuple, list)) else yb
        assert ((ya - yb).abs().max() / yb.abs().max()).item() < 1e-4, (family, mode)


@pytest.mark.gpu
def test_fused_bn_packed_relu_mask_is_bitwise_equivalent(dev):
    """norm + residual + ReLU: the backward reads the packed sign bits the forward wrote (1/32 of y) instead of y; the
    mask bit IS y > 0, so every gradient is bitwise the one of the y-reading path.  HW % 256 != 0 keeps reading y."""
    import mscs_amd.models.fused_bn as fb
    torch.manual_seed(33)
    for shape in ((2, 48, 16, 32), (3, 20, 32, 64), (2, 16, 10, 12)):
        outs = []
        for packed in (True, False):
            fb._PACKED_RELU_MASK = packed
            try:
                bn = fb.FusedBatchNorm2d(shape[1]).to(dev).train()
                with torch.no_grad():
                    bn.weight.copy_(torch.linspace(0.5, 1.5, shape[1]))
                    bn.bias.copy_(torch.linspace(-0.3, 0.3, shape[1]))
                g = torch.Generator(device=dev).manual_seed(5)
                x = torch.randn(shape, device=dev, generator=g).requires_grad_(True)
                r = torch.randn(shape, device=dev, generator=g).requires_grad_(True)
                gy = torch.randn(shape, device=dev, generator=g)
                y = bn(x, residual=r, relu=True)
                y.backward(gy)
                outs.append((y.detach(), x.grad, r.grad, bn.weight.grad, bn.bias.grad))
            finally:
                fb._PACKED_RELU_MASK = True
        for a, b in zip(*outs):
            assert torch.equal(a, b), shape


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(12, 64, 256, 16, 64), (2, 256, 64, 9, 32), (3, 48, 48, 17, 24), (1, 96, 256, 5, 8),
                                   (2, 384, 48, 8, 16), (2, 720, 19, 6, 40), (1, 40, 72, 7, 23), (2, 192, 192, 4, 64)])
def test_direct_conv1x1_matches_fp64(dev, shape):
    """1x1 convolution on the one-tap mode of the direct kernel (forward, data gradient) and k_wgrad1x1d (weight
    gradient) against float64 (3e-6 of max); the weight gradient is bitwise reproducible."""
    from mscs_amd.models import ops
    n, ci, co, h, w = shape
    torch.manual_seed(sum(shape) + 3)
    x = torch.randn(n, ci, h, w, device=dev).relu_() * 1.5
    wt = torch.randn(co, ci, 1, 1, device=dev) * (2.0 / ci) ** 0.5
    gy = torch.randn(n, co, h, w, device=dev) * 2e-4
    x64, w64, g64 = x.double().cpu(), wt.double().cpu(), gy.double().cpu()
    y64 = torch.nn.functional.conv2d(x64, w64)
    gx64 = torch.nn.functional.conv_transpose2d(g64, w64)
    gw64 = torch.einsum("nohw,nihw->oi", g64, x64).view(co, ci, 1, 1)
    y = ops.conv1x1_direct(x, wt)
    gx = ops.conv1x1_direct(gy, wt, transposed=True)
    assert y.shape == y64.shape and gx.shape == gx64.shape
    assert ((y.double().cpu() - y64).abs().max() / y64.abs().max()).item() < 3e-6
    assert ((gx.double().cpu() - gx64).abs().max() / gx64.abs().max()).item() < 3e-6
    if ops.conv1x1_wgrad_supported(x, co):
        gw = ops.conv1x1_wgrad(x, gy)
        assert ((gw.double().cpu() - gw64).abs().max() / gw64.abs().max()).item() < 3e-6
        assert torch.equal(gw, ops.conv1x1_wgrad(x, gy))
    # every tile configuration computes the same convolution
    from mscs_amd.models.amax import amax_of
    wamax, xamax = amax_of(wt), amax_of(x)
    wp = ops.conv3x3_pack(wt, wamax)
    for r in (1, 2, 3):
        for p_ in (1, 2, 4):
            out = torch.full_like(y, float("nan"))
            ops.conv1x1_launch(x, wp, co, xamax, wamax, out, r, p_)
            assert ((out.double().cpu() - y64).abs().max() / y64.abs().max()).item() < 3e-6, (r, p_)


@pytest.mark.gpu
def test_direct_conv1x1_module_autograd(dev):
    """DirectConv2d for a 1x1 nn.Conv2d (with and without bias, channel counts with and without the weight-gradient
    kernel's multiple-of-16 requirement): y, dx, dW, db against float64."""
    from mscs_amd.models import ops
    torch.manual_seed(21)
    for (ci, co, bias) in ((64, 256, False), (96, 40, True), (720, 19, True)):
        conv = torch.nn.Conv2d(ci, co, 1, bias=bias).to(dev)
        ops.use_direct_conv1x1(conv)
        assert isinstance(conv, ops.DirectConv2d)
        x = (torch.randn(3, ci, 12, 24, device=dev)).requires_grad_(True)
        gy = torch.randn(3, co, 12, 24, device=dev)
        conv(x).backward(gy)
        ref = torch.nn.Conv2d(ci, co, 1, bias=bias).double()
        ref.load_state_dict({k: v.double().cpu() for k, v in conv.state_dict().items()})
        x64 = x.detach().double().cpu().requires_grad_(True)
        ref(x64).backward(gy.double().cpu())
        pairs = [(conv(x), ref(x64)), (x.grad, x64.grad), (conv.weight.grad, ref.weight.grad)]
        if bias:
            pairs.append((conv.bias.grad, ref.bias.grad))
        for got, want in pairs:
            assert ((got.detach().double().cpu() - want.detach()).abs().max() / want.detach().abs().max()).item() < 3e-6


@pytest.mark.gpu
def test_gemm_conv1x1_matches_library(dev):
    """GemmConv1x1 (1x1 convolution as batched GEMMs) against nn.Conv2d: y, dx, dW, db to fp32 round-off."""
    from mscs_amd.models import ops
    torch.manual_seed(8)
    a = torch.nn.Conv2d(96, 40, 1).to(dev)
    b = torch.nn.Conv2d(96, 40, 1).to(dev)
    b.load_state_dict(a.state_dict())
    ops.use_gemm_conv1x1(a)
    assert isinstance(a, ops.GemmConv1x1)
    x1 = torch.randn(3, 96, 17, 23, device=dev, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    gy = torch.randn(3, 40, 17, 23, device=dev)
    a(x1).backward(gy)
    b(x2).backward(gy)
    for got, want in ((a(x1), b(x2)), (x1.grad, x2.grad), (a.weight.grad, b.weight.grad), (a.bias.grad, b.bias.grad)):
        assert ((got - want).abs().max() / want.abs().max()).item() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("cls", ["direct", "gemm"])
def test_wide_conv1x1_on_split_f16_gemm_matches_fp64(dev, cls):
    """1x1 convolutions whose produced channel count fills a GEMM tile (>= 192: HRNet layer1 64 -> 256 / 256 -> 64, the
    projector's 192 -> 256, the UPerNet decoder's C -> 512 laterals) run as one batched dcl_gemm_f16x3 over the images --
    DirectConv2d: forward / data gradient; GemmConv1x1: all three directions -- against float64 (3e-6 of max), bias included."""
    from mscs_amd.models import ops
    torch.manual_seed(23)
    for (ci, co, bias, h, w) in ((64, 256, False, 12, 24), (256, 64, False, 12, 24), (192, 256, True, 8, 16), (96, 512, True, 16, 32),
                                 (768, 512, False, 8, 8)):
        conv = torch.nn.Conv2d(ci, co, 1, bias=bias).to(dev)
        (ops.use_direct_conv1x1 if cls == "direct" else ops.use_gemm_conv1x1)(conv)
        x = torch.randn(3, ci, h, w, device=dev).requires_grad_(True)
        gy = torch.randn(3, co, h, w, device=dev) * 1e-3
        assert ops._conv1x1_by_gemm(co, ci, x, False) or ops._conv1x1_by_gemm(ci, co, gy, True)
        conv(x).backward(gy)
        ref = torch.nn.Conv2d(ci, co, 1, bias=bias).double()
        ref.load_state_dict({k: v.double().cpu() for k, v in conv.state_dict().items()})
        x64 = x.detach().double().cpu().requires_grad_(True)
        ref(x64).backward(gy.double().cpu())
        pairs = [(conv(x), ref(x64)), (x.grad, x64.grad), (conv.weight.grad, ref.weight.grad)]
        if bias:
            pairs.append((conv.bias.grad, ref.bias.grad))
        for k, (got, want) in enumerate(pairs):
            err = ((got.detach().double().cpu() - want.detach()).abs().max() / want.detach().abs().max()).item()
            assert err < 3e-6, (cls, ci, co, k, err)


@pytest.mark.gpu
@pytest.mark.parametrize("align", [True, False])
def test_upsample_concat_matches_cat_of_interpolates(dev, align):
    """upsample_concat (every map written straight into its channel slice, gradient read in place) against
    torch.cat of F.interpolate, forward and all input gradients."""
    from mscs_amd.models import ops
    torch.manual_seed(17)
    shapes = [(2, 48, 32, 64), (2, 96, 16, 32), (2, 40, 8, 16), (2, 24, 5, 7)]
    a = [torch.randn(s, device=dev, requires_grad=True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    y = ops.upsample_concat(a, align)
    ref = torch.cat([b[0]] + [torch.nn.functional.interpolate(t, size=(32, 64), mode="bilinear", align_corners=align)
                              for t in b[1:]], 1)
    assert y.shape == ref.shape and (y - ref).abs().max().item() < 1e-5
    gy = torch.randn_like(ref)
    y.backward(gy)
    ref.backward(gy)
    for t, r in zip(a, b):
        assert (t.grad - r.grad).abs().max().item() < 1e-4 * max(1.0, r.grad.abs().max().item())





@pytest.mark.gpu
def test_fan_out_sums_consumer_gradients_in_one_kernel(dev):
    """ops.fan_out: k aliases of a tensor, backward = ONE fused sum of the k incoming gradients (None entries skipped),
    bitwise equal to the left-to-right sum; the absmax tag travels with the aliases."""
    from mscs_amd.models import ops, amax as _amax
    torch.manual_seed(12)
    for k in (3, 4, 6):
        x = torch.randn(2, 5, 7, 9, device=dev, requires_grad=True)
        _amax.tag(x, x.detach().abs().amax(dim=(2, 3)).flatten().contiguous())
        outs = ops.fan_out(x, k)
        assert len(outs) == k and all(o.data_ptr() == x.data_ptr() for o in outs)
        assert all(_amax.amax_of(o) is _amax.amax_of(x) for o in outs)
        ws = [torch.randn_like(x) for _ in range(k)]
        used = [i for i in range(k) if i != 1]                          # consumer 1 contributes no gradient
        sum((outs[i] * ws[i]).sum() for i in used).backward()
        ref = ws[used[0]].clone()
        for i in used[1:]:
            ref = ref + ws[i]
        assert (x.grad - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("chans,H,W,Co,bias", [((16, 32, 48, 64), 32, 64, 48, True), ((16, 16, 32, 32), 24, 40, 32, False),
                                               ((48, 96, 192, 384), 64, 128, 96, True),
                                               ((32, 64, 96, 128), 64, 64, 80, False)])
def test_head_conv_over_upsampled_matches_fp64(dev, align, chans, H, W, Co, bias):
    """ops.conv3x3_over_upsampled (the channel products of the coarse maps at LOW resolution + the tap-wise bilinear gather
    k_tapup_fwd / k_tapup_bwd + a direct convolution of the fine maps) against the reference formulation -- F.conv2d over
    torch.cat of the F.interpolate'd maps (models/HRNet.py:549-553, :596-600) -- in float64: output and the gradients of
    every map, the weight and the bias, 3e-6 / 1e-5 of max; both align_corners settings, sizes that are not multiples of
    the tiles, four pyramid levels (scales 1, 2, 4, 8) and a level count where only one map is coarse; the last two cases
    take the split-f16 GEMM for the coarse maps' channel products (channel counts % 32; 9 x 80 = 720 is a ragged
    contraction), the others the library's."""
    import torch.nn.functional as F
    from mscs_amd.models import ops
    torch.manual_seed(3)
    n = 2
    ts = [torch.randn(n, c, max(H >> i, 1), max(W >> i, 1), device=dev).requires_grad_(True) for i, c in enumerate(chans)]
    wt = (torch.randn(Co, sum(chans), 3, 3, device=dev) * 0.05).requires_grad_(True)
    b = torch.randn(Co, device=dev).requires_grad_(True) if bias else None
    gy = torch.randn(n, Co, H, W, device=dev)
    y = ops.conv3x3_over_upsampled(ts, align, wt, b)
    y.backward(gy)
    got = [y.detach()] + [t.grad for t in ts] + [wt.grad] + ([b.grad] if bias else [])
    ts64 = [t.detach().double().requires_grad_(True) for t in ts]
    w64 = wt.detach().double().requires_grad_(True)
    b64 = b.detach().double().requires_grad_(True) if bias else None
    cat = torch.cat([ts64[0]] + [F.interpolate(t, size=(H, W), mode="bilinear", align_corners=align) for t in ts64[1:]], 1)
    y64 = F.conv2d(cat, w64, b64, padding=1)
    y64.backward(gy.double())
    ref = [y64.detach()] + [t.grad for t in ts64] + [w64.grad] + ([b64.grad] if bias else [])
    for k, (a, r) in enumerate(zip(got, ref)):
        err = ((a.double() - r).abs().max() / r.abs().max()).item()
        assert err < (3e-6 if k == 0 else 1e-5), (k, err)


@pytest.mark.gpu
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("chans,H,W,Co,min_scale", [((32, 256, 64), 64, 96, 64, None), ((48, 96, 192), 40, 72, 32, 2),
                                                    ((16, 256), 128, 256, 32, None)])
def test_head_conv_over_upsampled_two_x_level(dev, align, chans, H, W, Co, min_scale):
    """The map that is only 2x coarser than the output through the tap products as well (round 4: automatic for >= 256 channels --
    UPerNet's P3 -- or forced with min_scale = 2): three coarse maps (pairs (8x, 4x) and the 2x map alone), and a 128 x 256 output
    whose 2x source fills a forward tile's LDS window beyond 64 KB (one workgroup per CU).  Against float64 as the test above."""
    import torch.nn.functional as F
    from mscs_amd.models import ops
    torch.manual_seed(4)
    n = 2
    ts = [torch.randn(n, c, max(H >> i, 1), max(W >> i, 1), device=dev).requires_grad_(True) for i, c in enumerate(chans)]
    wt = (torch.randn(Co, sum(chans), 3, 3, device=dev) * 0.05).requires_grad_(True)
    gy = torch.randn(n, Co, H, W, device=dev)
    seen = []
    orig = ops._HeadSplit.apply
    try:
        ops._HeadSplit.apply = staticmethod(lambda *a: (seen.append(len(a) - 7), orig(*a))[1])
        y = ops.conv3x3_over_upsampled(ts, align, wt, None, min_scale=min_scale)
    finally:
        ops._HeadSplit.apply = orig
    assert seen == [len(chans) - 1]                 # every map but the full-resolution one went through the tap products
    y.backward(gy)
    got = [y.detach()] + [t.grad for t in ts] + [wt.grad]
    ts64 = [t.detach().double().requires_grad_(True) for t in ts]
    w64 = wt.detach().double().requires_grad_(True)
    cat = torch.cat([ts64[0]] + [F.interpolate(t, size=(H, W), mode="bilinear", align_corners=align) for t in ts64[1:]], 1)
    y64 = F.conv2d(cat, w64, None, padding=1)
    y64.backward(gy.double())
    ref = [y64.detach()] + [t.grad for t in ts64] + [w64.grad]
    # the library's own fp32 distance from float64 on the same formulation: 2 M outputs of ~2 400 products each put its maximum at
    # a few 1e-6 as well -- the bar is 3e-6 or twice the library's error, whichever is larger
    with torch.no_grad():
        cat32 = torch.cat([ts[0]] + [F.interpolate(t, size=(H, W), mode="bilinear", align_corners=align) for t in ts[1:]], 1)
        e_lib = ((F.conv2d(cat32, wt, None, padding=1).double() - y64).abs().max() / y64.abs().max()).item()
    for k, (a, r) in enumerate(zip(got, ref)):
        err = ((a.double() - r).abs().max() / r.abs().max()).item()
        assert err < (max(3e-6, 2 * e_lib) if k == 0 else 1e-5), (k, err, e_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("align", [True, False])
def test_head_conv_over_upsampled_any_map_order(dev, align):
    """UPerNet's fusion convolution concatenates [P2, P5, P4, P3] (reference models/UPerNet.py:96-101): the coarse maps sit
    BETWEEN the fine ones in the weight's input channels.  conv3x3_over_upsampled against float64 for that order (output,
    every map's gradient, weight gradient), with the overlapped backward (the default) and without."""
    import torch.nn.functional as F
    from mscs_amd.models import ops
    torch.manual_seed(9)
    n, H, W, Co = 2, 64, 64, 64
    sizes = [(64, 64), (8, 8), (16, 16), (32, 32)]
    for overlap in (2, 0):
        prev = ops._HeadSplit.overlap
        ops._HeadSplit.overlap = overlap
        try:
            ts = [torch.randn(n, 64, h, w, device=dev).requires_grad_(True) for (h, w) in sizes]
            wt = (torch.randn(Co, 256, 3, 3, device=dev) * 0.05).requires_grad_(True)
            gy = torch.randn(n, Co, H, W, device=dev)
            y = ops.conv3x3_over_upsampled(ts, align, wt, None)
            y.backward(gy)
            got = [y.detach()] + [t.grad for t in ts] + [wt.grad]
            ts64 = [t.detach().double().requires_grad_(True) for t in ts]
            w64 = wt.detach().double().requires_grad_(True)
            cat = torch.cat([ts64[0]] + [F.interpolate(t, size=(H, W), mode="bilinear", align_corners=align) for t in ts64[1:]], 1)
            y64 = F.conv2d(cat, w64, None, padding=1)
            y64.backward(gy.double())
            ref = [y64.detach()] + [t.grad for t in ts64] + [w64.grad]
            for k, (a, r) in enumerate(zip(got, ref)):
                err = ((a.double() - r).abs().max() / r.abs().max()).item()
                assert err < (3e-6 if k == 0 else 1e-5), (overlap, k, err)
        finally:
            ops._HeadSplit.overlap = prev


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 2])
def test_weight_gradient_kernel_variants_match_fp64(dev, variant):
    """The two stride-1 weight-gradient kernels of the library -- MFMA-order loads (0, dcl_wgrad3x3.hip: the fallback) and
    LDS-DMA staging (2, dcl_wgrad3x3d.hip, the default) -- against float64 (3e-6 of max) and bitwise reproducible,
    ragged strips and channel tiles included; widths that are not multiples of 8 go in zero-padded (ops.conv3x3_wgrad)."""
    from mscs_amd import _lib
    from mscs_amd.models import ops
    L = _lib.lib()
    torch.manual_seed(40 + variant)
    try:
        L.dcl_wgrad3x3_set_variant(variant)
        for (n, ci, co, h, w) in [(2, 48, 96, 19, 40), (3, 96, 48, 16, 64), (1, 32, 64, 9, 72), (2, 192, 192, 8, 32),
                                  (1, 80, 112, 5, 24), (2, 32, 48, 20, 20), (1, 64, 32, 5, 12)]:    # last two: W % 8 != 0
            x = torch.randn(n, ci, h, w, device=dev).relu_() * 2.0
            gy = torch.randn(n, co, h, w, device=dev) * 1e-4
            ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), torch.zeros(co, ci, 3, 3, device=dev).double(),
                                                     None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
            gw = ops.conv3x3_wgrad(x, gy)
            assert ((gw.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6, (variant, n, ci, co, h, w)
            assert torch.equal(gw, ops.conv3x3_wgrad(x, gy))
    finally:
        L.dcl_wgrad3x3_set_variant(-1)


@pytest.mark.gpu
def test_weight_gradient_wave_level_splits_match_fp64_and_workgroup_form(dev):
    """129 .. 256 tile pairs of the (3, 1) tile (the head's 144 -> 720 launch: 135; 384 -> 384: 192): the pixel splits go to
    single waves, XCD by XCD (k_wgrad3x3d<3, 1, true>).  Against float64 (3e-6 of max), bitwise reproducible, and equal to
    the one-workgroup-per-pair form to f16x3 round-off; a tiny image (fewer row steps than splits) keeps the old form."""
    from mscs_amd import _lib
    from mscs_amd.models import ops
    L = _lib.lib()
    torch.manual_seed(91)
    try:
        for (n, ci, co, h, w) in [(2, 144, 720, 12, 40), (1, 384, 384, 9, 32), (3, 144, 720, 5, 72), (1, 144, 720, 1, 8),
                                  (1, 240, 432, 7, 24)]:
            x = torch.randn(n, ci, h, w, device=dev).relu_() * 2.0
            gy = torch.randn(n, co, h, w, device=dev) * 1e-4
            ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), torch.zeros(co, ci, 3, 3, device=dev).double(),
                                                     None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
            for mode in (1, 2):
                L.dcl_wgrad3x3_set_wave_mode(mode)
                slabs = L.dcl_wgrad3x3_splits(n, ci, co, h, w, 1)
                gw = ops.conv3x3_wgrad(x, gy)
                assert ((gw.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6, (mode, n, ci, co, h, w)
                assert torch.equal(gw, ops.conv3x3_wgrad(x, gy))
            L.dcl_wgrad3x3_set_wave_mode(0)
            assert (L.dcl_wgrad3x3_splits(n, ci, co, h, w, 1) == 1) == (slabs > 1 or h * w == 8), (slabs, n, ci, co, h, w)
            old = ops.conv3x3_wgrad(x, gy)
            assert ((gw - old).abs().max() / ref.abs().max()).item() < 3e-6
    finally:
        L.dcl_wgrad3x3_set_wave_mode(2)


@pytest.mark.gpu
def test_weight_gradient_adjacent_strip_grouping_matches_fp64(dev):
    """The four waves of a workgroup walk four (W % 128 == 0), two (W % 64 == 0) adjacent 32-pixel strips over the same rows, or
    four row ranges of one strip (dcl_wgrad3x3_set_strip_group): every form against float64 (3e-6 of max), bitwise
    reproducible, ragged last strips and fewer rows than splits included."""
    from mscs_amd import _lib
    from mscs_amd.models import ops
    L = _lib.lib()
    torch.manual_seed(17)
    try:
        for (n, ci, co, h, w) in [(2, 48, 48, 24, 256), (3, 64, 64, 7, 128), (2, 96, 96, 9, 64), (1, 48, 96, 5, 192),
                                  (2, 32, 32, 3, 96), (1, 64, 48, 2, 120), (12, 48, 48, 2, 128)]:
            x = torch.randn(n, ci, h, w, device=dev).relu_() * 2.0
            gy = torch.randn(n, co, h, w, device=dev) * 1e-4
            ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), torch.zeros(co, ci, 3, 3, device=dev).double(),
                                                     None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
            for on in (1, 0):
                L.dcl_wgrad3x3_set_strip_group(on)
                gw = ops.conv3x3_wgrad(x, gy)
                assert ((gw.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6, (on, n, ci, co, h, w)
                assert torch.equal(gw, ops.conv3x3_wgrad(x, gy))
    finally:
        L.dcl_wgrad3x3_set_strip_group(1)


@pytest.mark.gpu
def test_stride2_gradient_formulations_agree(dev):
    """Stride-2 data gradient by output parity classes (default) against the stride-1 tile over the zero-inserted gradient
    (dcl_conv3x3_set_up2_phases(0)), and the weight gradient over the output pixels against the zero-inserted dY operand
    (dcl_wgrad3x3_set_stride2(0)): same results to f16x3 round-off, odd sizes included."""
    from mscs_amd import _lib
    from mscs_amd.models import ops
    L = _lib.lib()
    torch.manual_seed(77)
    try:
        for (n, ci, co, h, w) in [(2, 48, 96, 32, 64), (1, 32, 32, 9, 48), (2, 64, 48, 7, 16)]:
            x = torch.randn(n, ci, h, w, device=dev).relu_()
            wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
            gy = torch.randn(n, co, (h - 1) // 2 + 1, (w - 1) // 2 + 1, device=dev) * 1e-3
            res = {}
            for mode in (1, 0):
                L.dcl_conv3x3_set_up2_phases(mode)
                L.dcl_wgrad3x3_set_stride2(mode)
                res[mode] = (ops.conv3x3_direct(gy, wt, transposed=True, stride=2, out_hw=(h, w)),
                             ops.conv3x3_wgrad(x, gy, stride=2))
            for a, b in zip(res[1], res[0]):
                assert ((a - b).abs().max() / b.abs().max()).item() < 3e-6, (n, ci, co, h, w)
    finally:
        L.dcl_conv3x3_set_up2_phases(1)
        L.dcl_wgrad3x3_set_stride2(1)


# ---- split-f16 GEMM (csrc/dcl_gemm.hip) ---------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
def test_gemm_f16x3_every_layout_matches_fp64(dev, tile):
    """dcl_gemm_f16x3: all four operand layouts (contraction index contiguous / row index contiguous), ragged M and N,
    one to many k-steps, batch, bias, accumulate, k-split slabs and the absmax side output, every workgroup tile, against
    float64 (3e-6 of max: the bound of the direct convolutions; the library's fp32 GEMM is measured beside it)."""
    from mscs_amd import _lib
    from mscs_amd.models import ops
    from mscs_amd.models.amax import amax_of
    L = _lib.lib()
    torch.manual_seed(5 + tile)
    try:
        L.dcl_gemm_set_tile(tile)
        for (M, N, K, batch) in [(100, 36, 32, 1), (260, 520, 96, 1), (512, 256, 64, 2), (36, 700, 416, 3), (1028, 132, 160, 1)]:
            for akm in (True, False):
                for bkm in (True, False):
                    A = torch.randn((batch, M, K) if akm else (batch, K, M), device=dev)
                    B = torch.randn((batch, N, K) if bkm else (batch, K, N), device=dev) * 0.03
                    bias = torch.randn(N, device=dev)
                    C0 = torch.randn(batch, M, N, device=dev)
                    Ad = A.double() if akm else A.double().transpose(1, 2)
                    Bd = B.double() if bkm else B.double().transpose(1, 2)
                    ref = Ad @ Bd.transpose(1, 2)
                    for splitk in (1, 2) if K >= 64 else (1,):
                        for acc in (False, True):
                            out = C0.clone() if acc else torch.full((batch, M, N), float("nan"), device=dev)
                            ca = torch.zeros(1, device=dev)
                            ops.gemm_f16x3(A, akm, K if akm else M, B, bkm, K if bkm else N, M, N, K, out, N, amax_of(A),
                                           amax_of(B), bias=bias, batch=batch, strides=(M * K, N * K, M * N),
                                           accumulate=acc, c_amax=ca, splitk=splitk)
                            want = ref + bias.double() + (C0.double() if acc else 0)
                            err = ((out.double() - want).abs().max() / ref.abs().max()).item()
                            assert err < 3e-6, (tile, M, N, K, batch, akm, bkm, splitk, acc, err)
                            assert abs(ca.item() - out.abs().max().item()) <= 1e-6 * out.abs().max().item()
                    if not akm and batch == 1 and M % 4 == 0:
                        # the row sums of a row-contiguous A (the bias gradient riding on a weight-gradient GEMM)
                        for splitk in (1, 2) if K >= 64 else (1,):
                            out = torch.empty(1, M, N, device=dev)
                            rsum = torch.full((M,), float("nan"), device=dev)
                            ops.gemm_f16x3(A, akm, M, B, bkm, K if bkm else N, M, N, K, out, N, amax_of(A), amax_of(B),
                                           splitk=splitk, a_rowsum=rsum)
                            want_rs = A.double().sum(1)[0]
                            assert ((rsum.double() - want_rs).abs().max() / want_rs.abs().max()).item() < 1e-5, (tile, M, N, K, splitk)
    finally:
        L.dcl_gemm_set_tile(0)


@pytest.mark.gpu
def test_gemm_f16x3_linear_triplet_at_swin_shapes(dev):
    """The three GEMMs of a token-major Linear (forward, data gradient, slab-wise weight gradient) at a Swin stage-3 shape:
    each at least as close to float64 as the library's fp32 GEMM, and bitwise reproducible (fixed-order slab sums)."""
    from mscs_amd.models import ops
    torch.manual_seed(17)
    m, k, n = 6400, 768, 2304
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.02
    b = torch.randn(n, device=dev)
    gy = torch.randn(m, n, device=dev) * 1e-3
    for mine, lib, ref in [
            (lambda: ops.linear_f16x3(x, w, b), lambda: torch.nn.functional.linear(x, w, b), lambda: x.double() @ w.double().t() + b.double()),
            (lambda: ops.linear_dgrad_f16x3(gy, w), lambda: gy.mm(w), lambda: gy.double() @ w.double()),
            (lambda: ops.linear_wgrad_f16x3(gy, x), lambda: gy.t().mm(x), lambda: gy.double().t() @ x.double())]:
        r = ref()
        a, a2, l = mine(), mine(), lib()
        e_mine = ((a.double() - r).abs().max() / r.abs().max()).item()
        e_lib = ((l.double() - r).abs().max() / r.abs().max()).item()
        assert e_mine < 3e-6 and e_mine <= 2 * e_lib + 1e-7, (e_mine, e_lib)
        assert torch.equal(a, a2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 64, 64, 128), (1, 3, 32, 37, 50), (3, 1, 64, 20, 300), (2, 2, 48, 9, 7), (1, 3, 64, 512, 1024)])
def test_stem_weight_gradient_matches_fp64_and_is_reproducible(dev, shape):
    """csrc k_wgrad_stem (weight gradient of the stem's 3-channel stride-2 convolution, reference models/HRNet.py:404-405) against
    aten::convolution_backward in float64 -- 2e-6 of max: fp32 products and sums -- over odd sizes, ragged segments, 1 .. 3 input and
    32 .. 64 output channels; and bitwise equal from call to call (the library kernel it replaces uses atomics)."""
    from mscs_amd.models import ops
    n, ci, co, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(n * 1000 + h)
    x = torch.randn(n, ci, h, w, generator=g).to(dev)
    gy = torch.randn(n, co, (h - 1) // 2 + 1, (w - 1) // 2 + 1, generator=g).to(dev)
    assert ops.stem_wgrad_supported(x, co, 2)
    dw = ops.stem_wgrad(x, gy)
    ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), torch.zeros(co, ci, 3, 3, device=dev, dtype=torch.float64), None,
                                              [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert ((dw.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
    for _ in range(3):
        assert torch.equal(ops.stem_wgrad(x, gy), dw)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 48, 96, 32, 64), (2, 48, 48, 20, 48), (1, 32, 96, 7, 16), (3, 16, 48, 33, 80), (1, 96, 192, 64, 128),
                                   (2, 32, 32, 12, 32)])
def test_stride2_weight_gradient_lds_dma_form_is_bitwise_the_load_form(dev, shape):
    """csrc k_wgrad3x3_s2d (x rows of the stride-2 weight gradient staged by LDS-DMA, round 5; reference models/HRNet.py:216-261
    fuse layers) runs the arithmetic of k_wgrad3x3_s2 in the same order: bitwise equal results over odd heights, ragged strips,
    one to three co tiles per wave; and 2e-6 of max from float64."""
    from mscs_amd import _lib
    from mscs_amd.models import ops
    n, ci, co, h, w = shape
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(h * 100 + w)
    x = torch.randn(n, ci, h, w, generator=g).to(dev)
    gy = torch.randn(n, co, (h - 1) // 2 + 1, w // 2, generator=g).to(dev)
    try:
        assert L.dcl_wgrad3x3_set_stride2(2) == 0
        old = ops.conv3x3_wgrad(x, gy, 2)
        assert L.dcl_wgrad3x3_set_stride2(1) == 0
        new = ops.conv3x3_wgrad(x, gy, 2)
    finally:
        L.dcl_wgrad3x3_set_stride2(1)
    assert torch.equal(old, new)
    ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), torch.zeros(co, ci, 3, 3, device=dev, dtype=torch.float64), None,
                                              [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert ((new.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(5, 32, 1, 12, 24), (5, 32, 2, 12, 32), (12, 48, 1, 9, 16), (3, 32, 1, 10, 16)])
def test_narrow_input_convolution_weight_gradient_is_deterministic_and_right(dev, cfg):
    """DirectConv2d with fewer than 16 input channels that is NOT the stem's case (stride 1, or more than 3 channels): the weight
    gradient takes the split-f16 kernel on channels zero-padded to 16 (models/ops_conv.py) instead of aten::convolution_backward --
    2e-6 of max from float64, bitwise equal from call to call."""
    from mscs_amd.models.ops import DirectConv2d
    ci, co, st, h, w = cfg
    torch.manual_seed(ci * 10 + st)
    conv = DirectConv2d(ci, co, 3, st, 1, bias=False).to(dev)
    x = torch.randn(2, ci, h, w, device=dev, requires_grad=True)
    assert conv.eligible(x)
    y = conv(x)
    gy = torch.randn_like(y)
    gw = [torch.autograd.grad(y, conv.weight, gy, retain_graph=True)[0] for _ in range(3)]
    assert torch.equal(gw[0], gw[1]) and torch.equal(gw[0], gw[2])
    ref = torch.ops.aten.convolution_backward(gy.double(), x.detach().double(), conv.weight.detach().double(), None, [st, st], [1, 1], [1, 1],
                                              False, [0, 0], 1, [False, True, False])[1]
    assert ((gw[0].double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
