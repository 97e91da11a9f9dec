"""Swin window attention kernels through the C ABI against the reference's formulation
(models/Swin.py:198-230 inside :286-318: zero-pad -> roll -> window_partition -> (q * scale) k^T + relative-position
bias + shift mask -> softmax -> v -> window_reverse -> roll back -> crop), restated here with torch ops in fp64.
Tolerance: forward 2e-6 of max, gradients 1e-5 of max (fp32 round-off; the arithmetic is fp32 FMAs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _partition(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def _reverse(w, ws, H, W):
    B = w.shape[0] // ((H // ws) * (W // ws))
    return w.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def _shift_mask(Hp, Wp, ws, ss, device, dtype):
    region = torch.zeros((1, Hp, Wp, 1), dtype=dtype)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
        for wsl in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
            region[:, hs, wsl, :] = cnt
            cnt += 1
    mw = _partition(region, ws).reshape(-1, ws * ws)
    diff = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff)).to(device)


def reference_block_attention(xn, wqkv, bqkv, bias, H, W, heads, shift, scale):
    """xn [B, H*W, C] normalised tokens -> attention output [B, H*W, C], the reference's data flow."""
    B, L, C = xn.shape
    ws = 7
    x = xn.view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = H + pad_b, W + pad_r
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    win = _partition(x, ws)                                             # [B_, 49, C]
    B_, N, _ = win.shape
    qkv = F.linear(win, wqkv, bqkv).reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * scale, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if shift:
        mask = _shift_mask(Hp, Wp, ws, shift, xn.device, xn.dtype)
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = attn.softmax(-1)
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    x = _reverse(out, ws, Hp, Wp)
    if shift:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    return x[:, :H, :W, :].reshape(B, H * W, C)


@pytest.mark.parametrize("B,H,W,heads,shift", [(2, 14, 21, 3, 0), (2, 14, 21, 3, 3), (1, 16, 20, 2, 3), (2, 9, 5, 1, 3),
                                               (1, 32, 32, 6, 3), (3, 7, 7, 2, 0), (1, 5, 6, 1, 3)])
def test_window_attention_matches_reference_formulation(B, H, W, heads, shift):
    from mscs_amd.models.ops import window_attention
    dev = torch.device("cuda:0")
    C = 32 * heads
    gen = torch.Generator().manual_seed(H * 100 + W + shift)
    xn = torch.randn(B, H * W, C, generator=gen, dtype=torch.float64)
    wqkv = torch.randn(3 * C, C, generator=gen, dtype=torch.float64) / C ** 0.5
    bqkv = torch.randn(3 * C, generator=gen, dtype=torch.float64) * 0.5
    bias = torch.randn(heads, 49, 49, generator=gen, dtype=torch.float64)
    gout = torch.randn(B, H * W, C, generator=gen, dtype=torch.float64)
    scale = 32 ** -0.5
    # fp64 reference through autograd
    rx, rw, rb, rbias = [t.clone().requires_grad_(True) for t in (xn, wqkv, bqkv, bias)]
    ref = reference_block_attention(rx, rw, rb, rbias, H, W, heads, shift, scale)
    ref.backward(gout)
    # HIP path: projection on the natural token order + the kernel
    hx, hw, hb, hbias = [t.float().to(dev).requires_grad_(True) for t in (xn, wqkv, bqkv, bias)]
    qkv = F.linear(hx, hw, hb)
    out = window_attention(qkv, hb, hbias, H, W, heads, shift, scale)
    out.backward(gout.float().to(dev))

    def close(got, want, tol, what):
        err = (got.detach().double().cpu() - want.detach()).abs().max().item()
        assert err <= tol * max(1.0, want.detach().abs().max().item()), (what, err)
    close(out, ref, 2e-6, "out")
    close(hx.grad, rx.grad, 1e-5, "dx")
    close(hw.grad, rw.grad, 1e-5, "dW")
    close(hb.grad, rb.grad, 1e-5, "db (incl. the padded tokens)")
    close(hbias.grad, rbias.grad, 1e-5, "d relative-position bias")
    # bitwise run-to-run determinism (fixed-order partial sums)
    hx2, hb2, hbias2 = [t.detach().clone().requires_grad_(True) for t in (hx, hb, hbias)]
    out2 = window_attention(F.linear(hx2, hw.detach(), hb2), hb2, hbias2, H, W, heads, shift, scale)
    out2.backward(gout.float().to(dev))
    assert torch.equal(out2, out) and torch.equal(hbias2.grad, hbias.grad) and torch.equal(hx2.grad, hx.grad)


def test_swin_block_hip_path_matches_library_path():
    """SwinTransformerBlock with the HIP attention against the same block on the library path (pad / roll /
    partition / SDPA): forward and all parameter gradients, shifted and unshifted, non-multiple-of-7 map."""
    from mscs_amd.models.Swin import SwinTransformerBlock
    dev = torch.device("cuda:0")
    for shift in (0, 3):
        torch.manual_seed(3)
        blk = SwinTransformerBlock(dim=96, num_heads=3, window_size=7, shift_size=shift).to(dev)
        torch.nn.init.normal_(blk.attn.relative_position_bias_table, std=0.5)
        H, W = 18, 25
        blk.H, blk.W = H, W
        x = torch.randn(2, H * W, 96, device=dev)
        from mscs_amd.models.Swin import BasicLayer
        layer = BasicLayer(dim=96, depth=2, num_heads=3)
        mask = layer._shift_mask(H, W, dev)
        res = {}
        for hip in (True, False):
            blk.hip_attention = hip
            for p in blk.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            y = blk(xi, mask)
            (y * torch.cos(torch.arange(y.numel(), device=dev).view_as(y) * 0.37)).sum().backward()
            res[hip] = (y.detach(), xi.grad, {k: p.grad.clone() for k, p in blk.named_parameters()})
        ya, ga, pa = res[True]
        yb, gb, pb = res[False]
        assert (ya - yb).abs().max().item() <= 2e-5 * yb.abs().max().item()
        assert (ga - gb).abs().max().item() <= 1e-4 * gb.abs().max().item()
        for k in pa:
            assert (pa[k] - pb[k]).abs().max().item() <= 1e-4 * max(pb[k].abs().max().item(), 1e-6), k


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,bias", [(65536, 96, 288, True), (32768, 192, 96, False), (3 * 16384, 64, 64, True), (6400, 1536, 384, True)])
def test_token_linear_on_split_f16_gemm(M, K, N, bias):
    """TokenLinear (the Swin port's nn.Linear: models/Swin.py qkv / proj / fc1 / fc2 / reduction, reference Swin.py:
    62-76, 198-230): same parameters / state_dict as nn.Linear; forward, data gradient and weight gradient run on
    dcl_gemm_f16x3 and are each at least as close to float64 as the library's fp32 GEMM; bitwise reproducible."""
    import mscs_amd  # noqa: F401
    from mscs_amd.models.ops import TokenLinear
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    lin = TokenLinear(K, N, bias=bias).to(dev)
    ref = torch.nn.Linear(K, N, bias=bias).to(dev)
    ref.load_state_dict(lin.state_dict())
    assert list(lin.state_dict().keys()) == list(ref.state_dict().keys())
    x = torch.randn(4, M // 4, K, generator=g).to(dev)
    gy = torch.randn(4, M // 4, N, generator=g).to(dev)
    outs = []
    for mod in (lin, ref, lin):
        xi = x.clone().requires_grad_(True)
        mod.zero_grad()
        y = mod(xi)
        y.backward(gy)
        outs.append((y.detach(), xi.grad, mod.weight.grad.clone(), mod.bias.grad.clone() if bias else None))
    w64, b64 = lin.weight.detach().double(), (lin.bias.detach().double() if bias else 0)
    want = (x.double() @ w64.t() + b64, gy.double() @ w64, gy.double().view(-1, N).t().mm(x.double().view(-1, K)))
    for i, name in enumerate(("y", "dx", "dw")):
        den = want[i].abs().max()
        e_mine = ((outs[0][i].double() - want[i]).abs().max() / den).item()
        e_lib = ((outs[1][i].double() - want[i]).abs().max() / den).item()
        assert e_mine < 3e-6 and e_mine <= 2 * e_lib + 1e-7, (name, e_mine, e_lib)
        assert torch.equal(outs[0][i], outs[2][i]), name
    if bias:
        assert torch.allclose(outs[0][3], outs[1][3], rtol=1e-5, atol=1e-4)
    # few rows, or no grad: nn.Linear itself
    small = torch.randn(8, 49, K, device=dev)
    assert torch.equal(lin(small), ref(small))


@pytest.mark.gpu
@pytest.mark.parametrize("M,C", [(4 * 3137, 96), (2 * 1031, 192), (777, 384), (130, 768), (65, 1536), (50, 32), (9, 2048)])
def test_fused_layernorm_matches_fp64(M, C):
    """FusedLayerNorm (csrc/dcl_layernorm.hip; the Swin port's norm layers, reference Swin.py:251-332, :357-362,
    :452-455, :560-565) against nn.LayerNorm in fp64: forward and all three gradients at ragged row counts, and at
    least as close as ATen's fp32 kernels; unsupported row lengths fall through to nn.LayerNorm."""
    import mscs_amd  # noqa: F401
    from mscs_amd.models.ops import FusedLayerNorm
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    ln = FusedLayerNorm(C).to(dev)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(C, generator=g) + 0.5)
        ln.bias.copy_(torch.randn(C, generator=g))
    ref32 = torch.nn.LayerNorm(C).to(dev)
    ref32.load_state_dict(ln.state_dict())
    ref64 = torch.nn.LayerNorm(C).to(dev).double()
    ref64.load_state_dict(ln.state_dict())
    x = (torch.randn(M, C, generator=g) * 2.0 + 3.0).to(dev)
    gy = torch.randn(M, C, generator=g).to(dev)
    res = []
    for mod, dt in ((ln, torch.float32), (ref32, torch.float32), (ref64, torch.float64)):
        xi = x.detach().clone().to(dt).requires_grad_(True)
        y = mod(xi)
        y.backward(gy.to(dt))
        res.append([t.double() for t in (y.detach(), xi.grad, mod.weight.grad, mod.bias.grad)])
    for k, name in enumerate(("y", "dx", "dweight", "dbias")):
        den = res[2][k].abs().max()
        e_hip = ((res[0][k] - res[2][k]).abs().max() / den).item()
        e_lib = ((res[1][k] - res[2][k]).abs().max() / den).item()
        assert e_hip < 3e-6 and e_hip <= 3 * e_lib + 1e-7, (name, e_hip, e_lib)
    # determinism of the partial-row reduction
    xi = x.clone().requires_grad_(True)
    ln.weight.grad = None
    ln(xi).backward(gy)
    assert torch.equal(ln.weight.grad.double(), res[0][2])
    assert list(ln.state_dict().keys()) == list(ref32.state_dict().keys())


@pytest.mark.gpu
def test_fused_layernorm_fallbacks():
    import mscs_amd  # noqa: F401
    from mscs_amd.models.ops import FusedLayerNorm
    dev = torch.device("cuda:0")
    ln = FusedLayerNorm(100).to(dev)            # 25 vectors: no (V, G) plan
    x = torch.randn(7, 100, device=dev)
    assert torch.equal(ln(x), F.layer_norm(x, (100,), ln.weight, ln.bias, ln.eps))
    ln = FusedLayerNorm(96).to(dev)
    xt = torch.randn(96, 64, device=dev).t()    # not contiguous
    assert torch.equal(ln(xt), F.layer_norm(xt, (96,), ln.weight, ln.bias, ln.eps))




@pytest.mark.gpu
def test_linear_weight_tags_and_tag_carrying():
    """LinearTagGroup: one multi-tensor launch tags every TokenLinear weight with its exact absmax and re-tags after an
    in-place update; amax.carry hands a tag to views (the GEMM wrappers find operand scales without a pass); the tagged GELU
    bounds its output and, backward, its input gradient."""
    import mscs_amd  # noqa: F401
    from mscs_amd.models import amax, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    net = torch.nn.Sequential(ops.TokenLinear(64, 96), ops.TokenLinear(96, 32, bias=False)).to(dev)
    grp = ops.LinearTagGroup(net)
    grp.refresh()
    for m in net:
        t = amax.tag_of(m.weight)
        assert t is not None and abs(t.max().item() - m.weight.abs().max().item()) == 0.0
    with torch.no_grad():
        net[0].weight.mul_(3.0)
    assert amax.tag_of(net[0].weight) is None            # stale after the in-place update
    grp.refresh()
    assert abs(amax.tag_of(net[0].weight).max().item() - net[0].weight.abs().max().item()) == 0.0
    x = torch.randn(4, 512, 64, device=dev)
    amax.amax_of(x)
    v = amax.carry(x, x.view(-1, 64))
    assert amax.tag_of(v) is amax.tag_of(x)
    h = torch.randn(2048, 96, device=dev, requires_grad=True)
    amax.amax_of(h)
    g = ops.tagged_gelu(h)
    assert amax.tag_of(g) is amax.tag_of(h) and g.abs().max() <= amax.tag_of(g).max()
    gy = torch.randn_like(g)
    amax.amax_of(gy)
    g.backward(gy)
    assert torch.allclose(h.grad, torch.ops.aten.gelu_backward(gy, h.detach(), approximate="none"))


@pytest.mark.parametrize("std", [1.0, 3.0])
def test_window_attention_forward_at_size_against_fp64(std):
    """Forward of both kernel variants (matrix cores = default, vector ALU) on 8 x 70 x 70 tokens x 6 heads (7.5 M outputs, 4 800
    window-heads) against float64: a rare-event check -- an MFMA reading a register before the high half of an inline-asm
    split had landed showed up as 1e-4 errors in one of ~10^5 scores, invisible at the sizes of the tests above."""
    import mscs_amd  # noqa: F401
    from mscs_amd import _lib
    from mscs_amd.models import ops
    L = _lib.lib()
    dev = "cuda"
    B, H, W, heads = 8, 70, 70, 6
    C = 32 * heads
    torch.manual_seed(1)
    qkv = torch.randn(B, H * W, 3 * C, device=dev) * std
    qb = torch.zeros(3 * C, device=dev)
    bias = torch.randn(heads, 49, 49, device=dev) * 0.5
    x = qkv.double().view(B, H // 7, 7, W // 7, 7, 3, heads, 32).permute(5, 0, 1, 3, 6, 2, 4, 7).reshape(3, -1, heads, 49, 32)
    att = ((x[0] * 32 ** -0.5) @ x[1].transpose(-1, -2) + bias.double()).softmax(-1) @ x[2]
    ref = att.view(B, H // 7, W // 7, heads, 7, 7, 32).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, H * W, C)
    try:
        for mask in (0, 1):
            L.dcl_winattn_set_mfma(mask)
            out = ops.window_attention(qkv, qb, bias, H, W, heads, 0, 32 ** -0.5)
            err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 4e-6, (mask, std, err)
    finally:
        L.dcl_winattn_set_mfma(3)
