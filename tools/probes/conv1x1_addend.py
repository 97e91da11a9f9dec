"""Layer 1's 256-channel data gradient with the residual gradient: tile kernel's fused addend vs library GEMM with beta = 1."""
import sys, torch
sys.path.insert(0, "/root/repo")
import mscs_amd
from mscs_amd.models import ops, amax as _amax
dev = torch.device("cuda:0")
n, ci, co, h, w = 12, 256, 64, 128, 256            # conv 256 -> 64; its data gradient produces 256 channels
gy = torch.randn(n, co, h, w, device=dev) * 1e-3
wt = torch.randn(co, ci, 1, 1, device=dev) * 0.05
add = torch.randn(n, ci, h, w, device=dev) * 1e-3
wa, ga = _amax.amax_of(wt), _amax.amax_of(gy)
wpt = ops.conv3x3_pack(wt, wa, True)
gx = torch.empty(n, ci, h, w, device=dev)


def t(f, it=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


a2 = add.clone()
print("tile kernel, fused addend  us", round(t(lambda: ops.conv1x1_launch(gy, wpt, ci, ga, wa, gx, addend=add)), 1))
print("library baddbmm_ (beta = 1) us", round(t(lambda: a2.view(n, ci, h * w).baddbmm_(wt.view(co, ci).t().unsqueeze(0).expand(n, ci, co), gy.view(n, co, h * w))), 1))
print("library matmul + add_       us", round(t(lambda: (torch.matmul(wt.view(co, ci).t(), gy.view(n, co, h * w), out=gx.view(n, ci, h * w)), gx.add_(add))), 1))
ref = add.double() + torch.einsum("oc,nop->ncp", wt.view(co, ci).double(), gy.view(n, co, -1).double()).view(n, ci, h, w)
ops.conv1x1_launch(gy, wpt, ci, ga, wa, gx, addend=add)
a3 = add.clone(); a3.view(n, ci, h * w).baddbmm_(wt.view(co, ci).t().unsqueeze(0).expand(n, ci, co), gy.view(n, co, h * w))
print("err tile", ((gx.double() - ref).abs().max() / ref.abs().max()).item(), "err library", ((a3.double() - ref).abs().max() / ref.abs().max()).item())
