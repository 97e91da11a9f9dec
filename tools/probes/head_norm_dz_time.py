"""Stand-alone time and error of dcl_head_norm_dz at the benchmark head (12 x 720 x 128 x 256, 19 classes):  gpurun -- python tools/probes/head_norm_dz_time.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import mscs_amd
from mscs_amd import _lib as P
from mscs_amd.models import amax as A
L = P.lib(); dev = torch.device("cuda:0")
n, k, c, hw = 12, 19, 720, 32768
dl = torch.randn(n, k, hw, device=dev) * 1e-3; z = torch.randn(n, c, hw, device=dev)
wt = torch.zeros(c, 20, device=dev); wt[:, :k] = torch.randn(c, k, device=dev) * 0.04
c0 = torch.randn(c, device=dev) * 1e-4; c1 = torch.randn(c, device=dev) * 1e-4
dz = torch.empty_like(z); am = torch.zeros(64, device=dev)
def f(): P.check(L.dcl_head_norm_dz(P.ptr(dl), P.ptr(z), P.ptr(wt), P.ptr(c0), P.ptr(c1), n, k, c, hw, P.ptr(dz), P.ptr(am), P.stream_ptr(dev)), "x")
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"k_head_norm_dz {ms*1e3:.1f} us = {(2*n*c*hw*4 + n*k*hw*4)/ms/1e6:.0f} GB/s")
ref = torch.einsum('ck,nkp->ncp', wt[:, :k].double(), dl.double()) + c1.double().view(1, c, 1) * z.double() + c0.double().view(1, c, 1)
print("err", ((dz.double() - ref).abs().max() / ref.abs().max()).item(), "amax ok", abs(am.max().item() - dz.abs().max().item()) < 1e-12)
