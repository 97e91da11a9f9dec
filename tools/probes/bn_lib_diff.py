"""Bitwise differences between two builds of the norm kernels (DCL_LIB_PATH selects the build):

    DCL_LIB_PATH=a.so python tools/probes/bn_lib_diff.py save /tmp/a.pt;  DCL_LIB_PATH=b.so python ... save /tmp/b.pt
    python tools/probes/bn_lib_diff.py cmp /tmp/a.pt /tmp/b.pt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k].double() - b[k].double()).abs()
        n = int((a[k] != b[k]).sum())
        print(f"{k:28s} {n:8d} of {a[k].numel():8d} differ, max |d| {d.max().item():.3e} (max |a| {a[k].abs().max().item():.3e})")
    sys.exit(0)

import mscs_amd  # noqa: F401,E402
from mscs_amd.models import fused_bn  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for tag, (n, c, h, w, res, relu) in {"c64_relu": (2, 64, 32, 64, False, True), "c256_res_relu": (2, 256, 32, 64, True, True),
                                      "c48_plain": (2, 48, 64, 96, False, False), "c64_odd": (2, 64, 15, 17, False, True)}.items():
    torch.manual_seed(1)
    bn = fused_bn.FusedBatchNorm2d(c).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = (torch.randn(n, c, h, w, device=dev) * 2 + 0.3).requires_grad_(True)
    r = torch.randn(n, c, h, w, device=dev) if res else None
    dy = torch.randn(n, c, h, w, device=dev)
    y = bn(x, residual=r, relu=relu)
    y.backward(dy)
    torch.cuda.synchronize()
    for k, t in (("y", y), ("dx", x.grad), ("dgamma", bn.weight.grad), ("dbeta", bn.bias.grad), ("rmean", bn.running_mean), ("rvar", bn.running_var)):
        out[f"{tag}.{k}"] = t.detach().cpu()
torch.save(out, sys.argv[2])
