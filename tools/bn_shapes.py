#!/usr/bin/env python3
"""Fused BN (+ residual + ReLU) forward / backward on the HRNet-W48 branch shapes at batch 12 (run under
tools/prof_cmd.sh for the per-kernel times)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa
from mscs_amd.models.fused_bn import FusedBatchNorm2d
from per_shape_roofline import timeit
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
shapes = [(48, 128, 256), (96, 64, 128), (192, 32, 64), (384, 16, 32), (256, 128, 256)]
if len(sys.argv) > 1:
    shapes = [shapes[int(sys.argv[1])]]
for (c, h, w) in shapes:
    bn = FusedBatchNorm2d(c).to(dev).train()
    bn.emit_amax = os.environ.get('BN_NO_AMAX') is None
    x = torch.randn(12, c, h, w, device=dev, generator=gen).requires_grad_(True)
    r = torch.randn(12, c, h, w, device=dev, generator=gen).requires_grad_(True)
    gy = torch.randn(12, c, h, w, device=dev, generator=gen)
    mb = x.numel() * 4 / 1e6
    for res in (False, True):
        def fwd():
            return bn(x, residual=r if res else None, relu=True)
        y = fwd()
        tf = timeit(fwd, 20)
        def bwd():
            x.grad = None
            r.grad = None
            y.backward(gy, retain_graph=True)
        tb = timeit(bwd, 20)
        pf, pb = (4 if res else 3), (6 if res else 5)
        print(f"{c:3d} ch {h}x{w} ({mb:4.0f} MB) residual={res}: fwd {tf * 1e3:6.1f} us ({pf * mb / tf / 1e6:4.1f} TB/s over {pf} passes), "
              f"bwd {tb * 1e3:6.1f} us ({pb * mb / tb / 1e6:4.1f} TB/s over {pb} passes)", flush=True)
