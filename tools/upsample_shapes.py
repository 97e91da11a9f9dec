#!/usr/bin/env python3
"""Bilinear up-sampling kernels on the HRNet-W48 shapes at batch 12: time per launch and HBM rate."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa
from mscs_amd.models import ops
from per_shape_roofline import timeit
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
for (c, h, w, H, W) in [(48, 64, 128, 128, 256), (48, 32, 64, 128, 256), (48, 16, 32, 128, 256), (96, 32, 64, 64, 128),
                        (96, 64, 128, 128, 256), (384, 16, 32, 128, 256)]:
    x = torch.randn(12, c, h, w, device=dev, generator=gen).requires_grad_(True)
    add = torch.randn(12, c, H, W, device=dev, generator=gen)
    y = ops.upsample_bilinear(x, (H, W), True, add=add, relu=True)
    gy = torch.randn_like(y)
    tf = timeit(lambda: ops.upsample_bilinear(x, (H, W), True, add=add, relu=True), 20)
    def bwd():
        x.grad = None
        y.backward(gy, retain_graph=True)
    tb = timeit(bwd, 20)
    mb = y.numel() * 4 / 1e6
    print(f"{c:3d} ch {h}x{w} -> {H}x{W} ({mb:5.0f} MB out): fwd {tf * 1e3:6.1f} us ({(2 * mb + x.numel() * 4e-6) / (tf * 1e3):4.1f} TB/s r+w), "
          f"bwd (threshold + rows kernel) {tb * 1e3:6.1f} us", flush=True)
