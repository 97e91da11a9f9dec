#!/usr/bin/env python3
"""Fused up-sampling + cross-entropy at the two benchmark shapes: forward / backward time per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa
from mscs_amd.models.ops import UpsampledLogits
from per_shape_roofline import timeit
dev = torch.device("cuda:0")
for (n, C, h, w, H, W, al) in [(12, 19, 128, 256, 512, 1024, True), (16, 150, 128, 128, 512, 512, False),
                              (16, 150, 32, 32, 512, 512, False)]:
    z = torch.randn(n, C, h, w, device=dev, requires_grad=True)
    t = torch.randint(0, C + 1, (n, H, W), device=dev)
    def fwd():
        return UpsampledLogits(z, (H, W), al).cross_entropy(t, ignore_index=C)
    loss = fwd()
    tf = timeit(fwd, 10)
    def bwd():
        z.grad = None
        loss.backward(retain_graph=True)
    tb = timeit(bwd, 10)
    print(f"{n}x{C}x{h}x{w} -> {H}x{W}: fwd {tf * 1e3:7.1f} us, bwd {tb * 1e3:7.1f} us", flush=True)
