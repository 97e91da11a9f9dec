#!/usr/bin/env python3
"""Cost of reducing the per-plane absmax partials (N * C values) in the prologue of the convolution kernels: the same
launches with a one-value tag and with the N * C-value tag the fused BN kernels emit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa
from mscs_amd import _lib
from mscs_amd.models import ops, amax as _amax
from per_shape_roofline import timeit
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
n = 12
for (c, h, w) in [(48, 128, 256), (96, 64, 128), (192, 32, 64), (384, 16, 32)]:
    x = torch.randn(n, c, h, w, device=dev, generator=gen).relu_()
    gy = torch.randn(n, c, h, w, device=dev, generator=gen) * 1e-3
    wt = torch.randn(c, c, 3, 3, device=dev, generator=gen) * 0.05
    wa = _amax.amax_of(wt)
    wp = ops.conv3x3_pack(wt, wa)
    y = torch.empty_like(x)
    res = []
    for planes in (False, True):
        for t_ in (x, gy):
            if planes:
                _amax.tag(t_, t_.abs().amax(dim=(2, 3)).flatten().contiguous())
            else:
                t_._dcl_amax = None
        xa = _amax.amax_of(x)
        tf = timeit(lambda: ops.conv3x3_launch(x, wp, c, xa, wa, y), 30)
        tw = timeit(lambda: ops.conv3x3_wgrad(x, gy), 30)
        res.append(f"{'per-plane' if planes else 'one value'} tags ({xa.numel()}): forward {tf * 1e3:6.1f} us, wgrad {tw * 1e3:6.1f} us")
    print(f"C={c:3d} {h}x{w}: " + " | ".join(res), flush=True)
