#!/usr/bin/env python3
"""Forward 3x3 convolution time against the amount of work (batch): fixed cost of a launch vs per-image cost."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa
from mscs_amd.models import ops, amax as _amax
from per_shape_roofline import timeit
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
for (c, h, w) in [(48, 128, 256), (96, 64, 128), (192, 32, 64), (384, 16, 32)]:
    wt = torch.randn(c, c, 3, 3, device=dev, generator=gen) * 0.05
    wa = _amax.amax_of(wt)
    wp = ops.conv3x3_pack(wt, wa)
    res = []
    for n in (1, 2, 4, 8, 12, 24, 48):
        x = torch.randn(n, c, h, w, device=dev, generator=gen).relu_()
        y = torch.empty_like(x)
        xa = _amax.amax_of(x)
        t = timeit(lambda: ops.conv3x3_launch(x, wp, c, xa, wa, y), 20)
        res.append(f"N={n}: {t * 1e3:6.1f}")
    print(f"C={c:3d} {h}x{w} us: " + " | ".join(res), flush=True)
