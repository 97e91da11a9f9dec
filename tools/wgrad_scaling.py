#!/usr/bin/env python3
"""Weight-gradient time against the amount of work (batch): separates the fixed cost of a launch (prologue, slab
write, slab reduction) from the per-pixel cost."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa
from mscs_amd import _lib
from mscs_amd.models import ops
from per_shape_roofline import timeit
L = _lib.lib()
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
L.dcl_wgrad3x3_set_variant(int(os.environ.get("DCL_WGRAD_VARIANT", "-1")))
for (c, h, w) in [(48, 128, 256), (96, 64, 128), (192, 32, 64), (384, 16, 32)]:
    res = []
    for n in (1, 2, 4, 8, 12, 24, 48):
        x = torch.randn(n, c, h, w, device=dev, generator=gen).relu_()
        gy = torch.randn(n, c, h, w, device=dev, generator=gen) * 1e-3
        t = timeit(lambda: ops.conv3x3_wgrad(x, gy), 20)
        res.append(f"N={n}: {t * 1e3:6.1f}")
    print(f"C={c:3d} {h}x{w} us: " + " | ".join(res), flush=True)
