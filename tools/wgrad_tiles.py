#!/usr/bin/env python3
"""Weight-gradient kernel: time every (co tiles, ci tiles) per-wave tile on the HRNet-W48 branch shapes."""
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
VARIANT = int(os.environ.get("DCL_WGRAD_VARIANT", "-1"))
L.dcl_wgrad3x3_set_variant(VARIANT)
print("variant", VARIANT)
for (c, h, w) in [(48, 128, 256), (96, 64, 128), (192, 32, 64), (384, 16, 32), (64, 128, 256), (720, 128, 256)]:
    x = torch.randn(12 if c != 720 else 12, c, h, w, device=dev, generator=gen).relu_()
    gy = torch.randn(12, c, h, w, device=dev, generator=gen) * 1e-3
    flops = 2.0 * 12 * c * c * 9 * h * w
    res = []
    for nco in (1, 2, 3):
        for nci in (1, 2):
            if (c // 16) % nco or ((c // 16) % nci and nci == 2 and (c // 16) % 2):
                continue
            L.dcl_wgrad3x3_set_tile(nco, nci)
            try:
                t = timeit(lambda: ops.conv3x3_wgrad(x, gy), 20)
                res.append(f"({nco},{nci}) {t * 1e3:6.1f} us {flops / t / 1e9:5.0f} TF")
            except Exception as e:  # noqa
                res.append(f"({nco},{nci}) failed")
    L.dcl_wgrad3x3_set_tile(0, 0)
    t = timeit(lambda: ops.conv3x3_wgrad(x, gy), 20)
    print(f"C={c:3d} {h}x{w}: auto {t * 1e3:6.1f} us | " + " | ".join(res), flush=True)
