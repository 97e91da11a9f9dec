#!/usr/bin/env python3
"""1x1 convolutions of HRNet-W48 at batch 12: f16x3 direct kernels against the batched fp32 library GEMMs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa
from mscs_amd.models import ops, amax as _amax
from per_shape_roofline import timeit
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
n = 12
shapes = [("layer1 64->256", 64, 256, 128, 256), ("layer1 256->64", 256, 64, 128, 256), ("layer1 64->64", 64, 64, 128, 256),
          ("proj 48->256", 48, 256, 128, 256), ("proj 96->256", 96, 256, 64, 128), ("proj 192->256", 192, 256, 32, 64),
          ("fuse 96->48", 96, 48, 64, 128), ("fuse 384->48", 384, 48, 16, 32), ("fuse 384->192", 384, 192, 16, 32),
          ("cls 720->19", 720, 19, 128, 256)]
for name, ci, co, h, w in shapes:
    x = torch.randn(n, ci, h, w, device=dev, generator=gen).relu_()
    gy = torch.randn(n, co, h, w, device=dev, generator=gen) * 1e-3
    wt = torch.randn(co, ci, 1, 1, device=dev, generator=gen) * 0.05
    wa, xa, ga = _amax.amax_of(wt), _amax.amax_of(x), _amax.amax_of(gy)
    wp, wpt = ops.conv3x3_pack(wt, wa), ops.conv3x3_pack(wt, wa, True)
    y, gx = torch.empty_like(gy), torch.empty_like(x)
    tf = timeit(lambda: ops.conv1x1_launch(x, wp, co, xa, wa, y), 20)
    td = timeit(lambda: ops.conv1x1_launch(gy, wpt, ci, ga, wa, gx), 20)
    tw = timeit(lambda: ops.conv1x1_wgrad(x, gy), 20) if ops.conv1x1_wgrad_supported(x, co) else float("nan")
    w2 = wt.view(co, ci)
    lf = timeit(lambda: torch.matmul(w2, x.view(n, ci, h * w), out=y.view(n, co, h * w)), 20)
    ld = timeit(lambda: torch.matmul(w2.t(), gy.view(n, co, h * w), out=gx.view(n, ci, h * w)), 20)
    lw = timeit(lambda: torch.bmm(gy.view(n, co, h * w), x.view(n, ci, h * w).transpose(1, 2)).sum(0), 20)
    hw = h * w
    gf = gd = float("nan")
    if ci % 32 == 0 and hw % 4 == 0:
        gf = timeit(lambda: ops.gemm_f16x3(w2, True, ci, x, False, hw, co, hw, ci, y, hw, wa, xa, batch=n,
                                           strides=(0, ci * hw, co * hw), splitk=1), 20)
    if co % 4 == 0 and ci % 4 == 0:
        gd = timeit(lambda: ops.gemm_f16x3(w2, False, ci, gy, False, hw, ci, hw, co, gx, hw, wa, ga, batch=n,
                                           strides=(0, co * hw, ci * hw), splitk=1), 20)
    mb = (x.numel() + gy.numel()) * 4 / 1e6
    print(f"{name:16s} {mb:6.0f} MB  f16x3 fwd {tf * 1e3:6.1f} dgrad {td * 1e3:6.1f} wgrad {tw * 1e3:6.1f} us | "
          f"library fwd {lf * 1e3:6.1f} dgrad {ld * 1e3:6.1f} wgrad {lw * 1e3:6.1f} us | "
          f"dcl_gemm fwd {gf * 1e3:6.1f} dgrad {gd * 1e3:6.1f} us", flush=True)
