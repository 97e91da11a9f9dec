#!/usr/bin/env python3
"""Forward tile sweep of the direct 3x3 kernel on the UPerNet decoder's shapes (config 4: batch 16, 512 channels at
128^2 ... 16^2, fusion 2048 -> 512 at 128^2, auxiliary head 384 -> 256 at 32^2).   python tools/conv_tiles_decoder.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa: E402,F401
from mscs_amd.models import ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402
from bench_conv3x3 import timeit  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (n, ci, co, h, w) in [(16, 512, 512, 128, 128), (16, 512, 512, 64, 64), (16, 512, 512, 32, 32),
                          (16, 512, 512, 16, 16), (16, 2048, 512, 128, 128), (16, 384, 256, 32, 32)]:
    x = torch.randn(n, ci, h, w, device=dev).relu_()
    wt = torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (9 * ci)) ** 0.5
    sx, sw = amax_of(x), amax_of(wt)
    wp = ops.conv3x3_pack(wt, sw)
    out = torch.empty(n, co, h, w, device=dev)
    flops = 2.0 * n * ci * co * 9 * h * w
    t = timeit(lambda: ops.conv3x3_launch(x, wp, co, sx, sw, out), 5)
    print(f"{ci}->{co} {h}x{w}: plan {t:.3f} ms ({flops / t / 1e9:.0f} TF)", flush=True)
    for r in (2, 3):
        for p in (1, 2, 4):
            t = timeit(lambda: ops.conv3x3_launch(x, wp, co, sx, sw, out, r, p), 5)
            print(f"    R={r} P={p}: {t:.3f} ms ({flops / t / 1e9:.0f} TF)", flush=True)
    del x, wt, out
