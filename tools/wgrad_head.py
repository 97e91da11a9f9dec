#!/usr/bin/env python3
"""Weight gradient of the 720-channel head convolution: tile / partition variants of the shared-dY kernel against the
per-wave kernel (time, and agreement of the results)."""
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
shapes = [(720, 128, 256)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (c, h, w) in shapes:
    x = torch.randn(12, c, h, w, device=dev, generator=gen).relu_()
    gy = torch.randn(12, c, h, w, device=dev, generator=gen) * 1e-3
    flops = 2.0 * 12 * c * c * 9 * h * w
    L.dcl_wgrad3x3_set_variant(0)
    ref = ops.conv3x3_wgrad(x, gy)
    t = timeit(lambda: ops.conv3x3_wgrad(x, gy), 10)
    print(f"C={c} {h}x{w} per-wave kernel: {t * 1e3:8.1f} us {flops / t / 1e9:5.0f} TF", flush=True)
    L.dcl_wgrad3x3_set_variant(1)
    for (nco, nci) in [(3, 2), (3, 1), (5, 1), (2, 2), (2, 1)]:
        if (c // 16) % nco:
            continue
        for sk, nwg in [(0, 256), (1, 256), (1, 512)]:
            if nwg == 512 and nco * nci > 3:
                continue
            L.dcl_wgrad3x3_set_tile(nco, nci)
            L.dcl_wgrad3x3_set_partition(sk, nwg)
            out = ops.conv3x3_wgrad(x, gy)
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            t = timeit(lambda: ops.conv3x3_wgrad(x, gy), 10)
            print(f"  shared ({nco},{nci}) {'stream-K' if sk else 'equal   '} nwg={nwg}: {t * 1e3:8.1f} us "
                  f"{flops / t / 1e9:5.0f} TF  max diff vs per-wave {err:.1e}", flush=True)
    L.dcl_wgrad3x3_set_tile(0, 0)
    L.dcl_wgrad3x3_set_partition(-1, 0)
