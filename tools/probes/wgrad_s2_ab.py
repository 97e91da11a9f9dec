"""Stride-2 weight gradient: LDS-DMA staging of the x rows (k_wgrad3x3_s2d) against MFMA-order loads (k_wgrad3x3_s2) -- bitwise
comparison and HIP-event times on the fuse-layer shapes of HRNet-W48 at batch 12, and odd sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: E402,F401
from mscs_amd import _lib  # noqa: E402
from mscs_amd.models import ops  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (n, ci, co, h, w) in [(12, 48, 96, 128, 256), (12, 96, 192, 64, 128), (12, 192, 384, 32, 64), (12, 48, 48, 128, 256),
                          (2, 48, 48, 20, 48), (1, 32, 96, 7, 16), (3, 16, 48, 33, 80)]:
    x = torch.randn(n, ci, h, w, device=dev)
    gy = torch.randn(n, co, (h - 1) // 2 + 1, w // 2, device=dev)
    res = {}
    for variant in (0, -1):
        L.dcl_wgrad3x3_set_variant(variant)
        for _ in range(2):
            dw = ops.conv3x3_wgrad(x, gy, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dw = ops.conv3x3_wgrad(x, gy, 2)
        e1.record()
        torch.cuda.synchronize()
        res[variant] = (dw.clone(), e0.elapsed_time(e1) / 20 * 1e3)
    L.dcl_wgrad3x3_set_variant(-1)
    ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), torch.zeros(co, ci, 3, 3, device=dev, dtype=torch.float64), None,
                                              [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    err = ((res[-1][0].double() - ref).abs().max() / ref.abs().max()).item()
    flops = 2.0 * n * ci * co * 9 * gy.shape[2] * gy.shape[3]
    print(f"{n}x({ci}->{co})x{h}x{w} s2: loads {res[0][1]:7.1f} us, LDS-DMA {res[-1][1]:7.1f} us ({flops / res[-1][1] / 1e6 / 833.3:.3f} of the roofline), "
          f"bitwise equal {torch.equal(res[0][0], res[-1][0])}, distance to float64 {err:.1e}", flush=True)
