"""Tile sweep of the interleaved kernel on the 48- and 64-channel shapes (branch 0 / layer1: the step's critical chain)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: E402,F401
from mscs_amd import _lib  # noqa: E402
from mscs_amd.models import ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (n, c, h, w) in [(12, 48, 128, 256), (12, 64, 128, 256), (12, 96, 64, 128)]:
    x = torch.randn(n, c, h, w, device=dev).relu_()
    wt = torch.randn(c, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5
    sx, sw = amax_of(x), amax_of(wt)
    wp = ops.conv3x3_pack(wt, sw)
    out = torch.empty_like(x)
    for il in (1, 0):
        L.dcl_conv3x3_set_interleave(il)
        row = []
        for (r, p) in [(0, 0), (2, 2), (2, 4), (2, 1), (1, 4), (1, 2), (3, 4), (3, 2)]:
            for _ in range(2):
                ops.conv3x3_launch(x, wp, c, sx, sw, out, r, p)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.conv3x3_launch(x, wp, c, sx, sw, out, r, p)
            e1.record()
            torch.cuda.synchronize()
            row.append(f"({r},{p}) {e0.elapsed_time(e1) / 30 * 1e3:6.1f}")
        print(f"  C={c} {h}x{w} il={il}: " + "  ".join(row), flush=True)
L.dcl_conv3x3_set_interleave(1)
