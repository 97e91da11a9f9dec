"""Times the forward direct convolution on the BasicBlock shapes (see conv_bounds.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: E402,F401
from mscs_amd.models import ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (n, c, h, w) in [(12, 48, 128, 256), (12, 96, 64, 128), (12, 192, 32, 64), (12, 384, 16, 32), (12, 720, 128, 256)]:
    x = torch.randn(n, c, h, w, device=dev).relu_()
    wt = torch.randn(c, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5
    sx, sw = amax_of(x), amax_of(wt)
    wp = ops.conv3x3_pack(wt, sw)
    out = torch.empty_like(x)
    it = 3 if c == 720 else 30
    for _ in range(2):
        ops.conv3x3_launch(x, wp, c, sx, sw, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        ops.conv3x3_launch(x, wp, c, sx, sw, out)
    e1.record()
    torch.cuda.synchronize()
    print(f"  C={c:3d} {h}x{w}: {e0.elapsed_time(e1) / it * 1e3:8.1f} us", flush=True)
    del x, wt, out
    torch.cuda.empty_cache()
