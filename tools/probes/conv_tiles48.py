"""A/B of the interleaved (2, 2) tile against its wave-split form (mode 2) on the 48- and 64-channel shapes (branch 0 /
layer1: the step's critical chain).  The two variants are timed ALTERNATELY, one launch each per round, medians over the
rounds: back-to-back blocks of one variant drift by 10 % with the chip's clock."""
import os
import statistics
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
modes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
for (n, c, h, w) in [(12, 48, 128, 256), (12, 64, 128, 256)]:
    x = torch.randn(n, c, h, w, device=dev).relu_()
    wt = torch.randn(c, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5
    sx, sw = amax_of(x), amax_of(wt)
    wp = ops.conv3x3_pack(wt, sw)
    out = torch.empty_like(x)
    times = {m: [] for m in modes}
    for rnd in range(60):
        for m in modes:
            L.dcl_conv3x3_set_interleave(m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.conv3x3_launch(x, wp, c, sx, sw, out)
            e1.record()
            torch.cuda.synchronize()
            if rnd >= 10:
                times[m].append(e0.elapsed_time(e1) / 4 * 1e3)
    print(f"  C={c} {h}x{w}: " + "  ".join(f"mode {m}: median {statistics.median(t):6.1f} us (min {min(t):6.1f})" for m, t in times.items()),
          flush=True)
L.dcl_conv3x3_set_interleave(2)
